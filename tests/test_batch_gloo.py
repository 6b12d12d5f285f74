"""World-size-2 gloo test (CPU) of the batched-registration sharding + result gather.  The worker is
injected (the CPU oracle), because the product's own worker needs a GPU; what is under test is the
partition, the ordering and the single all_gather."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("point-cloud-process_amd")
        batch = importlib.import_module("point-cloud-process_amd.batch")
        oracle = importlib.import_module("oracle.oracle_np")
        syn = pkg.synthetic
        pairs = []
        for i in range(n_pairs):
            src, tgt, _ = syn.perturbed_pair(600, seed=100 + i)
            pairs.append((src, tgt, None))
        calls = []

        def fn(slot, src, tgt, T0):
            calls.append(slot)
            r = oracle.icp_point2point(src, tgt, np.eye(4))
            return {"T": r["T"], "iters": r["iters"], "status": int(r["failed"]), "n_assoc": 0, "cost": 0.0, "mean_d2": 0.0}

        res = batch.register_batch(pairs, register_fn=fn)
        lo, hi = batch.shard_range(n_pairs, rank, world)
        assert len(calls) == hi - lo  # only the local share was computed
        assert [r["pair"] for r in res] == list(range(n_pairs))  # full ordered list on every rank
        np.save(os.path.join(out_dir, f"T_{rank}.npy"), np.stack([r["T"] for r in res]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 2])
def test_register_batch_two_ranks_gloo(tmp_path, n_pairs):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_pairs, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "T_0.npy")
    b = np.load(tmp_path / "T_1.npy")
    assert a.shape == (n_pairs, 4, 4) and np.array_equal(a, b)
    # same answers as a serial run of the same worker
    oracle = importlib.import_module("oracle.oracle_np")
    syn = importlib.import_module("point-cloud-process_amd.synthetic")
    for i in range(n_pairs):
        src, tgt, _ = syn.perturbed_pair(600, seed=100 + i)
        assert np.array_equal(a[i], oracle.icp_point2point(src, tgt, np.eye(4))["T"])


def test_shard_range_covers_everything():
    batch = importlib.import_module("point-cloud-process_amd.batch")
    for n in (0, 1, 7, 256):
        for world in (1, 2, 3, 8):
            spans = [batch.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_register_batch_never_shares_a_slot():
    """A slot (= one pcr context / HIP stream, not thread-safe) must never run two pairs at once, whatever order the
    worker threads finish in (slots bound to the task index let a fast thread reuse a busy context)."""
    import threading
    import time

    batch = importlib.import_module("point-cloud-process_amd.batch")
    busy, lock, overlaps, seen = set(), threading.Lock(), [], []

    def fn(slot, src, tgt, T0):
        with lock:
            if slot in busy:
                overlaps.append(slot)
            busy.add(slot)
            seen.append(slot)
        time.sleep(0.002 * float(src[0, 0]))   # unequal durations: tasks finish out of order
        with lock:
            busy.discard(slot)
        return {"T": np.eye(4) * float(src[0, 0]), "iters": 1, "status": 0}

    fn.streams = 3
    durations = [9, 1, 1, 1, 7, 1, 1, 1, 1, 5, 1, 1]
    pairs = [(np.full((4, 3), float(d)), np.zeros((4, 3)), None) for d in durations]
    res = batch.register_batch(pairs, register_fn=fn)
    assert not overlaps
    assert set(seen) == {0, 1, 2}
    assert [r["pair"] for r in res] == list(range(len(pairs)))
    assert [r["T"][0, 0] for r in res] == [float(d) for d in durations]
