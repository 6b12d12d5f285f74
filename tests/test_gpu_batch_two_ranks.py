"""GPU: the real worker + shard + gather across two processes (Registration/main.py:190 is the loop being
sharded).  Two fresh ranks (gloo rendezvous, both on device 0 -- the only on-hardware rehearsal possible without a
multi-GPU node) call register_batch(pairs) with its DEFAULT worker: the rank's share goes through the native entry point
pcr_icp_batch (fused batch stages), the records through one all_gather.  Every rank's table must equal a serial
single-process run bit for bit, and a rank may only have computed its own share."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_PAIRS = 10


def _pairs(syn):
    rng = np.random.default_rng(11)
    pairs = []
    for i in range(N_PAIRS):
        n = int(rng.choice([500, 2500, 9000, 21000]))
        s, t, _ = syn.perturbed_pair(n, seed=700 + i, angle_deg=float(rng.uniform(0.5, 5.0)), t=tuple(rng.uniform(-0.6, 0.6, 3) * [1, 1, 0.1]))
        pairs.append((s, t, None))
    return pairs


KW = dict(mode="total", max_iter=25, r_thres=1e-4, t_thres=1e-4)


def _rank(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("point-cloud-process_amd")
        batch = importlib.import_module("point-cloud-process_amd.batch")
        pairs = _pairs(pkg.synthetic)
        seen = []
        real = batch.native_register_share

        def spy(share, **kw):
            seen.append(len(share))
            return real(share, **kw)

        batch.native_register_share = spy
        res = batch.register_batch(pairs, device=0, streams=2, **KW)
        lo, hi = batch.shard_range(N_PAIRS, rank, world)
        assert seen == [hi - lo]                                    # one native call, the local share only
        assert [r["pair"] for r in res] == list(range(N_PAIRS))     # the full ordered list on every rank
        np.save(os.path.join(out_dir, f"T_{rank}.npy"), np.stack([r["T"] for r in res]))
        np.save(os.path.join(out_dir, f"it_{rank}.npy"), np.array([[r["iters"], r["n_assoc"], r["status"]] for r in res]))
    finally:
        dist.destroy_process_group()


def test_register_batch_two_ranks_native_worker(tmp_path, pcp, syn):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "T_0.npy"), np.load(tmp_path / "T_1.npy")
    ia, ib = np.load(tmp_path / "it_0.npy"), np.load(tmp_path / "it_1.npy")
    assert a.shape == (N_PAIRS, 4, 4) and np.array_equal(a, b) and np.array_equal(ia, ib)
    # serial single-process run, one pair at a time through pcr_icp (not the batch entry point)
    for i, (s, t, _) in enumerate(_pairs(syn)):
        index = pcp.TargetIndex(pcp.DeviceCloud.upload(t))
        sd = pcp.DeviceCloud.upload(s)
        r = pcp.icp_device(sd, index, np.eye(4), **KW)
        sd.free()
        index.free()
        assert np.array_equal(a[i], r["T"]), i
        assert (ia[i] == [r["iters"], r["n_assoc"], r["status"]]).all(), i
    assert len(set(ia[:, 0])) >= 2          # the pairs really differ in work
