"""Batched scan-pair registration sharded across the GPUs of one node (BASELINE config 4).

The reference loops over pairs serially (Registration/main.py:190-216); pairs are independent, so
they are dealt to ranks in contiguous blocks (one process per GPU), every rank registers its own
share -- optionally several pairs in flight on separate HIP streams to hide the per-iteration host
round trip -- and the fixed-size result records are exchanged with ONE all_gather (RCCL over xGMI
on GPUs, gloo in the CPU tests).  There is no per-iteration communication and a single pair is
never split.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

RECORD = 24  # doubles per result record: pair id, 16 x T, iters, status, n_assoc, cost, mean_d2, 2 spare


def shard_range(n_items, rank, world):
    """Contiguous block partition: rank r gets [lo, hi); sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_result(pair_id, res):
    rec = np.zeros(RECORD)
    rec[0] = pair_id
    rec[1:17] = np.asarray(res["T"], dtype=np.float64).reshape(16)
    rec[17] = res.get("iters", 0)
    rec[18] = res.get("status", 0)
    rec[19] = res.get("n_assoc", 0)
    rec[20] = res.get("cost", 0.0)
    rec[21] = res.get("mean_d2", 0.0)
    return rec


def unpack_results(table):
    table = np.asarray(table).reshape(-1, RECORD)
    T = table[:, 1:17].reshape(-1, 4, 4).copy()   # one copy for all pairs; the dicts hold views of it
    ids, iters, status, n_assoc = (table[:, c].astype(np.int64).tolist() for c in (0, 17, 18, 19))
    cost, mean_d2 = table[:, 20].tolist(), table[:, 21].tolist()
    return [{"pair": ids[i], "T": T[i], "iters": iters[i], "status": status[i], "n_assoc": n_assoc[i], "cost": cost[i], "mean_d2": mean_d2[i]}
            for i in range(len(table))]


def gpu_register_fn(device=0, nn="grid", mode="compat", streams=1, **icp_kw):
    """Default worker: ICP on `device` through libpcr (one Context = one HIP stream per worker thread)."""
    from .device import Context, DeviceCloud, TargetIndex, icp_device

    ctxs = [Context(device) for _ in range(max(1, int(streams)))]

    def run(slot, src, tgt, T0):
        ctx = ctxs[slot]  # the caller owns `slot` for the duration of the call (register_batch hands slots out through a queue)
        sd = DeviceCloud.upload(src, ctx)
        index = TargetIndex(DeviceCloud.upload(tgt, ctx), kind=nn, ctx=ctx)
        try:
            return icp_device(sd, index, np.eye(4) if T0 is None else T0, mode=mode, **icp_kw)
        finally:
            sd.free()
            index.free()

    run.streams = len(ctxs)
    run.device = int(device)
    return run


# NumPy mirrors of pcr_pair / pcr_icp_result (include/pcr.h); their sizes are asserted against the ctypes structures on first use
_PAIR_DT = np.dtype([("src", "u8"), ("n_src", "i8"), ("stride_src", "i8"), ("tgt", "u8"), ("n_tgt", "i8"), ("stride_tgt", "i8"), ("T0", "u8")])
_RESULT_DT = np.dtype([("T", "f8", 16), ("T_total", "f8", 16), ("iters", "i4"), ("status", "i4"), ("n_assoc", "i8"), ("cost", "f8"), ("mean_d2", "f8"),
                       ("r_diff", "f8", 256), ("t_diff", "f8", 256), ("device_ms", "f8"), ("nn_kernel_ms", "f8"), ("nn_launches", "i4"), ("reserved", "i4")])
_ctx_pool = {}
_ctx_pool_lock = threading.Lock()
_batch_lock = threading.Lock()


def _pooled_contexts(device, n):
    """n contexts on `device`, created once per process and reused by later batches: a context owns a stream, pinned staging
    buffers and a 256-MiB device arena -- creating a dozen of them per call cost more than registering 100 pairs."""
    from .device import Context

    with _ctx_pool_lock:
        have = _ctx_pool.setdefault(int(device), [])
        while len(have) < n:
            have.append(Context(device))
        return have[:n]


def native_register_share(pairs, device=0, streams=8, mode="compat", max_iter=100, r_thres=0.5, t_thres=0.5, max_d2=5.0,
                          r_metric="frobenius", min_iter=0, nn="grid", as_table=False, first_id=0):
    """The local share of a batch through ONE C call (pcr_icp_batch): `streams` contexts on `device` and as many native
    threads, which pack the clouds of a sub-batch (64 pairs) together and run every stage -- keys, one sort, the grids of all
    targets, every ICP pass, the Procrustes steps -- as one launch for all its pairs (csrc/pcr_batch.hip); no interpreter in
    the loop, results bit-identical to pcr_icp on each pair alone.
    `pairs`: (src (N,>=3) float32, tgt (M,>=3) float32, T0 or None).  Returns result dicts in input order."""
    import ctypes as C

    from . import _lib as L
    if nn != "grid":
        raise ValueError("the native batch path uses the grid index")
    assert _PAIR_DT.itemsize == C.sizeof(L.Pair) and _RESULT_DT.itemsize == C.sizeof(L.IcpResult), "batch.py dtypes out of step with include/pcr.h"
    n = len(pairs)
    if n == 0:
        return []
    ctxs = _pooled_contexts(device, max(1, min(int(streams), n)))
    with _batch_lock:   # the pooled contexts are not thread-safe: one batch at a time per process
        # the pair table and the result table are NumPy structured arrays laid out like pcr_pair / pcr_icp_result: filling and
        # reading them costs a few microseconds per pair instead of ~200 through ctypes attribute access
        # (columns, not rows: a structured row assignment costs ~3 us, an __array_interface__ dictionary ~1.8 us; at 40 000 pairs/s
        # the table building was a quarter of the call)
        cols = [[0] * n for _ in range(7)]
        keep = []
        f32 = np.dtype(np.float32)
        for i, (src, tgt, T0) in enumerate(pairs):
            s = src if (type(src) is np.ndarray and src.dtype == f32 and src.flags.c_contiguous) else np.ascontiguousarray(src, dtype=np.float32)
            t = tgt if (type(tgt) is np.ndarray and tgt.dtype == f32 and tgt.flags.c_contiguous) else np.ascontiguousarray(tgt, dtype=np.float32)
            ss, ts = s.shape, t.shape
            if len(ss) != 2 or len(ts) != 2 or ss[1] < 3 or ts[1] < 3:
                raise ValueError("pairs must hold (N, >= 3) arrays")
            keep.append((s, t))
            cols[0][i], cols[1][i], cols[2][i] = s.ctypes.data, ss[0], ss[1]
            cols[3][i], cols[4][i], cols[5][i] = t.ctypes.data, ts[0], ts[1]
            if T0 is not None:
                T0c = L.as_f64(T0).reshape(16)
                keep.append(T0c)
                cols[6][i] = T0c.ctypes.data
        parr = np.zeros(n, dtype=_PAIR_DT)
        for name, col in zip(_PAIR_DT.names, cols):
            parr[name] = col
        p = L.IcpParams()
        L.lib().pcr_icp_default_params(C.byref(p))
        p.max_iter, p.r_thres, p.t_thres, p.max_d2, p.min_iter = int(max_iter), float(r_thres), float(t_thres), float(max_d2), int(min_iter)
        p.mode = L.PCR_ICP_COMPAT_MAIN if mode == "compat" else L.PCR_ICP_TOTAL
        p.r_metric = L.PCR_RMETRIC_GEODESIC if r_metric == "geodesic" else L.PCR_RMETRIC_FROBENIUS
        res = np.zeros(n, dtype=_RESULT_DT)
        status = np.zeros(n, dtype=np.int32)
        handles = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        rc = L.lib().pcr_icp_batch(handles, len(ctxs), parr.ctypes.data_as(C.POINTER(L.Pair)), n, C.byref(p),
                                   res.ctypes.data_as(C.POINTER(L.IcpResult)), L.iptr(status))
        L.check(rc, ctxs[0].handle)
        if as_table:   # (n, RECORD) rows like pack_result's, built column-wise
            table = np.zeros((n, RECORD))
            table[:, 0] = np.arange(first_id, first_id + n)
            table[:, 1:17] = res["T"]
            table[:, 17], table[:, 18], table[:, 19] = res["iters"], res["status"], res["n_assoc"]
            table[:, 20], table[:, 21] = res["cost"], res["mean_d2"]
            return table
        T, Tt = res["T"].reshape(n, 4, 4), res["T_total"].reshape(n, 4, 4)
        return [{"T": T[i], "T_total": Tt[i], "iters": int(res["iters"][i]), "status": int(res["status"][i]), "n_assoc": int(res["n_assoc"][i]),
                 "cost": float(res["cost"][i]), "mean_d2": float(res["mean_d2"][i])} for i in range(n)]


def register_batch(pairs, register_fn=None, group=None, device=None, streams=8, **kw):
    """Register ``pairs`` = sequence of (src (N,3+), tgt (M,3+), T0 or None).

    Without torch.distributed (or world size 1) everything runs on this process's GPU.  Inside an
    initialised process group every rank must call this with the SAME ``pairs`` list (or at least a
    list of the same length: only the local share is touched); each rank gets the full, ordered
    result list back.
    """
    dist = None
    rank, world = 0, 1
    try:
        import torch.distributed as dist  # noqa: PLC0415

        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        else:
            dist = None
    except Exception:
        dist = None
    n = len(pairs)
    lo, hi = shard_range(n, rank, world)
    native = False
    if register_fn is None:
        if device is None:
            import os

            device = int(os.environ.get("LOCAL_RANK", "0"))
        # float32 records (what the dataset readers return) go through the native batch entry point; anything else (float64
        # clouds, the brute-force index) through the per-pair Python worker
        native = kw.get("nn", "grid") == "grid" and all(
            getattr(pairs[i][0], "dtype", None) == np.float32 and getattr(pairs[i][1], "dtype", None) == np.float32 for i in range(lo, hi))
        if not native:
            register_fn = gpu_register_fn(device=device, streams=streams, **kw)
    workers = int(getattr(register_fn, "streams", 1))
    local = np.zeros((hi - lo, RECORD))
    # A pcr_ctx is not thread-safe: a slot (= one context, one HIP stream) belongs to exactly one task at a time.
    # Slots are taken from a queue when a task STARTS and given back when it ends -- binding them to the task index
    # would let a fast thread start task i + workers on a context a slow thread is still using.
    slots = queue.Queue()
    for w in range(workers):
        slots.put(w)

    def one(i):
        src, tgt, T0 = pairs[i]
        slot = slots.get()
        try:
            res = register_fn(slot, src, tgt, T0)
        finally:
            slots.put(slot)
        local[i - lo] = pack_result(i, res)

    if native:
        if hi > lo:
            local = native_register_share(pairs[lo:hi], device=device, streams=streams, as_table=True, first_id=lo, **kw)
    elif workers > 1 and hi - lo > 1:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(one, range(lo, hi)))
    else:
        for i in range(lo, hi):
            one(i)
    if dist is None or world == 1:
        return unpack_results(local)
    # one collective: fixed-size records, padded to the largest share
    import torch

    share = (n + world - 1) // world
    use_cuda = dist.get_backend(group) == "nccl"
    if use_cuda:
        # the gather buffer lives on the SAME device as this rank's pcr contexts (not on whatever torch's current device is)
        dev_id = getattr(register_fn, "device", device) if register_fn is not None else device
        if dev_id is None:
            import os

            dev_id = int(os.environ.get("LOCAL_RANK", "0"))
        dev = torch.device("cuda", int(dev_id))
    else:
        dev = torch.device("cpu")
    buf = torch.full((share, RECORD), -1.0, dtype=torch.float64, device=dev)
    if hi > lo:
        buf[: hi - lo] = torch.from_numpy(local).to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    table = torch.cat(gathered).cpu().numpy()
    table = table[table[:, 0] >= 0]
    table = table[np.argsort(table[:, 0], kind="stable")]
    return unpack_results(table)
