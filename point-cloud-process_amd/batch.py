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
import weakref
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


# NumPy mirrors of pcr_cloud_ref / pcr_pair_ref / pcr_icp_result (include/pcr.h); their sizes are asserted against the ctypes structures on first use
_PAIR_DT = np.dtype([("src", "u8"), ("n_src", "i8"), ("stride_src", "i8"), ("tgt", "u8"), ("n_tgt", "i8"), ("stride_tgt", "i8"), ("T0", "u8")])   # pcr_pair (pcr_icp_batch)
_CLOUD_DT = np.dtype([("xyz", "u8"), ("n", "i8"), ("stride", "i8")])
_PAIRREF_DT = np.dtype([("src", "i4"), ("tgt", "i4"), ("T0", "u8")])
_RESULT_DT = np.dtype([("T", "f8", 16), ("T_total", "f8", 16), ("iters", "i4"), ("status", "i4"), ("n_assoc", "i8"), ("cost", "f8"), ("mean_d2", "f8"),
                       ("r_diff", "f8", 256), ("t_diff", "f8", 256), ("device_ms", "f8"), ("nn_kernel_ms", "f8"), ("nn_launches", "i4"), ("reserved", "i4")])
_ctx_pool = {}
_ctx_pool_lock = threading.Lock()
_batch_lock = threading.Lock()
_array_facts = {}   # see native_register_share
native_calls = []   # (first pair id, pairs, scans, with global initialisation) of every native batch call of this process: what the tests' spies read


def _pooled_contexts(device, n):
    """n contexts on `device`, created once per process and reused by later batches: a context owns a stream, pinned staging
    buffers and a 256-MiB device arena -- creating a dozen of them per call cost more than registering 100 pairs."""
    from .device import Context

    with _ctx_pool_lock:
        have = _ctx_pool.setdefault(int(device), [])
        while len(have) < n:
            have.append(Context(device))
        return have[:n]


def global_params(voxel_size=2.0, seed=0, **kw):
    """pcr_global_params with the values of Registration/main.py:33-84,196 (voxel 2.0 -> normals 4.0 / 30, FPFH 10.0 / 100, RANSAC 3.0 m,
    0.9, 100 000 / 0.999); keyword overrides: normal_radius, fpfh_radius, normal_max_nn, fpfh_max_nn, mutual_filter, max_iteration,
    confidence, max_distance, edge_similarity, check_distance."""
    import ctypes as C

    from . import _lib as L
    g = L.GlobalParams()
    L.check(L.lib().pcr_global_default_params(float(voxel_size), C.byref(g)))
    g.ransac.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    for k, v in kw.items():
        if hasattr(g, k) and k != "ransac":
            setattr(g, k, type(getattr(g, k))(v))
        elif hasattr(g.ransac, k):
            setattr(g.ransac, k, type(getattr(g.ransac, k))(v))
        else:
            raise TypeError(f"unknown global-initialisation parameter {k!r}")
    return g


def native_register_share(pairs, device=0, streams=8, mode="compat", max_iter=100, r_thres=0.5, t_thres=0.5, max_d2=5.0,
                          r_metric="frobenius", min_iter=0, nn="grid", as_table=False, first_id=0, global_init=None, return_init=False):
    """The local share of a batch through ONE C call (pcr_register_pairs): `streams` contexts on `device` and as many native
    threads.  The clouds of the share form a SCAN TABLE -- arrays that are the same object (or the same memory) are the same scan,
    uploaded and, with `global_init`, preprocessed once however many pairs use it (Registration/reg_result.txt: 342 pairs over 504
    scans) --; `global_init` (a dict for global_params(), or True) runs prepare_dataset + execute_global_registration
    (main.py:196-203) for every pair without a T0; then the fused batch stages register every pair (csrc/pcr_batch.hip): no
    interpreter in the loop, results bit-identical to pcr_icp on each pair alone.
    `pairs`: (src (N,>=3) float32, tgt (M,>=3) float32, T0 or None).  Returns result dicts in input order."""
    import ctypes as C

    from . import _lib as L
    if nn != "grid":
        raise ValueError("the native batch path uses the grid index")
    assert (_CLOUD_DT.itemsize == C.sizeof(L.CloudRef) and _PAIRREF_DT.itemsize == C.sizeof(L.PairRef) and
            _RESULT_DT.itemsize == C.sizeof(L.IcpResult)), "batch.py dtypes out of step with include/pcr.h"
    n = len(pairs)
    if n == 0:
        return []
    ctxs = _pooled_contexts(device, max(1, min(int(streams), n)))
    with _batch_lock:   # the pooled contexts are not thread-safe: one batch at a time per process
        # scan table + pair table as NumPy structured arrays laid out like the C structures, filled column-wise (a ctypes attribute
        # access costs ~200 ns, a structured row assignment ~3 us: at 40 000 pairs/s the table building was a quarter of the call)
        scans = {}          # id(array) -> row (an array handed in for several pairs is one scan)
        by_mem = {}         # (address, n, stride) -> row (views of the same memory too)
        c_ptr, c_n, c_st = [], [], []
        p_src, p_tgt, p_T0 = [0] * n, [0] * n, [0] * n
        keep = []
        f32 = np.dtype(np.float32)
        known = _array_facts    # id(array) -> (weak reference, address, rows, columns) of float32 C-contiguous arrays seen before

        def scan_row_slow(a, ia):
            s = a if (type(a) is np.ndarray and a.dtype is f32 and a.flags.c_contiguous) else np.ascontiguousarray(a, dtype=np.float32)
            sh = s.shape
            if len(sh) != 2 or sh[1] < 3:
                raise ValueError("pairs must hold (N, >= 3) arrays")
            key = (s.__array_interface__["data"][0], sh[0], sh[1])
            if s is a:
                if len(known) > 65536:
                    known.clear()
                known[ia] = (weakref.ref(a, lambda _r, ia=ia: known.pop(ia, None)),) + key
            row = by_mem.get(key)
            if row is None:
                row = len(c_ptr)
                by_mem[key] = row
                c_ptr.append(key[0]); c_n.append(key[1]); c_st.append(key[2])
            keep.append(s)
            if s is a:
                scans[ia] = row     # (a converted copy is a fresh scan each time: its source may be anything)
            return row

        # (the loop is written out: at 40 000 pairs/s a Python function call per cloud is a tenth of the batch)
        for i, (src, tgt, T0) in enumerate(pairs):
            ia = id(src)
            row = scans.get(ia)
            if row is None:
                fact = known.get(ia)
                if fact is not None and fact[0]() is src:   # (asking NumPy for an array's address costs ~2 us; the same scan objects come back call after call)
                    row = scans[ia] = len(c_ptr)
                    c_ptr.append(fact[1]); c_n.append(fact[2]); c_st.append(fact[3])
                else:
                    row = scan_row_slow(src, ia)
            p_src[i] = row
            ia = id(tgt)
            row = scans.get(ia)
            if row is None:
                fact = known.get(ia)
                if fact is not None and fact[0]() is tgt:
                    row = scans[ia] = len(c_ptr)
                    c_ptr.append(fact[1]); c_n.append(fact[2]); c_st.append(fact[3])
                else:
                    row = scan_row_slow(tgt, ia)
            p_tgt[i] = row
            if T0 is not None:
                T0c = L.as_f64(T0).reshape(16)
                keep.append(T0c)
                p_T0[i] = T0c.__array_interface__["data"][0]
        carr = np.zeros(len(c_ptr), dtype=_CLOUD_DT)
        carr["xyz"], carr["n"], carr["stride"] = c_ptr, c_n, c_st
        parr = np.zeros(n, dtype=_PAIRREF_DT)
        parr["src"], parr["tgt"], parr["T0"] = p_src, p_tgt, p_T0
        p = L.IcpParams()
        L.lib().pcr_icp_default_params(C.byref(p))
        p.max_iter, p.r_thres, p.t_thres, p.max_d2, p.min_iter = int(max_iter), float(r_thres), float(t_thres), float(max_d2), int(min_iter)
        p.mode = L.PCR_ICP_COMPAT_MAIN if mode == "compat" else L.PCR_ICP_TOTAL
        p.r_metric = L.PCR_RMETRIC_GEODESIC if r_metric == "geodesic" else L.PCR_RMETRIC_FROBENIUS
        gp = None
        if global_init:
            gp = global_init if isinstance(global_init, L.GlobalParams) else global_params(**(global_init if isinstance(global_init, dict) else {}))
        res = np.zeros(n, dtype=_RESULT_DT)
        status = np.zeros(n, dtype=np.int32)
        T_init = np.zeros((n, 16)) if (return_init or gp is not None) else None
        handles = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
        native_calls.append((int(first_id), n, len(c_ptr), gp is not None))
        del native_calls[:-64]
        rc = L.lib().pcr_register_pairs(handles, len(ctxs), carr.ctypes.data_as(C.POINTER(L.CloudRef)), len(c_ptr),
                                        parr.ctypes.data_as(C.POINTER(L.PairRef)), n, C.byref(gp) if gp is not None else None, C.byref(p),
                                        res.ctypes.data_as(C.POINTER(L.IcpResult)), L.iptr(status), L.dptr(T_init) if T_init is not None else None)
        L.check(rc, ctxs[0].handle)
        if as_table:   # (n, RECORD) rows like pack_result's, built column-wise
            table = np.zeros((n, RECORD))
            table[:, 0] = np.arange(first_id, first_id + n)
            table[:, 1:17] = res["T"]
            table[:, 17], table[:, 18], table[:, 19] = res["iters"], res["status"], res["n_assoc"]
            table[:, 20], table[:, 21] = res["cost"], res["mean_d2"]
            return (table, T_init.reshape(n, 4, 4)) if return_init else table
        T, Tt = res["T"].reshape(n, 4, 4), res["T_total"].reshape(n, 4, 4)
        out = [{"T": T[i], "T_total": Tt[i], "iters": int(res["iters"][i]), "status": int(res["status"][i]), "n_assoc": int(res["n_assoc"][i]),
                "cost": float(res["cost"][i]), "mean_d2": float(res["mean_d2"][i])} for i in range(n)]
        if T_init is not None:
            for i, o in enumerate(out):
                o["T_init"] = T_init[i].reshape(4, 4)
        return out


def register_batch(pairs, register_fn=None, group=None, device=None, streams=8, global_init=None, **kw):
    """Register ``pairs`` = sequence of (src (N,3+), tgt (M,3+), T0 or None).

    ``global_init`` (True, or a dict of global_params() overrides): pairs without a T0 start from the reference's own
    initialisation (prepare_dataset + execute_global_registration, Registration/main.py:196-203), computed INSIDE the rank's
    native call for its own share only; needs float32 clouds (the native path).

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
        f32 = np.dtype(np.float32)
        native = kw.get("nn", "grid") == "grid" and all(
            getattr(pairs[i][0], "dtype", None) is f32 and getattr(pairs[i][1], "dtype", None) is f32 for i in range(lo, hi))
        if not native:
            if global_init:
                raise ValueError("global_init needs the native batch path: float32 clouds and the grid index")
            register_fn = gpu_register_fn(device=device, streams=streams, **kw)
    elif global_init:
        raise ValueError("global_init is computed by the native worker: do not pass register_fn with it")
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
            local = native_register_share(pairs[lo:hi], device=device, streams=streams, as_table=True, first_id=lo, global_init=global_init, **kw)
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
