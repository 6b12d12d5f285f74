#!/usr/bin/env python3
"""Benchmark of the registration hot path (BASELINE.json metric).

A "step" is one point-to-point ICP iteration on a 120 000 x 120 000-point
KITTI-shaped synthetic scan pair (BASELINE.json configs[1]): in-place source
transform + exact 1-NN of every source point + d2 < 5 gate + Procrustes moment
accumulation + 3x3 Procrustes step + convergence test, all on the GPU in ONE
launch per iteration (the loop state lives on the device; the host only enqueues
passes).  Inputs and the target index are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nn grid|brute]

--gpus N > 1 without a launcher: this process starts N ranks ITSELF (a child
`python -m torch.distributed.run --nproc-per-node N ... bench.py`, before any
GPU call is made here) and relays their output; under torch.distributed.run
(WORLD_SIZE set) it is one of the ranks.  One process per GPU: every rank
registers its own independent 120k pair (the path shards across pairs, never
inside one), results are gathered with one all_gather over
RCCL, the timing is the max over ranks.  Further blocks of the same JSON line:
"batch256" = BASELINE configs[3] (256 registration_dataset-shaped pairs of
20 000 6-float records) through register_batch -> pcr_icp_batch (fused batch
stages), block-sharded over the ranks, one RCCL all_gather of the result
records; "config3" = configs[2] (0.2 m voxel filter of both 120k clouds -> ICP);
"config5" = configs[4] (1 M-point ISS + coarse-to-fine ICP), N = 1 only.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_POINTS = 120_000
MAX_D2 = 5.0
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix; scripts/mfma_f64_peak measures the achievable rate
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_PEAK_GBS = 63.0           # PCIe 5.0 x16, one direction
# algorithmic HBM bytes per correspondence of the grid pass: SURVEY 8d's figure (16 B query + 16 B matched target + 8 B result);
# the binary64 record layout actually moves 100 B (32 B source record read + 32 B written back in place + 32 B matched target
# record + 4 B result slot): reported beside it as `frac_f64_layout`
GRID_BYTES_PER_CORR = 40.0
GRID_LAYOUT_BYTES_PER_CORR = 100.0
# VALU issue ceiling (MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 binary32 VALU instruction issues over 2 cycles; binary64
# ones take twice as long, so this ceiling is generous for this kernel's mix)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2    # wave64 VALU instructions per second: one per SIMD every 2 cycles
BATCH_PAIRS, BATCH_POINTS = 256, 20_000


def profile_file(name):
    """Newest committed rocprofv3 counter summary of that name under profiles/ (r03_..., else r02_...): counters cannot be
    collected inside this process, so the per-launch HBM traffic and instruction counts come from the committed passes."""
    for tag in ("r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", f"{tag}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", "r04_" + name)


def launch_ranks(n):
    """Start n ranks of this script under torch.distributed.run and relay their output.  Runs BEFORE this process makes
    any GPU call (a process that has initialised the GPU must never be replaced or re-exec'd)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(src, tgt, budget_s=12.0, max_iters=40):
    """Reference-equivalent CPU path (oracle, kind "port"): scipy cKDTree exact 1-NN on all
    host cores + NumPy mean-subtraction Procrustes, same gate and loop as main.py:105-146."""
    oracle = importlib.import_module("oracle.oracle_np")
    from scipy.spatial import cKDTree

    cores = len(os.sched_getaffinity(0))
    s = src.astype(np.float64)
    t = tgt.astype(np.float64)
    t0 = time.perf_counter()
    tree = cKDTree(t)
    build_s = time.perf_counter() - t0
    T = np.eye(4)
    iters = 0
    t0 = time.perf_counter()
    while iters < max_iters and (time.perf_counter() - t0) < budget_s:
        s = s @ T[:3, :3].T + T[:3, 3]
        _, j = tree.query(s, k=1, workers=-1)
        d2 = oracle.dist2_direct(s, t[j])
        keep = d2 < MAX_D2
        R, tt, _ = oracle.procrustes(s[keep].T, t[j[keep]].T)
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = tt.squeeze()
        iters += 1
    el = time.perf_counter() - t0
    return {
        "value": len(src) * iters / el,
        "unit": "correspondences/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{iters} ICP iterations on the same 120k x 120k pair (scipy cKDTree workers=-1 + NumPy Procrustes), "
                  f"{el:.1f} s; one-off tree build {build_s * 1e3:.0f} ms not counted",
        "ms_per_iter": 1e3 * el / iters,
    }


def parity_gates(pkg, index, src, tgt):
    """The gates SURVEY 8d asks to report next to every timing, on the bench pair itself (CPU oracle = checker only):
    exact 1-NN of all 120k source points against scipy's tree (index mismatches away from ties, squared distances
    bit-identical) and the composed transform after 10 ICP iterations against the CPU restatement."""
    oracle = importlib.import_module("oracle.oracle_np")
    s64, t64 = src.astype(np.float64), tgt.astype(np.float64)
    idx, d2 = index.nn1(src)
    oi, od2, margin = oracle.nn1_exact(s64, t64, workers=-1)
    clear = margin > 1e-12
    T_cpu, _ = oracle.icp_total(s64, t64, max_iteration=10, R_diff_thres=-1.0, t_diff_thres=-1.0)
    sd = pkg.DeviceCloud.upload(src, index.ctx)
    r = pkg.icp_device(sd, index, np.eye(4), mode="total", max_iter=10, r_thres=-1.0, t_thres=-1.0, max_d2=MAX_D2, min_iter=10)
    sd.free()
    return {"nn_index_mismatch_away_from_ties": int((idx[clear] != oi[clear]).sum()), "nn_ties_excluded": int((~clear).sum()),
            "nn_d2_bit_identical": bool(np.array_equal(d2, od2)),
            "Rt_frobenius_vs_cpu_after_10_iters": float(np.linalg.norm(r["T_total"] - T_cpu)), "tolerance": 1e-4}


def run_icp_steps(pkg, index, src_host, steps, ctx, sd=None):
    """Exactly `steps` ICP iterations (thresholds off), device-resident inputs.  `sd`: a source cloud already uploaded
    and laid out (the timed run hands one in so that the set-up stays outside the timed region).  Returns dict."""
    out = {"iters": 0, "device_ms": 0.0, "nn_kernel_ms": 0.0, "nn_launches": 0}
    own_sd = sd is None
    if own_sd:
        sd = pkg.DeviceCloud.upload(src_host, ctx).prepare(index)  # resident + laid out before the timed region
    T0 = np.eye(4)
    ctx.sync()
    t0 = time.perf_counter()
    left = steps
    last = None
    while left > 0:
        k = min(left, 256)
        r = pkg.icp_device(sd, index, T0, mode="total", max_iter=k, r_thres=-1.0, t_thres=-1.0, max_d2=MAX_D2, min_iter=k)
        T0 = np.eye(4)  # later chunks continue from the already-transformed source
        for key in ("iters", "device_ms", "nn_kernel_ms", "nn_launches"):
            out[key] += r[key]
        last = r
        left -= k
    ctx.sync()
    out["wall_s"] = time.perf_counter() - t0
    out["n_assoc"] = last["n_assoc"]
    out["T_total"] = last["T_total"]
    if own_sd:
        sd.free()   # (a cloud handed in is the caller's to release -- after its timed region: releasing device memory is not a step)
    return out


def concurrent_leg(pkg, dev_id, src, tgt, n_pairs, steps):
    """Aggregate throughput with `n_pairs` independent copies of the workload in flight on ONE GPU (one context =
    one HIP stream per pair, one host thread each): what a rank of the batched job (BASELINE configs[3]) does.  A single
    pair is latency-bound; this leg shows the throughput bound (contexts carry the shared-device hint)."""
    import threading

    ctxs = [pkg.Context(dev_id, shared=True) for _ in range(n_pairs)]
    work = []
    for c in ctxs:
        idx = pkg.TargetIndex(pkg.DeviceCloud.upload(tgt, c), ctx=c)
        work.append((pkg.DeviceCloud.upload(src, c).prepare(idx), idx))

    def run(i, k):
        sd, idx = work[i]
        pkg.icp_device(sd, idx, np.eye(4), mode="total", max_iter=k, r_thres=-1.0, t_thres=-1.0, max_d2=MAX_D2, min_iter=k)

    def wave(k):
        th = [threading.Thread(target=run, args=(i, k)) for i in range(n_pairs)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for c in ctxs:
            c.sync()

    wave(5)
    t0 = time.perf_counter()
    wave(steps)
    el = time.perf_counter() - t0
    for sd, idx in work:
        sd.free()
        idx.free()
    for c in ctxs:
        c.close()
    return {"pairs_in_flight": n_pairs, "steps_per_pair": steps, "value": float(len(src)) * steps * n_pairs / el,
            "unit": "correspondences/s", "ms_per_icp_iter_per_pair": 1e3 * el / steps}


def filter_pairs_per_pass(pkg, dev_id, src, tgt, cell):
    """(query, staged candidate) pairs the tile kernel's filter evaluates in one pass over the bench pair: read from the
    kernel's own per-tile stamps (PCR_DEBUG_STAMPS) on a throw-away context, untimed.  The counters are a run-time switch of the
    stand-alone tile kernel only (in the one-launch pass they are compiled in by -DPCR_PASS_DIAG=1), so this ICP runs the
    tile -> hard -> accumulate launches (PCR_ICP_NO_FUSED): the same tile code on the same queries."""
    import ctypes as C

    os.environ["PCR_DEBUG_STAMPS"] = "1"
    try:
        c = pkg.Context(dev_id)
    finally:
        del os.environ["PCR_DEBUG_STAMPS"]
    idx = pkg.TargetIndex(pkg.DeviceCloud.upload(tgt, c), cell=cell, ctx=c)
    sd = pkg.DeviceCloud.upload(src, c).prepare(idx)
    os.environ["PCR_ICP_NO_FUSED"] = "1"
    try:
        pkg.icp_device(sd, idx, np.eye(4), mode="total", max_iter=2, r_thres=-1.0, t_thres=-1.0, max_d2=MAX_D2, min_iter=2)
    finally:
        del os.environ["PCR_ICP_NO_FUSED"]
    nb = (len(src) + 63) // 64
    buf = np.zeros(nb * 4, dtype=np.uint64)
    L = pkg._lib
    L.check(L.lib().pcr_debug_read(c.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
    pairs = buf.reshape(nb, 4)[:, 1].astype(np.float64)   # (query, staged candidate) pairs of every block of 4 wave tiles
    sd.free()
    idx.free()
    c.close()
    return float(pairs.sum()), float(pairs.sum() / len(src))


def batch_leg(pkg, torch, dist, rank, world, dev_id, tdev, streams):
    """BASELINE configs[3]: 256 pairs x 20 000-point 6-float records (Registration/main.py:190-216 is the loop being
    sharded) through register_batch with its default worker: the rank's share goes through ONE C call, pcr_icp_batch
    (fused batch stages: host packing, upload, index build and ICP of every pair inside the timed region), then one
    all_gather of the result records.  Two passes: the reference's own stopping rule (compat, thresholds 0.5: 1-2
    iterations) and a tight one (composed transform, thresholds 1e-3, at most 30 iterations).  At N = 1 also the
    per-pair path (PCR_BATCH_PER_PAIR=1: upload -> index build -> pcr_icp per pair on the same contexts) and a profiled,
    untimed run for the dominant kernel's roofline."""
    batch = importlib.import_module("point-cloud-process_amd.batch")
    lo, hi = batch.shard_range(BATCH_PAIRS, rank, world)
    mine = pkg.synthetic.registration_batch_6f(BATCH_PAIRS, BATCH_POINTS, seed=1000, indices=range(lo, hi))
    pairs = [None] * BATCH_PAIRS
    truth = {}
    for i, (s6, t6, Tt) in zip(range(lo, hi), mine):
        pairs[i] = (s6, t6, None)
        truth[i] = Tt
    for i in range(BATCH_PAIRS):   # other ranks' pairs are never touched: placeholders keep the list length
        if pairs[i] is None:
            pairs[i] = (None, None, None)
    out = {"pairs": BATCH_PAIRS, "points_per_cloud": BATCH_POINTS, "record": "6 x f32 (x,y,z,nx,ny,nz)", "streams_per_gpu": streams,
           "pairs_per_gpu": hi - lo, "entry_point": "pcr_icp_batch (fused batch stages) via register_batch(pairs, device=..., streams=...)"}
    runs = []
    modes = (("compat", dict(mode="compat")), ("tight", dict(mode="total", max_iter=30, r_thres=1e-3, t_thres=1e-3)))

    def timed(kw, reps):
        best, res = None, None
        runs.clear()
        for _ in range(reps):
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            res = batch.register_batch(pairs, device=dev_id, streams=streams, **kw)
            if dist is not None:
                dist.barrier()
            el = time.perf_counter() - t0
            if dist is not None:
                tm = torch.tensor([el], dtype=torch.float64, device=tdev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                el = float(tm.item())
            best = el if best is None else min(best, el)
            runs.append(el)
        return best, res

    for tag, kw in modes:
        batch.native_register_share(pairs[lo:hi], device=dev_id, streams=streams, **kw)   # warm the pooled contexts (arenas, pinned buffers, code objects), untimed
        el, res = timed(kw, 8)
        runs_s = list(runs)
        iters = np.array([r["iters"] for r in res])
        errs = [float(np.linalg.norm(res[i]["T"] - truth[i])) for i in truth] if tag == "tight" else []
        out[tag] = {"seconds": el, "pairs_per_s": BATCH_PAIRS / el, "pairs_per_s_per_gpu": BATCH_PAIRS / el / world,
                    "timing": "best of 8 runs of the whole batch (host packing + H2D + index builds + ICP + gather)", "runs_s": runs_s,
                    # both clouds of every pair as 16-B records over the wall time, against N x 8 TB/s
                    "hbm_frac_algorithmic": BATCH_PAIRS * 2 * BATCH_POINTS * 16 / el / (HBM_PEAK_GBS * 1e9 * world),
                    "pcie_bytes": (hi - lo) * 2 * BATCH_POINTS * 12,
                    # what bounds the batch: both clouds of every pair cross PCIe once as packed float32 coordinates (12 B per point)
                    "roofline_pcie": {"bound": "pcie", "achieved": (hi - lo) * 2 * BATCH_POINTS * 12 / el / 1e9, "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                                      "frac": (hi - lo) * 2 * BATCH_POINTS * 12 / el / 1e9 / PCIE_PEAK_GBS,
                                      "note": "host -> device bytes of this rank over the whole batch's wall time, against the 63 GB/s of a PCIe 5.0 x16 link"},
                    "correspondences_per_s": float(iters.sum()) * BATCH_POINTS / el, "mean_iters": float(iters.mean()),
                    "results_gathered": len(res)}
        if errs:
            # point-to-point ICP on sparse ring-structured sweeps keeps a few decimetres of bias (the CPU oracle lands on the same
            # transform: tests/test_gpu_voxel_knn_iss.py); reported for orientation, not a parity gate
            out[tag]["median_T_error_vs_truth_local_share"] = float(np.median(errs))
        if world == 1:
            # the per-pair path on the same contexts (what round 2 shipped), and bitwise agreement of the two
            os.environ["PCR_BATCH_PER_PAIR"] = "1"
            try:
                el_pp, res_pp = timed(kw, 8)
            finally:
                del os.environ["PCR_BATCH_PER_PAIR"]
            out[tag]["per_pair_path"] = {"seconds": el_pp, "pairs_per_s": BATCH_PAIRS / el_pp,
                                         "results_bitwise_equal_to_fused": bool(all(np.array_equal(a["T"], b["T"]) and a["iters"] == b["iters"]
                                                                                    for a, b in zip(res, res_pp)))}
            # dominant kernel of the fused path, HIP events around every launch of ONE sub-batch stream (untimed, profiled run)
            ctx0 = batch._pooled_contexts(dev_id, 1)[0]
            ctx0.profile(True)
            n_prof = min(32, hi - lo)
            batch.native_register_share(pairs[lo:lo + n_prof], device=dev_id, streams=1, **kw)
            ms, passes = ctx0.profile_read()
            ctx0.profile(False)
            if passes:
                k_us = [1e3 * m / passes for m in ms[:3]]
                algo = 40.0 * n_prof * BATCH_POINTS     # 16 B query + 16 B target + 8 B result per correspondence
                out[tag]["roofline"] = {"bound": "hbm", "kernel": "batch_pass_kernel", "achieved": algo / (k_us[0] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": algo / (k_us[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                        "pairs_per_launch": n_prof, "passes_profiled": passes,
                                        "kernel_us": {"batch_pass_kernel": k_us[0], "batch_drain_kernel": k_us[1], "batch_finish_kernel": k_us[2]},
                                        "note": "one launch serves every pair of a sub-batch; VALU-issue / latency bound like grid_pass_kernel, "
                                                "rocprofv3 stats of the whole batch in profiles/r03_batch_*_kernel_stats.csv"}
    # algorithmic HBM bytes of the batch: both clouds of every pair as 16-B records
    out["algorithmic_bytes"] = BATCH_PAIRS * 2 * BATCH_POINTS * 16
    return out


def setup_leg(pkg, ctx, src, tgt, cell):
    """The set-up of one registration, outside the timed region of `value` (Registration/main.py:105 builds its KD-tree once per
    pair and stops after 1-2 iterations: for the reference's own usage the set-up is most of a registration): best of 10, ms."""
    def best(fn, free=True, reps=10):
        t_best = None
        for _ in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            r = fn()
            ctx.sync()
            el = 1e3 * (time.perf_counter() - t0)
            t_best = el if t_best is None else min(t_best, el)
            if free:
                r.free()
        return t_best

    up_t = best(lambda: pkg.DeviceCloud.upload(tgt, ctx))
    dt = pkg.DeviceCloud.upload(tgt, ctx)
    build = best(lambda: pkg.TargetIndex(dt, cell=cell, ctx=ctx))
    index = pkg.TargetIndex(dt, cell=cell, ctx=ctx)
    up_s = best(lambda: pkg.DeviceCloud.upload(src, ctx))
    clouds = [pkg.DeviceCloud.upload(src, ctx) for _ in range(10)]
    it = iter(clouds)
    prep = best(lambda: next(it).prepare(index), free=False)
    for c in clouds:
        c.free()
    one = best(lambda: pkg.icp_device(pkg.DeviceCloud.upload(src, ctx), index, np.eye(4)), free=False)
    index.free()
    dt.free()
    return {"target_upload_ms": up_t, "index_build_ms": build, "source_upload_ms": up_s, "prepare_ms": prep,
            "compat_registration_on_built_index_ms": one,
            "note": "wall, best of 10; uploads include the host packing + bounding box and the PCIe transfer (1.44 MB per 120k cloud)"}


def config3_leg(pkg, ctx, src, tgt):
    """BASELINE configs[2]: 0.2 m voxel filter (Pca_and_Voxel_filter/voxel_filter.py:10-68 as called at :87-90) of both
    120k clouds, device resident, then point-to-point ICP on the filtered clouds (Registration/main.py:211)."""
    ds0, dt0 = pkg.DeviceCloud.upload(src, ctx), pkg.DeviceCloud.upload(tgt, ctx)
    pkg.voxel_filter_device(ds0, 0.2).free()   # warm
    ctx.sync()
    reps = 10
    ctx.timer_start()
    t0 = time.perf_counter()
    outs = [pkg.voxel_filter_device(ds0, 0.2) for _ in range(reps)]
    dev_ms = ctx.timer_stop_ms() / reps
    wall_ms = 1e3 * (time.perf_counter() - t0) / reps
    rows_s = outs[0].n
    for o in outs[1:]:
        o.free()
    dsv, dtv = outs[0], pkg.voxel_filter_device(dt0, 0.2)
    rows_t = dtv.n
    t0 = time.perf_counter()
    index = pkg.TargetIndex(dtv, ctx=ctx)
    ctx.sync()
    build_ms = 1e3 * (time.perf_counter() - t0)
    dtv.free()
    pts_s = dsv.download()
    run_icp_steps(pkg, index, pts_s, 10, ctx)
    k = 50
    t0 = time.perf_counter()
    r = run_icp_steps(pkg, index, pts_s, k, ctx, sd=dsv.prepare(index))
    el = time.perf_counter() - t0
    # 16 B in + 8 B key out per input point + 24 B per output voxel
    algo = 24.0 * len(src) + 24.0 * rows_s
    res = {"leaf_m": 0.2, "points_in": [int(len(src)), int(len(tgt))], "rows_out": [int(rows_s), int(rows_t)],
           "voxel_filter_ms": wall_ms, "voxel_filter_device_ms": dev_ms, "index_build_ms": build_ms,
           "icp_ms_per_iter": 1e3 * el / k, "icp_device_ms_per_iter": r["device_ms"] / r["iters"],
           "correspondences_per_s": rows_s * k / el,
           "roofline": {"bound": "hbm", "kernel": "voxel filter, all launches of one call (HIP events on the library's stream)",
                        "achieved": algo / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes": algo},
           "roofline_icp": {"bound": "hbm", "kernel": "grid_pass_kernel", "achieved": 40.0 * rows_s / (r["device_ms"] / r["iters"] * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 40.0 * rows_s / (r["device_ms"] / r["iters"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "note": "40 B per correspondence (SURVEY 8d) over the device time of one iteration; cache resident, latency bound"}}
    index.free()
    ds0.free()
    dt0.free()
    return res


def global_init_leg(pkg, ctx, src, tgt, dev_id, streams, with_batch=True):
    """The stage in front of ICP in the reference's pair loop (Registration/main.py:196-203: prepare_dataset ->
    execute_global_registration; Open3D there, "parity unpinned" here): ms per step on the bench's 120k pair through the C ABI
    (device-resident where the ABI allows it), the whole stage from host arrays, and BASELINE configs[3]'s 256 pairs with the
    initialisation inside the rank's one native call (pcr_register_pairs)."""
    batch = importlib.import_module("point-cloud-process_amd.batch")
    glob = importlib.import_module("point-cloud-process_amd.global_registration")

    def best(fn, reps=7):
        fn()
        t_best = None
        for _ in range(reps):
            ctx.sync()
            t0 = time.perf_counter()
            r = fn()
            ctx.sync()
            el = 1e3 * (time.perf_counter() - t0)
            t_best = el if t_best is None else min(t_best, el)
        return t_best, r

    out = {"voxel_size_m": 2.0, "points": [int(len(src)), int(len(tgt))]}
    ds, dt = pkg.DeviceCloud.upload(src, ctx), pkg.DeviceCloud.upload(tgt, ctx)
    out["upload_ms"], _ = best(lambda: pkg.DeviceCloud.upload(src, ctx).free())
    down_ms, down = best(lambda: glob.voxel_down_sample_device(ds, 2.0, ctx=ctx))
    out["down_sample_ms"] = down_ms
    out["rows_after_down_sample"] = [int(down.n), int(glob.voxel_down_sample_device(dt, 2.0, ctx=ctx).n)]
    out["normals_ms"], nrm = best(lambda: pkg.estimate_normals_hybrid(down, 4.0, 30, ctx=ctx))
    out["fpfh_ms"], _ = best(lambda: pkg.compute_fpfh_feature(down, nrm, 10.0, 100, ctx=ctx))
    out["preprocess_ms"], ps = best(lambda: pkg.preprocess_point_cloud(ds, 2.0, ctx=ctx))
    pt = pkg.preprocess_point_cloud(dt, 2.0, ctx=ctx)
    fs, ft = ps[1].data, pt[1].data
    out["match_both_ways_ms"], _ = best(lambda: pkg.find_matchings(fs, ft, ctx=ctx))
    out["registration_ms"], res = best(lambda: pkg.execute_global_registration(ps[0], pt[0], ps[1], pt[1], 2.0, seed=1, ctx=ctx, evaluate=False))
    out["ransac"] = {k: res.info[k] for k in ("iterations", "n_valid", "best_iteration", "n_correspondences", "corr_fitness")}

    def whole():
        a = pkg.preprocess_point_cloud(pkg.DeviceCloud.upload(src, ctx), 2.0, ctx=ctx)
        b = pkg.preprocess_point_cloud(pkg.DeviceCloud.upload(tgt, ctx), 2.0, ctx=ctx)
        return pkg.execute_global_registration(a[0], b[0], a[1], b[1], 2.0, seed=1, ctx=ctx, evaluate=False)

    out["whole_stage_ms_per_pair_from_host_arrays"], _ = best(whole)
    out["note"] = ("wall ms, best of 7, one context; down_sample / normals / fpfh / match are the single-step entry points (small host copies "
                   "in and out included), preprocess = pcr_preprocess (down-sample + normals + FPFH, nothing leaves the device), registration = "
                   "pcr_global_registration (matching both ways + mutual filter + RANSAC loop on the device, one synchronisation)")
    ds.free()
    dt.free()
    if not with_batch:   # (--no-batch: the profiling runs, where eight launching threads under the tracer are not wanted)
        return out
    # configs[3] with the initialisation: every pair 20 degrees / 2.2 m apart, so that the stage has something to find
    P = BATCH_PAIRS
    pairs = []
    for i in range(P):
        s_, t_, _ = pkg.synthetic.perturbed_pair(BATCH_POINTS, seed=3000 + i, angle_deg=20.0 + (i % 7), t=(2.0 + 0.1 * (i % 5), -1.0, 0.05))
        pairs.append((s_, t_, None))
    kw = dict(mode="total", max_iter=30, r_thres=1e-3, t_thres=1e-3)
    rows = {}
    for tag, gi in (("icp_only", None), ("with_global_init", True)):
        batch.register_batch(pairs, device=dev_id, streams=streams, global_init=gi, **kw)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = batch.register_batch(pairs, device=dev_id, streams=streams, global_init=gi, **kw)
            runs.append(time.perf_counter() - t0)
        rows[tag] = {"seconds": min(runs), "pairs_per_s": P / min(runs), "runs_s": runs, "mean_iters": float(np.mean([x["iters"] for x in r]))}
    rows["global_init_ms_per_pair"] = 1e3 * (rows["with_global_init"]["seconds"] - rows["icp_only"]["seconds"]) / P
    rows["pairs"], rows["points_per_cloud"], rows["streams_per_gpu"] = P, BATCH_POINTS, streams
    rows["entry_point"] = ("register_batch(pairs, global_init=True) -> pcr_register_pairs: the initialisation fused for the whole share (every scan down-sampled by one sort, "
                           "normals / SPFH / FPFH one launch each, matching on the f64 matrix cores and the RANSAC loop for all pairs side by side), T0 into the fused ICP batch")
    out["batch256"] = rows
    return out


def config5_leg(pkg, ctx):
    """BASELINE configs[4]: 1 M-point synthetic scan (8 KITTI-shaped frames of one world): ISS keypoints with radius-NN
    covariance (Keypoint_detection_ISS/ISS.py:35-73) and a coarse-to-fine ICP that ends on 1 M x 1 M points."""
    syn = pkg.synthetic
    poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
    frames = [syn.kitti_like_scan(125000, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
    world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])
    radius = 0.09    # mean neighbour count ~ 38 (k ~ 40)
    cloud = pkg.DeviceCloud.upload(world, ctx)
    pkg.iss_keypoints(cloud, radius=radius, non_max_radius=radius, iss_count=20)   # warm
    ctx.sync()
    kp, lam, counts = pkg.iss_keypoints(cloud, radius=radius, non_max_radius=radius, iss_count=20, return_details=True)
    t0 = time.perf_counter()
    pkg.iss_keypoints(cloud, radius=radius, non_max_radius=radius, iss_count=20, return_details=True)
    iss_wall_details = time.perf_counter() - t0
    ctx.sync()
    ctx.timer_start()
    t0 = time.perf_counter()
    kp2 = pkg.iss_keypoints(cloud, radius=radius, non_max_radius=radius, iss_count=20)   # what ISS.py produces: the keypoint list
    iss_wall = time.perf_counter() - t0
    iss_dev_ms = ctx.timer_stop_ms()
    assert kp2 == kp
    cloud.free()
    T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
    src = (world - T_off[:3, 3]) @ T_off[:3, :3]
    src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
    c2f_runs = []
    pkg.coarse_to_fine_icp(src, world, leaves=(2.0, 0.5, 0.0), max_iteration=30)   # untimed: the first 1 M-point run of a process grows the arena and the staging blocks
    for _ in range(3):   # median of three (a single call can be hit by the platform's 30-50 ms stalls: DESIGN section 3.1.7); all three listed
        t0 = time.perf_counter()
        T, logs = pkg.coarse_to_fine_icp(src, world, leaves=(2.0, 0.5, 0.0), max_iteration=30)
        c2f_runs.append(time.perf_counter() - t0)
    c2f_wall = sorted(c2f_runs)[1]
    # steady-state iteration at 1 M x 1 M, inputs resident
    index = pkg.TargetIndex(pkg.DeviceCloud.upload(world, ctx), ctx=ctx)
    run_icp_steps(pkg, index, src, 2, ctx)   # untimed: the context's arena grows to the 1 M-point scratch here (hipMalloc), not in the timed call
    # Three 20-iteration calls from the 0.8 m / 3 degree offset; the MEDIAN is the headline and every call is listed with what the
    # library's own per-pass log says about it (pcr_icp_pass_log: the kernels' 100-MHz clock): the tile and drain launches of every
    # pass, the queries the tiles handed to the queue, and the call's phases on the host.  (Rounds 2-3 took the best of three because
    # single calls came out 2-4 x slower now and then; with the log a slow call says where it was slow: DESIGN section 3.1.7.)
    runs, calls = [], []
    for _ in range(3):
        x = run_icp_steps(pkg, index, src, 20, ctx)
        lg = ctx.pass_log()
        runs.append(x)
        calls.append({"device_ms": x["device_ms"], "wall_ms": 1e3 * x["wall_s"], "kernels_ms_by_their_own_clock": 1e-3 * (sum(lg["tile_us"]) + sum(lg["drain_us"])),
                      "tile_launch_us": {"first_pass": lg["tile_us"][0], "later_mean": float(np.mean(lg["tile_us"][1:]))},
                      "drain_launch_us": {"first_pass": lg["drain_us"][0], "passes_2_to_10_mean": float(np.mean(lg["drain_us"][1:10])), "passes_11_to_20_mean": float(np.mean(lg["drain_us"][10:]))},
                      "queue_items": {"first_pass": lg["items"][0], "pass_2": lg["items"][1], "pass_10": lg["items"][9], "pass_20": lg["items"][-1]},
                      "per_pass_us": [round(a + b, 1) for a, b in zip(lg["tile_us"], lg["drain_us"])],
                      "host_us": {k: round(float(v), 1) for k, v in lg["host_us"].items()}})
    r = sorted(runs, key=lambda x: x["device_ms"])[1]
    index.free()
    algo = 52.0 * len(world)   # 52 B per point (two passes over the records + counts + eigenvalues out)
    return {"points": int(len(world)), "iss_radius_m": radius, "mean_neighbours": float(counts.mean()), "keypoints": len(kp),
            "iss_ms": 1e3 * iss_wall, "iss_ms_with_per_point_eigenvalues_and_counts": 1e3 * iss_wall_details, "iss_device_ms": iss_dev_ms,
            "coarse_to_fine_icp_s": c2f_wall, "coarse_to_fine_runs_s": c2f_runs, "coarse_to_fine_levels": [{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in lg.items()} for lg in logs],
            "T_error_vs_truth_max_abs": float(np.abs(T - T_off).max()),
            "icp_1m_calls": calls,
            "icp_1m_calls_ms": [x["device_ms"] for x in runs],
            "icp_1m_headline": "median of the three calls (device time by the kernels' own clock: first kernel of the call .. end of its last pass)",
            "icp_1m_ms_per_iter": r["device_ms"] / r["iters"], "icp_1m_correspondences_per_s": len(world) / (r["device_ms"] / r["iters"] * 1e-3),
            "roofline": {"bound": "hbm", "kernel": "pcr_iss (keypoints only), all launches incl. the grid build (HIP events on the library's stream)",
                         "achieved": algo / (iss_dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / (iss_dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": algo},
            "roofline_icp": {"bound": "hbm", "kernel": "grid_pass_kernel + grid_drain_kernel (two launches per pass above 131 072 points)",
                             "achieved": 40.0 * len(world) / (r["device_ms"] / r["iters"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 40.0 * len(world) / (r["device_ms"] / r["iters"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--nn", default="grid", choices=["grid", "brute"])
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device of this rank (default LOCAL_RANK)")
    ap.add_argument("--cell", type=float, default=0.0, help="level-0 cell size of the grid index in metres (0 = automatic)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-brute", action="store_true", help="skip the brute-force MFMA leg")
    ap.add_argument("--in-flight", type=int, default=4, help="pairs in flight for the supplementary concurrent leg (0 = skip)")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch256 block (BASELINE configs[3])")
    ap.add_argument("--batch-streams", type=int, default=8, help="contexts (= native worker threads, one sub-batch in flight each) per GPU in the batch256 block")
    ap.add_argument("--no-configs", action="store_true", help="skip the config3 / config5 blocks (BASELINE configs[2] and configs[4])")
    a = ap.parse_args()

    if a.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        dev_id = a.device if a.device >= 0 else local_rank
        torch.cuda.set_device(dev_id)
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend=a.dist_backend)
    dev_id = a.device if a.device >= 0 else local_rank
    tdev = f"cuda:{dev_id}" if (world == 1 or a.dist_backend == "nccl") else "cpu"
    pkg = importlib.import_module("point-cloud-process_amd")
    ctx = pkg.Context(dev_id)
    syn = pkg.synthetic

    # every rank registers its own pair (different seed); same size => weak scaling
    src, tgt, T_true = syn.perturbed_pair(a.points, seed=rank)
    index = pkg.TargetIndex(pkg.DeviceCloud.upload(tgt, ctx), kind=a.nn, cell=a.cell, ctx=ctx)
    ctx.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    if world > 1 and a.device >= 0:   # rehearsal: several ranks on ONE GPU (tell the library, see pcr_ctx_set_shared)
        ctx.set_shared(True)
    # the set-up leg (uploads, index builds, one compat registration: what the reference's own usage consists of) is measured FIRST, on every
    # rank: it is a result of its own, and the device's clocks are up when the timed region starts -- the first call of a process on a
    # device that has idled measured 2 us per iteration more than the following ones (scripts/cold_clock.py), W = 5 warm-up steps are 0.2 ms
    setup_block = setup_leg(pkg, ctx, src, tgt, a.cell) if a.nn == "grid" else None
    run_icp_steps(pkg, index, src, a.warmup, ctx)  # untimed warm-up
    if dist is not None:  # warm the collective too (communicator and kernel set-up are one-off costs)
        wrec = torch.zeros(18, dtype=torch.float64, device=tdev)
        dist.all_gather([torch.zeros_like(wrec) for _ in range(world)], wrec)
    # set-up of the timed run, reported separately: upload of the source (PCIe) + Morton lay-out for this index
    ctx.sync()
    ts = time.perf_counter()
    sd_timed = pkg.DeviceCloud.upload(src, ctx).prepare(index)
    ctx.sync()
    setup_ms = 1e3 * (time.perf_counter() - ts)
    barrier()
    t0 = time.perf_counter()
    r = run_icp_steps(pkg, index, src, a.steps, ctx, sd=sd_timed)
    # result gather: 16 doubles + iters + n_assoc per rank (RCCL all_gather), inside the timed region
    if dist is not None:
        rec = torch.zeros(18, dtype=torch.float64, device=tdev)
        rec[:16] = torch.from_numpy(r["T_total"].reshape(16))
        rec[16] = r["iters"]
        rec[17] = r["n_assoc"]
        allrec = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
    barrier()
    elapsed = time.perf_counter() - t0
    sd_timed.free()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    batch256 = None
    if not a.no_batch and a.nn == "grid":
        batch256 = batch_leg(pkg, torch, dist, rank, world, dev_id, tdev, a.batch_streams)

    if rank == 0:
        n_src = a.points
        corr_total = float(n_src) * a.steps * world
        value = corr_total / elapsed
        # the pass kernel's average launch period: HIP events on the library's stream around a repeat of the timed call (same
        # arguments, untimed) / its launches -- ONE launch per iteration, back to back inside the device-resident loop, so this is
        # the kernel's duration plus the gap to the next launch and the call's init kernel: an upper bound that rocprofv3's
        # per-kernel average (profiles/r04_grid_kernel_stats.csv) must sit just below.  (The per-launch events of the profiled run
        # below bracket one launch on an IDLE stream each -- the state is read back after every pass there -- and so include the
        # ~8 us from the event to the kernel's start: kept as `kernel_us`, not used for the roofline.)
        sd_ev = pkg.DeviceCloud.upload(src, ctx).prepare(index)
        ctx.sync()
        ctx.timer_start()
        r_ev = run_icp_steps(pkg, index, src, a.steps, ctx, sd=sd_ev)
        loop_event_us = ctx.timer_stop_ms() * 1e3 / max(r_ev["iters"], 1)
        sd_ev.free()
        # per-kernel profile (HIP events on the library's stream) in a separate, untimed run
        ctx.profile(True)
        rp = run_icp_steps(pkg, index, src, min(a.steps, 50), ctx)
        prof_ms, passes = ctx.profile_read()
        ctx.profile(False)
        kern = {}
        if a.nn == "grid":
            names = ["grid_pass_kernel", "-", "-", "-"]   # ONE launch per iteration: tiles, hard queue, moments, Procrustes, convergence test
        else:
            names = ["brute_nn_kernel", "brute_merge_kernel+brute_exact_kernel", "brute_final_kernel", "brute_reduce_partials_kernel"]
        for nm, ms in zip(names, prof_ms):
            if nm != "-":
                kern[nm] = ms / max(passes, 1) * 1e3  # us per launch
        dom = max(kern, key=kern.get)
        dom_s = kern[dom] * 1e-6
        if a.nn == "grid":
            # the TIMED call's own device time / its launches: the kernels stamp their 100 MHz clock (s_memrealtime) -- first kernel of the
            # call .. end of its last pass; HIP events recorded by the host around the same call also contain the host's launch latency
            # in front of the first kernel (`launch_us_hip_events`).  rocprofv3's per-kernel average sits just below both.
            dom_s = r["device_ms"] / max(r["iters"], 1) * 1e-3
            algo_bytes = GRID_BYTES_PER_CORR * n_src
            traffic = None  # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
            try:
                pmc = json.load(open(profile_file("pmc_traffic.json")))
                traffic = pmc["kernels"][dom].get("hbm_bytes") if a.points == N_POINTS else None
            except Exception:
                traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": algo_bytes / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo_bytes / dom_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_correspondence": GRID_BYTES_PER_CORR,
                        "launch_us": dom_s * 1e6, "launch_us_hip_events": loop_event_us,
                        "launch_us_source": f"the timed {a.steps}-iteration call's device time by the kernels' own clock / {a.steps} launches (incl. launch gaps, the call's init "
                                            "kernel and its unseeded first pass); launch_us_hip_events: HIP events on the library's stream around a repeat of that call "
                                            "(adds the host's launch latency in front of the first kernel); rocprofv3 --kernel-trace average of the same kernel: "
                                            "profiles/r04_grid_kernel_stats.csv",
                        "frac_f64_layout": GRID_LAYOUT_BYTES_PER_CORR * n_src / dom_s / 1e9 / HBM_PEAK_GBS,
                        "traffic_source": os.path.relpath(profile_file("pmc_traffic.json"), ROOT) + " (separate rocprofv3 --pmc passes of the same command), not measured in this run",
                        "note": "working set (2 x 3.8 MB) is L2/MALL resident; this kernel is VALU-issue and latency bound, not HBM bound: see roofline_valu"}
        else:
            flops = 8.0 * n_src * a.points
            roofline = {"bound": "mfma", "kernel": dom, "achieved": flops / dom_s / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dom_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None}
        line = {
            "metric": "correspondence-pairs/sec + ms/ICP-iter, 120k-pt KITTI pair",
            "value": value,
            "unit": "correspondences/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"point-to-point ICP iteration, {n_src} x {a.points} KITTI-shaped synthetic scan pair per GPU "
                                   f"(BASELINE configs[1]); exact 1-NN via {a.nn}", "nn": a.nn, "pairs_per_gpu": 1,
                       "points_src": n_src, "points_tgt": a.points, "max_d2": MAX_D2, "cell_m": index.cell},
            "ms_per_icp_iter": 1e3 * elapsed / a.steps,
            "device_ms_per_iter": r["device_ms"] / r["iters"],
            "pass_kernels_ms_per_iter": rp["nn_kernel_ms"] / max(rp["nn_launches"], 1),  # from the profiled (untimed) run
            "kernel_us": kern,
            "setup_ms_not_timed": setup_ms,
            "setup": setup_block,
            "n_assoc_last": int(r["n_assoc"]),
            "roofline": roofline,
        }
        if a.nn == "grid" and "grid_pass_kernel" in kern:
            # what actually bounds the pass: VALU issue.  Wave instructions per launch from the committed SQ counter passes (rocprofv3
            # cannot run inside this process), over this run's kernel duration, against one wave64 VALU instruction per SIMD every 2 cycles.
            try:
                sqc = json.load(open(profile_file("sq_counters.json")))["kernels"]["grid_pass_kernel"]
                n_valu = sqc["SQ_INSTS_VALU"] if a.points == N_POINTS else None
                fpairs, mean_p = filter_pairs_per_pass(pkg, dev_id, src, tgt, a.cell)
                pass_s = kern["grid_pass_kernel"] * 1e-6
                line["roofline_valu"] = {"bound": "valu-issue", "kernel": "grid_pass_kernel",
                                         "achieved": None if n_valu is None else n_valu / pass_s / 1e9, "peak": VALU_ISSUE_PEAK / 1e9,
                                         "unit": "G wave-instructions/s", "frac": None if n_valu is None else n_valu / pass_s / VALU_ISSUE_PEAK,
                                         "valu_instructions_per_launch": n_valu,
                                         "source": os.path.relpath(profile_file("sq_counters.json"), ROOT) + " (SQ_INSTS_VALU, separate rocprofv3 --pmc pass of the same pair)",
                                         "filter_pairs_per_launch": fpairs, "mean_candidates_per_query": mean_p,
                                         "note": "whole launch incl. the tail in which most waves wait for the slowest tiles; during the tile stage "
                                                 "(first ~12 us) the SIMDs issue VALU back to back (DESIGN.md 3.1); the filter itself runs on the "
                                                 "matrix cores (v_mfma_f32_32x32x2_f32)"}
            except Exception as e:  # diagnostics only
                line["roofline_valu"] = {"error": repr(e)}
        if batch256 is not None:
            line["batch256"] = batch256
        if a.nn == "brute":
            line["candidate_pairs_per_s"] = float(n_src) * a.points * a.steps * world / elapsed
        # brute-force MFMA leg (candidate pairs/s + MFMA roofline), N = 1 only
        if world == 1 and a.nn == "grid" and not a.no_brute:
            ib = pkg.TargetIndex(pkg.DeviceCloud.upload(tgt, ctx), kind="brute", ctx=ctx)
            run_icp_steps(pkg, ib, src, 2, ctx)
            ctx.profile(True)
            kb = 10
            t1 = time.perf_counter()
            rb = run_icp_steps(pkg, ib, src, kb, ctx)
            eb = time.perf_counter() - t1
            pm, pp = ctx.profile_read()
            ctx.profile(False)
            sweep_s = pm[0] / max(pp, 1) * 1e-3
            flops = 8.0 * n_src * a.points
            line["brute_mfma"] = {
                "ms_per_icp_iter": 1e3 * eb / kb,
                "correspondences_per_s": n_src * kb / eb,
                "candidate_pairs_per_s": float(n_src) * a.points / sweep_s,
                "roofline": {"bound": "mfma", "kernel": "brute_nn_kernel", "achieved": flops / sweep_s / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": flops / sweep_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                             "dtype": "f64 (v_mfma_f64_16x16x4_f64)"},
            }
            ib.free()
        if world == 1 and a.nn == "grid" and not a.no_configs and a.points == N_POINTS:
            line["config3"] = config3_leg(pkg, ctx, src, tgt)
            line["config5"] = config5_leg(pkg, ctx)
            line["global_init"] = global_init_leg(pkg, ctx, src, tgt, dev_id, a.batch_streams, with_batch=not a.no_batch)
        if world == 1 and a.nn == "grid" and a.in_flight > 1:
            line["concurrent_pairs"] = concurrent_leg(pkg, dev_id, src, tgt, a.in_flight, min(a.steps, 100))
        if world == 1 and not a.no_cpu:
            line["cpu_baseline"] = cpu_baseline(src, tgt)
            if a.nn == "grid" and a.points <= 200_000:
                line["parity"] = parity_gates(pkg, index, src, tgt)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
