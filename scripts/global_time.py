"""Timing of the global-initialisation stage (Registration/main.py:196-203) on a 120k KITTI-shaped pair and as a batch of pairs
(run on the GPU box).  STAGES=0 skips the per-stage calls (for rocprofv3 runs of the native path alone)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
import importlib
batch = importlib.import_module("point-cloud-process_amd.batch")

N = int(os.environ.get("N", 120000))
src, tgt, T_true = pcr.synthetic.perturbed_pair(N, seed=4, angle_deg=35.0, t=(4.0, -2.0, 0.1))
ctx = pcr.default_context()
def stage(name, fn, reps=5):
    fn(); ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    ctx.sync()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"{name}: {ms:.3f} ms", flush=True)
    return out
if os.environ.get("STAGES", "1") != "0":
    s64, t64 = src.astype(np.float64), tgt.astype(np.float64)
    sd = stage("voxel_down_sample(2.0) [host arrays in and out]", lambda: pcr.voxel_down_sample(s64, 2.0))
    td = pcr.voxel_down_sample(t64, 2.0)
    print("down-sampled sizes", len(sd), len(td))
    ns = stage("normals hybrid r=4 nn=30 [host in/out]", lambda: pcr.estimate_normals_hybrid(sd, 4.0, 30))
    nt = pcr.estimate_normals_hybrid(td, 4.0, 30)
    fs = stage("fpfh r=10 nn=100 [host in/out]", lambda: pcr.compute_fpfh_feature(sd, ns, 10.0, 100))
    ft = pcr.compute_fpfh_feature(td, nt, 10.0, 100)
    stage("feature match both ways [host in/out]", lambda: pcr.find_matchings(fs.data, ft.data))
dsrc, dtgt = pcr.DeviceCloud.upload(src, ctx), pcr.DeviceCloud.upload(tgt, ctx)
ps = stage("preprocess_point_cloud (native pcr_preprocess, cloud resident)", lambda: pcr.preprocess_point_cloud(dsrc, 2.0))
pt = pcr.preprocess_point_cloud(dtgt, 2.0)
res = stage("execute_global_registration (native, prepared clouds, no final evaluation)", lambda: pcr.execute_global_registration(ps[0], pt[0], ps[1], pt[1], 2.0, seed=1, evaluate=False))
print(res, res.info)
res = stage("execute_global_registration (native + final whole-cloud evaluation)", lambda: pcr.execute_global_registration(ps[0], pt[0], ps[1], pt[1], 2.0, seed=1))
print(res, res.info)
def whole():
    a = pcr.preprocess_point_cloud(pcr.DeviceCloud.upload(src, ctx), 2.0)
    b = pcr.preprocess_point_cloud(pcr.DeviceCloud.upload(tgt, ctx), 2.0)
    return pcr.execute_global_registration(a[0], b[0], a[1], b[1], 2.0, seed=1, evaluate=False)
stage("whole stage from host arrays: 2 x (upload + preprocess) + registration", whole)
R = res.transformation[:3, :3] @ T_true[:3, :3].T
print("rotation error deg %.3f, translation error %.3f m" % (np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))), np.linalg.norm(res.transformation[:3, 3] - T_true[:3, 3])))
# the pair loop with initialisation: P pairs of 20 000-point scans (BASELINE configs[3] shape) through ONE native call
P = int(os.environ.get("PAIRS", 64))
pairs = []
for i in range(P):
    s, t, _ = pcr.synthetic.perturbed_pair(20000, seed=3000 + i, angle_deg=20.0 + (i % 7), t=(2.0 + 0.1 * (i % 5), -1.0, 0.05))
    pairs.append((s, t, None))
for gi, tag in ((None, "ICP only"), (True, "global init + ICP")):
    batch.native_register_share(pairs, device=0, streams=8, global_init=gi)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = batch.native_register_share(pairs, device=0, streams=8, global_init=gi)
        best = min(best, time.perf_counter() - t0)
    print(f"{P} pairs x 20 000 points, {tag}: {best * 1e3:.2f} ms = {P / best:.0f} pairs/s ({best / P * 1e3:.3f} ms per pair)", flush=True)
