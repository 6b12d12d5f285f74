"""Timing of the global-initialisation stage (main.py:196-203) on a 120k KITTI-shaped pair (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr

src, tgt, T_true = pcr.synthetic.perturbed_pair(120000, seed=4, angle_deg=35.0, t=(4.0, -2.0, 0.1))
def stage(name, fn, reps=3):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    print(f"{name}: {(time.perf_counter() - t) / reps * 1e3:.2f} ms", flush=True)
    return out
sd = stage("voxel_down_sample(2.0)", lambda: pcr.voxel_down_sample(src, 2.0))
td = pcr.voxel_down_sample(tgt, 2.0)
print("down-sampled sizes", len(sd), len(td))
ns = stage("normals hybrid r=4 nn=30", lambda: pcr.estimate_normals_hybrid(sd, 4.0, 30))
nt = pcr.estimate_normals_hybrid(td, 4.0, 30)
fs = stage("fpfh r=10 nn=100", lambda: pcr.compute_fpfh_feature(sd, ns, 10.0, 100))
ft = pcr.compute_fpfh_feature(td, nt, 10.0, 100)
stage("feature match both ways", lambda: pcr.find_matchings(fs.data, ft.data))
res = stage("ransac (execute_global_registration)", lambda: pcr.execute_global_registration(pcr.PointCloud(sd), pcr.PointCloud(td), fs, ft, 2.0, seed=1))
print(res, res.info)
tot = stage("whole stage: 2x preprocess + ransac", lambda: pcr.execute_global_registration(*sum((list(pcr.preprocess_point_cloud(pcr.PointCloud(c), 2.0)) for c in (src, tgt)), [])[::2], *[pcr.preprocess_point_cloud(pcr.PointCloud(c), 2.0)[1] for c in (src, tgt)], 2.0, seed=1), reps=1)
