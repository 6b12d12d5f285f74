"""Timing of pcr_normals / pcr_pca on the bench clouds (run on the GPU box)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr

for name, pts in (("object10k", pcr.synthetic.object_cloud(10000, seed=1)), ("scan120k", pcr.synthetic.kitti_like_scan(120000, seed=1))):
    pts = pts.astype(np.float64)
    dc = pcr.DeviceCloud.upload(pts)
    for k in (5, 8, 16):
        pcr.estimate_normals(dc, k)
        t = time.perf_counter()
        for _ in range(5):
            pcr.estimate_normals(dc, k)
        dt = (time.perf_counter() - t) / 5
        print(f"{name} k={k}: {dt*1e3:.2f} ms  ({len(pts)/dt/1e6:.1f} Mpts/s)", flush=True)
    t = time.perf_counter()
    for _ in range(20):
        pcr.PCA(dc)
    print(f"{name} PCA: {(time.perf_counter()-t)/20*1e3:.3f} ms", flush=True)
