"""Set-up path of one registration (what Registration/main.py:105 `KDTreeFlann(target)` and the source's lay-out cost here):
wall ms of target upload, index build, source upload, prepare, at N points.  usage: python scripts/setup_time.py [N]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
src, tgt, _ = pcp.synthetic.perturbed_pair(N, seed=0)
ctx = pcp.default_context()
def t(fn, reps=20):
    fn(); ctx.sync()
    best = 1e9
    for _ in range(reps):
        ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync(); best = min(best, time.perf_counter() - t0)
        if hasattr(r, "free"): r.free()
    return best * 1e3
up = t(lambda: pcp.DeviceCloud.upload(tgt, ctx))
dt = pcp.DeviceCloud.upload(tgt, ctx)
build = t(lambda: pcp.TargetIndex(dt, ctx=ctx))
index = pcp.TargetIndex(dt, ctx=ctx)
def prep():
    return pcp.DeviceCloud.upload(src, ctx).prepare(index)
both = t(prep)
one = t(lambda: pcp.icp_point2point(pcp.PointCloud(src), index, np.eye(4)), 10)
full = t(lambda: pcp.icp_point2point(pcp.PointCloud(src), tgt, np.eye(4)), 10)
print(f"N={N}: upload {up:.3f} ms, index_build {build:.3f} ms, source upload+prepare {both:.3f} ms, icp_point2point on a built index {one:.3f} ms, "
      f"icp_point2point from host arrays (upload + build + ICP + download) {full:.3f} ms")
