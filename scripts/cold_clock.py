"""Does the first timed call of a process run on a GPU that has not clocked up yet?  Wall of consecutive 20-iteration calls of the bench
pair from process start (each on a fresh source; 5 warm-up iterations first, as bench.py does with the driver's arguments)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
src, tgt, _ = pcp.synthetic.perturbed_pair(120000, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid")
out = []
for rep in range(12):
    sd = pcp.DeviceCloud.upload(src).prepare(index)
    if rep == 0:
        pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=5, r_thres=-1, t_thres=-1, min_iter=5)
    ctx.sync()
    t0 = time.perf_counter()
    r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=20, r_thres=-1, t_thres=-1, min_iter=20)
    ctx.sync()
    out.append((1e6 * (time.perf_counter() - t0) / 20, r["device_ms"] * 1e3 / 20))
    sd.free()
    if rep == 5:
        time.sleep(0.5)   # idle half a second: do the clocks fall back?
print("us per iteration, wall / device, consecutive calls (0.5 s idle before call 6):", " ".join("%.1f/%.1f" % v for v in out))
