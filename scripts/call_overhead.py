"""Fixed cost of one pcr_icp call around its passes: wall of the Python call, wall of the C call alone, the library's own
HIP-event time of the loop (device_ms), for ITERS passes of the bench pair (best / median of REPS calls)."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
IT = int(os.environ.get("ITERS", 20)); REPS = int(os.environ.get("REPS", 15))
src, tgt, _ = pcp.synthetic.perturbed_pair(120000, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid")
py, cc, dev = [], [], []
for rep in range(REPS + 2):
    sd = pcp.DeviceCloud.upload(src).prepare(index)
    ctx.sync()
    t0 = time.perf_counter()
    r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1, t_thres=-1, min_iter=IT)
    t1 = time.perf_counter()
    sd.free()
    # the C call alone
    sd = pcp.DeviceCloud.upload(src).prepare(index)
    ctx.sync()
    p = L.IcpParams(); L.lib().pcr_icp_default_params(C.byref(p))
    p.max_iter = IT; p.min_iter = IT; p.r_thres = -1.0; p.t_thres = -1.0; p.max_d2 = 5.0; p.mode = L.PCR_ICP_TOTAL
    res = L.IcpResult(); T0 = np.eye(4).reshape(16).copy()
    t2 = time.perf_counter()
    L.lib().pcr_icp(index.ctx.handle, sd.handle, index.handle, C.byref(p), L.dptr(T0), C.byref(res))
    t3 = time.perf_counter()
    sd.free()
    if rep >= 2:
        py.append(1e6 * (t1 - t0)); cc.append(1e6 * (t3 - t2)); dev.append(1e3 * res.device_ms)
f = lambda v: "best %.1f median %.1f" % (min(v), float(np.median(v)))
print("%d passes: Python call %s us | C call %s us | device (HIP events over the loop) %s us" % (IT, f(py), f(cc), f(dev)))
print("   per pass: Python %.2f, C %.2f, device %.2f us; fixed cost of a call over its device time: %.0f us (C), %.0f us (Python)" % (
    min(py) / IT, min(cc) / IT, min(dev) / IT, min(cc) - min(dev), min(py) - min(dev)))
lg = ctx.pass_log()
print("   host phases of the last C call (us):", {k: round(float(v), 1) for k, v in lg["host_us"].items()})
print("   passes by the device clock (us):", " ".join("%.1f" % v for v in lg["tile_us"][:IT]))
