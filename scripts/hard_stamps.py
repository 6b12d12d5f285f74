"""Per-item stamps of the exact descent (queue items of the one-launch ICP pass / the stand-alone hard stage).
For the one-launch pass build scripts/build_variant.sh pdiag "-DPCR_PASS_DIAG=1" and run with
PCR_LIB_PATH=scripts/bin/libpcr_pdiag.so: the product build compiles the pass kernel's stamps out."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
src, tgt, Tt = pcp.synthetic.perturbed_pair(120000, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid")
sd = pcp.DeviceCloud.upload(src).prepare(index)
IT = int(os.environ.get("ITERS", 4))
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1, t_thres=-1, min_iter=IT)
hb = np.zeros((1 << 17) + 60000 * 4, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, hb.ctypes.data_as(C.POINTER(C.c_uint64)), hb.size))
h = hb[(1 << 17):].reshape(-1, 4)
h = h[h[:, 0] > 0]
hc = h[:, 0].astype(np.float64); steps = (h[:, 1] >> np.uint64(32)).astype(int); pts = h[:, 2].astype(int)
known = (h[:, 3] & np.uint64(1)).astype(bool); sl = ((h[:, 3] >> np.uint64(8)).astype(int) - 1)
print("hard items", len(h), "cycles pct 50/90/99/max", np.percentile(hc, [50, 90, 99, 100]))
print(" steps mean %.1f max %d | pts scanned mean %.0f p99 %.0f max %d" % (steps.mean(), steps.max(), pts.mean(), np.percentile(pts, 99), pts.max()))
print(" with candidate: %d (median cycles %.0f, p99 %.0f) | without: %d (median %.0f, p99 %.0f)" % (known.sum(), np.median(hc[known]), np.percentile(hc[known], 99), (~known).sum(), np.median(hc[~known]) if (~known).any() else 0, np.percentile(hc[~known], 99) if (~known).any() else 0))
print(" start level histogram", np.bincount(sl + 1))
o = np.argsort(-hc)[:10]
print(" slowest:", [(int(hc[i]), int(steps[i]), int(pts[i]), bool(known[i]), int(sl[i])) for i in o])
for lo, hi in ((0, 3), (3, 5), (5, 8), (8, 12), (12, 100)):
    m = (steps >= lo) & (steps < hi)
    if m.any():
        print(" steps in [%2d, %3d): %5d items, cycles median %.0f p90 %.0f, points median %.0f" % (lo, hi, m.sum(), np.median(hc[m]), np.percentile(hc[m], 90), np.median(pts[m])))
