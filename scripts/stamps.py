"""Distribution of per-tile cycles / staged points (PCR_DEBUG_STAMPS=1)."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
NPTS = int(os.environ.get('N', 120000))
src, tgt, Tt = pcp.synthetic.perturbed_pair(NPTS, seed=0)
cell = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid", cell=cell)
sd = pcp.DeviceCloud.upload(src).prepare(index)
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=3, r_thres=-1, t_thres=-1, min_iter=3)
nb = (NPTS + 63) // 64
buf = np.zeros((1 << 16) + nb * 8, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
ph = buf[(1 << 16):].reshape(nb, 8)[:, :5].astype(np.float64)
b = buf[: nb * 4].reshape(nb, 4)
cyc = b[:, 0].astype(np.float64); P = b[:, 1].astype(np.float64); lvl = (b[:, 2] >> np.uint64(32)).astype(int) - 1; ncell = (b[:, 2] & np.uint64(0xffffffff)).astype(int)
c0 = b[:, 3].astype(int); c1 = c0 * 0; c2 = c0 * 0
print("cell", index.cell, "blocks", nb)
print("cycles(100MHz ticks?) pct 50/90/99/max", np.percentile(cyc, [50, 90, 99, 100]), "sum", cyc.sum())
print("P pct 50/90/99/max", np.percentile(P, [50, 90, 99, 100]), "mean", P.mean())
print("levels", np.bincount(lvl + 1), "ncell mean", ncell.mean())
print("lists ring1/ring2/hard totals", c0.sum(), c1.sum(), c2.sum(), "max per block", c0.max(), c1.max(), c2.max())
o = np.argsort(-cyc)[:8]
print("slowest blocks:", [(int(cyc[i]), int(P[i]), int(lvl[i]), int(ncell[i])) for i in o])
print("corr cycles~P", np.corrcoef(cyc, P)[0, 1])
print("phase stamps (cycles since start) median: load+box %.0f | lookups %.0f | scan %.0f | staged(last round) %.0f | eval done %.0f | end %.0f" % (*np.median(ph, axis=0), np.median(cyc)))
print("phase stamps p90:", np.percentile(ph, 90, axis=0))
one = P <= 512
d = np.diff(np.c_[np.zeros(nb), ph, cyc], axis=1)
names = ["load+box", "lookups", "scan", "staging", "eval", "merge+lists"]
print("single-round tiles (P<=512): n=%d, median P=%.0f" % (one.sum(), np.median(P[one])))
for k, nm in enumerate(names):
    print("   %-12s median %7.0f  p90 %7.0f cycles" % (nm, np.median(d[one, k]), np.percentile(d[one, k], 90)))
# hard-stage items
hb = np.zeros((1 << 17) + 60000 * 4, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, hb.ctypes.data_as(C.POINTER(C.c_uint64)), hb.size))
h = hb[(1 << 17):].reshape(-1, 4)
h = h[h[:, 0] > 0]
hc = h[:, 0].astype(np.float64); sc_ = (h[:, 1] >> np.uint64(32)).astype(int); ex = (h[:, 1] & np.uint64(0xffffffff)).astype(int); pts = h[:, 2].astype(int)
known = (h[:, 3] & np.uint64(1)).astype(bool); sl = ((h[:, 3] >> np.uint64(8)).astype(int) - 1)
print("hard items", len(h), "cycles pct 50/90/99/max", np.percentile(hc, [50, 90, 99, 100]), "sum", hc.sum())
print("  scans/item mean %.1f max %d | expands/item mean %.1f max %d | pts scanned/item mean %.0f max %d" % (sc_.mean(), sc_.max(), ex.mean(), ex.max(), pts.mean(), pts.max()))
print("  with prior bound: %d (median cycles %.0f) | without: %d (median cycles %.0f)" % (known.sum(), np.median(hc[known]), (~known).sum(), np.median(hc[~known]) if (~known).any() else 0))
print("  start level histogram", np.bincount(sl + 1))
