"""Per-block (4 wave tiles) cycles / filter pairs / unresolved counts of the wave-tile kernel (PCR_DEBUG_STAMPS=1)."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
NPTS = int(os.environ.get('N', 120000))
src, tgt, Tt = pcp.synthetic.perturbed_pair(NPTS, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid", cell=float(os.environ.get("CELL", 0)))
sd = pcp.DeviceCloud.upload(src).prepare(index)
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=3, r_thres=-1, t_thres=-1, min_iter=3)
nb = (NPTS + 63) // 64
buf = np.zeros((1 << 16) + nb * 8, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
b = buf[: nb * 4].reshape(nb, 4).astype(np.float64)
ph = buf[(1 << 16):].reshape(nb, 8).astype(np.float64)
print("cell", index.cell, "blocks", nb)
print("block cycles pct 50/90/99/max", np.percentile(b[:, 0], [50, 90, 99, 100]), "sum", b[:, 0].sum())
print("pairs/block pct 50/90/99/max", np.percentile(b[:, 1], [50, 90, 99, 100]), "total", b[:, 1].sum(), "per query", b[:, 1].sum() / NPTS)
print("open after the last pass by reason [clamped/none, level, too many points, ambiguous, ball out of box]:", buf[(1 << 15):(1 << 15) + 5])
print("unresolved total", b[:, 3].sum(), "max/block", b[:, 3].max())
print("pass stats (slot 2):", np.percentile(b[:, 2], [50, 90, 99, 100]))
names = ["load+xform", "cube+level", "directory", "prefix", "staging", "filter", "merge+verify", "append"]
print("wave-0 phase cycles (sum over passes/rounds): median / p90 / mean")
for i, nm in enumerate(names):
    print("   %-14s %8.0f %8.0f %8.0f" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].mean()))
print("   total mean %.0f" % ph.sum(axis=1).mean())
