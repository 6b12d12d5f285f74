"""Per-block (4 wave tiles) cycles / filter pairs / unresolved counts and per-tile phase cycles of the wave tiles of the one-launch
ICP pass.  Needs a diagnostic build: scripts/build_variant.sh diag "-DPCR_WT_DIAG -DPCR_PASS_DIAG=1", then
PCR_LIB_PATH=scripts/bin/libpcr_diag.so ITERS=20 python3 scripts/wt_stamps.py (the product build compiles the stamps out)."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
NPTS = int(os.environ.get('N', 120000)); IT = int(os.environ.get('ITERS', 3))
src, tgt, Tt = pcp.synthetic.perturbed_pair(NPTS, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid", cell=float(os.environ.get("CELL", 0)))
sd = pcp.DeviceCloud.upload(src).prepare(index)
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1, t_thres=-1, min_iter=IT)
nb = (NPTS + 63) // 64
buf = np.zeros((1 << 16) + nb * 32, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
b = buf[: nb * 4].reshape(nb, 4).astype(np.float64)
ph = buf[(1 << 16):].reshape(nb * 4, 8).astype(np.float64)
staged = ph[:, 7].copy(); ph[:, 7] = 0
live = ph.sum(axis=1) > 0; ph = ph[live]; staged = staged[live]
print("cell", index.cell, "blocks", nb)
print("block cycles pct 50/90/99/max", np.percentile(b[:, 0], [50, 90, 99, 100]), "sum", b[:, 0].sum())
print("pairs/block pct 50/90/99/max", np.percentile(b[:, 1], [50, 90, 99, 100]), "total", b[:, 1].sum(), "per query", b[:, 1].sum() / NPTS)
print("open after the last pass by reason [clamped/none, level, too many points, ambiguous, ball out of box]:", buf[(1 << 15):(1 << 15) + 5])
print("unresolved total", b[:, 3].sum(), "max/block", b[:, 3].max())
print("pass stats (slot 2):", np.percentile(b[:, 2], [50, 90, 99, 100]))
names = ["load+xform", "cube+level", "directory", "prefix", "staging", "filter", "merge+verify", "append"]
print("phase cycles of every wave tile of the LAST pass (sum over passes/rounds; s_memtime ticks = shader cycles): median / p90 / mean")
for i, nm in enumerate(names):
    print("   %-14s %8.0f %8.0f %8.0f" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].mean()))
print("   total mean %.0f" % ph.sum(axis=1).mean())
print("by points staged (class: waves | median ticks per phase | total):")
for lo, hi in ((0, 65), (65, 129), (129, 193), (193, 385), (385, 577), (577, 769), (769, 100000)):
    sel = (staged >= lo) & (staged < hi)
    if sel.any():
        print("   [%4d, %6d): %5d | %s | %.0f" % (lo, hi, sel.sum(), " ".join("%s %.0f" % (nm.split("+")[0][:6], np.median(ph[sel, i])) for i, nm in enumerate(names[:7])), np.median(ph[sel].sum(axis=1))))
tot = ph.sum(axis=1)
print("slowest 16 tiles (total | staged | phases):")
for k in np.argsort(-tot)[:16]:
    print("   %6.0f | %4d | %s" % (tot[k], staged[k], " ".join("%s %.0f" % (nm.split("+")[0][:6], ph[k, i]) for i, nm in enumerate(names[:7]))))
