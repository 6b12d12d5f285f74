"""Per-wave timeline of the one-kernel ICP pass (PCR_DEBUG_STAMPS=1): when a wave's tile ended, when its moments were in,
when it left the work queue, how many items it served, polls and lost compare-and-swaps; 100-MHz real-time stamps.
Needs scripts/build_variant.sh pdiag "-DPCR_PASS_DIAG=1" and PCR_LIB_PATH=scripts/bin/libpcr_pdiag.so: the product build compiles the stamps out."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
L = pcp._lib
N = int(os.environ.get("N", 120000))
src, tgt, Tt = pcp.synthetic.perturbed_pair(N, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind="grid", cell=float(os.environ.get("CELL", 0)))
sd = pcp.DeviceCloud.upload(src).prepare(index)
it = int(os.environ.get("ITERS", 4))
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=it, r_thres=-1, t_thres=-1, min_iter=it)
nw = (N + 127) // 128 * 4
buf = np.zeros((1 << 19) + nw * 8, dtype=np.uint64)
L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
w = buf[(1 << 19):].reshape(nw, 8).astype(np.int64)
t0 = w[:, 0].min()
us = lambda c: (c - t0) / 100.0
pc = lambda v: np.percentile(v, [50, 90, 99, 100]).round(1)
print("waves", nw, "(last pass of", it, ")")
print("start      us pct 50/90/99/max", pc(us(w[:, 0])))
print("tile end   us", pc(us(w[:, 1])), " tile duration", pc((w[:, 1] - w[:, 0]) / 100.0))
print("moments in us", pc(us(w[:, 2])), " duration", pc((w[:, 2] - w[:, 1]) / 100.0))
print("queue exit us", pc(us(w[:, 3])), " duration", pc((w[:, 3] - w[:, 2]) / 100.0))
print("ticket     us", pc(us(w[:, 7])), " duration", pc((w[:, 7] - w[:, 3]) / 100.0))
print("finishing wave: first sum in %.2f, sums exchanged %.2f, Procrustes done %.2f, state written %.2f us" % tuple((int(buf[(1 << 19) - k]) - t0) / 100.0 for k in (4, 3, 2, 1)))
print("items total", w[:, 4].sum(), "per wave max", w[:, 4].max(), "waves with items", (w[:, 4] > 0).sum())
print("polls total", (w[:, 5] & 0xffffffff).sum(), "max", (w[:, 5] & 0xffffffff).max(), " lost CAS total", (w[:, 5] >> 32).sum())
pairs = w[:, 6] & ((1 << 40) - 1); passes = (w[:, 6] >> 40) & 0xff; unres = (w[:, 6] >> 48)
dur = (w[:, 1] - w[:, 0]) / 100.0
print("tile duration by passes:", {int(k): (int((passes == k).sum()), round(float(np.median(dur[passes == k])), 1), round(float(dur[passes == k].max()), 1)) for k in np.unique(passes)})
staged = pairs / 32.0
for lo, hi in [(0, 193), (193, 385), (385, 577), (577, 769), (769, 1e9)]:
    m = (staged >= lo) & (staged < hi)
    if m.any():
        print("  staged points (sum over passes) in [%d, %d): %5d waves, duration median %.1f p90 %.1f max %.1f, unresolved mean %.1f" % (lo, min(hi, 99999), m.sum(), np.median(dur[m]), np.percentile(dur[m], 90), dur[m].max(), unres[m].mean()))
o = np.argsort(-dur)[:12]
print("slowest tiles (us, passes, staged, unresolved):", [(round(float(dur[i]), 1), int(passes[i]), int(staged[i]), int(unres[i])) for i in o])
hb = buf[(1 << 17):(1 << 17) + 240000].reshape(-1, 4)
hb = hb[hb[:, 0] > 0]
if len(hb):
    print("hard items", len(hb), "cycles pct", pc(hb[:, 0].astype(float)), " start offset (wave clock / 16) pct", pc((hb[:, 1] & np.uint64(0xffffffff)).astype(float)))
nb = (N + 127) // 128
ph = buf[(1 << 16):(1 << 16) + nb * 8].reshape(nb, 8).astype(np.float64)
if ph.sum() > 0:   # -DPCR_WT_DIAG build: phase cycles of wave 0 of every block (shader clock)
    names = ["load+xform", "cube+level", "directory", "prefix", "staging", "filter", "merge+verify", "-"]
    print("wave-0 tile phases, cycles (sum over passes/rounds): median / p90 / max")
    for i, nm in enumerate(names[:7]):
        print("   %-14s %8.0f %8.0f %8.0f" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
    print("   total median %.0f" % np.median(ph.sum(axis=1)))
    tot = ph.sum(axis=1)
    heavy = tot >= np.percentile(tot, 97)
    print("the heaviest 3 %% of these waves (%d): mean cycles per phase" % heavy.sum(), {nm: int(ph[heavy, i].mean()) for i, nm in enumerate(names[:7])}, "total", int(tot[heavy].mean()))
# per queue group (wave id after the XCD remap, mod 32): items served and when the group's last wave left
nblk = nw // 4
per = nblk >> 3; main = per << 3
raw = np.arange(nw)
blk = raw // 4
blk2 = np.where(blk < main, (blk & 7) * per + (blk >> 3), blk)
wid = blk2 * 4 + (raw & 3)
grp = wid % 32
it_g = np.bincount(grp, weights=w[:, 4], minlength=32)
ex_g = np.array([us(w[grp == g, 3]).max() for g in range(32)])
last_tile_g = np.array([us(w[grp == g, 1]).max() for g in range(32)])
print("items per group min/median/max", int(it_g.min()), int(np.median(it_g)), int(it_g.max()), "| group's last queue exit us min/median/max %.1f %.1f %.1f" % (ex_g.min(), np.median(ex_g), ex_g.max()),
      "| last tile end per group min/max %.1f %.1f" % (last_tile_g.min(), last_tile_g.max()))
busy = w[:, 4] > 0
print("waves that served items: exit us pct 50/90/99/max", pc(us(w[busy, 3])), " idle ones:", pc(us(w[~busy, 3])))
