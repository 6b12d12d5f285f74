"""Work-queue items of the two-launch ICP pass at N points (default 1 M x 1 M from the bench's 0.8 m / 3 degree offset): for the last pass
of calls of 1, 2, 3 ... iterations -- items published, cycles / descent steps / points scanned per item (the drain kernel's own
stamps, PCR_DEBUG_STAMPS=1; s_memtime ticks at 100 MHz)."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("point-cloud-process_amd")
syn, L = pkg.synthetic, pkg._lib
N = int(os.environ.get("N", 1000000))
if os.environ.get("WORLD", "frames") == "frames":   # the bench's config5 world: eight overlapping frames
    poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
    frames = [syn.kitti_like_scan(N // 8, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
    world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])
else:
    world = syn.kitti_like_scan(N, seed=11)
T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
src = (world - T_off[:3, 3]) @ T_off[:3, :3]
src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
ctx = pkg.Context(0)
index = pkg.TargetIndex(pkg.DeviceCloud.upload(world, ctx), ctx=ctx)
print("cell0", index.cell)
for IT in [int(x) for x in os.environ.get("ITS", "1,2,3,5,8,12,20").split(",")]:
    sd = pkg.DeviceCloud.upload(src, ctx).prepare(index)
    r = pkg.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1.0, t_thres=-1.0, max_d2=float(os.environ.get("MAX_D2", 5.0)), min_iter=IT)
    buf = np.zeros(1 << 19, dtype=np.uint64)
    L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
    total = int(buf[(1 << 19) - 8])
    it = buf[(1 << 17):(1 << 17) + 4 * 60000].reshape(-1, 4)
    it = it[it[:, 0] > 0]
    cyc = it[:, 0].astype(np.float64) / float(os.environ.get("TICK_MHZ", 100.0))      # us if s_memtime ticks at TICK_MHZ
    steps = (it[:, 1] >> np.uint64(32)).astype(np.int64)
    pts = it[:, 2].astype(np.int64)
    cand = (it[:, 3] & np.uint64(1)).astype(np.int64)
    lvl = ((it[:, 3] >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64) - 1
    q = lambda a, p: float(np.percentile(a, p)) if len(a) else 0.0
    print(f"pass {IT}: items {total} of {N} ({100.0 * total / N:.1f} %), sample {len(it)}: us/item p50 {q(cyc,50):.1f} p90 {q(cyc,90):.1f} p99 {q(cyc,99):.1f} mean {cyc.mean() if len(cyc) else 0:.1f}; "
          f"steps p50 {q(steps,50):.0f} p90 {q(steps,90):.0f}; points p50 {q(pts,50):.0f} p90 {q(pts,90):.0f} p99 {q(pts,99):.0f} mean {pts.mean() if len(pts) else 0:.0f}; with candidate {cand.mean() if len(cand) else 0:.2f}; "
          f"start level hist {np.bincount(np.clip(lvl, 0, 11), minlength=6)[:6].tolist()}; n_assoc {r['n_assoc']} mean_d2 {r['mean_d2']:.4f} device ms {r['device_ms']:.2f}", flush=True)
    sd.free()
