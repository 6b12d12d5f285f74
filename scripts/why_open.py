"""Why the wave tiles of a pass leave queries open (two-launch ICP pass, N points; needs a diagnostic build:
scripts/build_variant.sh diag "-DPCR_WT_DIAG -DPCR_PASS_DIAG=1", PCR_LIB_PATH=scripts/bin/libpcr_diag.so): counts of the last pass of
calls of 1, 2, ... iterations -- 0 clamped / never took part, 1 no level fits the box, 2 too many points in the box, 3 ambiguous
filter result, 4 the ball reaches out of the staged box."""
import ctypes as C, importlib, os, sys
import numpy as np
os.environ["PCR_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("point-cloud-process_amd")
syn, L = pkg.synthetic, pkg._lib
N = int(os.environ.get("N", 1000000))
poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
frames = [syn.kitti_like_scan(N // 8, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])
T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
src = (world - T_off[:3, 3]) @ T_off[:3, :3]
src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
ctx = pkg.Context(0)
index = pkg.TargetIndex(pkg.DeviceCloud.upload(world, ctx), ctx=ctx)
for IT in [int(x) for x in os.environ.get("ITS", "1,2,5,12").split(",")]:
    sd = pkg.DeviceCloud.upload(src, ctx).prepare(index)
    r = pkg.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1.0, t_thres=-1.0, max_d2=5.0, min_iter=IT)
    buf = np.zeros(1 << 19, dtype=np.uint64)
    L.check(L.lib().pcr_debug_read(ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size))
    why = buf[(1 << 15):(1 << 15) + 5].astype(np.int64)
    lg = ctx.pass_log()
    print(f"pass {IT}: open by reason [other, level, points, ambiguous, ball-out] = {why.tolist()} sum {int(why.sum())}; items {lg['items'][-1]}; tile us {lg['tile_us'][-1]:.0f} drain us {lg['drain_us'][-1]:.0f}", flush=True)
    sd.free()
