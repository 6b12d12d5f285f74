"""The 1 M x 1 M registration of bench.py's config5 leg, repeated: device and wall ms per iteration of 20-iteration calls.
STAGE=none|iss|c2f|iss,c2f runs the leg's earlier stages first; WORLD=frames (the bench's eight overlapping frames) or scan."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("point-cloud-process_amd")
syn = pkg.synthetic
ctx = pkg.Context(0)
poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
frames = [syn.kitti_like_scan(125000, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)]) if os.environ.get("WORLD", "frames") == "frames" else syn.kitti_like_scan(1000000, seed=11)
stage = os.environ.get("STAGE", "iss,c2f")
if "iss" in stage:
    cloud = pkg.DeviceCloud.upload(world, ctx)
    kp = pkg.iss_keypoints(cloud, radius=0.09, non_max_radius=0.09, iss_count=20)
    cloud.free()
T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
src = (world - T_off[:3, 3]) @ T_off[:3, :3]
src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
if "c2f" in stage:
    T, logs = pkg.coarse_to_fine_icp(src, world, leaves=(2.0, 0.5, 0.0), max_iteration=30)
index = pkg.TargetIndex(pkg.DeviceCloud.upload(world, ctx), ctx=ctx)
for rep, IT in enumerate([int(x) for x in os.environ.get("SEQ", "20,20,20,20").split(",")]):
    sd = pkg.DeviceCloud.upload(src, ctx).prepare(index)
    ctx.sync(); t0 = time.perf_counter()
    r = pkg.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1.0, t_thres=-1.0, max_d2=5.0, min_iter=IT)
    ctx.sync(); w = time.perf_counter() - t0
    print(stage, "rep", rep, "iters", IT, "total device ms %.2f" % r["device_ms"], "device ms/iter %.3f wall %.3f" % (r["device_ms"] / r["iters"], 1e3 * w / r["iters"]), flush=True)
    print("   ", ctx.search_stats())
    lg = ctx.pass_log()
    print("    host us", lg["host_us"], "sum of kernels us %.0f" % (sum(lg["tile_us"]) + sum(lg["drain_us"])))
    if os.environ.get("PASSLOG", "1") != "0":
        print("   tile us ", " ".join("%.0f" % v for v in lg["tile_us"]))
        print("   drain us", " ".join("%.0f" % v for v in lg["drain_us"]))
        print("   items   ", " ".join("%d" % v for v in lg["items"]), flush=True)
    sd.free()
