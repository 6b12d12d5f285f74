import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pcp = importlib.import_module("point-cloud-process_amd")
syn = pcp.synthetic
world = syn.kitti_like_scan(1000000, seed=11)
T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
src = (world - T_off[:3, 3]) @ T_off[:3, :3]
src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
ctx = pcp.default_context()
index = pcp.TargetIndex(pcp.DeviceCloud.upload(world, ctx), ctx=ctx)
for rep in range(6):
    sd = pcp.DeviceCloud.upload(src, ctx).prepare(index)
    ctx.sync(); t0 = time.perf_counter()
    r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=20, r_thres=-1.0, t_thres=-1.0, max_d2=float(os.environ.get("MAX_D2", 1.0)), min_iter=20)
    ctx.sync(); w = time.perf_counter() - t0
    print("rep", rep, "device ms/iter %.3f wall ms/iter %.3f n_assoc %d" % (r["device_ms"] / r["iters"], 1e3 * w / r["iters"], r["n_assoc"]), flush=True)
    sd.free()
