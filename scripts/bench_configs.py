"""Timings of the other BASELINE.json configs on one GPU (parity-test cases, not bench lines)."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
syn = pcp.synthetic
ctx = pcp.default_context()
out = {}

def timed(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps, r

# C1: 2 048-pt object vs rotated copy
obj = syn.object_cloud(2048, seed=1)
Trot = syn.rigid_transform((0, 0, 1), np.deg2rad(10.0), (0.02, -0.01, 0.03))
src = ((obj.astype(np.float64) - Trot[:3, 3]) @ Trot[:3, :3]).astype(np.float32)
dt, T = timed(lambda: pcp.icp_point2point(pcp.PointCloud(src), obj, np.eye(4)), 10)
out["C1_icp_point2point_2048_ms"] = dt * 1e3

# C3: 0.2 m voxel downsample of a 120k pair, then ICP (device resident)
s120, t120, _ = syn.perturbed_pair(120000, seed=0)
ds0, dt0 = pcp.DeviceCloud.upload(s120), pcp.DeviceCloud.upload(t120)
tv, dsv = timed(lambda: pcp.voxel_filter_device(ds0, 0.2), 5)
out["C3_voxel_filter_120k_ms"] = tv * 1e3
out["C3_voxel_rows"] = dsv.n
dtv = pcp.voxel_filter_device(dt0, 0.2)
index = pcp.TargetIndex(dtv)
def icp_c3():
    s = pcp.voxel_filter_device(ds0, 0.2).prepare(index)
    return pcp.icp_device(s, index, np.eye(4), mode="total", max_iter=30, r_thres=-1, t_thres=-1, min_iter=30)
ti, r = timed(icp_c3, 3)
out["C3_downsample_plus_30_icp_iters_ms"] = ti * 1e3
out["C3_icp_ms_per_iter_device"] = r["device_ms"] / r["iters"]
out["C3_points"] = [dsv.n, dtv.n]

# C4: 256 pairs x 20 000 points (6-float records), compat ICP, one GPU, several streams
pairs = [syn.registration_pair_6f(20000, seed=1000 + i)[:2] + (None,) for i in range(32)]
pairs = pairs * 8
for streams in (1, 4):
    t0 = time.perf_counter()
    res = pcp.register_batch(pairs, streams=streams)
    el = time.perf_counter() - t0
    out[f"C4_256_pairs_20k_streams{streams}_s"] = el
    out[f"C4_pairs_per_s_streams{streams}"] = len(pairs) / el
out["C4_mean_iters"] = float(np.mean([r["iters"] for r in res]))

# C5: 1M-pt synthetic scan (8 frames), ISS
frames = [syn.kitti_like_scan(125000, seed=50 + i, sensor_pose=syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0))) for i in range(8)]
poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])
cloud = pcp.DeviceCloud.upload(world)
t0 = time.perf_counter()
kp, lam, counts = pcp.iss_keypoints(cloud, radius=0.6, non_max_radius=0.6, iss_count=20, return_details=True)
out["C5_iss_1M_points_s"] = time.perf_counter() - t0
out["C5_mean_neighbours"] = float(counts.mean())
out["C5_keypoints"] = len(kp)
# C5 continued: coarse-to-fine ICP of two 1M-point scans of the same world (8 frames each, second set shifted)
T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(3.0), (0.8, -0.4, 0.02))
world2 = (world - T_off[:3, 3]) @ T_off[:3, :3]          # world = T_off * world2
world2 = world2 + np.random.default_rng(7).normal(0, 0.01, world2.shape)
t0 = time.perf_counter()
Tc, logs = pcp.coarse_to_fine_icp(world2, world, leaves=(2.0, 0.5, 0.0), max_iteration=30)
out["C5_coarse_to_fine_icp_1M_s"] = time.perf_counter() - t0
out["C5_icp_levels"] = logs
out["C5_icp_error_vs_truth"] = float(np.abs(Tc - T_off).max())
print(json.dumps(out, indent=1))
