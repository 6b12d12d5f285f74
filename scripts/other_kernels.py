"""Driver for profiling the kernels outside the ICP pass (one op per process so rocprofv3 stats are per op).

  python scripts/other_kernels.py <op> [reps]
  ops: voxel120k voxel1m iss1m knn120k radius20k normals120k

Prints one JSON line: wall ms per call (device-resident inputs where the API allows) + the algorithmic bytes
SURVEY 8d assigns to the op.  scripts/profile_other.sh runs it under rocprofv3 (--kernel-trace --stats, then
FETCH_SIZE / WRITE_SIZE in separate passes) and folds everything into profiles/r03_other_configs.json."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
syn = pcp.synthetic
op = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = pcp.default_context()


def world_1m():
    poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
    frames = [syn.kitti_like_scan(125000, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
    return np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])


def timed(fn):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3, r


out = {"op": op, "reps": reps}
if op in ("voxel120k", "voxel1m"):
    pts = syn.perturbed_pair(120000, seed=0)[1].astype(np.float64) if op == "voxel120k" else world_1m()
    d = pcp.DeviceCloud.upload(pts)
    ms, r = timed(lambda: pcp.voxel_filter_device(d, 0.2))
    out.update(n=len(pts), rows=r.n, ms=ms,
               # 16 B in + 8 B key out per input point + 24 B per output voxel
               algorithmic_bytes=24 * len(pts) + 24 * r.n)
elif op == "iss1m":
    pts = world_1m()
    d = pcp.DeviceCloud.upload(pts)
    r = pcp.iss_keypoints(d, radius=0.09, non_max_radius=0.09, iss_count=20, return_details=True)
    ms_details, _ = timed(lambda: pcp.iss_keypoints(d, radius=0.09, non_max_radius=0.09, iss_count=20, return_details=True))
    ms, kp = timed(lambda: pcp.iss_keypoints(d, radius=0.09, non_max_radius=0.09, iss_count=20))   # the reference's output: the keypoint list
    assert kp == r[0]
    out.update(n=len(pts), ms=ms, ms_with_per_point_details=ms_details, mean_neighbours=float(r[2].mean()), keypoints=len(r[0]),
               # (16 B + 4 B count) per point per pass x 2 passes + 12 B out = 52 B per point
               algorithmic_bytes=52 * len(pts))
elif op == "knn120k":
    tgt = syn.perturbed_pair(120000, seed=0)[1].astype(np.float64)
    q = syn.perturbed_pair(120000, seed=0)[0].astype(np.float64)
    root = pcp.kdtree_construction(tgt, 32)
    ms, r = timed(lambda: pcp.knn_search_batch(root, q, 8))
    out.update(n=len(tgt), q=len(q), k=8, ms=ms,
               # 24 B query in + k x (24 B candidate record + 4 B index + 8 B distance out)
               algorithmic_bytes=len(q) * (24 + 8 * (24 + 12)))
elif op == "radius20k":
    tgt = syn.perturbed_pair(120000, seed=0)[1].astype(np.float64)
    q = syn.perturbed_pair(120000, seed=0)[0].astype(np.float64)[::6]
    root = pcp.kdtree_construction(tgt, 32)
    ms, r = timed(lambda: pcp.radius_search_batch(root, q, 1.0))
    m = int(r[0][-1])
    out.update(n=len(tgt), q=len(q), radius=1.0, ms=ms, neighbours_total=m,
               # count pass + fill pass: the query twice, every neighbour record twice, 12 B out per neighbour
               algorithmic_bytes=len(q) * 48 + m * (2 * 24 + 12))
elif op == "normals120k":
    pts = syn.perturbed_pair(120000, seed=0)[1].astype(np.float64)
    d = pcp.DeviceCloud.upload(pts)
    ms, r = timed(lambda: pcp.estimate_normals(d, 5))
    out.update(n=len(pts), k=5, ms=ms,
               # 24 B point in + k x 24 B neighbour records + 24 B normal + 24 B eigenvalues out
               algorithmic_bytes=len(pts) * (24 + 5 * 24 + 48))
else:
    raise SystemExit("unknown op " + op)
print(json.dumps(out))
