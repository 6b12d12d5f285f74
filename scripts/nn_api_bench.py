"""The reference's Kdtree_Octree/lesson2/benchmark.py:29-147, re-shaped for the drop-in query API.

Same configuration (leaf_size 32, min_extent 1e-4, k = 8, radius = 1, query = first point) and the same four
timings per structure -- build / k-NN / radius / brute -- on a KITTI-shaped synthetic scan of the real file's size
(124 668 points; Kdtree_Octree/000000.bin itself is not on the GPU box), in the CORRECT orientation: db is (N,3)
(benchmark.py reads (3,N) through read_velodyne_bin and indexes the wrong axis: SURVEY section 6).
Printed per structure in the reference's format, plus two things a GPU index is for: the BATCHED forms
(all N points as queries in one call) and scipy's cKDTree on the host as the CPU baseline.

  python scripts/nn_api_bench.py [--json out.json]
"""
import argparse, importlib, json, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
from scipy.spatial import cKDTree  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--json", default="")
ap.add_argument("--points", type=int, default=124668)
a = ap.parse_args()

leaf_size, min_extent, k, radius = 32, 0.0001, 8, 1.0
db_np = pcp.synthetic.kitti_like_scan(a.points, seed=0).astype(np.float64)   # (N,3)
query = db_np[0, :]
ctx = pcp.default_context()
out = {"points": a.points, "k": k, "radius": radius, "structures": {}}


def ms(fn, reps=1):
    fn()  # warm-up (code objects, arenas)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    ctx.sync()
    return (time.perf_counter() - t0) * 1e3 / reps, r


def brute():
    diff = np.linalg.norm(np.expand_dims(query, 0) - db_np, axis=1)
    nn_idx = np.argsort(diff)
    return diff[nn_idx]


brute_ms, _ = ms(brute, 3)
for name, build, knn, rad in (
    ("octree", lambda: pcp.octree_construction(db_np, leaf_size, min_extent), pcp.octree_knn_search, pcp.octree_radius_search_fast),
    ("kdtree", lambda: pcp.kdtree_construction(db_np, leaf_size), pcp.kdtree_knn_search, pcp.kdtree_radius_search),
):
    build_ms, root = ms(build, 3)

    def one_knn():
        rs = pcp.KNNResultSet(capacity=k)
        knn(root, db_np, rs, query)
        return rs

    def one_rad():
        rs = pcp.RadiusNNResultSet(radius=radius)
        rad(root, db_np, rs, query)
        return rs

    knn_ms, rs_k = ms(one_knn, 20)
    rad_ms, rs_r = ms(one_rad, 20)
    print("%s --------------" % name)
    print("%s: build %.3f, knn %.3f, radius %.3f, brute %.3f" % (name.capitalize(), build_ms, knn_ms, rad_ms, brute_ms))
    # batched: every point of the scan as a query, one call
    bk_ms, (bi, bd) = ms(lambda: pcp.knn_search_batch(root, db_np, k), 3)
    sub = db_np[::12]
    br_ms, (off, ri, rd) = ms(lambda: pcp.radius_search_batch(root, sub, radius), 2)
    print("   batched: k-NN of all %d points %.3f ms (%.2f Mquery/s); radius of %d points %.3f ms (%d neighbours, %.1f Mneighbour/s)"
          % (len(db_np), bk_ms, len(db_np) / bk_ms / 1e3, len(sub), br_ms, int(off[-1]), off[-1] / br_ms / 1e3))
    out["structures"][name] = {"build_ms": build_ms, "knn_single_ms": knn_ms, "radius_single_ms": rad_ms, "brute_ms": brute_ms,
                               "knn_batch_all_points_ms": bk_ms, "knn_batch_Mquery_per_s": len(db_np) / bk_ms / 1e3,
                               "radius_batch_queries": len(sub), "radius_batch_ms": br_ms, "radius_batch_neighbours": int(off[-1]),
                               "knn_single_result_size": rs_k.size(), "radius_single_result_size": rs_r.size()}

print("scipy cKDTree (host, %d threads) --------------" % len(os.sched_getaffinity(0)))
t0 = time.perf_counter(); tree = cKDTree(db_np, leafsize=leaf_size); b = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); tree.query(query, k); kq = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); tree.query_ball_point(query, radius); rq = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); tree.query(db_np, k, workers=-1); kb = (time.perf_counter() - t0) * 1e3
sub = db_np[::12]
t0 = time.perf_counter(); cnt = tree.query_ball_point(sub, radius, workers=-1, return_length=True); rb = (time.perf_counter() - t0) * 1e3
print("Kdtree: build %.3f, knn %.3f, radius %.3f, brute %.3f" % (b, kq, rq, brute_ms))
print("   batched: k-NN of all points %.3f ms (%.2f Mquery/s); radius counts of %d points %.3f ms" % (kb, len(db_np) / kb / 1e3, len(sub), rb))
out["structures"]["scipy_ckdtree_host"] = {"build_ms": b, "knn_single_ms": kq, "radius_single_ms": rq, "knn_batch_all_points_ms": kb,
                                           "radius_batch_count_only_ms": rb, "threads": len(os.sched_getaffinity(0))}
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(out, open(a.json, "w"), indent=1)
