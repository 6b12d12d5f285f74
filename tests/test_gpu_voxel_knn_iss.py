"""GPU parity: voxel filter (bit-exact vs the reference's outputs), k-NN / radius API vs the reference's
kd-tree and octree results, ISS vs the CPU restatement."""
import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def test_voxel_filter_bit_exact_vs_reference(pcp):
    g = load_golden("voxel_filter.npz")
    for tag in g["cases"]:
        tag = str(tag)
        cname, leaf = tag.split("_leaf")
        pts = g[f"{cname}_in"]
        h, D = pcp.voxel_keys(pts, float(leaf))
        assert np.array_equal(D, g[f"{tag}_D"]), tag
        assert np.array_equal(h, g[f"{tag}_h"]), tag  # voxel indices: bit-exact float64
        cen = pcp.voxel_filter(pts, float(leaf), "centroid")
        ref = g[f"{tag}_centroid"]
        assert cen.shape == ref.shape, tag  # occupied voxels - 1
        assert np.array_equal(cen, ref), tag  # np.mean's pairwise summation reproduced bitwise
        rnd = pcp.voxel_filter(pts, float(leaf), "random", seed=7)
        assert rnd.shape == ref.shape
        # every random pick is a member of the voxel whose centroid sits in the same output row
        hs = np.sort(np.unique(h))[:-1]
        for row, key in zip(rnd[:: max(1, len(rnd) // 200)], hs[:: max(1, len(rnd) // 200)]):
            members = pts[h == key]
            assert (members == row).all(axis=1).any()


def test_voxel_filter_edge_cases(pcp):
    g = load_golden("voxel_filter.npz")
    out = pcp.voxel_filter(g["onevoxel_in"], 1.0, "centroid")
    assert out.shape == (0,)  # single occupied voxel -> the reference returns an empty array
    with pytest.raises(ValueError):
        pcp.voxel_filter(np.zeros((0, 3)), 0.1, "centroid")
    assert pcp.voxel_filter(g["onevoxel_in"], 1.0, "nonsense").shape == (0,)
    # float32 input is widened exactly: same result as float64 of the same values
    pts = g["uni1000_in"].astype(np.float32)
    assert np.array_equal(pcp.voxel_filter(pts, 0.1, "centroid"), pcp.voxel_filter(pts.astype(np.float64), 0.1, "centroid"))


def test_voxel_filter_120k_properties(pcp, oracle, syn):
    pts = syn.kitti_like_scan(120000, seed=3).astype(np.float64)
    out = pcp.voxel_filter(pts, 0.2, "centroid")
    ref, order, starts, ends = oracle.voxel_filter(pts, 0.2, "centroid")
    assert np.array_equal(out, ref)
    # idempotence-like property: every centroid lies inside the bounding box of the cloud
    assert (out >= pts.min(0)).all() and (out <= pts.max(0)).all()


def test_downsample_then_icp_config3(pcp, oracle, syn):
    """BASELINE config 3 at its full size: 0.2 m voxel downsample of both 120 000-point scans (bit-exact against the
    restatement pinned to voxel_filter.py), then ICP on the device-resident results against the CPU oracle."""
    src, tgt, Tt = syn.perturbed_pair(120000, seed=2)
    ds = pcp.voxel_filter_device(pcp.DeviceCloud.upload(src), 0.2)
    dt = pcp.voxel_filter_device(pcp.DeviceCloud.upload(tgt), 0.2)
    s_host, t_host = ds.download(), dt.download()
    assert np.array_equal(s_host, oracle.voxel_filter(src.astype(np.float64), 0.2)[0])
    index = pcp.TargetIndex(dt)
    res = pcp.icp_device(ds, index, np.eye(4), mode="compat")
    ref = oracle.icp_point2point(s_host, t_host, np.eye(4))
    assert res["iters"] == ref["iters"]
    assert np.linalg.norm(res["T"] - ref["T"]) < 1e-9


def test_knn_and_radius_match_reference_trees(pcp):
    g = load_golden("nn_api.npz")
    for name, rads in (("rand64", (0.25, 0.5)), ("kitti4000", (0.5, 1.0))):
        db = g[f"{name}_db"]
        kroot = pcp.kdtree_construction(db, leaf_size=4)
        oroot = pcp.octree_construction(db, 4, 0.0001)
        for qi, q in enumerate(g[f"{name}_queries"]):
            for k in (1, 8):
                for tree, fn, root in (("kd", pcp.kdtree_knn_search, kroot), ("oct", pcp.octree_knn_search, oroot)):
                    rs = pcp.KNNResultSet(capacity=k)
                    ret = fn(root, db, rs, q)
                    # the searches' boolean return (kdtree.py: always False; octree.py:187,212,...: "ball inside the octant")
                    assert bool(ret) == bool(g[f"{name}_{tree}_knn{k}_ret"][qi]), (name, tree, k, qi)
                    assert [x.index for x in rs.dist_index_list] == g[f"{name}_{tree}_knn{k}_idx"][qi].tolist(), (name, tree, k, qi)
                    assert np.allclose([x.distance for x in rs.dist_index_list], g[f"{name}_{tree}_knn{k}_dist"][qi], rtol=1e-14, atol=0)
            for rad in rads:
                for tree, fn, root in (("kd", pcp.kdtree_radius_search, kroot), ("oct", pcp.octree_radius_search, oroot),
                                       ("octfast", pcp.octree_radius_search_fast, oroot)):
                    rs = pcp.RadiusNNResultSet(radius=rad)
                    ret = fn(root, db, rs, q)
                    assert bool(ret) == bool(g[f"{name}_{tree}_rad{rad}_q{qi}_ret"][0]), (name, tree, rad, qi)
                    lst = sorted(rs.dist_index_list)
                    assert [x.index for x in lst] == g[f"{name}_{tree}_rad{rad}_q{qi}_idx"].tolist(), (name, tree, rad, qi)
                    assert rs.count == int(g[f"{name}_{tree}_rad{rad}_q{qi}_count"][0])


def test_knn_batch_properties(pcp, oracle, syn):
    db = syn.kitti_like_scan(20000, seed=4).astype(np.float64)
    root = pcp.kdtree_construction(db, 32)
    rng = np.random.default_rng(1)
    q = db[rng.integers(0, len(db), 300)] + rng.normal(0, 0.05, (300, 3))
    idx, dist = pcp.knn_search_batch(root, q, 8)
    assert (np.diff(dist, axis=1) >= 0).all()  # ascending
    for i in range(0, 300, 25):
        oi, od = oracle.knn_bruteforce(db, q[i], 8)
        assert np.array_equal(idx[i], oi) and np.allclose(dist[i], od, rtol=1e-14, atol=0)
    off, ridx, rdist = pcp.radius_search_batch(root, q[:50], 0.7)
    for i in range(50):
        oi, od = oracle.radius_bruteforce(db, q[i], 0.7)
        assert np.array_equal(ridx[off[i]:off[i + 1]], oi)
    # k larger than the cloud: unfilled slots keep (1e10, 0) like KNNResultSet
    small = pcp.kdtree_construction(db[:5], 4)
    idx, dist = pcp.knn_search_batch(small, q[:3], 8)
    assert (dist[:, 5:] == 1e10).all() and (idx[:, 5:] == 0).all()
    # Open3D-style wrapper used by main.py:117 returns SQUARED distances
    tree = pcp.KDTreeFlann(db)
    k, ii, dd = tree.search_knn_vector_3d(q[0], 1)
    assert k == 1 and abs(dd[0] - oracle.dist2_direct(q[0], db[ii[0]])) < 1e-15


def test_iss_matches_the_reference_script_golden(pcp, golden):
    """tests/golden/iss.npz holds what Keypoint_detection_ISS/ISS.py:17-75 itself computed (run unmodified in the build
    container, oracle/ref_harness.py gen_iss): the printed keypoint list, the candidates of the ratio tests and their
    lambda3 in sorted order.  pcr_iss must give the same keypoints, the same candidate set and lambda3 to 1e-9."""
    g = golden("iss.npz")
    for tag in ("object", "sparse", "six"):
        pts = g[f"{tag}_points"]
        radius, l21, l32, nmr, cap = g[f"{tag}_params"]
        kp, lam, counts = pcp.iss_keypoints(pts[:, :3], radius=radius, lambda21=l21, lambda32=l32, non_max_radius=nmr,
                                            iss_count=int(cap), return_details=True)
        assert kp == g[f"{tag}_iss_idx"].tolist(), tag
        cand = np.nonzero((lam[:, 1] / lam[:, 0] < l21) & (lam[:, 2] / lam[:, 1] < l32))[0]
        assert np.array_equal(cand, g[f"{tag}_cand_idx"]), tag
        srt = g[f"{tag}_sorted_idx"]
        assert np.allclose(lam[srt, 2], g[f"{tag}_sorted_lambda3"], rtol=1e-9, atol=0), tag
        # the weights the script used are 1 / len(query_ball_point(pj, radius)) (ISS.py:49): counts are implied
        from scipy.spatial import cKDTree
        assert np.array_equal(counts, cKDTree(pts[:, :3]).query_ball_point(pts[:, :3], radius, return_length=True)), tag
    # the defaults of the drop-in are the script's hyper-parameters (ISS.py:20-27)
    assert pcp.iss_keypoints(g["sparse_points"]) == g["sparse_iss_idx"].tolist()


def test_iss_matches_cpu_restatement(pcp, oracle, syn):
    """A second size/radius against the CPU restatement (itself pinned to the reference script's output by
    tests/test_oracle_golden.py::test_iss_oracle_matches_the_reference_script)."""
    pts = syn.object_cloud(6000, seed=2).astype(np.float64)
    kp, lam, counts = pcp.iss_keypoints(pts, radius=0.08, non_max_radius=0.08, iss_count=20, return_details=True)
    okp, olam, ocounts = oracle.iss_oracle(pts, radius=0.08, non_max_radius=0.08, iss_count=20)
    assert np.array_equal(counts, ocounts)
    assert np.allclose(lam, olam, rtol=1e-9, atol=1e-15)
    assert kp == okp
    assert len(kp) <= 21


def test_register_batch_single_gpu_streams(pcp, oracle, syn):
    """BASELINE config 4 on one GPU: several pairs in flight on separate HIP streams, ordered results."""
    pairs = []
    for i in range(6):
        s6, t6, _ = syn.registration_pair_6f(4000, seed=1000 + i)  # 6-float records like registration_dataset/*.bin
        pairs.append((s6, t6, None))
    res = pcp.register_batch(pairs, streams=3)
    assert [r["pair"] for r in res] == list(range(6))
    for (s6, t6, _), r in zip(pairs, res):
        ref = oracle.icp_point2point(s6[:, :3], t6[:, :3], np.eye(4))
        assert r["iters"] == ref["iters"]
        assert np.linalg.norm(r["T"] - ref["T"]) < 1e-9


def test_register_batch_unequal_pairs_match_serial(pcp, syn):
    """12 pairs of very different sizes and iteration counts on 3 streams: every result equals the serial run bit for
    bit (a context shared by two in-flight pairs would corrupt the pinned read-back or the scratch)."""
    rng = np.random.default_rng(5)
    pairs = []
    for i in range(12):
        n = int(rng.choice([600, 3000, 12000, 25000]))
        s, t, _ = syn.perturbed_pair(n, seed=300 + i, angle_deg=float(rng.uniform(0.5, 6.0)), t=tuple(rng.uniform(-0.8, 0.8, 3) * [1, 1, 0.1]))
        pairs.append((s, t, None))
    kw = dict(mode="total", max_iter=40, r_thres=1e-4, t_thres=1e-4)
    serial = pcp.register_batch(pairs, streams=1, **kw)
    threaded = pcp.register_batch(pairs, streams=3, **kw)
    assert len({r["iters"] for r in serial}) >= 2         # the pairs really differ in work (sizes differ by 40x as well)
    for a, b in zip(serial, threaded):
        assert a["pair"] == b["pair"] and a["iters"] == b["iters"] and a["n_assoc"] == b["n_assoc"]
        assert np.array_equal(a["T"], b["T"])


def test_icp_batch_fused_stages_bitwise_equal_per_pair_path(pcp, syn, monkeypatch):
    """pcr_icp_batch runs a share in fused stages (one launch per stage for all pairs of a sub-batch: csrc/pcr_batch.hip); with
    PCR_BATCH_PER_PAIR=1 it runs upload -> index build -> pcr_icp per pair.  Unequal pairs (40 .. 25 000 points, given and
    default T0, one pair that finds no association at all: main.py:125-127), three stopping rules, several sub-batch sizes
    and worker counts: every field of every result must agree BIT FOR BIT."""
    batch = __import__("importlib").import_module("point-cloud-process_amd.batch")
    rng = np.random.default_rng(5)
    pairs = []
    for i in range(14):
        n = int(rng.choice([40, 600, 3000, 12000, 25000]))
        s, t, _ = syn.perturbed_pair(n, seed=300 + i, angle_deg=float(rng.uniform(0.5, 6.0)), t=tuple(rng.uniform(-0.8, 0.8, 3) * [1, 1, 0.1]))
        pairs.append((s, t, None if i % 3 else syn.rigid_transform((0, 0, 1), 0.01, (0.05, 0, 0))))
    pairs.append((pairs[1][0] + np.float32(500.0), pairs[1][1], None))
    keys = ("iters", "status", "n_assoc", "cost", "mean_d2")
    for kw in (dict(mode="compat"), dict(mode="total", max_iter=40, r_thres=1e-4, t_thres=1e-4), dict(mode="total", max_iter=3, r_thres=1e-9, t_thres=1e-9)):
        monkeypatch.setenv("PCR_BATCH_PER_PAIR", "1")
        ref = batch.native_register_share(pairs, device=0, streams=1, **kw)
        monkeypatch.setenv("PCR_BATCH_PER_PAIR", "0")
        assert ref[-1]["status"] == 1 and ref[-1]["iters"] == 0          # PCR_E_TOO_FEW_ASSOC: soft, the batch goes on
        assert len({r["iters"] for r in ref}) >= 2 or kw.get("max_iter") == 3
        for sub, streams in ((4, 1), (5, 3), (64, 2)):
            monkeypatch.setenv("PCR_BATCH_SUB", str(sub))
            got = batch.native_register_share(pairs, device=0, streams=streams, **kw)
            for i, (a, b) in enumerate(zip(ref, got)):
                assert all(a[k] == b[k] for k in keys), (kw, sub, i, {k: (a[k], b[k]) for k in keys})
                assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["T_total"], b["T_total"]), (kw, sub, i)
    monkeypatch.delenv("PCR_BATCH_SUB")


def test_icp_batch_scans_shared_between_pairs(pcp, syn, monkeypatch):
    """Registration/reg_result.txt registers 342 pairs over 504 scans: a scan that several pairs of a call use is packed and brought
    over ONCE (by the first sub-batch that uses it; later sub-batches read its device copy behind an event, a later slot of the same
    sub-batch reads the same staging rows).  Chains of pairs over eight scans -- also one whose owner pair is not taken by the fused
    stages (its other cloud is empty) and a scan with a NaN that two pairs share -- through several sub-batch sizes and worker
    counts: every result bit for bit what the per-pair path gives."""
    batch = __import__("importlib").import_module("point-cloud-process_amd.batch")
    poses = [syn.rigid_transform((0, 0, 1), 0.015 * i, (0.25 * i, 0.05 * i, 0)) for i in range(8)]
    scans = [syn.kitti_like_scan(int(n), seed=900 + i, sensor_pose=P) for i, (n, P) in enumerate(zip([9000, 700, 15000, 4000, 12000, 2500, 20000, 6000], poses))]
    nan_scan = scans[5].copy()
    nan_scan[11, 2] = np.nan
    empty = np.zeros((0, 3), np.float32)
    order = [(1, 0), (2, 1), (2, 0), (3, 2), (4, 3), (0, 4), (6, 4), (7, 6), (6, 2), (1, 7), (5, 6), (3, 0)]
    pairs = [(scans[a], scans[b], None if i % 4 else syn.rigid_transform((0, 0, 1), 0.01, (0.05, 0, 0))) for i, (a, b) in enumerate(order)]
    pairs.insert(2, (nan_scan, scans[3], None))
    pairs.append((scans[2], nan_scan, None))
    kw = dict(mode="total", max_iter=12, r_thres=1e-4, t_thres=1e-4)
    keys = ("iters", "status", "n_assoc", "cost", "mean_d2")
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "1")
    ref = batch.native_register_share(pairs, device=0, streams=1, **kw)
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "0")
    assert batch.native_calls[-1][1:3] == (len(pairs), 9)          # 14 pairs over 9 scans (the eight + the NaN copy)
    for sub, streams in ((1, 3), (2, 2), (3, 4), (5, 1), (64, 2)):
        monkeypatch.setenv("PCR_BATCH_SUB", str(sub))
        got = batch.native_register_share(pairs, device=0, streams=streams, **kw)
        for i, (a, b) in enumerate(zip(ref, got)):
            assert all(a[k] == b[k] for k in keys), (sub, i, {k: (a[k], b[k]) for k in keys})
            assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["T_total"], b["T_total"]), (sub, i)
    # the owner of a shared scan is a pair the fused stages do not take (empty source): the scan still reaches the later sub-batch
    monkeypatch.setenv("PCR_BATCH_SUB", "1")
    import ctypes as C
    L = pcp._lib
    use = [(empty, scans[0]), (scans[1], scans[0]), (scans[2], scans[0])]
    parr = np.zeros(3, dtype=batch._PAIR_DT)
    for i, (s_, t_) in enumerate(use):
        parr[i] = (s_.ctypes.data if len(s_) else 0, s_.shape[0], 3, t_.ctypes.data, t_.shape[0], 3, 0)
    p = L.IcpParams()
    L.lib().pcr_icp_default_params(C.byref(p))
    res = np.zeros(3, dtype=batch._RESULT_DT)
    status = np.zeros(3, dtype=np.int32)
    ctxs = batch._pooled_contexts(0, 2)
    handles = (C.c_void_p * 2)(*[c.handle for c in ctxs])
    rc = L.lib().pcr_icp_batch(handles, 2, parr.ctypes.data_as(C.POINTER(L.Pair)), 3, C.byref(p), res.ctypes.data_as(C.POINTER(L.IcpResult)), L.iptr(status))
    assert rc == L.PCR_E_EMPTY and status[0] == L.PCR_E_EMPTY and status[1] == 0 and status[2] == 0
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "1")
    one = batch.native_register_share([(scans[1], scans[0], None), (scans[2], scans[0], None)], device=0, streams=1)
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "0")
    monkeypatch.delenv("PCR_BATCH_SUB")
    for i in (0, 1):
        assert np.array_equal(res["T"][i + 1].reshape(4, 4), one[i]["T"]) and res["iters"][i + 1] == one[i]["iters"]


def test_icp_batch_bad_pairs_do_not_poison_the_batch(pcp, syn, monkeypatch):
    """An empty cloud inside a batch: that pair's status is a hard error (and the call's return value), every other pair still
    gets its result (per-pair failure must not poison a batch).  A NaN coordinate is not an error of the path (such a
    point is never associated, like in the per-pair call): the fused stages hand that pair to the per-pair path, same bits."""
    import ctypes as C

    batch = __import__("importlib").import_module("point-cloud-process_amd.batch")
    L = pcp._lib
    good = [syn.perturbed_pair(3000, seed=400 + i)[:2] for i in range(5)]
    nan_src = good[1][0].copy()
    nan_src[17, 1] = np.nan
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "1")
    ref = batch.native_register_share([(s, t, None) for s, t in good], device=0, streams=2)
    ref_nan = batch.native_register_share([(nan_src, good[1][1], None)], device=0, streams=1)[0]
    monkeypatch.setenv("PCR_BATCH_PER_PAIR", "0")
    clouds = [(good[0][0], good[0][1]), (nan_src, good[1][1]), (good[2][0], good[2][1]), (np.zeros((0, 3), np.float32), good[3][1]), (good[4][0], good[4][1])]
    n = len(clouds)
    parr = np.zeros(n, dtype=batch._PAIR_DT)
    for i, (s, t) in enumerate(clouds):
        parr[i] = (s.ctypes.data if len(s) else 0, s.shape[0], 3, t.ctypes.data, t.shape[0], 3, 0)
    p = L.IcpParams()
    L.lib().pcr_icp_default_params(C.byref(p))
    res = np.zeros(n, dtype=batch._RESULT_DT)
    status = np.zeros(n, dtype=np.int32)
    ctxs = batch._pooled_contexts(0, 2)
    handles = (C.c_void_p * 2)(*[c.handle for c in ctxs])
    rc = L.lib().pcr_icp_batch(handles, 2, parr.ctypes.data_as(C.POINTER(L.Pair)), n, C.byref(p), res.ctypes.data_as(C.POINTER(L.IcpResult)), L.iptr(status))
    assert rc == L.PCR_E_EMPTY                         # the first hard error is the call's return value
    assert status[3] == L.PCR_E_EMPTY
    for i in (0, 2, 4):
        assert status[i] == 0 and np.array_equal(res["T"][i].reshape(4, 4), ref[i]["T"]) and res["iters"][i] == ref[i]["iters"]
    assert status[1] == ref_nan["status"] and np.array_equal(res["T"][1].reshape(4, 4), ref_nan["T"]) and res["n_assoc"][1] == ref_nan["n_assoc"]
    assert res["n_assoc"][1] < 3000


def test_icp_batch_native_entry_point(pcp, oracle, syn):
    """pcr_icp_batch (the pair loop of Registration/main.py:190-216 as one C call, native worker threads): results come back in
    pair order whatever thread ran them, equal the reference semantics (compat goldens' oracle), an empty batch is a no-op and a
    malformed pair raises instead of crashing."""
    batch = __import__("importlib").import_module("point-cloud-process_amd.batch")
    pairs = []
    for i in range(9):
        s6, t6, _ = syn.registration_pair_6f(3000 + 500 * (i % 3), seed=2000 + i)
        pairs.append((s6, t6, None if i % 2 else np.eye(4)))
    res = batch.native_register_share(pairs, device=0, streams=4)
    assert len(res) == 9
    for (s6, t6, _), r in zip(pairs, res):
        ref = oracle.icp_point2point(s6[:, :3], t6[:, :3], np.eye(4))
        assert r["iters"] == ref["iters"] and np.linalg.norm(r["T"] - ref["T"]) < 1e-9
    assert batch.native_register_share([], device=0) == []
    with pytest.raises(ValueError):
        batch.native_register_share([(np.zeros((5, 2), np.float32), np.zeros((5, 3), np.float32), None)], device=0)
    # an empty cloud inside a batch is a hard error of that pair (PCR_E_EMPTY), reported, not a crash
    with pytest.raises(Exception):
        batch.native_register_share([(np.zeros((0, 3), np.float32), pairs[0][1], None)], device=0)


def test_dbscan_matches_reference_labels(pcp, oracle, syn):
    """Cluster_dbscan/dbscan.py: labels equal the reference's own output (goldens), numbering and noise quirks included."""
    g = load_golden("dbscan.npz")
    for tag in ("blobs", "blobs_tight", "scan"):
        r, m = g[f"{tag}_param"]
        d = pcp.DBSCAN(radius=float(r), Min_Pts=int(m))
        d.fit(g[f"{tag}_pts"])
        assert d.predict().dtype == np.int32
        assert np.array_equal(d.predict(), g[f"{tag}_labels"]), tag
    pts = syn.kitti_like_scan(60000, seed=6).astype(np.float64)
    d = pcp.DBSCAN(radius=0.6, Min_Pts=8)
    d.fit(pts)
    assert np.array_equal(d.predict(), oracle.dbscan(pts, 0.6, 8))


def test_kdtreeflann_radius_search(pcp, oracle, syn):
    pts = syn.kitti_like_scan(20000, seed=7).astype(np.float64)
    tree = pcp.KDTreeFlann(pcp.PointCloud(pts))
    k, idx, d2 = tree.search_radius_vector_3d(pts[123] + 0.01, 1.5)
    ref = np.flatnonzero(np.linalg.norm(pts - (pts[123] + 0.01), axis=1) <= 1.5)
    assert k == len(ref) and set(idx) == set(ref.tolist()) and np.all(np.diff(d2) >= 0)
    k2, idx2, _ = tree.search_hybrid_vector_3d(pts[123] + 0.01, 1.5, 10)
    assert k2 == min(10, k) and idx2 == idx[:k2]


def test_coarse_to_fine_icp_config5(pcp, syn):
    """BASELINE config 5's refinement: down-sampled levels first, each starting from the previous transform; a 6 degree /
    1.2 m offset that full-resolution ICP alone resolves more slowly is brought inside the evaluator's success gate."""
    src, tgt, T_true = syn.perturbed_pair(120000, seed=5, angle_deg=6.0, t=(1.2, -0.6, 0.05))
    T, logs = pcp.coarse_to_fine_icp(src, tgt, leaves=(1.0, 0.4, 0.0), max_iteration=40)
    assert [l["leaf"] for l in logs] == [1.0, 0.4, 0.0]
    ok, rte, rre = pcp.is_registration_successful(T, T_true)
    assert ok and rte < 0.5 and rre < 1.0, (rte, rre, logs)
    assert logs[-1]["n_assoc"] > 100000
