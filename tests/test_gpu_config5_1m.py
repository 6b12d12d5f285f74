"""GPU: BASELINE configs[4] at its full size -- a 1 000 000-point synthetic scan (8 KITTI-shaped frames of one static
world), ISS keypoints + radius-NN covariance on the GPU, feeding the global initialisation and a coarse-to-fine ICP.
At this size the CPU restatement cannot run in test time, so the checks are size-independent properties: neighbour
counts against scipy on a 1-in-997 sample, eigenvalue ordering, the reference's keypoint cap (ISS.py:72-73), and the
recovered rigid transform against the known truth.  ISS parity itself is "unpinned" by the reference (script body,
input file absent: DESIGN.md section 4); the small-size restatement test lives in test_gpu_voxel_knn_iss.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ISS_RADIUS = 0.09     # mean neighbour count ~ 38 on this cloud (SURVEY 8d asks for k ~ 40)


@pytest.fixture(scope="module")
def world_1m(syn):
    poses = [syn.rigid_transform((0, 0, 1), 0.02 * i, (3.0 * i, 0.2 * i, 0)) for i in range(8)]
    frames = [syn.kitti_like_scan(125000, seed=50 + i, sensor_pose=P) for i, P in enumerate(poses)]
    world = np.concatenate([f.astype(np.float64) @ P[:3, :3].T + P[:3, 3] for f, P in zip(frames, poses)])
    assert world.shape == (1_000_000, 3)
    return world


def test_iss_1m_counts_eigenvalues_and_cap(pcp, world_1m):
    from scipy.spatial import cKDTree

    cloud = pcp.DeviceCloud.upload(world_1m)
    kp, lam, counts = pcp.iss_keypoints(cloud, radius=ISS_RADIUS, non_max_radius=ISS_RADIUS, iss_count=20, return_details=True)
    cloud.free()
    assert lam.shape == (1_000_000, 3) and counts.shape == (1_000_000,)
    # |N(p)| (ISS.py:43: radius query, the point itself included) against scipy on a 1-in-997 sample
    tree = cKDTree(world_1m)
    sample = np.arange(0, 1_000_000, 997)
    ref = np.array([len(x) for x in tree.query_ball_point(world_1m[sample], ISS_RADIUS)])
    assert np.array_equal(counts[sample], ref)
    assert 25.0 < counts.mean() < 60.0          # the k ~ 40 regime of SURVEY 8d, not the 2 000-neighbour run of round 1
    assert counts.min() >= 1                    # every point is its own neighbour
    # eigenvalues of a weighted scatter matrix: real, descending, non-negative up to rounding
    assert np.isfinite(lam).all()
    assert (lam[:, 0] >= lam[:, 1]).all() and (lam[:, 1] >= lam[:, 2]).all()
    assert (lam[:, 2] >= -1e-12 * np.maximum(lam[:, 0], 1e-300)).all()
    assert (lam[:, 0] <= ISS_RADIUS ** 2 * (1 + 1e-9)).all()     # a scatter of offsets no longer than the radius
    # keypoints: the reference stops once MORE than iss_count are collected (ISS.py:72-73); all pass the ratio tests;
    # no two lie within the suppression radius of each other; collected in descending lambda_3
    assert 1 <= len(kp) <= 21 and len(set(kp)) == len(kp)
    k = np.asarray(kp)
    assert (lam[k, 1] / lam[k, 0] < 0.5).all() and (lam[k, 2] / lam[k, 1] < 0.5).all()
    assert (np.diff(lam[k, 2]) <= 0).all()
    d = np.linalg.norm(world_1m[k][:, None, :] - world_1m[k][None, :, :], axis=2)
    assert (d[np.triu_indices(len(k), 1)] > ISS_RADIUS).all()
    # the first keypoint is the global maximum of lambda_3 among the points that pass the ratio tests
    ok = (lam[:, 1] / np.maximum(lam[:, 0], 1e-300) < 0.5) & (lam[:, 2] / np.maximum(lam[:, 1], 1e-300) < 0.5) & (lam[:, 0] > 0)
    assert lam[k[0], 2] == lam[ok, 2].max()


def test_iss_init_then_coarse_to_fine_icp_1m(pcp, syn, world_1m):
    """The pipeline of configs[4]: ISS keypoints (GPU) are the feature-detection step of the template's global
    initialisation (icp_template.py:56-71), whose transform starts a coarse-to-fine ICP that ends on 1M x 1M points."""
    T_off = syn.rigid_transform((0.05, 0.0, 1.0), np.deg2rad(25.0), (3.0, -2.0, 0.1))
    src = (world_1m - T_off[:3, 3]) @ T_off[:3, :3]            # world = T_off * src
    src = src + np.random.default_rng(7).normal(0, 0.01, src.shape)
    R0, t0, info = pcp.ransac_init(pcp.PointCloud(src), pcp.PointCloud(world_1m), voxel_size=2.0, seed=3, detector="iss",
                                   iss_count=400, return_info=True)
    assert info["detector"] == "iss" and 3 <= info["n_src_keypoints"] <= 401 and 3 <= info["n_tgt_keypoints"] <= 401
    assert info["n_src_keypoints"] < info["n_src"]             # a real selection, not the whole down-sampled cloud
    T0 = np.eye(4)
    T0[:3, :3], T0[:3, 3] = R0, t0[:, 0]
    dR = R0 @ T_off[:3, :3].T
    ang0 = np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
    assert ang0 < 6.0 and np.linalg.norm(t0[:, 0] - T_off[:3, 3]) < 2.5, (ang0, t0.T, info)
    T, logs = pcp.coarse_to_fine_icp(src, world_1m, leaves=(2.0, 0.5, 0.0), init=T0, max_iteration=30)
    assert [lg["leaf"] for lg in logs] == [2.0, 0.5, 0.0]
    assert logs[-1]["n_assoc"] > 990_000                       # the last level associates (almost) all of the 1M points
    assert np.abs(T - T_off).max() < 1e-3, np.abs(T - T_off).max()
    # and without the ISS-based initialisation the same refinement does NOT get there from identity (the init matters)
    T_id, _ = pcp.coarse_to_fine_icp(src, world_1m, leaves=(2.0, 0.5, 0.0), max_iteration=30)
    assert np.abs(T_id - T_off).max() > 1e-2
