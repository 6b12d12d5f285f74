"""Global initialisation in front of ICP: the Open3D stage of Registration/main.py:33-84 and the template surface
icp_template.py:20-41,56-110 -- voxel down-sample, hybrid-radius normals, FPFH, feature matching, 3-point RANSAC.

Function names and arguments mirror the reference's (``preprocess_point_cloud``, ``prepare_dataset``,
``execute_global_registration`` of main.py; ``find_matchings``, ``ransac_init`` of icp_template.py).  Open3D itself is
absent and unpinned in the reference, so this stage is "parity unpinned": it follows Open3D's published behaviour
and is checked in tests/ against a CPU restatement (same counter-based random stream -> same hypotheses).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .device import DeviceCloud, TargetIndex, default_context, points_of
from .registration import PointCloud, read_bin_velodyne

__all__ = [
    "Feature", "RegistrationResult", "voxel_down_sample", "voxel_down_sample_device", "estimate_normals_hybrid", "compute_fpfh_feature", "find_matchings",
    "registration_ransac_based_on_feature_matching", "preprocess_point_cloud", "prepare_dataset", "execute_global_registration",
    "ransac_init",
]


class Feature:
    """o3d.pipelines.registration.Feature: ``.data`` is (dim, N) like Open3D's (main.py:44-46)."""

    def __init__(self, data):
        self.data = np.asarray(data, dtype=np.float64)

    def dimension(self):
        return self.data.shape[0]

    def num(self):
        return self.data.shape[1]


class RegistrationResult:
    """o3d.pipelines.registration.RegistrationResult: what main.py:211 reads is ``.transformation``."""

    def __init__(self, transformation, fitness=0.0, inlier_rmse=0.0, correspondence_set=None, info=None):
        self.transformation = np.asarray(transformation, dtype=np.float64)
        self.fitness = float(fitness)
        self.inlier_rmse = float(inlier_rmse)
        self.correspondence_set = np.zeros((0, 2), dtype=np.int32) if correspondence_set is None else correspondence_set
        self.info = info or {}

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


class _Prep:
    """Owner of a pcr_prep handle (preprocess_point_cloud's result, resident on the device)."""

    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle
        self.n = int(L.lib().pcr_prep_size(handle))

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("preprocessed cloud freed")
        return self._h

    def download(self, points=False, normals=False, features=False):
        out = [np.empty((self.n, 3)) if points else None, np.empty((self.n, 3)) if normals else None, np.empty((self.n, 33)) if features else None]
        L.check(L.lib().pcr_prep_download(self.ctx.handle, self.handle, *[L.dptr(a) if a is not None else None for a in out]), self.ctx.handle)
        return out

    def cloud(self):
        """Non-owning DeviceCloud view of the down-sampled cloud."""
        c = DeviceCloud(self.ctx, C.c_void_p(L.lib().pcr_prep_cloud(self.handle)), self.n)
        c.free = lambda: None   # the prep owns it
        return c

    def free(self):
        if self._h:
            L.lib().pcr_prep_free(self.ctx.handle, self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class PreparedCloud(PointCloud):
    """pcd_down of preprocess_point_cloud (main.py:35-40): ``.points`` / ``.normals`` are fetched from the device on first use."""

    def __init__(self, prep):
        self._prep = prep
        self._points = self._normals = None

    @property
    def points(self):
        if self._points is None:
            self._points = self._prep.download(points=True)[0]
        return self._points

    @points.setter
    def points(self, v):   # (transform() and callers may replace them: the object then no longer stands for the device copy)
        self._points = np.asarray(v, dtype=np.float64)
        self._prep = None

    @property
    def normals(self):
        if self._normals is None and self._prep is not None:
            self._normals = self._prep.download(normals=True)[1]
        return self._normals

    @normals.setter
    def normals(self, v):
        self._normals = None if v is None else np.asarray(v, dtype=np.float64)

    def __len__(self):
        return self._prep.n if self._prep is not None else len(self._points)


class PreparedFeature(Feature):
    """pcd_fpfh of preprocess_point_cloud (main.py:44-46): ``.data`` (33, N) is fetched from the device on first use."""

    def __init__(self, prep):
        self._prep = prep
        self._data = None

    @property
    def data(self):
        if self._data is None:
            self._data = np.ascontiguousarray(self._prep.download(features=True)[2].T)
        return self._data

    def dimension(self):
        return 33

    def num(self):
        return self._prep.n


def _prep_of(down, feature, ctx):
    """The device-resident prep behind (pcd_down, pcd_fpfh) when both still stand for it (same device)."""
    a, b = getattr(down, "_prep", None), getattr(feature, "_prep", None)
    return a if (a is not None and a is b and a._h and a.ctx.device == ctx.device) else None


def _cloud(points, ctx):
    if isinstance(points, DeviceCloud):
        return points, None
    c = DeviceCloud.upload(np.ascontiguousarray(points_of(points)[:, :3], dtype=np.float64), ctx)
    return c, c


def voxel_down_sample(points, voxel_size, ctx=None):
    """pcd.voxel_down_sample(voxel_size) (main.py:35): origin min - voxel/2, centroid of every occupied voxel,
    rows ordered by voxel key -> (M,3) float64."""
    ctx = ctx or default_context()
    pc = np.ascontiguousarray(points_of(points)[:, :3], dtype=np.float64)
    if pc.shape[0] == 0:
        raise ValueError("empty cloud")
    out = np.empty_like(pc)
    n_out = C.c_int64()
    L.check(L.lib().pcr_voxel_filter(ctx.handle, L.dptr(pc), pc.shape[0], float(voxel_size), 2, C.c_uint64(0), L.dptr(out), C.byref(n_out)),
            ctx.handle)
    return out[: n_out.value].copy()


def voxel_down_sample_device(cloud, voxel_size, ctx=None):
    """voxel_down_sample on a DeviceCloud -> DeviceCloud (rows ordered by voxel key); nothing leaves the device."""
    ctx = ctx or cloud.ctx
    h = C.c_void_p()
    L.check(L.lib().pcr_voxel_filter_cloud(ctx.handle, cloud.handle, float(voxel_size), 2, C.c_uint64(0), C.byref(h)), ctx.handle)
    return DeviceCloud(ctx, h, L.lib().pcr_cloud_size(h))


def estimate_normals_hybrid(points, radius, max_nn=30, orient=True, viewpoint=(0.0, 0.0, 0.0), ctx=None):
    """estimate_normals(KDTreeSearchParamHybrid(radius, max_nn)) (main.py:39-40) -> (N,3).  ``orient`` flips the
    normals toward ``viewpoint`` (the sensor origin of a scan); Open3D leaves the sign to its eigen-solver."""
    ctx = ctx or default_context()
    cloud, own = _cloud(points, ctx)
    out = np.empty((cloud.n, 3), dtype=np.float64)
    vp = L.as_f64(np.asarray(viewpoint, dtype=np.float64).reshape(3))
    L.check(L.lib().pcr_normals_hybrid(ctx.handle, cloud.handle, float(radius), int(max_nn), 1 if orient else 0, L.dptr(vp), L.dptr(out)),
            ctx.handle)
    if own is not None:
        own.free()
    return out


def compute_fpfh_feature(points, normals=None, radius=10.0, max_nn=100, ctx=None):
    """o3d.pipelines.registration.compute_fpfh_feature(pcd, KDTreeSearchParamHybrid(radius, max_nn)) (main.py:44-46).
    ``points`` may carry ``.normals``; returns a Feature with ``.data`` (33, N)."""
    ctx = ctx or default_context()
    if normals is None:
        normals = getattr(points, "normals", None)
    if normals is None:
        raise ValueError("compute_fpfh_feature needs normals (Open3D raises too when the cloud has none)")
    cloud, own = _cloud(points, ctx)
    nrm = L.as_f64(np.asarray(normals, dtype=np.float64).reshape(-1, 3))
    if nrm.shape[0] != cloud.n:
        raise ValueError("normals do not match the cloud")
    out = np.empty((cloud.n, 33), dtype=np.float64)
    L.check(L.lib().pcr_fpfh(ctx.handle, cloud.handle, L.dptr(nrm), float(radius), int(max_nn), L.dptr(out)), ctx.handle)
    if own is not None:
        own.free()
    return Feature(out.T)


def _match(a, b, ctx):
    a = L.as_f64(a)
    b = L.as_f64(b)
    idx = np.empty(a.shape[0], dtype=np.int32)
    d2 = np.empty(a.shape[0], dtype=np.float64)
    L.check(L.lib().pcr_feature_match(ctx.handle, L.dptr(a), a.shape[0], L.dptr(b), b.shape[0], a.shape[1], L.iptr(idx), L.dptr(d2)), ctx.handle)
    return idx, d2


def find_matchings(src_features, tgt_features, src_tree=None, tgt_tree=None, dist_thres=None, mutual=True, ctx=None):
    """icp_template.py:20-41: features are (length of feature, #points); returns an (M,2) array of
    (src_idx, tgt_idx): nearest target feature of every source feature, kept when it is mutual (``mutual``) and,
    if ``dist_thres`` is given, closer than it (Euclidean feature distance).  The tree arguments are accepted for
    signature compatibility and ignored (the search is exhaustive on the device)."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(np.asarray(src_features, dtype=np.float64).T)
    b = np.ascontiguousarray(np.asarray(tgt_features, dtype=np.float64).T)
    ij, d2 = _match(a, b, ctx)
    keep = np.ones(len(ij), dtype=bool)
    if mutual:
        ji, _ = _match(b, a, ctx)
        keep &= ji[ij] == np.arange(len(ij))
    if dist_thres is not None:
        keep &= np.sqrt(d2) < dist_thres
    src = np.flatnonzero(keep)
    return np.stack([src, ij[src]], axis=1).astype(np.int64).reshape(-1, 2)


def _ransac(src_cloud, tgt_cloud, corr, max_distance, edge_similarity, check_distance, max_iteration, confidence, seed, ctx):
    p = L.RansacParams()
    L.lib().pcr_ransac_default_params(C.byref(p))
    p.max_iteration = int(max_iteration)
    p.confidence = float(confidence)
    p.max_distance = float(max_distance)
    p.edge_similarity = float(edge_similarity)
    p.check_distance = 1 if check_distance else 0
    p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    res = L.RansacResult()
    c = np.ascontiguousarray(corr, dtype=np.int32).reshape(-1, 2)
    st = L.lib().pcr_ransac(ctx.handle, src_cloud.handle, tgt_cloud.handle, L.iptr(c), c.shape[0], C.byref(p), C.byref(res))
    L.check(st, ctx.handle)
    return st, res


def registration_ransac_based_on_feature_matching(source, target, source_feature, target_feature, mutual_filter=True,
                                                  max_correspondence_distance=3.0, ransac_n=3, edge_length_similarity=0.9,
                                                  check_distance=True, max_iteration=100000, confidence=0.999, seed=0, ctx=None, evaluate=True):
    """o3d.pipelines.registration.registration_ransac_based_on_feature_matching as called at main.py:73-83
    (point-to-point estimation without scaling, ransac_n = 3, edge-length + distance checkers,
    RANSACConvergenceCriteria(max_iteration, confidence)).  The random stream is counter-based and seeded."""
    if ransac_n != 3:
        raise ValueError("only ransac_n = 3 (main.py:76) is implemented")
    ctx = ctx or default_context()
    ps, pt = _prep_of(source, source_feature, ctx), _prep_of(target, target_feature, ctx)
    if ps is not None and pt is not None:
        # both sides are still on the device: matching, mutual filter and the RANSAC loop in one native call
        p = L.RansacParams()
        L.lib().pcr_ransac_default_params(C.byref(p))
        p.max_iteration, p.confidence, p.max_distance = int(max_iteration), float(confidence), float(max_correspondence_distance)
        p.edge_similarity, p.check_distance, p.seed = float(edge_length_similarity), 1 if check_distance else 0, int(seed) & 0xFFFFFFFFFFFFFFFF
        res = L.RansacResult()
        st = L.lib().pcr_global_registration(ctx.handle, ps.handle, pt.handle, C.byref(p), 1 if mutual_filter else 0, C.byref(res))
        L.check(st, ctx.handle)
        T = np.array(res.T, dtype=np.float64).reshape(4, 4)
        info = {"iterations": res.iterations, "n_valid": res.n_valid, "best_iteration": res.best_iteration, "corr_fitness": res.corr_fitness,
                "corr_rmse": res.corr_rmse, "n_correspondences": res.reserved_i, "status": st}
        if not evaluate:
            return RegistrationResult(T, res.corr_fitness, res.corr_rmse, None, info)
        return _evaluate(ps.cloud(), pt.cloud(), T, max_correspondence_distance, info, ctx)
    fa = np.ascontiguousarray(np.asarray(getattr(source_feature, "data", source_feature), dtype=np.float64).T)
    fb = np.ascontiguousarray(np.asarray(getattr(target_feature, "data", target_feature), dtype=np.float64).T)
    ij, _ = _match(fa, fb, ctx)
    corr = np.stack([np.arange(len(ij)), ij], axis=1)
    if mutual_filter:
        ji, _ = _match(fb, fa, ctx)
        mutual = corr[ji[ij] == np.arange(len(ij))]
        if len(mutual) >= ransac_n * 3:  # Open3D falls back to the one-way set when too few survive
            corr = mutual
    src, own_s = _cloud(source, ctx)
    tgt, own_t = _cloud(target, ctx)
    st, res = _ransac(src, tgt, corr, max_correspondence_distance, edge_length_similarity, check_distance, max_iteration, confidence, seed, ctx)
    T = np.array(res.T, dtype=np.float64).reshape(4, 4)
    info = {"iterations": res.iterations, "n_valid": res.n_valid, "best_iteration": res.best_iteration, "corr_fitness": res.corr_fitness,
            "corr_rmse": res.corr_rmse, "n_correspondences": len(corr), "status": st}
    try:
        return _evaluate(src, tgt, T, max_correspondence_distance, info, ctx) if evaluate else RegistrationResult(T, res.corr_fitness, res.corr_rmse, None, info)
    finally:
        for own in (own_s, own_t):
            if own is not None:
                own.free()


def _evaluate(src, tgt, T, max_correspondence_distance, info, ctx):
    """Open3D's final evaluation over the whole source cloud (GetRegistrationResultAndCorrespondences): fitness, inlier_rmse and
    the correspondence set of the RegistrationResult (main.py:211 itself only reads .transformation)."""
    index = TargetIndex(tgt, ctx=ctx)
    idx, d2 = index.nn1(src, T=T)
    index.free()
    inl = np.flatnonzero(d2 < max_correspondence_distance * max_correspondence_distance)
    fitness = len(inl) / max(src.n, 1)
    rmse = float(np.sqrt(d2[inl].sum() / len(inl))) if len(inl) else 0.0
    cset = np.stack([inl, idx[inl]], axis=1).astype(np.int32)
    return RegistrationResult(T, fitness, rmse, cset, info)


def preprocess_point_cloud(pcd, voxel_size, ctx=None):
    """main.py:33-47 -> (pcd_down with .points/.normals, pcd_fpfh): down-sample at voxel_size, normals with radius
    2*voxel_size / 30 neighbours, FPFH with radius 5*voxel_size / 100 neighbours -- one native call (pcr_preprocess); both
    results stay on the device and come over only when their arrays are looked at."""
    ctx = ctx or default_context()
    full, own = _cloud(pcd, ctx)
    h = C.c_void_p()
    try:
        L.check(L.lib().pcr_preprocess(ctx.handle, full.handle, float(voxel_size), float(voxel_size) * 2, 30, float(voxel_size) * 5, 100, C.byref(h)), ctx.handle)
    finally:
        if own is not None:
            own.free()
    prep = _Prep(ctx, h)
    return PreparedCloud(prep), PreparedFeature(prep)


def prepare_dataset(path_src, path_trg, voxel_size, ctx=None):
    """main.py:50-65 -> (source, target, source_down, target_down, source_fpfh, target_fpfh).  The full-resolution
    normals main.py:54,57 computes are only used by its dead point-to-plane refinement and are not computed here."""
    source = PointCloud(read_bin_velodyne(path_src))
    target = PointCloud(read_bin_velodyne(path_trg))
    source_down, source_fpfh = preprocess_point_cloud(source, voxel_size, ctx=ctx)
    target_down, target_fpfh = preprocess_point_cloud(target, voxel_size, ctx=ctx)
    return source, target, source_down, target_down, source_fpfh, target_fpfh


def execute_global_registration(source_down, target_down, source_fpfh, target_fpfh, voxel_size, seed=0, ctx=None, evaluate=True):
    """main.py:68-84: distance threshold 1.5 * voxel_size, 100000 iterations / 0.999 confidence.  ``evaluate=False`` skips the
    whole-cloud fitness / inlier_rmse of the returned RegistrationResult (main.py:211 reads .transformation only)."""
    return registration_ransac_based_on_feature_matching(
        source_down, target_down, source_fpfh, target_fpfh, True, voxel_size * 1.5, 3, 0.9, True, 100000, 0.999, seed=seed, ctx=ctx, evaluate=evaluate)


def ransac_init(src_cloud, tgt_cloud, voxel_size=2.0, seed=0, ctx=None, detector="voxel", iss_radius=None, iss_count=400,
                return_info=False):
    """icp_template.py:56-110 -> (R (3,3), t (3,1)): feature DETECTION, feature DESCRIPTION (FPFH, 33-d),
    correspondence = mutual nearest features, then RANSAC with procrustes_transformation on 3 samples.

    detector="voxel": every point of the voxel_size down-sampled cloud is a feature point (what main.py:33-84 does).
    detector="iss":   the template's own plan (icp_template.py:56-71 "feature detection"): ISS keypoints
                      (Keypoint_detection_ISS/ISS.py:35-73 on the GPU, pcr_iss) of the down-sampled cloud, at most
                      iss_count + 1 per cloud; descriptors are computed on the whole down-sampled cloud (their support) and
                      only the keypoints' rows are matched and sampled by RANSAC.
    """
    s_down, s_f = preprocess_point_cloud(src_cloud, voxel_size, ctx=ctx)
    t_down, t_f = preprocess_point_cloud(tgt_cloud, voxel_size, ctx=ctx)
    info = {"detector": detector, "n_src": len(s_down.points), "n_tgt": len(t_down.points)}
    if detector == "iss":
        from .iss import iss_keypoints

        r = float(iss_radius) if iss_radius else 2.5 * voxel_size
        ks = np.asarray(iss_keypoints(s_down.points, radius=r, non_max_radius=r, iss_count=iss_count, ctx=ctx), dtype=np.int64)
        kt = np.asarray(iss_keypoints(t_down.points, radius=r, non_max_radius=r, iss_count=iss_count, ctx=ctx), dtype=np.int64)
        info.update(n_src_keypoints=len(ks), n_tgt_keypoints=len(kt), iss_radius=r)
        if len(ks) < 3 or len(kt) < 3:
            raise ValueError(f"ISS found {len(ks)} / {len(kt)} keypoints: need at least 3 per cloud (lower the eigenvalue-ratio "
                             "thresholds, change iss_radius, or use detector='voxel')")
        s_kp, t_kp = PointCloud(np.asarray(s_down.points)[ks]), PointCloud(np.asarray(t_down.points)[kt])
        s_kf, t_kf = Feature(np.asarray(s_f.data)[:, ks]), Feature(np.asarray(t_f.data)[:, kt])
        res = execute_global_registration(s_kp, t_kp, s_kf, t_kf, voxel_size, seed=seed, ctx=ctx)
    elif detector == "voxel":
        res = execute_global_registration(s_down, t_down, s_f, t_f, voxel_size, seed=seed, ctx=ctx)
    else:
        raise ValueError("detector must be 'voxel' or 'iss'")
    T = res.transformation
    R, t = T[:3, :3].copy(), T[:3, 3].reshape(3, 1).copy()
    if return_info:
        info.update(fitness=res.fitness, inlier_rmse=res.inlier_rmse, ransac=res.info)
        return R, t, info
    return R, t
