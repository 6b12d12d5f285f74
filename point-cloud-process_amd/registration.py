"""Drop-in call surface of Registration/main.py and Registration/icp_template.py.

Same names, argument meaning, return shapes and error behaviour as the
reference; the association search, the in-place source transform and the
Procrustes moment accumulation run as HIP kernels behind libpcr.so.

Reference lines are cited as file:line relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np

from . import _lib as L
from .device import Context, DeviceCloud, TargetIndex, default_context, icp_device, points_of

__all__ = [
    "PointCloud",
    "KDTreeFlann",
    "icp_point2point",
    "ICP", "coarse_to_fine_icp",
    "find_associations",
    "procrustes_transformation",
    "rotmat2quaternion",
    "homo2tq",
    "copysign",
    "read_bin_velodyne",
    "read_oxford_bin",
    "read_velodyne_bin",
    "write_reg_result",
]


class PointCloud:
    """Minimal stand-in for o3d.geometry.PointCloud: ``.points`` (N,3) float64 and an
    in-place ``.transform(T)`` (what Registration/main.py:52-56,110 uses)."""

    def __init__(self, points=None):
        self.points = None if points is None else np.array(points, dtype=np.float64)[:, :3]

    def transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        self.points = self.points @ T[:3, :3].T + T[:3, 3]
        return self


def _set_points(obj, pts):
    """Write transformed points back into a cloud-like object (main.py:110 mutates `source`)."""
    if hasattr(obj, "points"):
        try:
            obj.points = pts
            return
        except Exception:
            pass
        try:  # Open3D Vector3dVector supports slice assignment through numpy view
            np.asarray(obj.points)[:] = pts
        except Exception:
            pass
    elif isinstance(obj, np.ndarray) and obj.dtype == np.float64 and obj.shape == pts.shape:
        obj[:] = pts


class KDTreeFlann:
    """o3d.geometry.KDTreeFlann(target) (main.py:105): device-resident exact NN index."""

    def __init__(self, target, kind="grid", cell=0.0, ctx=None):
        self.index = target if isinstance(target, TargetIndex) else TargetIndex(points_of(target), kind=kind, cell=cell, ctx=ctx)

    def search_knn_vector_3d(self, query, k):
        """-> [k, idx list, SQUARED distance list] like Open3D (main.py:117-119)."""
        q = np.asarray(query, dtype=np.float64).reshape(1, 3)
        idx, dist = self.index.knn(q, int(k))
        kk = min(int(k), self.index.n)
        return [kk, idx[0, :kk].tolist(), (dist[0, :kk] ** 2).tolist()]

    def search_radius_vector_3d(self, query, radius):
        """-> [k, idx list, SQUARED distance list], ascending distance (Final_Project/scripts/extract.py:519-524)."""
        q = np.asarray(query, dtype=np.float64).reshape(1, 3)
        _, idx, dist = self.index.radius(q, float(radius))
        return [len(idx), idx.tolist(), (dist ** 2).tolist()]

    def search_hybrid_vector_3d(self, query, radius, max_nn):
        """-> the max_nn nearest of the radius result (Open3D KDTreeSearchParamHybrid)."""
        k, idx, d2 = self.search_radius_vector_3d(query, radius)
        k = min(k, int(max_nn))
        return [k, idx[:k], d2[:k]]


def _as_index(tgt):
    if isinstance(tgt, TargetIndex):
        return tgt, False
    if isinstance(tgt, KDTreeFlann):
        return tgt.index, False
    return TargetIndex(points_of(tgt)), True


def icp_point2point(source, target, transformation, *, nn="grid", cell=0.0, ctx=None, return_info=False):
    """Registration/main.py:97-156, same semantics bit for bit in control flow:

    * parameters max_iteration=100, R/t_diff_thres=0.5, dist_thres=5 on the SQUARED distance (main.py:98-103);
    * every iteration first applies the current ``transformation`` to ``source`` IN PLACE (main.py:110);
    * exact 1-NN, keep pairs with d2 < 5 (main.py:116-121); fewer than 3 pairs prints
      "ICP failed, cannot find enough associations!" and stops (main.py:125-127);
    * R = U V^T with no reflection fix, t = mean(B - R A) (main.py:131-141);
    * the first t_diff is the Frobenius norm of the (3,1)-(3,) broadcast (main.py:100,150);
    * returns the LAST incremental 4x4, not the composed transform (main.py:143-146,156).
    """
    ctx = ctx or default_context()
    T0 = np.array(transformation, dtype=np.float64).reshape(4, 4)
    src_dev = DeviceCloud.upload(points_of(source), ctx)
    if isinstance(target, (TargetIndex, KDTreeFlann)):
        index, own = _as_index(target)
    else:
        index, own = TargetIndex(points_of(target), kind=nn, cell=cell, ctx=ctx), True
    try:
        res = icp_device(src_dev, index, T0, mode="compat", max_iter=100, r_thres=0.5, t_thres=0.5, max_d2=5.0)
        if res["status"] == L.PCR_E_TOO_FEW_ASSOC:
            print("ICP failed, cannot find enough associations!")
        _set_points(source, src_dev.download())
    finally:
        src_dev.free()
        if own:
            index.free()
    return (res["T"], res) if return_info else res["T"]


def ICP(src_cloud, tgt_cloud, *, init=None, max_iteration=50, R_diff_thres=1e-5, t_diff_thres=1e-5, dist_thres=5.0,
        r_metric="geodesic", nn="grid", cell=0.0, ctx=None):
    """Registration/icp_template.py:128-200 filled in: returns ``(homo_mat_total, log)``.

    The template leaves max_iteration / thresholds / dist_thres as ``None``
    (icp_template.py:157-159,115); the defaults here are tight thresholds so the
    loop actually converges, R_diff is the geodesic angle the template links to
    (icp_template.py:184).  ``init`` replaces ``ransac_init`` (icp_template.py:145-152):
    a 4x4 initial guess applied to the source and folded into ``homo_mat_total``; ``init="ransac"`` runs
    ``ransac_init`` itself like the template's ``init_use_ransac = True``.
    """
    ctx = ctx or default_context()
    if isinstance(init, str):
        if init != "ransac":
            raise ValueError("init must be a 4x4 transform, None or 'ransac'")
        from .global_registration import ransac_init

        R_init, t_init = ransac_init(src_cloud, tgt_cloud, ctx=ctx)
        init = np.eye(4)
        init[:3, :3], init[:3, 3] = R_init, t_init[:, 0]
    T0 = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    src_dev = DeviceCloud.upload(points_of(src_cloud), ctx)
    if isinstance(tgt_cloud, (TargetIndex, KDTreeFlann)):
        index, own = _as_index(tgt_cloud)
    else:
        index, own = TargetIndex(points_of(tgt_cloud), kind=nn, cell=cell, ctx=ctx), True
    try:
        res = icp_device(src_dev, index, T0, mode="total", max_iter=max_iteration, r_thres=R_diff_thres, t_thres=t_diff_thres,
                         max_d2=dist_thres, r_metric=r_metric)
        if res["status"] == L.PCR_E_TOO_FEW_ASSOC:
            print("ICP failed, cannot find enough associations!")
    finally:
        src_dev.free()
        if own:
            index.free()
    log = defaultdict(lambda: [])
    log["R_diff"] = list(res["R_diff"])
    log["t_diff"] = list(res["t_diff"])
    log["n_assoc"] = [res["n_assoc"]]
    log["cost"] = [res["cost"]]
    return res["T"], log


def find_associations(src_points, tgt_tree=None, dist_thres=5.0):
    """icp_template.py:113-126: ``src_points`` is (3,N); returns a (K,2) int array of
    (src_idx, tgt_idx) for every source point whose exact nearest target is closer than
    ``dist_thres`` (squared distance, strict <, like main.py:119)."""
    src = np.ascontiguousarray(np.asarray(src_points, dtype=np.float64).T)
    index, own = _as_index(tgt_tree)
    idx, _ = index.nn1(src, None, dist_thres)
    if own:
        index.free()
    keep = np.nonzero(idx >= 0)[0]
    return np.stack([keep, idx[keep].astype(np.int64)], axis=1) if keep.size else np.zeros((0, 2), dtype=np.int64)


def procrustes_transformation(A, B):
    """icp_template.py:43-54 / main.py:131-141: A, B are (3,K); returns R (3,3), t (3,1), cost."""
    A = L.as_f64(A)
    B = L.as_f64(B)
    if A.shape != B.shape or A.ndim != 2 or A.shape[0] != 3:
        raise ValueError("A and B must both be (3, K)")
    R = np.zeros(9)
    t = np.zeros(3)
    cost = C.c_double()
    L.check(L.lib().pcr_procrustes(L.dptr(A), L.dptr(B), A.shape[1], L.dptr(R), L.dptr(t), C.byref(cost)))
    return R.reshape(3, 3), t.reshape(3, 1), cost.value


def coarse_to_fine_icp(src_cloud, tgt_cloud, leaves=(1.0, 0.4, 0.0), *, init=None, max_iteration=30, R_diff_thres=1e-5,
                       t_diff_thres=1e-5, dist_thres=5.0, ctx=None):
    """BASELINE config 5's refinement: ICP (icp_template.py:128-200 semantics, composed transform) on voxel-filtered copies
    of both clouds, coarsest leaf first, each level starting from the previous level's transform; leaf 0 = full
    resolution.  Everything stays on the device between levels.  Returns (homo_mat_total, [log per level])."""
    from .voxel_filter import voxel_filter_device

    ctx = ctx or default_context()
    T = np.eye(4) if init is None else np.array(init, dtype=np.float64).reshape(4, 4)
    src_full = DeviceCloud.upload(points_of(src_cloud), ctx)
    tgt_full = DeviceCloud.upload(points_of(tgt_cloud), ctx)
    logs = []
    leaves = tuple(leaves)
    src_alive = True
    try:
        for k, leaf in enumerate(leaves):
            if leaf > 0:
                s = voxel_filter_device(src_full, leaf)
            elif k == len(leaves) - 1:
                s = src_full            # the last level registers the full cloud itself (ICP moves its source in place: nobody needs it afterwards);
                src_alive = False       # a copy through the host was 2-3 ms of a 10-ms refinement at 1 M points
            else:
                s = DeviceCloud.upload(src_full.download(), ctx)
            t = voxel_filter_device(tgt_full, leaf) if leaf > 0 else tgt_full
            index = TargetIndex(t, ctx=ctx)
            try:
                res = icp_device(s, index, T, mode="total", max_iter=max_iteration, r_thres=R_diff_thres, t_thres=t_diff_thres,
                                 max_d2=dist_thres, r_metric="geodesic")
            finally:
                index.free()
                s.free()
                if t is not tgt_full:
                    t.free()
            T = res["T"]
            logs.append({"leaf": leaf, "iters": res["iters"], "n_assoc": res["n_assoc"], "mean_d2": res["mean_d2"]})
    finally:
        if src_alive:
            src_full.free()
        tgt_full.free()
    return T, logs


def copysign(v, s):
    """main.py:176-180."""
    if v * s < 0:
        v *= -1
    return v


def homo2tq(homo_mat):
    """main.py:170-174 -> (tx, ty, tz, qw, qx, qy, qz)."""
    T = L.as_f64(homo_mat).reshape(16)
    out = np.zeros(7)
    L.check(L.lib().pcr_homo2tq(L.dptr(T), L.dptr(out)))
    return tuple(float(x) for x in out)


def rotmat2quaternion(m):
    """main.py:158-168 -> (qw, qx, qy, qz)."""
    T = np.eye(4)
    T[:3, :3] = np.asarray(m, dtype=np.float64)[:3, :3]
    return homo2tq(T)[3:]


# ------------------------------------------------------------------- I/O
def read_bin_velodyne(path):
    """main.py:10-17: 6 x float32 records -> (N,3) float32 (np.fromfile instead of struct.iter_unpack)."""
    data = np.fromfile(path, dtype=np.float32)
    return np.ascontiguousarray(data[: (data.size // 6) * 6].reshape(-1, 6)[:, :3])


def read_oxford_bin(bin_path):
    """icp_template.py:11-17: -> (6,N) float32 [x,y,z,nx,ny,nz]."""
    data_np = np.fromfile(bin_path, dtype=np.float32)
    return np.transpose(np.reshape(data_np, (int(data_np.shape[0] / 6), 6)))


def read_velodyne_bin(path):
    """Kdtree_Octree/lesson2/benchmark.py:16-27: KITTI 4 x float32 records.  The reference
    returns the TRANSPOSED array, shape (3,N); kept for drop-in behaviour (use ``.T`` for (N,3))."""
    data = np.fromfile(path, dtype=np.float32)
    return np.ascontiguousarray(data[: (data.size // 4) * 4].reshape(-1, 4)[:, :3]).T


def write_reg_result(path, rows):
    """main.py:220-222: rows = (idx1, idx2, tx, ty, tz, qw, qx, qy, qz)."""
    np.savetxt(path, np.asarray(rows, dtype=np.float64), delimiter=",", header="idx1,idx2,t_x,t_y,t_z,q_w,q_x,q_y,q_z",
               fmt="%i,%i,%f,%f,%f,%f,%f,%f,%f")
