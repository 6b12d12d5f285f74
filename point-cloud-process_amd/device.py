"""Thin object wrappers over the C-ABI handles (context, device cloud, target index)."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib as L

_default_ctx = {}
_lock = threading.Lock()


class Context:
    """One device + one HIP stream (pcr_ctx).  Not thread-safe, like the C object."""

    def __init__(self, device=0, shared=False):
        """`shared`: other contexts keep the same device busy while this one registers pairs (see pcr_ctx_set_shared)."""
        self._h = C.c_void_p()
        self.device = int(device)
        L.check(L.lib().pcr_ctx_create(self.device, C.byref(self._h)))
        if shared:
            self.set_shared(True)

    def set_shared(self, shared=True):
        L.check(L.lib().pcr_ctx_set_shared(self.handle, 1 if shared else 0), self.handle)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context destroyed")
        return self._h

    def sync(self):
        L.check(L.lib().pcr_ctx_sync(self.handle), self.handle)

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        hbm = C.c_int64()
        L.check(L.lib().pcr_ctx_device_info(self.handle, name, C.byref(cu), C.byref(hbm)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": hbm.value}

    def timer_start(self):
        L.check(L.lib().pcr_timer_start(self.handle), self.handle)

    def timer_stop_ms(self):
        ms = C.c_double()
        L.check(L.lib().pcr_timer_stop_ms(self.handle, C.byref(ms)), self.handle)
        return ms.value

    def profile(self, on=True):
        L.check(L.lib().pcr_profile_enable(self.handle, 1 if on else 0), self.handle)

    def profile_read(self):
        ms = np.zeros(4)
        n = C.c_int()
        L.check(L.lib().pcr_profile_read(self.handle, L.dptr(ms), C.byref(n)), self.handle)
        return ms, n.value

    def pass_log(self):
        """Per-pass log of the last device-resident ICP loop on this context: {"tile_us": [...], "drain_us": [...], "items": [...]}."""
        t, d, it, n = np.zeros(L.PCR_ICP_MAX_LOG), np.zeros(L.PCR_ICP_MAX_LOG), np.zeros(L.PCR_ICP_MAX_LOG, dtype=np.int64), C.c_int()
        h = np.zeros(6)
        L.check(L.lib().pcr_icp_pass_log(self.handle, L.PCR_ICP_MAX_LOG, L.dptr(t), L.dptr(d), L.lptr(it), C.byref(n), L.dptr(h)), self.handle)
        k = min(n.value, L.PCR_ICP_MAX_LOG)
        return {"tile_us": t[:k].tolist(), "drain_us": d[:k].tolist(), "items": it[:k].tolist(),
                "host_us": {"setup": h[0], "enqueue_first_chunk": h[1], "waiting_for_device": h[2], "whole_loop": h[3], "events_start_to_last_kernel": h[4], "last_wait": h[5]}}

    def search_stats(self):
        """Diagnostics: {"brute_fallback": queries the last brute-force search re-did by the exact sweep, "arena_allocations" /
        "arena_allocation_us" / "arenas": device arenas this context has allocated so far, the host time that took, arenas held}."""
        out = np.zeros(4, dtype=np.int64)
        L.check(L.lib().pcr_search_stats(self.handle, L.lptr(out)), self.handle)
        return {"brute_fallback": int(out[0]), "arena_allocations": int(out[1]), "arena_allocation_us": int(out[2]), "arenas": int(out[3])}

    def close(self):
        if self._h:
            L.lib().pcr_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def default_context(device=0):
    with _lock:
        ctx = _default_ctx.get(device)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[device] = ctx
        return ctx


def points_of(obj):
    """(N,3) array of anything cloud-like: an object with ``.points`` (Open3D style,
    Registration/main.py:52) or an array-like (N,3)."""
    pts = getattr(obj, "points", obj)
    arr = np.asarray(pts)
    if arr.ndim != 2 or arr.shape[1] < 3:
        raise ValueError(f"expected an (N,3) point array, got shape {arr.shape}")
    return arr


class DeviceCloud:
    """Device-resident float64 cloud (pcr_cloud)."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = int(n)

    @classmethod
    def upload(cls, points, ctx=None):
        ctx = ctx or default_context()
        arr = points_of(points)
        h = C.c_void_p()
        if arr.shape[0] == 0:
            raise L.PcrError(L.PCR_E_EMPTY)
        if arr.dtype == np.float32:
            a = np.ascontiguousarray(arr)
            st = L.lib().pcr_cloud_upload_f32(ctx.handle, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1], C.byref(h))
        else:
            a = np.ascontiguousarray(arr, dtype=np.float64)
            st = L.lib().pcr_cloud_upload_f64(ctx.handle, L.dptr(a), a.shape[0], a.shape[1], C.byref(h))
        L.check(st, ctx.handle)
        return cls(ctx, h, arr.shape[0])

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("cloud freed")
        return self._h

    def download(self):
        out = np.empty((self.n, 3), dtype=np.float64)
        L.check(L.lib().pcr_cloud_download_f64(self.ctx.handle, self.handle, L.dptr(out)), self.ctx.handle)
        return out

    def transform(self, T):
        T = L.as_f64(T).reshape(16)
        L.check(L.lib().pcr_cloud_transform(self.ctx.handle, self.handle, L.dptr(T)), self.ctx.handle)
        return self

    def prepare(self, index):
        """Lay the records out for queries against `index` (done lazily otherwise)."""
        L.check(L.lib().pcr_cloud_prepare(self.ctx.handle, self.handle, index.handle), self.ctx.handle)
        return self

    def free(self):
        if self._h:
            L.lib().pcr_cloud_free(self.ctx.handle, self._h)
            self._h = C.c_void_p()

    def __len__(self):
        return self.n

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


class TargetIndex:
    """Device-resident NN index over a target cloud (pcr_index): the stand-in for
    o3d.geometry.KDTreeFlann(target) (Registration/main.py:105) and for the roots
    returned by kdtree_construction / octree_construction."""

    KINDS = {"grid": L.PCR_INDEX_GRID, "brute": L.PCR_INDEX_BRUTE}

    def __init__(self, target, kind="grid", cell=0.0, ctx=None):
        ctx = ctx or (target.ctx if isinstance(target, DeviceCloud) else default_context())
        self.ctx = ctx
        self._own_cloud = None
        if not isinstance(target, DeviceCloud):
            target = DeviceCloud.upload(target, ctx)
            self._own_cloud = target
        self.n = target.n
        self.kind = kind
        h = C.c_void_p()
        L.check(L.lib().pcr_index_build(ctx.handle, target.handle, self.KINDS[kind], float(cell), C.byref(h)), ctx.handle)
        self._h = h
        if self._own_cloud is not None:
            self._own_cloud.free()
            self._own_cloud = None

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("index freed")
        return self._h

    @property
    def cell(self):
        return L.lib().pcr_index_cell(self.handle)

    def nn1(self, queries, T=None, max_d2=0.0):
        """Exact nearest target of every query -> (idx int32 (Q,), d2 float64 (Q,)); idx -1 = gated out."""
        own = None
        if not isinstance(queries, DeviceCloud):
            queries = own = DeviceCloud.upload(queries, self.ctx)
        idx = np.empty(queries.n, dtype=np.int32)
        d2 = np.empty(queries.n, dtype=np.float64)
        Tp = None
        if T is not None:
            Tc = L.as_f64(T).reshape(16)
            Tp = L.dptr(Tc)
        L.check(L.lib().pcr_nn1(self.ctx.handle, self.handle, queries.handle, Tp, float(max_d2), L.iptr(idx), L.dptr(d2)), self.ctx.handle)
        if own is not None:
            own.free()
        return idx, d2

    def knn(self, queries, k):
        q = L.as_f64(np.atleast_2d(queries))[:, :3].copy()
        idx = np.empty((q.shape[0], k), dtype=np.int32)
        dist = np.empty((q.shape[0], k), dtype=np.float64)
        L.check(L.lib().pcr_knn(self.ctx.handle, self.handle, L.dptr(q), q.shape[0], int(k), L.iptr(idx), L.dptr(dist)), self.ctx.handle)
        return idx, dist

    def radius(self, queries, r):
        """-> (offsets int64 (Q+1,), idx int32 (M,), dist float64 (M,)), ascending distance per query."""
        q = L.as_f64(np.atleast_2d(queries))[:, :3].copy()
        nq = q.shape[0]
        counts = np.zeros(nq, dtype=np.int64)
        if 1 <= nq <= 8:   # the reference's one-query-per-call API: one launch, results straight into pinned memory
            cap = 8192
            idx = np.empty(nq * cap, dtype=np.int32)
            dist = np.empty(nq * cap, dtype=np.float64)
            st = L.lib().pcr_radius_small(self.ctx.handle, self.handle, L.dptr(q), nq, float(r), cap, L.lptr(counts), L.iptr(idx), L.dptr(dist))
            if st == L.PCR_OK:
                offsets = np.zeros(nq + 1, dtype=np.int64)
                np.cumsum(counts, out=offsets[1:])
                m = int(offsets[-1])
                return offsets, idx[:m], dist[:m]
            if st != L.PCR_E_UNSUPPORTED:
                L.check(st, self.ctx.handle)
        L.check(L.lib().pcr_radius(self.ctx.handle, self.handle, L.dptr(q), nq, float(r), L.lptr(counts), None, None, None), self.ctx.handle)
        offsets = np.zeros(nq + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        m = int(offsets[-1])
        idx = np.empty(max(m, 1), dtype=np.int32)
        dist = np.empty(max(m, 1), dtype=np.float64)
        L.check(L.lib().pcr_radius(self.ctx.handle, self.handle, L.dptr(q), nq, float(r), L.lptr(counts), L.lptr(offsets), L.iptr(idx), L.dptr(dist)), self.ctx.handle)
        return offsets, idx[:m], dist[:m]

    def moments(self, source, T=None, max_d2=5.0):
        """One fused association+accumulation pass -> (moments[18], origin[3], sum_d2)."""
        own = None
        if not isinstance(source, DeviceCloud):
            source = own = DeviceCloud.upload(source, self.ctx)
        m = np.zeros(18)
        o = np.zeros(3)
        s = C.c_double()
        Tc = L.as_f64(np.eye(4) if T is None else T).reshape(16)
        L.check(L.lib().pcr_icp_moments(self.ctx.handle, source.handle, self.handle, L.dptr(Tc), float(max_d2), L.dptr(m), L.dptr(o), C.byref(s)), self.ctx.handle)
        if own is not None:
            own.free()
        return m, o, s.value

    def free(self):
        if self._h:
            L.lib().pcr_index_free(self.ctx.handle, self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass


def icp_device(source: DeviceCloud, index: TargetIndex, T0, *, mode="compat", max_iter=100, r_thres=0.5, t_thres=0.5,
               max_d2=5.0, r_metric="frobenius", min_iter=0):
    """pcr_icp on device-resident inputs -> dict with T, T_total, iters, status, log, timings."""
    p = L.IcpParams()
    L.lib().pcr_icp_default_params(C.byref(p))
    p.max_iter = int(max_iter)
    p.r_thres = float(r_thres)
    p.t_thres = float(t_thres)
    p.max_d2 = float(max_d2)
    p.mode = L.PCR_ICP_COMPAT_MAIN if mode == "compat" else L.PCR_ICP_TOTAL
    p.r_metric = L.PCR_RMETRIC_GEODESIC if r_metric == "geodesic" else L.PCR_RMETRIC_FROBENIUS
    p.min_iter = int(min_iter)
    res = L.IcpResult()
    T0c = L.as_f64(T0).reshape(16)
    st = L.lib().pcr_icp(index.ctx.handle, source.handle, index.handle, C.byref(p), L.dptr(T0c), C.byref(res))
    L.check(st, index.ctx.handle)
    n = res.iters
    return {
        "T": np.array(res.T[:]).reshape(4, 4),
        "T_total": np.array(res.T_total[:]).reshape(4, 4),
        "iters": n,
        "status": res.status,
        "n_assoc": res.n_assoc,
        "cost": res.cost,
        "mean_d2": res.mean_d2,
        "R_diff": list(res.r_diff[:n]),
        "t_diff": list(res.t_diff[:n]),
        "device_ms": res.device_ms,
        "nn_kernel_ms": res.nn_kernel_ms,
        "nn_launches": res.nn_launches,
    }
