"""Drop-in for Pca_and_Voxel_filter/voxel_filter.py: ``voxel_filter(point_cloud, leaf_size, type)``.

Same quirks as the reference: D = (max-min)//leaf with no +1, float64 keys
h = hx + hy*Dx + hz*Dx*Dy, stable order inside a voxel, the voxel with the largest key is never
emitted (voxel_filter.py:42-51), "centroid" = np.mean over the group (bitwise), "random" = one
member of the group (the reference uses the unseeded global ``random``; here an explicit seed).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .device import DeviceCloud, default_context

__all__ = ["voxel_filter", "voxel_keys", "voxel_filter_device"]

_MODES = {"centroid": 0, "random": 1}


def _as_points(point_cloud):
    pc = np.asarray(point_cloud)
    if pc.ndim != 2 or pc.shape[1] < 3:
        raise ValueError(f"expected an (N,3) array, got {pc.shape}")
    if pc.shape[0] == 0:
        # the reference raises on an empty cloud too (np.max of an empty array, voxel_filter.py:20)
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    return np.ascontiguousarray(pc[:, :3], dtype=np.float64)


def voxel_keys(point_cloud, leaf_size, ctx=None):
    """Per-point voxel key h (float64, bit-exact with voxel_filter.py:20-33) and D = (Dx, Dy, Dz)."""
    ctx = ctx or default_context()
    pc = _as_points(point_cloud)
    h = np.empty(pc.shape[0], dtype=np.float64)
    D = np.zeros(3)
    L.check(L.lib().pcr_voxel_keys(ctx.handle, L.dptr(pc), pc.shape[0], float(leaf_size), L.dptr(h), L.dptr(D)), ctx.handle)
    return h, D


def voxel_filter(point_cloud, leaf_size, type, seed=0, ctx=None):  # noqa: A002 - the reference names the argument `type`
    """voxel_filter.py:10-68 -> (occupied voxels - 1, 3) float64."""
    if type not in _MODES:
        return np.array([], dtype=np.float64)  # the reference falls through both `if`s and returns an empty array
    ctx = ctx or default_context()
    pc = _as_points(point_cloud)
    out = np.empty((pc.shape[0], 3), dtype=np.float64)
    n_out = C.c_int64()
    L.check(L.lib().pcr_voxel_filter(ctx.handle, L.dptr(pc), pc.shape[0], float(leaf_size), _MODES[type], C.c_uint64(int(seed)),
                                     L.dptr(out), C.byref(n_out)), ctx.handle)
    res = out[: n_out.value].copy()
    return res if n_out.value else np.array([], dtype=np.float64)


def voxel_filter_device(cloud: DeviceCloud, leaf_size, type="centroid", seed=0):  # noqa: A002
    """Device-resident variant for the downsample -> ICP pipeline (BASELINE config 3)."""
    h = C.c_void_p()
    L.check(L.lib().pcr_voxel_filter_cloud(cloud.ctx.handle, cloud.handle, float(leaf_size), _MODES[type], C.c_uint64(int(seed)), C.byref(h)),
            cloud.ctx.handle)
    n = L.lib().pcr_cloud_size(h)
    return DeviceCloud(cloud.ctx, h, n)
