"""PCA of a cloud and per-point surface normals: Pca_and_Voxel_filter/pca_normal.py.

PCA(data, correlation=False, sort=True) keeps the reference's signature (pca_normal.py:10) -- `correlation`
is accepted and, exactly like the reference (which calls np.cov in both cases), ignored.  The moments are
reduced on the device; eigenvectors are defined up to sign (the reference's come from LAPACK)."""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .device import DeviceCloud, default_context, points_of

__all__ = ["PCA", "estimate_normals"]


def PCA(data, correlation=False, sort=True, ctx=None):  # noqa: N802 (reference name)
    """Returns (eigenvalues (3,), eigenvectors (3,3) with eigenvectors as COLUMNS), eigenvalues descending
    (pca_normal.py:26-34).  sort=False returns them ascending, the order np.linalg.eigh gives."""
    ctx = ctx or default_context()
    own = None
    cloud = data
    if not isinstance(data, DeviceCloud):
        cloud = own = DeviceCloud.upload(points_of(data), ctx)
    ev = np.empty(3, dtype=np.float64)
    vec = np.empty((3, 3), dtype=np.float64)
    L.check(L.lib().pcr_pca(ctx.handle, cloud.handle, L.dptr(ev), L.dptr(vec), None), ctx.handle)
    if own is not None:
        own.free()
    if not sort:
        ev, vec = ev[::-1].copy(), vec[:, ::-1].copy()
    return ev, vec


def estimate_normals(points, k=5, ctx=None, return_details=False):
    """The loop of pca_normal.py:85-90 for every point at once: normal_i = eigenvector of the smallest
    eigenvalue of np.cov of the k nearest neighbours of point i (itself included).  Returns (N,3) float64;
    with return_details also (eigenvalues (N,3) descending, neighbour indices (N,k))."""
    ctx = ctx or default_context()
    own = None
    cloud = points
    if not isinstance(points, DeviceCloud):
        cloud = own = DeviceCloud.upload(points_of(points), ctx)
    n = cloud.n
    normals = np.empty((n, 3), dtype=np.float64)
    ev = np.empty((n, 3), dtype=np.float64) if return_details else None       # (only read back when asked for)
    nbr = np.empty((n, int(k)), dtype=np.int32) if return_details else None
    L.check(L.lib().pcr_normals(ctx.handle, cloud.handle, int(k), L.dptr(normals), L.dptr(ev) if return_details else None,
                                L.iptr(nbr) if return_details else None), ctx.handle)
    if own is not None:
        own.free()
    return (normals, ev, nbr) if return_details else normals
