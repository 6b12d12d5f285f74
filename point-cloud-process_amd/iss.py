"""ISS keypoints: the script body of Keypoint_detection_ISS/ISS.py:35-73 as a function."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .device import DeviceCloud, default_context, points_of

__all__ = ["iss_keypoints"]


def iss_keypoints(points, radius=0.5, lambda21=0.5, lambda32=0.5, non_max_radius=0.5, iss_count=20, ctx=None, return_details=False):
    """Defaults are the reference's hyper-parameters (ISS.py:20-27).  Returns the list of keypoint
    indices (at most iss_count + 1, ISS.py:72-73); with return_details also (lambdas (N,3), counts (N,))."""
    ctx = ctx or default_context()
    own = None
    cloud = points
    if not isinstance(points, DeviceCloud):
        cloud = own = DeviceCloud.upload(points_of(points), ctx)
    n = cloud.n
    # the per-point eigenvalues and neighbour counts (24 + 4 bytes per point over PCIe: most of the call at 1 M points) are only
    # read back when asked for; the keypoints need the candidates' lambda_3 only, which the library fetches as a compact list
    lam = np.empty((n, 3), dtype=np.float64) if return_details else None
    counts = np.empty(n, dtype=np.int32) if return_details else None
    kp = np.empty(int(iss_count) + 1, dtype=np.int32)
    nk = C.c_int()
    L.check(L.lib().pcr_iss(ctx.handle, cloud.handle, float(radius), float(lambda21), float(lambda32), float(non_max_radius), int(iss_count),
                            L.dptr(lam) if return_details else None, L.iptr(counts) if return_details else None, L.iptr(kp), C.byref(nk)), ctx.handle)
    if own is not None:
        own.free()
    idx = [int(i) for i in kp[: nk.value]]
    return (idx, lam, counts) if return_details else idx
