"""DBSCAN over the device radius search: Cluster_dbscan/dbscan.py (same class, same attributes)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .device import DeviceCloud, default_context, points_of

__all__ = ["DBSCAN"]


class DBSCAN(object):
    """dbscan.py:4-38: ``DBSCAN(radius, Min_Pts).fit(data)`` then ``.predict()`` / ``.labels_`` (int32, -1 = noise).
    Labels equal the reference's, cluster numbering and noise quirks included (see include/pcr.h:pcr_dbscan)."""

    def __init__(self, radius=0.5, Min_Pts=10, ctx=None):
        self.radius = radius
        self.Min_Pts = Min_Pts
        self.labels_ = None
        self._ctx = ctx

    def fit(self, data):
        ctx = self._ctx or default_context()
        own = None
        cloud = data
        if not isinstance(data, DeviceCloud):
            cloud = own = DeviceCloud.upload(np.ascontiguousarray(points_of(data)[:, :3], dtype=np.float64), ctx)
        labels = np.empty(cloud.n, dtype=np.int32)
        nc = C.c_int32()
        L.check(L.lib().pcr_dbscan(ctx.handle, cloud.handle, float(self.radius), int(self.Min_Pts), L.iptr(labels), C.byref(nc)), ctx.handle)
        if own is not None:
            own.free()
        self.labels_ = labels
        self.n_clusters_ = nc.value

    def predict(self):
        return self.labels_
