"""Drop-in for Kdtree_Octree/lesson2/octree.py: octree_construction / octree_knn_search /
octree_radius_search / octree_radius_search_fast.  Same contract as the kd-tree functions;
the boolean the reference returns ("the query ball lies inside the octant", octree.py:187,212)
is reproduced for the ROOT octant: centre = mean of the points, extent = half the largest
axis range (octree.py:319-322)."""
from __future__ import annotations

import numpy as np

from .device import TargetIndex
from .result_set import KNNResultSet, RadiusNNResultSet

__all__ = ["octree_construction", "octree_knn_search", "octree_radius_search", "octree_radius_search_fast", "Octant"]


class Octant:
    """Opaque root octant: keeps the reference's root geometry (octree.py:11-26, 319-322)."""

    def __init__(self, index, center, extent, n):
        self.index = index
        self.center = center
        self.extent = extent
        self.children = [None] * 8
        self.point_indices = range(n)
        self.is_leaf = False


def octree_construction(db_np, leaf_size, min_extent, ctx=None):
    """octree.py:310-328 (leaf_size / min_extent accepted; the device index has no leaves)."""
    db = np.asarray(db_np)[:, :3]
    db_min = np.amin(db, axis=0)
    db_max = np.amax(db, axis=0)
    extent = np.max(db_max - db_min) * 0.5
    center = np.mean(db, axis=0)
    return Octant(TargetIndex(db, kind="grid", ctx=ctx), center, extent, db.shape[0])


def _inside(query, radius, octant):
    """octree.py:106-117."""
    possible_space = np.fabs(np.asarray(query, dtype=np.float64) - octant.center) + radius
    return bool(np.all(possible_space < octant.extent))


def octree_knn_search(root, db, result_set: KNNResultSet, query):
    """octree.py:262-306."""
    if root is None:
        return False
    idx, dist = root.index.knn(np.asarray(query, dtype=np.float64).reshape(1, 3), result_set.capacity)
    for j in range(min(result_set.capacity, root.index.n)):
        result_set.add_point(dist[0, j], int(idx[0, j]))
    return _inside(query, result_set.worstDist(), root)


def octree_radius_search(root, db, result_set: RadiusNNResultSet, query):
    """octree.py:216-259."""
    if root is None:
        return False
    off, idx, dist = root.index.radius(np.asarray(query, dtype=np.float64).reshape(1, 3), result_set.radius)
    result_set.add_points(dist, idx) if hasattr(result_set, "add_points") else [result_set.add_point(d, int(i)) for d, i in zip(dist, idx)]
    return _inside(query, result_set.worstDist(), root)


def octree_radius_search_fast(root, db, result_set: RadiusNNResultSet, query):
    """octree.py:166-212 (same result set as the plain search)."""
    return octree_radius_search(root, db, result_set, query)
