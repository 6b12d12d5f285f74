"""Drop-in for Kdtree_Octree/lesson2/kdtree.py: kdtree_construction / kdtree_knn_search /
kdtree_radius_search with the reference's signatures; the "root" is an opaque handle to the
device-resident voxel-hash index, the result sets are filled through their own add_point()
in ascending distance so their contents equal the reference's (comparison_counter excepted:
it counts the neighbours reported, not tree-node visits).

Single-query calls are launch-latency bound by construction; ``*_search_batch`` are the
additive batched forms.
"""
from __future__ import annotations

import numpy as np

from .device import TargetIndex
from .result_set import KNNResultSet, RadiusNNResultSet

__all__ = ["kdtree_construction", "kdtree_knn_search", "kdtree_radius_search", "knn_search_batch", "radius_search_batch", "Node"]


class Node:
    """Opaque root (kdtree.py:10-33 is a Python tree node; here a device index handle)."""

    def __init__(self, index: TargetIndex, leaf_size=None):
        self.index = index
        self.leaf_size = leaf_size
        self.axis = 0
        self.value = None
        self.left = None
        self.right = None
        self.point_indices = None

    def is_leaf(self):
        return False


def kdtree_construction(db_np, leaf_size, ctx=None):
    """kdtree.py:119-137: db_np is (N, dim>=3); leaf_size is accepted and ignored (no leaves here)."""
    return Node(TargetIndex(np.asarray(db_np)[:, :3], kind="grid", ctx=ctx), leaf_size)


def _index_of(root):
    return root.index if hasattr(root, "index") else root


def kdtree_knn_search(root, db, result_set: KNNResultSet, query):
    """kdtree.py:141-172.  Returns False like the reference."""
    if root is None:
        return False
    idx, dist = _index_of(root).knn(np.asarray(query, dtype=np.float64).reshape(1, 3), result_set.capacity)
    n = min(result_set.capacity, _index_of(root).n)
    for j in range(n):
        result_set.add_point(dist[0, j], int(idx[0, j]))
    return False


def kdtree_radius_search(root, db, result_set: RadiusNNResultSet, query):
    """kdtree.py:176-208.  Returns False like the reference."""
    if root is None:
        return False
    off, idx, dist = _index_of(root).radius(np.asarray(query, dtype=np.float64).reshape(1, 3), result_set.radius)
    result_set.add_points(dist, idx) if hasattr(result_set, "add_points") else [result_set.add_point(d, int(i)) for d, i in zip(dist, idx)]
    return False


def knn_search_batch(root, queries, k):
    """(Q,3) queries -> (idx (Q,k) int32, dist (Q,k) float64), ascending Euclidean distances."""
    return _index_of(root).knn(np.asarray(queries, dtype=np.float64), int(k))


def radius_search_batch(root, queries, radius):
    """(Q,3) queries -> (offsets (Q+1,), idx (M,), dist (M,)); per query ascending distance."""
    return _index_of(root).radius(np.asarray(queries, dtype=np.float64), float(radius))
