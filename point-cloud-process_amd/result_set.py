"""Result containers with the public surface of Kdtree_Octree/lesson2/result_set.py.

``KNNResultSet(capacity)`` / ``RadiusNNResultSet(radius)`` expose ``add_point``, ``worstDist``,
``size``, ``full`` (kNN only), ``dist_index_list`` (objects with ``.distance`` / ``.index``),
``count``, ``comparison_counter``.  Observable behaviour follows result_set.py:15-93:

* kNN: a distance strictly greater than the current worst is rejected, an equal one is accepted
  and placed AFTER the entries it ties with -- when the set is already full it therefore replaces
  the last entry (result_set.py:37-50); unfilled slots read (1e10, 0) (result_set.py:19-22);
* radius: ``dist > radius`` is rejected, i.e. the ball is closed (result_set.py:80).

The bookkeeping is a bisect over a plain list of keys rather than the reference's shifting loop.
"""
from __future__ import annotations

from bisect import bisect_right

_UNSET_DISTANCE = 1e10


class DistIndex:
    __slots__ = ("distance", "index")

    def __init__(self, distance, index):
        self.distance = distance
        self.index = index

    def __lt__(self, other):
        return self.distance < other.distance

    def __repr__(self):
        return f"DistIndex({self.distance!r}, {self.index!r})"


class KNNResultSet:
    def __init__(self, capacity):
        self.capacity = capacity
        self.count = 0
        self.worst_dist = _UNSET_DISTANCE
        self.comparison_counter = 0
        self._keys = []  # distances of the filled entries, ascending
        self.dist_index_list = [DistIndex(_UNSET_DISTANCE, 0) for _ in range(capacity)]

    def size(self):
        return self.count

    def full(self):
        return self.count == self.capacity

    def worstDist(self):
        return self.worst_dist

    def add_point(self, dist, index):
        self.comparison_counter += 1
        if dist > self.worst_dist or self.capacity == 0:
            return
        filled = self.dist_index_list[: self.count]
        if self.count == self.capacity:
            # the incoming entry takes the place of the current last one, then settles among the rest
            filled.pop()
            self._keys.pop()
        else:
            self.count += 1
        at = bisect_right(self._keys, dist)
        self._keys.insert(at, dist)
        filled.insert(at, DistIndex(dist, index))
        self.dist_index_list[: self.count] = filled
        self.worst_dist = self.dist_index_list[self.capacity - 1].distance

    def __str__(self):
        rows = ["%d - %.2f" % (e.index, e.distance) for e in self.dist_index_list]
        rows.append("In total %d comparison operations." % self.comparison_counter)
        return "\n".join(rows)


class RadiusNNResultSet:
    def __init__(self, radius):
        self.radius = radius
        self.worst_dist = radius
        self.count = 0
        self.comparison_counter = 0
        self.dist_index_list = []

    def size(self):
        return self.count

    def worstDist(self):
        return self.radius

    def add_point(self, dist, index):
        self.comparison_counter += 1
        if not dist > self.radius:
            self.dist_index_list.append(DistIndex(dist, index))
            self.count += 1

    def __str__(self):
        self.dist_index_list.sort()
        rows = ["%d - %.2f" % (e.index, e.distance) for e in self.dist_index_list]
        rows.append("In total %d neighbors within %f.\nThere are %d comparison operations." % (self.count, self.radius, self.comparison_counter))
        return "\n".join(rows)
