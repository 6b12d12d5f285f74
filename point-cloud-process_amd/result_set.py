"""Result containers with the public surface of Kdtree_Octree/lesson2/result_set.py.

``KNNResultSet(capacity)`` / ``RadiusNNResultSet(radius)`` expose ``add_point``, ``worstDist``,
``size``, ``full`` (kNN only), ``dist_index_list`` (objects with ``.distance`` / ``.index``),
``count``, ``comparison_counter``.  Observable behaviour follows result_set.py:15-93:

* kNN: a distance strictly greater than the current worst is rejected, an equal one is accepted
  and placed AFTER the entries it ties with -- when the set is already full it therefore replaces
  the last entry (result_set.py:37-50); unfilled slots read (1e10, 0) (result_set.py:19-22);
* radius: ``dist > radius`` is rejected, i.e. the ball is closed (result_set.py:80).

Both containers share one base (counters, report formatting); the k-NN set keeps a sorted key list and
places a new entry with ``bisect`` instead of the reference's shifting loop.
"""
from __future__ import annotations

from bisect import bisect_right

_UNSET_DISTANCE = 1e10


class DistIndex:
    __slots__ = ("distance", "index")

    def __init__(self, distance, index):
        self.distance, self.index = distance, index

    def __lt__(self, other):
        return self.distance < other.distance

    def __repr__(self):
        return f"DistIndex({self.distance!r}, {self.index!r})"


class _Collector:
    """What both result sets have in common: the two counters and the text report."""

    def __init__(self):
        self.count = 0
        self.comparison_counter = 0
        self.dist_index_list = []

    def size(self):
        return self.count

    def _report(self, tail):
        return "\n".join(["%d - %.2f" % (e.index, e.distance) for e in self.dist_index_list] + [tail])


class RadiusNNResultSet(_Collector):
    """result_set.py:63-93.  ``add_points(dists, indices)`` takes a whole query result at once (what the device search returns):
    the DistIndex objects of ``dist_index_list`` are then only made when the list is first looked at -- a radius-1 query on a
    KITTI scan returns ~1 500 neighbours, and one Python object per neighbour cost ten times the search itself."""

    def __init__(self, radius):
        super().__init__()
        self.radius = self.worst_dist = radius   # (the base class has set dist_index_list through the property below: _items, _pending)

    @property
    def dist_index_list(self):
        if self._pending:
            for d, i in self._pending:
                self._items.extend(map(DistIndex, d.tolist(), i.tolist()))
            self._pending = []
        return self._items

    @dist_index_list.setter
    def dist_index_list(self, v):
        self._items, self._pending = v, []

    def worstDist(self):
        return self.radius

    def add_point(self, dist, index):
        self.comparison_counter += 1
        if dist <= self.radius or dist != dist:  # closed ball; the reference's `dist > radius` test also lets NaN through
            self.count += 1
            self.dist_index_list.append(DistIndex(dist, index))

    def add_points(self, dists, indices):
        """add_point for every (dist, index) pair of two equally long arrays, in order."""
        import numpy as np

        d = np.asarray(dists, dtype=np.float64).reshape(-1)
        i = np.asarray(indices).reshape(-1)
        self.comparison_counter += len(d)
        keep = ~(d > self.radius)
        if not keep.all():
            d, i = d[keep], i[keep]
        self.count += len(d)
        if len(d):
            self._pending.append((d, i))

    def __str__(self):
        self.dist_index_list.sort()
        return self._report("In total %d neighbors within %f.\nThere are %d comparison operations."
                            % (self.count, self.radius, self.comparison_counter))

class KNNResultSet(_Collector):
    def __init__(self, capacity):
        super().__init__()
        self.capacity = capacity
        self.worst_dist = _UNSET_DISTANCE
        self.dist_index_list = [DistIndex(_UNSET_DISTANCE, 0) for _ in range(capacity)]
        self._keys = []  # distances of the filled entries, ascending

    def full(self):
        return self.count == self.capacity

    def worstDist(self):
        return self.worst_dist

    def add_point(self, dist, index):
        self.comparison_counter += 1
        if self.capacity == 0 or dist > self.worst_dist:
            return
        if self.full():
            del self._keys[-1]  # the newcomer displaces the current last entry, then settles among the rest
        else:
            self.count += 1
        at = bisect_right(self._keys, dist)
        self._keys.insert(at, dist)
        live = self.dist_index_list[: len(self._keys) - 1]
        live.insert(at, DistIndex(dist, index))
        self.dist_index_list[: len(live)] = live
        self.worst_dist = self.dist_index_list[-1].distance

    def __str__(self):
        return self._report("In total %d comparison operations." % self.comparison_counter)
