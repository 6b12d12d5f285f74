"""KNNResultSet / RadiusNNResultSet insertion semantics vs the reference's own classes (golden stream)."""
import numpy as np

from tests.conftest import load_golden


def test_result_sets_match_reference_stream(pcp):
    g = load_golden("nn_api.npz")
    for cap in (3, 12):
        r = pcp.KNNResultSet(capacity=cap)
        for d, i in zip(g["stream_d"], g["stream_i"]):
            r.add_point(float(d), int(i))
        assert np.array_equal([x.distance for x in r.dist_index_list], g[f"stream_cap{cap}_dist"])
        assert np.array_equal([x.index for x in r.dist_index_list], g[f"stream_cap{cap}_idx"])
        count, cmp_, worst = g[f"stream_cap{cap}_meta"]
        assert (r.count, r.comparison_counter, r.worstDist()) == (count, cmp_, worst)
        assert r.size() == count and r.full() == (count == cap)
    r = pcp.RadiusNNResultSet(radius=0.5)
    for d, i in zip(g["stream_d"], g["stream_i"]):
        r.add_point(float(d), int(i))
    assert np.array_equal([x.index for x in r.dist_index_list], g["stream_radius_idx"])
    count, cmp_, worst = g["stream_radius_meta"]
    assert (r.count, r.comparison_counter, r.worstDist()) == (count, cmp_, worst)


def test_knn_result_set_randomised_against_sorted_model(pcp):
    rng = np.random.default_rng(5)
    for cap in (1, 4, 9):
        d = np.round(rng.uniform(0, 1, 200), 1)  # many ties
        r = pcp.KNNResultSet(capacity=cap)
        for k, v in enumerate(d):
            r.add_point(float(v), k)
        got = [x.distance for x in r.dist_index_list]
        assert got == sorted(d.tolist())[:cap]
        assert r.worstDist() == got[-1]
