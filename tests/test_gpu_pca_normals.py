"""GPU parity: PCA and per-point normals vs outputs of the reference's own PCA (pca_normal.py)."""
import numpy as np
import pytest

from tests.conftest import load_golden
from tests.pca_checks import check_normals, check_pca

pytestmark = pytest.mark.gpu


def test_pca_matches_reference(pcp):
    g = load_golden("pca_normals.npz")
    for tag in ("object", "plane", "tiny", "scan"):
        w, v = pcp.PCA(g[f"{tag}_pts"])
        check_pca(w, v, g[f"{tag}_w"], g[f"{tag}_v"])
        w2, v2 = pcp.PCA(g[f"{tag}_pts"], sort=False)
        assert np.array_equal(w2, w[::-1]) and np.array_equal(v2, v[:, ::-1])


def test_normals_match_reference(pcp):
    g = load_golden("pca_normals.npz")
    for tag in ("object", "plane", "tiny", "scan"):
        pts = g[f"{tag}_pts"]
        nrm, evs, nbr = pcp.estimate_normals(pts, 5, return_details=True)
        check_normals(pts, nrm, evs, nbr.astype(np.int64), g[f"{tag}_normals"], g[f"{tag}_evs"], g[f"{tag}_nbrs"])


def test_normals_large_scan_vs_oracle(pcp, oracle, syn):
    from scipy.spatial import cKDTree

    pts = syn.kitti_like_scan(120000, seed=4).astype(np.float64)
    nrm, evs, nbr = pcp.estimate_normals(pts, 8, return_details=True)
    d, i = cKDTree(pts).query(pts[::97], k=8)
    got = np.linalg.norm(pts[nbr[::97]] - pts[::97, None, :], axis=2)
    assert np.allclose(got, d, rtol=1e-12, atol=0)          # the k nearest, ascending
    sub = np.arange(0, len(pts), 997)
    for j in sub:
        w, v = oracle.pca(pts[nbr[j]])
        assert np.allclose(evs[j], w, rtol=0, atol=1e-9 * max(w[0], 1e-300))
    with pytest.raises((RuntimeError, ValueError)):
        pcp.estimate_normals(pts[:100], 17)     # k > 16: PCR_E_UNSUPPORTED
    with pytest.raises((RuntimeError, ValueError)):
        pcp.estimate_normals(pts[:100], 1)      # a covariance needs two points
