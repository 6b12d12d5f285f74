"""Shared assertions for PCA / normal parity (eigenvectors are defined up to sign; a normal is only
well-conditioned where the two smallest eigenvalues are separated)."""
import numpy as np


def check_pca(w, v, w_ref, v_ref, rtol=1e-9):
    scale = max(abs(w_ref).max(), 1e-300)
    assert np.allclose(w, w_ref, rtol=0, atol=rtol * scale)
    assert np.allclose(v.T @ v, np.eye(3), atol=1e-12)
    gaps = [abs(w_ref[0] - w_ref[1]), min(abs(w_ref[0] - w_ref[1]), abs(w_ref[1] - w_ref[2])), abs(w_ref[1] - w_ref[2])]
    for c in range(3):
        if gaps[c] > 1e-6 * scale:
            assert abs(abs(v[:, c] @ v_ref[:, c]) - 1.0) < 1e-8, c


def check_normals(pts, normals, evs, nbrs, g_normals, g_evs, g_nbrs):
    n = len(pts)
    assert normals.shape == (n, 3) and np.allclose(np.linalg.norm(normals, axis=1), 1.0, atol=1e-12)
    # neighbour sets: equal up to ties in distance
    same = (np.sort(nbrs, axis=1) == np.sort(g_nbrs, axis=1)).all(axis=1)
    for i in np.flatnonzero(~same):
        d_a = np.sort(np.linalg.norm(pts[nbrs[i]] - pts[i], axis=1))
        d_b = np.sort(np.linalg.norm(pts[g_nbrs[i]] - pts[i], axis=1))
        assert np.allclose(d_a, d_b, rtol=1e-12, atol=0), i
    scale = np.maximum(np.abs(g_evs).max(axis=1), 1e-300)
    ok = same
    assert np.allclose(evs[ok], g_evs[ok], rtol=0, atol=1e-9 * scale[ok].max())
    assert (np.abs(evs[ok] - g_evs[ok]).max(axis=1) <= 1e-9 * scale[ok]).all()
    gap = (g_evs[:, 1] - g_evs[:, 2]) / scale
    well = ok & (gap > 1e-4)
    assert well.mean() > 0.5
    dots = np.abs(np.einsum("ij,ij->i", normals[well], g_normals[well]))
    assert (np.abs(dots - 1.0) < 1e-7).all()
    # everywhere: the normal is an eigenvector of the neighbourhood covariance for the smallest eigenvalue
    for i in range(0, n, max(1, n // 300)):
        c = np.cov(pts[nbrs[i]].T)
        r = c @ normals[i] - evs[i, 2] * normals[i]
        assert np.linalg.norm(r) <= 1e-9 * max(scale[i], 1e-300) + 1e-18, i
