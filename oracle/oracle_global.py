"""TEST INFRASTRUCTURE -- CPU restatement (NumPy) of the global-initialisation stage in front of ICP
(Registration/main.py:33-84; icp_template.py:20-41,56-110).  Only tests/ may import it.

PARITY UNPINNED: this stage of the reference is Open3D (absent, unpinned, randomised RANSAC); no output of the
reference exists for it.  The functions restate Open3D's published behaviour (>= 0.12: voxel_down_sample,
EstimateNormals with a hybrid search, ComputeFPFHFeature, RegistrationRANSACBasedOnFeatureMatching with
EvaluateRANSACBasedOnCorrespondence) with the same deterministic choices the device code documents in
include/pcr.h: neighbours ordered by (d^2, row), d^2 < r^2, counter-based random stream.
"""
from __future__ import annotations

import numpy as np

M64 = (1 << 64) - 1


def voxel_down_sample(points, voxel):
    """main.py:35.  Origin min - voxel/2; centroid = running sum in input order / count; rows by voxel key."""
    p = np.asarray(points, dtype=np.float64)
    mn = p.min(axis=0) - voxel * 0.5
    mx = p.max(axis=0)
    D = np.floor((mx - mn) / voxel) + 1.0
    h = np.floor((p - mn) / voxel)
    key = (h[:, 0] + h[:, 1] * D[0]) + (h[:, 2] * D[0]) * D[1]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[heads[1:], len(ks)]
    out = np.empty((len(heads), 3))
    for g, (s, e) in enumerate(zip(heads, ends)):
        acc = np.zeros(3)
        for r in order[s:e]:
            acc = acc + p[r]
        out[g] = acc / (e - s)
    return out


def _d2(p, q):
    d = p - q
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def hybrid_neighbours(points, radius, max_nn):
    """KDTreeSearchParamHybrid: the <= max_nn nearest rows with d^2 < radius^2, ascending (d^2, row)."""
    p = np.asarray(points, dtype=np.float64)
    out = []
    r2 = radius * radius
    for i in range(len(p)):
        d2 = _d2(p, p[i])
        sel = np.flatnonzero(d2 < r2)
        o = sel[np.lexsort((sel, d2[sel]))][:max_nn]
        out.append((o, d2[o]))
    return out


def normals_hybrid(points, radius, max_nn=30, orient=True, viewpoint=(0.0, 0.0, 0.0), nbrs=None):
    """main.py:39-40: covariance of the neighbourhood (cumulant form), eigenvector of the smallest eigenvalue;
    fewer than 3 neighbours -> (0,0,1)."""
    p = np.asarray(points, dtype=np.float64)
    nbrs = nbrs or hybrid_neighbours(p, radius, max_nn)
    vp = np.asarray(viewpoint, dtype=np.float64)
    out = np.zeros((len(p), 3))
    gap = np.zeros(len(p))
    for i, (idx, _) in enumerate(nbrs):
        if len(idx) < 3:
            out[i] = (0, 0, 1)
            continue
        x = p[idx] - p[i]
        m = x.mean(axis=0)
        S = x.T @ x / len(idx) - np.outer(m, m)
        w, v = np.linalg.eigh(S)
        n = v[:, 0]
        gap[i] = (w[1] - w[0]) / max(w[2], 1e-300)
        if orient and n @ (vp - p[i]) < 0:
            n = -n
        out[i] = n
    return out, gap


def pair_features(p1, n1, p2, n2):
    """Open3D ComputePairFeatures (Darboux frame of the pair)."""
    d = p2 - p1
    ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    if ln == 0.0:
        return np.zeros(3)
    a1 = ((n1[0] * d[0] + n1[1] * d[1]) + n1[2] * d[2]) / ln
    a2 = ((n2[0] * d[0] + n2[1] * d[1]) + n2[2] * d[2]) / ln
    if abs(a1) < abs(a2):  # acos(|a1|) > acos(|a2|)
        u, w2, d, f2 = n2, n1, -d, -a2
    else:
        u, w2, f2 = n1, n2, a1
    v = np.cross(d, u)
    vn = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    if vn == 0.0:
        return np.zeros(3)
    v = v / vn
    w = np.cross(u, v)
    f1 = (v[0] * w2[0] + v[1] * w2[1]) + v[2] * w2[2]
    f0 = np.arctan2((w[0] * w2[0] + w[1] * w2[1]) + w[2] * w2[2], (u[0] * w2[0] + u[1] * w2[1]) + u[2] * w2[2])
    return np.array([f0, f1, f2])


def _bin(x):
    return int(min(max(np.floor(x), 0), 10))


def spfh(points, normals, nbrs):
    p = np.asarray(points, dtype=np.float64)
    out = np.zeros((len(p), 33))
    for i, (idx, _) in enumerate(nbrs):
        if len(idx) <= 1:
            continue
        incr = 100.0 / (len(idx) - 1)
        for k in idx[1:]:
            f = pair_features(p[i], normals[i], p[k], normals[k])
            out[i, _bin(11 * (f[0] + np.pi) / (2.0 * np.pi))] += incr
            out[i, 11 + _bin(11 * (f[1] + 1.0) * 0.5)] += incr
            out[i, 22 + _bin(11 * (f[2] + 1.0) * 0.5)] += incr
    return out


def fpfh(points, normals, radius, max_nn=100, nbrs=None):
    """main.py:44-46 -> (N,33): SPFH of the point + 1/d^2-weighted SPFH of its neighbours, each of the three
    11-bin blocks of the weighted part renormalised to 100."""
    p = np.asarray(points, dtype=np.float64)
    nbrs = nbrs or hybrid_neighbours(p, radius, max_nn)
    s = spfh(p, normals, nbrs)
    out = np.zeros_like(s)
    for i, (idx, d2) in enumerate(nbrs):
        if len(idx) <= 1:
            continue
        acc = np.zeros(33)
        tot = np.zeros(3)
        for k, dd in zip(idx[1:], d2[1:]):
            if dd == 0.0:
                continue
            val = s[k] / dd
            acc += val
            for j in range(33):
                tot[j // 11] += val[j]
        scale = np.where(tot != 0.0, 100.0 / np.where(tot != 0.0, tot, 1.0), 0.0)
        out[i] = acc * np.repeat(scale, 11) + s[i]
    return out


def feature_match(A, B):
    """Nearest row of B for every row of A; squared L2 summed over the dimensions in order; ties to the lowest row."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    s = np.zeros((len(A), len(B)))
    for k in range(A.shape[1]):
        d = A[:, k, None] - B[None, :, k]
        s = s + d * d
    idx = s.argmin(axis=1)
    return idx.astype(np.int32), s[np.arange(len(A)), idx]


def mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def kabsch(A, B):
    """procrustes_transformation (icp_template.py:43-54) on (K,3) rows, proper rotation for the rank-2 case."""
    ca, cb = A.mean(axis=0), B.mean(axis=0)
    H = (B - cb).T @ (A - ca)
    U, _, Vt = np.linalg.svd(H)
    if np.linalg.det(U @ Vt) < 0:
        U[:, -1] = -U[:, -1]
    R = U @ Vt
    return R, cb - R @ ca


def ransac(src, tgt, corr, max_iteration=100000, confidence=0.999, max_distance=3.0, edge_similarity=0.9, check_distance=True, seed=0):
    """main.py:73-83 / icp_template.py:88-110: sequential loop with running best and confidence-based exit.
    Returns dict(T, best_iteration, iterations, n_valid, corr_fitness, corr_rmse, inliers)."""
    src = np.asarray(src, dtype=np.float64)
    tgt = np.asarray(tgt, dtype=np.float64)
    corr = np.asarray(corr)
    m = len(corr)
    S, Tg = src[corr[:, 0]], tgt[corr[:, 1]]
    best = dict(T=np.eye(4), best_iteration=-1, corr_fitness=0.0, corr_rmse=0.0, inliers=0)
    exit_itr = max_iteration
    n_valid = 0
    itr = 0
    while itr < exit_itr:
        c = [mix64(seed ^ mix64(itr * 3 + j)) % m for j in range(3)]
        s, t = S[c], Tg[c]
        ok = True
        if edge_similarity > 0:
            for i in range(3):
                for j in range(i + 1, 3):
                    ds = np.linalg.norm(s[i] - s[j])
                    dt = np.linalg.norm(t[i] - t[j])
                    if ds < dt * edge_similarity or dt < ds * edge_similarity:
                        ok = False
        if ok:
            R, tr = kabsch(s, t)
            if check_distance and (np.linalg.norm(s @ R.T + tr - t, axis=1) > max_distance).any():
                ok = False
        if ok:
            n_valid += 1
            dis = np.linalg.norm(S @ R.T + tr - Tg, axis=1)
            inl = dis < max_distance
            good = int(inl.sum())
            fit = good / m
            rmse = float(np.sqrt((dis[inl] ** 2).sum() / good)) if good else 0.0
            if fit > best["corr_fitness"] or (fit == best["corr_fitness"] and rmse < best["corr_rmse"]):
                T = np.eye(4)
                T[:3, :3], T[:3, 3] = R, tr
                best.update(T=T, best_iteration=itr, corr_fitness=fit, corr_rmse=rmse, inliers=good)
                x = 1.0 - fit ** 3
                k = 0.0 if x <= 0 else np.log(1.0 - confidence) / np.log(x)
                if k < max_iteration:
                    exit_itr = min(exit_itr, int(np.ceil(k)))
        itr += 1
    best.update(iterations=min(itr, exit_itr), n_valid=n_valid)
    return best
