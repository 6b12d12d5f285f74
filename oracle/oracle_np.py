"""TEST INFRASTRUCTURE -- CPU restatement (NumPy/SciPy) of the reference's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / reported baseline.  Nothing in
point-cloud-process_amd/ imports it; the product path has no CPU fallback.

Pinned against the reference itself: tests/test_oracle_golden.py checks every
function below against tests/golden/*.npz, which oracle/ref_harness.py produced
by running the reference's own functions (Registration/main.py:icp_point2point,
Pca_and_Voxel_filter/voxel_filter.py:voxel_filter, Kdtree_Octree/lesson2/*).
ISS (Keypoint_detection_ISS/ISS.py is a script body without importable
functions): ``iss_oracle`` restates ISS.py:41-73 line by line and is pinned too
-- tests/golden/iss.npz is what the script itself computes, run unmodified
through runpy on seeded input files (oracle/ref_harness.py:gen_iss).

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


# --------------------------------------------------------------- NN search
def dist2_direct(a, b):
    """(dx*dx + dy*dy) + dz*dz in binary64, the association metric of main.py:117-119
    (Open3D returns squared distances)."""
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def nn1_exact(queries, targets, workers=1):
    """Exact 1-NN of each query in targets -> (idx, d2, margin) where margin is the
    relative gap to the second-nearest (ties have margin ~ 0 and are excluded from
    index parity).  Stand-in for KDTreeFlann.search_knn_vector_3d(q, 1), main.py:117."""
    q = np.asarray(queries, dtype=np.float64)
    t = np.asarray(targets, dtype=np.float64)
    tree = cKDTree(t)
    k = 2 if len(t) > 1 else 1
    _, i = tree.query(q, k=k, workers=workers)
    i = i.reshape(len(q), k)
    d2a = dist2_direct(q, t[i[:, 0]])
    if k == 2:
        d2b = dist2_direct(q, t[i[:, 1]])
        swap = (d2b < d2a) | ((d2b == d2a) & (i[:, 1] < i[:, 0]))
        best = np.where(swap, i[:, 1], i[:, 0])
        lo = np.minimum(d2a, d2b)
        hi = np.maximum(d2a, d2b)
        margin = (hi - lo) / np.maximum(hi, 1e-300)
    else:
        best, lo, margin = i[:, 0], d2a, np.ones(len(q))
    return best.astype(np.int64), lo, margin


def nn1_bruteforce(queries, targets, chunk=2048):
    """O(N*M) exact 1-NN with the lowest-index tie rule (small cases only)."""
    q = np.asarray(queries, dtype=np.float64)
    t = np.asarray(targets, dtype=np.float64)
    idx = np.empty(len(q), dtype=np.int64)
    d2 = np.empty(len(q))
    for s in range(0, len(q), chunk):
        D = dist2_direct(q[s : s + chunk, None, :], t[None, :, :])
        j = np.argmin(D, axis=1)
        idx[s : s + chunk] = j
        d2[s : s + chunk] = D[np.arange(len(j)), j]
    return idx, d2


# --------------------------------------------------------------- Procrustes
def procrustes(A, B):
    """main.py:131-141 with the dense K x K centring matrix L = I - 11^T/K replaced by the
    algebraically identical A - mean(A) (L needs 124 GB at 120k points).
    A, B: (3,K).  Returns R (3,3) = U V^T (no reflection fix), t (3,1), cost."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    N = A.shape[1]
    Ap = A - A.mean(axis=1, keepdims=True)
    Bp = B - B.mean(axis=1, keepdims=True)
    u, s, vt = np.linalg.svd(Bp @ Ap.T)
    R = u @ vt
    t = (B - R @ A).sum(axis=1, keepdims=True) / N
    cost = np.linalg.norm(B - (R @ A + t))
    return R, t, cost


def moments(A, B, origin):
    """The 18 raw moments the fused GPU pass accumulates (about `origin`):
    K, Sa[3], Sb[3], Sba[9] (b_i a_j row-major), Saa, Sbb.  A, B: (K,3)."""
    a = np.asarray(A, dtype=np.float64) - origin
    b = np.asarray(B, dtype=np.float64) - origin
    m = np.zeros(18)
    m[0] = len(a)
    m[1:4] = a.sum(0)
    m[4:7] = b.sum(0)
    m[7:16] = (b.T @ a).reshape(9)
    m[16] = (a * a).sum()
    m[17] = (b * b).sum()
    return m


# ---------------------------------------------------------------------- ICP
def icp_point2point(src_pts, tgt_pts, T0, max_iteration=100, R_diff_thres=0.5, t_diff_thres=0.5, dist_thres=5.0):
    """Registration/main.py:97-156 restated on arrays.  Returns dict(T = last increment,
    T_total, iters, failed, src_after (the in-place mutated source, main.py:110))."""
    src = np.array(src_pts, dtype=np.float64)[:, :3]
    tgt = np.asarray(tgt_pts, dtype=np.float64)[:, :3]
    T = np.array(T0, dtype=np.float64).reshape(4, 4)
    R_last = T[:3, :3]
    t_last = T[:3, 3]  # shape (3,): main.py:100
    tree = cKDTree(tgt)
    T_total = np.eye(4)
    iters = 0
    failed = False
    for _ in range(max_iteration):
        src = src @ T[:3, :3].T + T[:3, 3]  # main.py:110 (in place)
        T_total = T @ T_total
        _, j = tree.query(src, k=1)
        d2 = dist2_direct(src, tgt[j])
        keep = d2 < dist_thres  # main.py:119, strict, squared
        if keep.sum() < 3:  # main.py:125-127
            failed = True
            break
        A = src[keep].T
        B = tgt[j[keep]].T
        R, t, cost = procrustes(A, B)
        T = np.zeros((4, 4))
        T[:3, :3] = R
        T[:3, 3] = t.squeeze()
        T[3, 3] = 1.0
        iters += 1
        R_diff = np.linalg.norm(R - R_last)
        t_diff = np.linalg.norm(t - t_last)  # first pass: (3,1)-(3,) broadcast -> 3x3 (main.py:150)
        R_last, t_last = R, t
        if R_diff <= R_diff_thres and t_diff <= t_diff_thres:
            break
    return {"T": T, "T_total": T_total, "iters": iters, "failed": failed, "src_after": src}


def icp_total(src_pts, tgt_pts, T_init=None, max_iteration=50, R_diff_thres=1e-5, t_diff_thres=1e-5, dist_thres=5.0, geodesic=True):
    """Registration/icp_template.py:128-200 filled in the way the product fills it."""
    src = np.array(src_pts, dtype=np.float64)[:, :3]
    tgt = np.asarray(tgt_pts, dtype=np.float64)[:, :3]
    T_init = np.eye(4) if T_init is None else np.asarray(T_init, dtype=np.float64)
    homo = T_init.copy()
    src = src @ T_init[:3, :3].T + T_init[:3, 3]
    R_last, t_last = T_init[:3, :3], T_init[:3, 3:4]
    tree = cKDTree(tgt)
    log = {"R_diff": [], "t_diff": []}
    for _ in range(max_iteration):
        _, j = tree.query(src, k=1)
        d2 = dist2_direct(src, tgt[j])
        keep = d2 < dist_thres
        if keep.sum() < 3:
            break
        R, t, cost = procrustes(src[keep].T, tgt[j[keep]].T)
        if geodesic:
            c = np.clip((np.trace(R_last.T @ R) - 1.0) / 2.0, -1.0, 1.0)
            R_diff = float(np.arccos(c))
        else:
            R_diff = float(np.linalg.norm(R - R_last))
        t_diff = float(np.linalg.norm(t - t_last))
        R_last, t_last = R, t
        log["R_diff"].append(R_diff)
        log["t_diff"].append(t_diff)
        if R_diff <= R_diff_thres and t_diff <= t_diff_thres:
            break
        src = src @ R.T + t.T
        Ti = np.eye(4)
        Ti[:3, :3] = R
        Ti[:3, 3] = t.squeeze()
        homo = Ti @ homo
    return homo, log


# -------------------------------------------------------------- pose utils
def rotmat2quaternion(m):
    """main.py:158-168."""
    trace = m[0][0] + m[1][1] + m[2][2]
    qw = np.sqrt(max(0, trace + 1)) / 2
    qx = np.sqrt(max(0, 1 + m[0][0] - m[1][1] - m[2][2])) / 2
    qy = np.sqrt(max(0, 1 - m[0][0] + m[1][1] - m[2][2])) / 2
    qz = np.sqrt(max(0, 1 - m[0][0] - m[1][1] + m[2][2])) / 2

    def cs(v, s):
        return -v if v * s < 0 else v

    return qw, cs(qx, m[2][1] - m[1][2]), cs(qy, m[0][2] - m[2][0]), cs(qz, m[1][0] - m[0][1])


def homo2tq(T):
    """main.py:170-174."""
    T = np.asarray(T, dtype=np.float64)
    qw, qx, qy, qz = rotmat2quaternion(T[:3, :3])
    return T[0, 3], T[1, 3], T[2, 3], qw, qx, qy, qz


# ------------------------------------------------------------ voxel filter
def voxel_keys(points, leaf):
    """voxel_filter.py:20-33: per-point key h (float64) and D = (Dx,Dy,Dz)."""
    pc = np.asarray(points, dtype=np.float64)
    mx = np.max(pc, axis=0)
    mn = np.min(pc, axis=0)
    D = (mx - mn) // leaf  # NumPy float floor_divide, no +1 (voxel_filter.py:22-24)
    hx = np.floor((pc[:, 0] - mn[0]) / leaf)
    hy = np.floor((pc[:, 1] - mn[1]) / leaf)
    hz = np.floor((pc[:, 2] - mn[2]) / leaf)
    h = hx + hy * D[0] + hz * D[0] * D[1]
    return h, D


def voxel_filter(points, leaf, mode="centroid", seed=0):
    """voxel_filter.py:10-68: stable sort by h, one output per group, LAST group dropped
    (voxel_filter.py:42-51).  centroid = np.mean over the group in input order."""
    pc = np.asarray(points, dtype=np.float64)
    h, _ = voxel_keys(pc, leaf)
    order = np.argsort(h, kind="stable")
    hs = h[order]
    starts = np.flatnonzero(np.r_[True, hs[1:] != hs[:-1]])
    ends = np.r_[starts[1:], len(hs)]
    out = []
    rng = np.random.default_rng(seed)
    for s, e in zip(starts[:-1], ends[:-1]):  # last group never emitted
        grp = pc[order[s:e]]
        if mode == "centroid":
            out.append(np.mean(np.ascontiguousarray(grp.T), axis=1))
        else:
            out.append(grp[rng.integers(0, e - s)])
    return np.array(out, dtype=np.float64).reshape(-1, 3), order, starts, ends


# ------------------------------------------------------- k-NN / radius API
def knn_bruteforce(db, query, k):
    """Result of kdtree_knn_search / octree_knn_search (kdtree.py:141-172, octree.py:262-306)
    with KNNResultSet (result_set.py:15-60): ascending Euclidean distances; unfilled slots
    keep (1e10, 0) (result_set.py:19-22)."""
    db = np.asarray(db, dtype=np.float64)
    d = np.linalg.norm(np.asarray(query, dtype=np.float64)[None, :] - db, axis=1)
    order = np.lexsort((np.arange(len(d)), d))[:k]
    dist = np.full(k, 1e10)
    idx = np.zeros(k, dtype=np.int64)
    dist[: len(order)] = d[order]
    idx[: len(order)] = order
    return idx, dist


def radius_bruteforce(db, query, r):
    """RadiusNNResultSet contents (result_set.py:63-93): distance <= r inclusive, sorted ascending."""
    db = np.asarray(db, dtype=np.float64)
    d = np.linalg.norm(np.asarray(query, dtype=np.float64)[None, :] - db, axis=1)
    sel = np.flatnonzero(d <= r)
    order = sel[np.lexsort((sel, d[sel]))]
    return order, d[order]


# ----------------------------------------------------------------------- ISS
def iss_oracle(points, radius=0.5, lambda21=0.5, lambda32=0.5, non_max_radius=0.5, iss_count=20):
    """Keypoint_detection_ISS/ISS.py:35-73 restated ("parity unpinned": see module docstring).
    Returns (iss_idx list, lambdas (N,3) descending, counts (N,))."""
    pts = np.asarray(points, dtype=np.float64)[:, :3]
    tree = cKDTree(pts)
    nbrs = tree.query_ball_point(pts, radius)
    counts = np.array([len(x) for x in nbrs])
    lam = np.zeros((len(pts), 3))
    lambda3 = {}
    for i in range(len(pts)):
        pj = pts[nbrs[i]]
        w = 1.0 / counts[nbrs[i]]
        d = pj - pts[i]
        nume = (d * w[:, None]).T @ d
        denom = w.sum()
        val = np.linalg.eigvalsh(nume / denom)  # symmetric by construction (ISS.py:52 uses eig)
        sv = np.sort(val)[::-1]
        lam[i] = sv
        if sv[1] / sv[0] < lambda21 and sv[2] / sv[1] < lambda32:
            lambda3[i] = sv[2]
    iss = []
    live = dict(lambda3)
    for idx in sorted(lambda3, key=lambda x: lambda3[x], reverse=True):
        if idx not in live:
            continue
        iss.append(idx)
        for nb in tree.query_ball_point(pts[idx], non_max_radius):
            live.pop(nb, None)
        if len(iss) > iss_count:
            break
    return iss, lam, counts


# ------------------------------------------------------------ PCA / normals
def pca(data, sort=True):
    """Pca_and_Voxel_filter/pca_normal.py:10-36: eigen-decomposition of the sample covariance (divisor N-1)
    of the (N,3) cloud; eigenvalues descending, eigenvectors as columns.  Pinned by tests/golden/pca_normals.npz."""
    x = np.asarray(data, dtype=np.float64)
    c = x - x.mean(axis=0, keepdims=True)            # pca_normal.py:22-23
    H = c.T @ c / (len(x) - 1)                        # np.cov, pca_normal.py:25
    w, v = np.linalg.eigh(H)                          # pca_normal.py:27
    if sort:
        o = w.argsort()[::-1]                         # pca_normal.py:30-33
        w, v = w[o], v[:, o]
    return w, v


def normals(points, k=5):
    """pca_normal.py:85-90: for every point, v[:, 2] of the PCA of its k nearest neighbours (itself included).
    Returns (normals (N,3), eigenvalues (N,3), neighbours (N,k))."""
    pts = np.asarray(points, dtype=np.float64)
    k = min(k, len(pts))
    nbr = np.empty((len(pts), k), dtype=np.int64)
    for i, q in enumerate(pts):
        idx, _ = knn_bruteforce(pts, q, k)
        nbr[i] = idx
    out = np.empty((len(pts), 3))
    evs = np.empty((len(pts), 3))
    for i in range(len(pts)):
        w, v = pca(pts[nbr[i]])
        out[i], evs[i] = v[:, 2], w
    return out, evs, nbr


# ------------------------------------------------------------------ DBSCAN
def dbscan(data, radius=0.5, min_pts=10):
    """Cluster_dbscan/dbscan.py:10-36 with a visited mask instead of the O(N) list membership tests (same traversal):
    seeds from the END of the index list (:18), seed needs >= Min_Pts (:20), reached points expand with > Min_Pts (:32),
    a point popped as a noise seed is never relabelled (:21-22,29).  Pinned by tests/golden/dbscan.npz."""
    pts = np.asarray(data, dtype=np.float64)
    tree = cKDTree(pts)
    nbrs = tree.query_ball_point(pts, radius)
    n = len(pts)
    labels = -np.ones(n, dtype=np.int32)
    visited = np.zeros(n, dtype=bool)
    label = -1
    for ind in range(n - 1, -1, -1):
        if visited[ind]:
            continue
        visited[ind] = True
        if len(nbrs[ind]) < min_pts:
            continue
        label += 1
        labels[ind] = label
        stack = list(nbrs[ind])
        while stack:
            cur = stack.pop()
            if visited[cur]:
                continue
            visited[cur] = True
            labels[cur] = label
            if len(nbrs[cur]) > min_pts:
                stack.extend(nbrs[cur])
    return labels
