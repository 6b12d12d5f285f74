"""Seeded synthetic clouds shaped like the inputs of the registration path.

The reference ships exactly one real cloud (Kdtree_Octree/000000.bin, a KITTI
Velodyne HDL-64 scan, 124 668 points) and none of the registration_dataset
clouds (Registration/registration_dataset.txt is a download link).  Nothing
under /root/reference exists on the GPU box, so every benchmark and test input
is generated here from ``numpy.random.default_rng(seed)``.

Statistics the KITTI-shaped generator is tuned to:
x in +-78 m, range p50 ~ 10 m / p95 ~ 38 m, ~82 % of returns within 20 m,
self-NN distance p50 ~ 3 cm, ground plane at z = -1.73 m.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "kitti_like_scan",
    "perturbed_pair",
    "rigid_transform",
    "object_cloud",
    "registration_pair_6f",
    "registration_batch_6f",
]


def rigid_transform(axis, angle_rad, t):
    """4x4 float64 homogeneous transform: rotation about ``axis`` then translation."""
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    C = 1.0 - c
    R = np.array(
        [
            [c + x * x * C, x * y * C - z * s, x * z * C + y * s],
            [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
            [z * x * C - y * s, z * y * C + x * s, c + z * z * C],
        ]
    )
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t, dtype=np.float64)
    return T


def _ray_boxes(origins, dirs, lo, hi):
    """Slab test of R rays against B axis-aligned boxes -> nearest entry distance (inf = miss)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dirs  # (R,3)
        t0 = (lo[None, :, :] - origins[:, None, :]) * inv[:, None, :]
        t1 = (hi[None, :, :] - origins[:, None, :]) * inv[:, None, :]
    tmin = np.minimum(t0, t1).max(axis=2)
    tmax = np.maximum(t0, t1).min(axis=2)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
    t = np.where(hit, tmin, np.inf)
    return t.min(axis=1)


def kitti_like_scan(n_points=120_000, seed=0, sensor_pose=None, pad=True):
    """One HDL-64-like sweep ray-cast against a ground plane and random boxes.

    Returns float32 (n_points, 3) in the sensor frame (like a KITTI .bin's xyz
    columns).  ``sensor_pose`` (4x4) moves the sensor inside the same static
    world, which is how scan *pairs* with real overlap are produced.
    """
    rng = np.random.default_rng(seed)
    world_rng = np.random.default_rng(1_000_003)  # the world is shared by all seeds
    n_box = 48
    centers = np.empty((n_box, 3))
    centers[:, 0] = world_rng.uniform(-70, 70, n_box)
    centers[:, 1] = world_rng.uniform(-45, 45, n_box)
    # keep a clear disc around the origin so the sensor is not inside a box
    near = np.hypot(centers[:, 0], centers[:, 1]) < 5.0
    centers[near, 0] += 12.0
    kind = world_rng.integers(0, 3, n_box)
    half = np.where(
        (kind == 0)[:, None],
        np.array([2.0, 0.9, 0.75]),
        np.where((kind == 1)[:, None], np.array([10.0, 0.15, 1.5]), np.array([0.15, 8.0, 1.5])),
    )
    ground_z = -1.73
    centers[:, 2] = ground_z + half[:, 2]
    lo, hi = centers - half, centers + half

    n_beams, n_az = 64, 2083
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    az = np.deg2rad(np.arange(n_az) * (360.0 / n_az))
    az = az[None, :] + rng.normal(0, 2e-4, (n_beams, n_az))
    el = elev[:, None] + rng.normal(0, 2e-4, (n_beams, n_az))
    d_local = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1).reshape(-1, 3)

    T = np.eye(4) if sensor_pose is None else np.asarray(sensor_pose, dtype=np.float64)
    R, o = T[:3, :3], T[:3, 3]
    d_world = d_local @ R.T
    origins = np.broadcast_to(o, d_world.shape)

    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (ground_z - origins[:, 2]) / d_world[:, 2]
    tg = np.where((d_world[:, 2] < 0) & (tg > 0), tg, np.inf)
    tb = np.full(d_world.shape[0], np.inf)
    step = 16384
    for s in range(0, d_world.shape[0], step):
        tb[s : s + step] = _ray_boxes(origins[s : s + step], d_world[s : s + step], lo, hi)
    t = np.minimum(tg, tb)
    keep = np.isfinite(t) & (t < 80.0) & (t > 1.3)
    keep &= rng.random(t.shape[0]) > 0.03
    t = t + rng.normal(0, 0.02, t.shape[0])
    pts = (d_local * t[:, None])[keep]

    if not pad and pts.shape[0] < n_points:
        return np.ascontiguousarray(pts, dtype=np.float32)   # every return of the sweep, no padding
    if pts.shape[0] >= n_points:
        sel = np.sort(rng.choice(pts.shape[0], n_points, replace=False))
        pts = pts[sel]
    else:
        # pad with jittered re-samples (never exact duplicates: NN ties are excluded from parity)
        extra = rng.integers(0, pts.shape[0], n_points - pts.shape[0])
        pad = pts[extra] + rng.normal(0, 0.015, (extra.shape[0], 3))
        pts = np.concatenate([pts, pad], axis=0)
    return np.ascontiguousarray(pts, dtype=np.float32)


def perturbed_pair(n_points=120_000, seed=0, angle_deg=1.15, t=(0.3, -0.15, 0.03)):
    """(source, target, T_true) with ``target ~= T_true * source``.

    Both clouds are separate sweeps of the same static world from two sensor
    poses (so they are not permutations of each other); ``T_true`` maps source
    sensor coordinates to target sensor coordinates.
    """
    T_true = rigid_transform((0.1, 0.2, 1.0), np.deg2rad(angle_deg), t)
    tgt = kitti_like_scan(n_points, seed=2 * seed, sensor_pose=np.eye(4))
    # the source sensor sits at pose T_true in the target frame: p_tgt = T_true * p_src
    src = kitti_like_scan(n_points, seed=2 * seed + 1, sensor_pose=T_true)
    return src, tgt, T_true


def object_cloud(n_points=2048, seed=0):
    """ModelNet40-like object: points on the surface of a few boxes inside the unit ball (float32)."""
    rng = np.random.default_rng(seed)
    n_parts = 5
    c = rng.uniform(-0.45, 0.45, (n_parts, 3))
    h = rng.uniform(0.05, 0.35, (n_parts, 3))
    part = rng.integers(0, n_parts, n_points)
    face = rng.integers(0, 6, n_points)
    u = rng.uniform(-1, 1, (n_points, 3))
    ax = face // 2
    sign = np.where(face % 2 == 0, -1.0, 1.0)
    u[np.arange(n_points), ax] = sign
    pts = c[part] + u * h[part]
    pts /= max(1.0, np.abs(pts).max())
    return np.ascontiguousarray(pts, dtype=np.float32)


def registration_pair_6f(n_points=20_000, seed=1000):
    """A registration_dataset-style pair: two (n,6) float32 records x,y,z,nx,ny,nz
    (Registration/main.py:10-17 reads 6-float records) plus the true transform."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0.5, 2.5)
    tt = rng.uniform(-0.4, 0.4, 3) * np.array([1.0, 1.0, 0.1])
    axis = np.array([rng.normal(0, 0.1), rng.normal(0, 0.1), 1.0])
    T_true = rigid_transform(axis, np.deg2rad(ang), tt)
    tgt = kitti_like_scan(n_points, seed=2 * seed, sensor_pose=np.eye(4))
    src = kitti_like_scan(n_points, seed=2 * seed + 1, sensor_pose=T_true)

    def six(p):
        out = np.zeros((p.shape[0], 6), dtype=np.float32)
        out[:, :3] = p
        out[:, 5] = 1.0
        return out

    return six(src), six(tgt), T_true


def registration_batch_6f(n_pairs=256, n_points=20_000, seed=1000, n_base=2, indices=None):
    """BASELINE configs[3]: ``n_pairs`` registration_dataset-style pairs of (n,6) float32 records.

    Ray-casting a sweep costs ~1 s, so the batch is derived from ``n_base`` base sweep pairs of the shared static
    world: pair i takes its own random ``n_points`` subset of each base sweep (different points, same scene), its own
    range noise and its own small rigid offset between source and target -- independent registration problems with
    the statistics of one scan pair.  ``indices`` limits generation to a rank's share (returns that many pairs).
    Returns a list of (src (n,6) f32, tgt (n,6) f32, T_true (4,4) f64).
    """
    base = []
    for b in range(n_base):
        T_b = rigid_transform((0.1, 0.2, 1.0), np.deg2rad(1.0 + 0.3 * b), (0.3 + 0.1 * b, -0.15, 0.03))
        tgt = kitti_like_scan(10**9, seed=2 * (seed + b), sensor_pose=np.eye(4), pad=False).astype(np.float64)
        src = kitti_like_scan(10**9, seed=2 * (seed + b) + 1, sensor_pose=T_b, pad=False).astype(np.float64)
        base.append((src, tgt, T_b))
    out = []
    for i in (range(n_pairs) if indices is None else indices):
        rng = np.random.default_rng(seed * 7919 + i)
        src, tgt, T_b = base[i % n_base]
        ang = rng.uniform(0.2, 1.5)
        axis = np.array([rng.normal(0, 0.1), rng.normal(0, 0.1), 1.0])
        D = rigid_transform(axis, np.deg2rad(ang), rng.uniform(-0.3, 0.3, 3) * np.array([1.0, 1.0, 0.1]))
        s = src[np.sort(rng.choice(len(src), n_points, replace=False))] + rng.normal(0, 0.01, (n_points, 3))
        t = tgt[np.sort(rng.choice(len(tgt), n_points, replace=False))] + rng.normal(0, 0.01, (n_points, 3))
        # p_tgt = T_b p_src; move the source by D^-1 so that T_true = T_b D
        s = (s - D[:3, 3]) @ D[:3, :3]
        rec_s = np.zeros((n_points, 6), dtype=np.float32)
        rec_t = np.zeros((n_points, 6), dtype=np.float32)
        rec_s[:, :3] = s
        rec_t[:, :3] = t
        rec_s[:, 5] = rec_t[:, 5] = 1.0
        out.append((rec_s, rec_t, T_b @ D))
    return out
