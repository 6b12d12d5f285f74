"""Soak of the fused global initialisation against the scan-by-scan path: SHARES random shares (3-14 pairs over a random scan table: scans of
60 .. 60 000 points, some used by several pairs, 3- / 4- / 6-float records, clusters far from the origin, duplicated points), both paths in
this process (PCR_INIT_PER_SCAN is read per call): T_init, T, iterations and status must be the same bits."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("point-cloud-process_amd")
batch = importlib.import_module("point-cloud-process_amd.batch")
syn = pkg.synthetic
SHARES = int(os.environ.get("SHARES", 24))
rng = np.random.default_rng(int(os.environ.get("SEED", 11)))
base = [syn.kitti_like_scan(60000, seed=700 + b).astype(np.float64) for b in range(4)]
bad = 0; t0 = time.time(); pairs_done = 0
for share in range(SHARES):
    n_scans = int(rng.integers(3, 10))
    scans = []
    for s in range(n_scans):
        w = base[int(rng.integers(0, 4))]
        n = int(rng.choice([60, 500, 3000, 9000, 20000, 60000]))
        sel = rng.choice(len(w), n, replace=False)
        T = syn.rigid_transform(rng.normal(size=3) * 0.05 + np.array([0, 0, 1.0]), np.deg2rad(rng.uniform(-25, 25)), rng.uniform(-3, 3, 3) * np.array([1, 1, 0.05]))
        p = w[sel] @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.01, (n, 3))
        if rng.random() < 0.2:
            p = p + np.array([4000.0, -2500.0, 30.0])            # far from the origin
        if rng.random() < 0.2:
            p[: n // 10] = p[n // 10: 2 * (n // 10)][: n // 10]   # duplicated points
        cols = int(rng.choice([3, 4, 6]))
        a = np.zeros((n, cols), dtype=np.float32)
        a[:, :3] = p
        scans.append(a)
    pairs = []
    for _ in range(int(rng.integers(3, 15))):
        i, j = rng.choice(n_scans, 2, replace=False)
        T0 = None if rng.random() < 0.85 else syn.rigid_transform((0, 0, 1), 0.05, (0.2, 0.1, 0.0))
        pairs.append((scans[i], scans[j], T0))
    os.environ.pop("PCR_INIT_PER_SCAN", None)
    a = batch.native_register_share(pairs, device=0, streams=4, global_init=True, return_init=True)
    os.environ["PCR_INIT_PER_SCAN"] = "1"
    b = batch.native_register_share(pairs, device=0, streams=4, global_init=True, return_init=True)
    os.environ.pop("PCR_INIT_PER_SCAN", None)
    for k, (x, y) in enumerate(zip(a, b)):
        same = np.array_equal(x["T_init"], y["T_init"]) and np.array_equal(x["T"], y["T"]) and x["iters"] == y["iters"] and x["status"] == y["status"]
        if not same:
            bad += 1
            print("MISMATCH share", share, "pair", k, "sizes", len(pairs[k][0]), len(pairs[k][1]), flush=True)
    pairs_done += len(pairs)
    if share % 6 == 5:
        print(f"share {share}: {pairs_done} pairs so far, mismatches {bad}, {time.time() - t0:.0f} s", flush=True)
print("DONE shares", SHARES, "pairs", pairs_done, "mismatches", bad)
sys.exit(1 if bad else 0)
