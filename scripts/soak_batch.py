"""Soak of the fused batch stages: random batches (pair counts, cloud sizes 3 .. 40 000, duplicates, given T0s, both modes, sub-batch sizes,
worker counts) through pcr_icp_batch and through the per-pair path; every field must agree bit for bit.  usage: soak_batch.py [rounds]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
batch = importlib.import_module("point-cloud-process_amd.batch")
syn = pcp.synthetic
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(12345)
keys = ("iters", "status", "n_assoc", "cost", "mean_d2")
bad = total = 0
t0 = time.time()
base = [syn.kitti_like_scan(40000, seed=900 + i) for i in range(3)] + [syn.object_cloud(4000, seed=7) * 20.0]
for r in range(rounds):
    n_pairs = int(rng.integers(1, 40))
    pairs = []
    for i in range(n_pairs):
        b = base[int(rng.integers(0, len(base)))]
        n = int(rng.choice([3, 5, 33, 64, 257, 1000, 5000, 20000, 40000]))
        n = min(n, len(b))
        tgt = b[np.sort(rng.choice(len(b), n, replace=False))].astype(np.float32)
        T = syn.rigid_transform(rng.normal(0, 1, 3), float(rng.uniform(0, 0.1)), rng.uniform(-0.5, 0.5, 3))
        m = int(rng.integers(max(3, n // 2), n + 1))
        src = ((tgt[:m].astype(np.float64) - T[:3, 3]) @ T[:3, :3] + rng.normal(0, 0.01, (m, 3))).astype(np.float32)
        if rng.random() < 0.1:
            src = np.concatenate([src, src[:5]])            # duplicate points
        if rng.random() < 0.2:                              # wider records (x,y,z,nx,ny,nz)
            src = np.hstack([src, rng.normal(0, 1, src.shape).astype(np.float32)])
        T0 = None if rng.random() < 0.5 else syn.rigid_transform((0, 0, 1), float(rng.uniform(0, 0.05)), rng.uniform(-0.2, 0.2, 3))
        pairs.append((np.ascontiguousarray(src), tgt, T0))
    kw = dict(mode="compat") if r % 2 == 0 else dict(mode="total", max_iter=int(rng.integers(1, 25)), r_thres=1e-4, t_thres=1e-4)
    os.environ["PCR_BATCH_PER_PAIR"] = "1"
    ref = batch.native_register_share(pairs, device=0, streams=1, **kw)
    os.environ["PCR_BATCH_PER_PAIR"] = "0"
    os.environ["PCR_BATCH_SUB"] = str(int(rng.choice([1, 3, 4, 7, 16, 64])))
    got = batch.native_register_share(pairs, device=0, streams=int(rng.integers(1, 6)), **kw)
    for i, (a, b) in enumerate(zip(ref, got)):
        total += 1
        if not (all(a[k] == b[k] for k in keys) and np.array_equal(a["T"], b["T"]) and np.array_equal(a["T_total"], b["T_total"])):
            bad += 1
            print("MISMATCH round", r, "pair", i, "sizes", pairs[i][0].shape, pairs[i][1].shape, {k: (a[k], b[k]) for k in keys}, flush=True)
    if r % 5 == 4:
        print(f"round {r + 1}: {total} pairs compared, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", total, "pairs")
