"""One-off soak: many random clouds / cell sizes / transforms through the exact 1-NN grid path (vs scipy) and repeated ICP
runs for bitwise reproducibility.  Prints a summary; not part of the test suite."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
oracle = importlib.import_module("oracle.oracle_np")
syn = pcr.synthetic
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = 0
t0 = time.time()
n_cases = int(os.environ.get("CASES", 150))
for case in range(n_cases):
    kind = rng.integers(0, 4)
    nt = int(rng.integers(500, 60000)); nq = int(rng.integers(100, 30000))
    if kind == 0:
        tgt = rng.uniform(-10, 10, (nt, 3)) * rng.uniform(0.1, 5, 3)
    elif kind == 1:
        tgt = syn.kitti_like_scan(nt, seed=int(rng.integers(0, 1000))).astype(np.float64)
    elif kind == 2:
        c = rng.uniform(-20, 20, (8, 3)); tgt = c[rng.integers(0, 8, nt)] + rng.normal(0, rng.uniform(0.01, 2), (nt, 3))
    else:
        tgt = np.round(rng.uniform(-3, 3, (nt, 3)) * 4) / 4 + rng.normal(0, 1e-3, (nt, 3)) * (rng.random() < 0.5)
    tgt = tgt + rng.uniform(-1e3, 1e3, 3) * (rng.random() < 0.3)
    q = tgt[rng.integers(0, nt, nq)] + rng.normal(0, rng.uniform(0.001, 1.0), (nq, 3))
    cell = 0.0 if rng.random() < 0.5 else float(rng.uniform(0.02, 3.0))
    T = syn.rigid_transform(rng.normal(size=3), rng.uniform(0, 0.2), rng.normal(0, 0.3, 3))
    index = pcr.TargetIndex(tgt, cell=cell)
    idx, d2 = index.nn1(q, T=T)
    qt = q @ T[:3, :3].T + T[:3, 3]
    # the device transforms with the same operation order as xform_apply: ((r0*x + r1*y) + r2*z) + t
    qt = np.stack([((T[i, 0] * q[:, 0] + T[i, 1] * q[:, 1]) + T[i, 2] * q[:, 2]) + T[i, 3] for i in range(3)], axis=1)
    oi, od2, margin = oracle.nn1_exact(qt, tgt, workers=-1)
    clear = margin > 1e-12
    ok = np.array_equal(d2, od2) and np.array_equal(idx[clear], oi[clear])
    if not ok:
        bad += 1
        print("MISMATCH case", case, "kind", kind, nt, nq, "cell", cell, "d2 diffs", int((d2 != od2).sum()), "idx diffs", int((idx[clear] != oi[clear]).sum()), flush=True)
    index.free()
print(f"nn1 soak: {n_cases} cases, {bad} bad, {time.time()-t0:.1f} s", flush=True)
# repeated ICP: bitwise equal
for seed in range(3):
    src, tgt, _ = syn.perturbed_pair(120000, seed=seed)
    index = pcr.TargetIndex(tgt)
    ref = None
    for rep in range(15):
        sd = pcr.DeviceCloud.upload(src)
        r = pcr.icp_device(sd, index, np.eye(4), mode="total", max_iter=40, r_thres=-1.0, t_thres=-1.0, min_iter=40)
        b = r["T_total"].tobytes()
        sd.free()
        if ref is None: ref = b
        elif b != ref:
            bad += 1; print("NONDETERMINISTIC pair", seed, "rep", rep, flush=True)
    index.free()
print("determinism soak done; total bad =", bad, flush=True)
