"""Fused batch stages (pcr_batch.hip) against the per-pair path: bitwise comparison on unequal pairs, then the
256 x 20 000 batch of BASELINE configs[3] timed both ways.  usage: python scripts/batch_fused_check.py [pairs] [points]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
batch = importlib.import_module("point-cloud-process_amd.batch")
syn = pcp.synthetic


def run(pairs, per_pair, sub=None, streams=4, **kw):
    os.environ["PCR_BATCH_PER_PAIR"] = "1" if per_pair else "0"
    if sub is None:
        os.environ.pop("PCR_BATCH_SUB", None)
    else:
        os.environ["PCR_BATCH_SUB"] = str(sub)
    return batch.native_register_share(pairs, device=0, streams=streams, **kw)


def same(a, b):
    return (a["iters"] == b["iters"] and a["n_assoc"] == b["n_assoc"] and a["status"] == b["status"] and np.array_equal(a["T"], b["T"])
            and np.array_equal(a["T_total"], b["T_total"]) and a["cost"] == b["cost"] and a["mean_d2"] == b["mean_d2"])


rng = np.random.default_rng(5)
pairs = []
for i in range(14):
    n = int(rng.choice([40, 600, 3000, 12000, 25000]))
    s, t, _ = syn.perturbed_pair(n, seed=300 + i, angle_deg=float(rng.uniform(0.5, 6.0)), t=tuple(rng.uniform(-0.8, 0.8, 3) * [1, 1, 0.1]))
    pairs.append((s, t, None if i % 3 else syn.rigid_transform((0, 0, 1), 0.01, (0.05, 0, 0))))
# a pair whose source is far away from its target: fewer than 3 associations -> the soft failure of main.py:125-127
pairs.append((pairs[1][0] + np.float32(500.0), pairs[1][1], None))
bad = 0
for kw in (dict(mode="compat"), dict(mode="total", max_iter=40, r_thres=1e-4, t_thres=1e-4), dict(mode="total", max_iter=3, r_thres=1e-9, t_thres=1e-9)):
    ref = run(pairs, True, streams=1, **kw)
    for sub, streams in ((4, 1), (5, 3), (64, 2)):
        got = run(pairs, False, sub=sub, streams=streams, **kw)
        n_bad = sum(0 if same(a, b) else 1 for a, b in zip(ref, got))
        bad += n_bad
        print(f"{kw.get('mode')} max_iter={kw.get('max_iter', 100)} sub={sub} streams={streams}: {len(pairs) - n_bad}/{len(pairs)} bitwise equal; iters",
              [r["iters"] for r in got], "status", [r["status"] for r in got])
        if n_bad:
            for i, (a, b) in enumerate(zip(ref, got)):
                if not same(a, b):
                    print("  pair", i, "iters", a["iters"], b["iters"], "n_assoc", a["n_assoc"], b["n_assoc"], "|dT|", np.abs(a["T"] - b["T"]).max())
print("BITWISE", "OK" if bad == 0 else f"FAILED ({bad})")

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
big = [(s6, t6, None) for s6, t6, _ in syn.registration_batch_6f(P, N, seed=1000)]
os.environ["PCR_BATCH_TIMING"] = "1"
for tag, kw in (("compat", dict(mode="compat")), ("tight", dict(mode="total", max_iter=30, r_thres=1e-3, t_thres=1e-3))):
    for per_pair, sub in ((True, None), (False, None), (False, 8), (False, 32), (False, 64)):
        run(big[:16], per_pair, sub=sub, streams=8, **kw)   # warm
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            res = run(big, per_pair, sub=sub, streams=8, **kw)
            best = min(best, time.perf_counter() - t0)
        print(f"{tag} {'per-pair' if per_pair else 'fused'} sub={sub}: {P / best:.0f} pairs/s ({best * 1e3:.1f} ms), mean iters {np.mean([r['iters'] for r in res]):.2f}", flush=True)
