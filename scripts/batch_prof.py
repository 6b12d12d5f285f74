"""BASELINE configs[3] through pcr_icp_batch (fused batch stages): driver for rocprofv3 kernel stats.
usage: python3 scripts/batch_prof.py [streams=8] [pairs=256] [compat|tight] ; PCR_BATCH_PER_PAIR=1 for the per-pair path"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
batch = importlib.import_module("point-cloud-process_amd.batch")
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else "compat"
kw = dict(mode="compat") if mode == "compat" else dict(mode="total", max_iter=30, r_thres=1e-3, t_thres=1e-3)
pairs = [(s, t, None) for s, t, _ in pcp.synthetic.registration_batch_6f(n_pairs, 20000, seed=1000)]
batch.native_register_share(pairs, device=0, streams=streams, **kw)   # creates the pooled contexts and their buffers, untimed
os.environ["PCR_BATCH_TIMING"] = "1"
best = 1e9
for rep in range(int(os.environ.get("REPS", "3"))):
    t0 = time.perf_counter()
    res = batch.native_register_share(pairs, device=0, streams=streams, **kw)
    best = min(best, time.perf_counter() - t0)
print("pairs/s %.0f (%.2f ms)  mean iters %.2f" % (len(pairs) / best, best * 1e3, np.mean([r["iters"] for r in res])))
