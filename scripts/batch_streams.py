"""Config 4 on one GPU: 256 pairs x 20 000 points, pairs per second against the number of pairs in flight."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
syn = pcr.synthetic
pairs = [syn.registration_pair_6f(20000, seed=1000 + i)[:2] + (None,) for i in range(32)] * 8
for streams in (1, 2, 4, 8, 12, 16):
    pcr.register_batch(pairs[:streams * 2], streams=streams)
    t0 = time.perf_counter()
    res = pcr.register_batch(pairs, streams=streams)
    el = time.perf_counter() - t0
    print(f"streams {streams}: {el:.3f} s  {len(pairs)/el:.0f} pairs/s  mean iters {np.mean([r['iters'] for r in res]):.1f}", flush=True)
