"""ms per ICP iteration on the 120k pair as a function of PCR_ICP_LANES (set in the environment before running)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
N = int(os.environ.get("N", 120000))
src, tgt, Tt = pcr.synthetic.perturbed_pair(N, seed=0)
ctx = pcr.default_context()
index = pcr.TargetIndex(tgt)
for rep in range(3):
    sd = pcr.DeviceCloud.upload(src).prepare(index)
    ctx.sync()
    t0 = time.perf_counter()
    r = pcr.icp_device(sd, index, np.eye(4), mode="total", max_iter=100, r_thres=-1, t_thres=-1, min_iter=100)
    wall = time.perf_counter() - t0
    sd.free()
print("lanes", os.environ.get("PCR_ICP_LANES", "1"), "N", N, "us/iter %.1f" % (1e6 * wall / r["iters"]), "n_assoc", r["n_assoc"],
      "T", np.round(r["T_total"][:3, 3], 6))
