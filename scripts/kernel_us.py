"""Per-kernel microseconds of the grid ICP pass (HIP events on the library's stream), 120k pair."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
src, tgt, _ = pcr.synthetic.perturbed_pair(int(os.environ.get("N", 120000)), seed=0)
ctx = pcr.default_context()
index = pcr.TargetIndex(tgt, cell=float(os.environ.get('CELL', 0)))
sd = pcr.DeviceCloud.upload(src).prepare(index)
pcr.icp_device(sd, index, np.eye(4), mode="total", max_iter=10, r_thres=-1, t_thres=-1, min_iter=10)
ctx.profile(True)
r = pcr.icp_device(sd, index, np.eye(4), mode="total", max_iter=50, r_thres=-1, t_thres=-1, min_iter=50)
ms, n = ctx.profile_read()
ctx.profile(False)
print("cell %.3f" % index.cell, "skip", os.environ.get("PCR_TILE_SKIP", "0"), "tile/hard/accumulate us:", np.round(ms[:3] / n * 1e3, 1), "n_assoc", r["n_assoc"])
