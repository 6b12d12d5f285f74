"""Small driver for rocprofv3: N ICP iterations on the 120k pair (grid or brute)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "grid"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cell = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
src, tgt, Tt = pcp.synthetic.perturbed_pair(120000, seed=0)
ctx = pcp.default_context()
index = pcp.TargetIndex(tgt, kind=kind, cell=cell)
sd = pcp.DeviceCloud.upload(src)
r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=iters, r_thres=-1, t_thres=-1, min_iter=iters)
print(kind, "cell", index.cell, "iters", r["iters"], "device ms/iter", r["device_ms"] / r["iters"], "n_assoc", r["n_assoc"])
