import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr
src, tgt, _ = pcr.synthetic.perturbed_pair(120000, seed=0)
index = pcr.TargetIndex(tgt)
outs = []
for rep in range(6):
    sd = pcr.DeviceCloud.upload(src)
    r = pcr.icp_device(sd, index, np.eye(4), mode="total", max_iter=100, r_thres=-1.0, t_thres=-1.0, min_iter=100)
    outs.append(r["T_total"].copy())
    sd.free()
    print(rep, r["n_assoc"], r["T_total"][:3, 3], "bitwise same as run 0:", outs[-1].tobytes() == outs[0].tobytes(), flush=True)
