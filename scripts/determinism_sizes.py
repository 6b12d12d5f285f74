"""Bitwise reproducibility of the ICP loop across cloud sizes (one and several wave generations per launch)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pcp = importlib.import_module("point-cloud-process_amd")
for n in [int(x) for x in os.environ.get("SIZES", "120000,140000,200000,400000,1000000").split(",")]:
    src, tgt, _ = pcp.synthetic.perturbed_pair(n, seed=0)
    index = pcp.TargetIndex(tgt)
    outs = []
    for rep in range(4):
        sd = pcp.DeviceCloud.upload(src)
        r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=8, r_thres=-1.0, t_thres=-1.0, min_iter=8)
        outs.append((r["T_total"].tobytes(), int(r["n_assoc"])))
        sd.free()
    index.free()
    print(n, "n_assoc", [o[1] for o in outs], "bitwise equal:", all(o == outs[0] for o in outs), "device us/iter %.1f" % (r["device_ms"] / r["iters"] * 1e3))
