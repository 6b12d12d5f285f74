import sys, time, importlib, numpy as np
sys.path.insert(0, "/root/repo")
pcp = importlib.import_module("point-cloud-process_amd")
syn = pcp.synthetic
src, tgt, Tt = syn.perturbed_pair(120000, seed=0)
ctx = pcp.default_context()
print(ctx.device_info())
for kind in ("grid", "brute"):
    t0 = time.time(); index = pcp.TargetIndex(tgt, kind=kind); ctx.sync(); print(kind, "build s", time.time() - t0, "cell", index.cell)
    t0 = time.time(); index2 = pcp.TargetIndex(tgt, kind=kind); ctx.sync(); print(kind, "build2 s", time.time() - t0)
    sd = pcp.DeviceCloud.upload(src)
    for rep in range(3):
        sd2 = pcp.DeviceCloud.upload(src)
        t0 = time.time()
        r = pcp.icp_device(sd2, index, np.eye(4), mode="total", max_iter=20, r_thres=0, t_thres=0, min_iter=20)
        wall = time.time() - t0
        print(kind, "iters", r["iters"], "wall ms/iter", 1e3 * wall / r["iters"], "device ms/iter", r["device_ms"] / r["iters"], "nn kernel ms/iter", r["nn_kernel_ms"] / r["nn_launches"], "n_assoc", r["n_assoc"])
        sd2.free()
