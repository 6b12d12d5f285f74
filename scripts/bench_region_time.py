"""The pieces of bench.py's timed region (sync, the icp_device call, sync, torch.cuda.synchronize) for a 5-pass warm-up call followed
by 20-pass calls, as the driver runs it (--steps 20 --warmup 5)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pcp = importlib.import_module("point-cloud-process_amd")
src, tgt, _ = pcp.synthetic.perturbed_pair(120000, seed=0)
torch.cuda.set_device(0); torch.cuda.synchronize()
ctx = pcp.Context(0) if hasattr(pcp, "Context") else pcp.default_context()
index = pcp.TargetIndex(pcp.DeviceCloud.upload(tgt, ctx), ctx=ctx)
for rep, IT in enumerate((5, 20, 20, 20)):
    sd = pcp.DeviceCloud.upload(src, ctx).prepare(index)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.sync(); a = time.perf_counter()
    r = pcp.icp_device(sd, index, np.eye(4), mode="total", max_iter=IT, r_thres=-1.0, t_thres=-1.0, max_d2=5.0, min_iter=IT)
    b = time.perf_counter()
    ctx.sync(); c = time.perf_counter()
    torch.cuda.synchronize(); d = time.perf_counter()
    print(IT, "passes: sync %.1f | icp_device %.1f (device %.1f) | sync %.1f | torch sync %.1f | total %.1f us" % (1e6*(a-t0), 1e6*(b-a), 1e3*r["device_ms"], 1e6*(c-b), 1e6*(d-c), 1e6*(d-t0)), flush=True)
    sd.free()
