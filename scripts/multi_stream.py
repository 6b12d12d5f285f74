"""Aggregate ICP-iteration throughput with S independent 120k pairs in flight on one GPU (one context/stream per pair)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcp_amd as pcr

N = int(os.environ.get("N", 120000))
ITERS = 60
pairs = [pcr.synthetic.perturbed_pair(N, seed=s) for s in range(4)]
for S in (1, 2, 3, 4, 6, 8):
    ctxs = [pcr.Context(0) for _ in range(S)]
    work = []
    for i, c in enumerate(ctxs):
        src, tgt, _ = pairs[i % len(pairs)]
        idx = pcr.TargetIndex(pcr.DeviceCloud.upload(tgt, c), ctx=c)
        sd = pcr.DeviceCloud.upload(src, c)
        sd.prepare(idx)
        work.append((sd, idx))
    def run(i, out):
        sd, idx = work[i]
        r = pcr.icp_device(sd, idx, np.eye(4), mode="total", max_iter=ITERS, r_thres=-1.0, t_thres=-1.0)
        out[i] = r["iters"]
    out = [0] * S
    for rep in range(2):
        bar = time.perf_counter()
        th = [threading.Thread(target=run, args=(i, out)) for i in range(S)]
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - bar
    its = sum(out)
    print(f"S={S}: {its} iterations in {dt*1e3:.2f} ms -> {dt/its*1e6*S:.1f} us/iter/pair latency, {its*N/dt/1e9:.2f} G corr/s aggregate", flush=True)
    for sd, idx in work:
        sd.free(); idx.free()
    for c in ctxs: c.close()
