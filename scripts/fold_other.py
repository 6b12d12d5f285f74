"""Fold the rocprofv3 outputs of scripts/profile_other.sh into one JSON (kernel stats, PMC traffic, roofline fraction)."""
import collections, csv, glob, json, os, sys

root = sys.argv[1]
HBM_PEAK = 8.0e12
res = {"note": "rocprofv3 --kernel-trace --stats and, in separate passes, --pmc FETCH_SIZE / --pmc WRITE_SIZE of "
               "`python3 scripts/other_kernels.py <op> 5` (6 calls of the op per process: 1 warm-up + 5); hbm_bytes = "
               "(2 x FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (gfx950 halves FETCH_SIZE on wide reads; uncalibrated for "
               "scattered 24..32-byte accesses; Infinity-Cache hits are counted); frac = algorithmic bytes / dominant-kernel time / 8 TB/s",
       "ops": {}}
for op in sorted(os.listdir(root)):
    d = os.path.join(root, op)
    if not os.path.isdir(d) or not os.path.exists(os.path.join(d, "wall.json")):
        continue
    try:
        wall = json.loads(open(os.path.join(d, "wall.json")).read().strip().splitlines()[-1])
    except Exception:
        continue
    calls = wall["reps"] + 1
    kernels = []
    for f in glob.glob(d + "/stats/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kernels.append({"name": row["Name"].split("(")[0][:90], "calls": int(row["Calls"]), "total_us": float(row["TotalDurationNs"]) / 1e3,
                            "avg_us": float(row["AverageNs"]) / 1e3})
    kernels.sort(key=lambda k: -k["total_us"])
    traffic = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
    for which in ("fetch", "write"):
        for f in glob.glob(d + f"/{which}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                traffic[row["Kernel_Name"].split("(")[0][:90]][row["Counter_Name"]] += float(row["Counter_Value"])
    tot_us = sum(k["total_us"] for k in kernels)
    per_call_us = tot_us / calls
    dom = kernels[0] if kernels else None
    entry = {"wall_ms_per_call": wall["ms"], "params": {k: v for k, v in wall.items() if k not in ("ms", "op", "reps", "algorithmic_bytes")},
             "algorithmic_bytes": wall["algorithmic_bytes"], "launches_per_call": sum(k["calls"] for k in kernels) / calls,
             "sum_kernel_us_per_call": per_call_us,
             "kernels": [{**k, "us_per_call": k["total_us"] / calls} for k in kernels[:8]]}
    if dom:
        dom_us = dom["total_us"] / calls
        t = traffic.get(dom["name"], {"FETCH_SIZE": 0, "WRITE_SIZE": 0})
        all_fetch = sum(v["FETCH_SIZE"] for v in traffic.values()) / calls
        all_write = sum(v["WRITE_SIZE"] for v in traffic.values()) / calls
        entry["roofline"] = {"bound": "hbm", "kernel": dom["name"], "kernel_us_per_call": dom_us,
                             "achieved_GBs": wall["algorithmic_bytes"] / (dom_us * 1e-6) / 1e9, "peak_GBs": HBM_PEAK / 1e9,
                             "frac": wall["algorithmic_bytes"] / (dom_us * 1e-6) / HBM_PEAK,
                             "frac_all_kernels": wall["algorithmic_bytes"] / (per_call_us * 1e-6) / HBM_PEAK,
                             "traffic_dominant_kernel_bytes": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / calls,
                             "traffic_all_kernels_bytes": (2 * all_fetch + all_write) * 1024}
    res["ops"][op] = entry
print(json.dumps(res, indent=1))
