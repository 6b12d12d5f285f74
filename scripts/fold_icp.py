"""Fold scripts/profile_icp.sh's rocprofv3 outputs into the files committed under profiles/ (<TAG>_*, TAG from the environment, default r03)."""
import collections, csv, glob, json, os, shutil, sys

out = sys.argv[1]
TAG = os.environ.get("TAG", "r03")


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items() if k.startswith("grid_")}


for name in ("grid", "brute", "bench"):
    for f in glob.glob(os.path.join(out, name + "_stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, f"{TAG}_{name}_kernel_stats.csv"))
traffic = {"note": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), `python3 scripts/prof_pass.py "
                   "grid 10` (ten ICP passes of the 120k x 120k pair inside the device-resident loop), mean per launch; "
                   "FETCH_SIZE / WRITE_SIZE in KiB as reported; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per MI355X_MICROARCH.md "
                   "(gfx950 halves FETCH_SIZE on 16-B/lane reads; uncalibrated for this kernel's scattered 16..64-byte accesses; "
                   "Infinity-Cache hits are counted)", "kernels": {}}
f, w, t = counters("pmc_FETCH_SIZE"), counters("pmc_WRITE_SIZE"), counters("pmc_TCC_HIT_sum_TCC_MISS_sum")
for k in f:
    e = {"FETCH_SIZE": f[k].get("FETCH_SIZE"), "WRITE_SIZE": w.get(k, {}).get("WRITE_SIZE"), "TCC_HIT_sum": t.get(k, {}).get("TCC_HIT_sum"),
         "TCC_MISS_sum": t.get(k, {}).get("TCC_MISS_sum")}
    if e["FETCH_SIZE"] is not None and e["WRITE_SIZE"] is not None:
        e["hbm_bytes"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if e["TCC_HIT_sum"] and e["TCC_MISS_sum"] is not None:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    traffic["kernels"][k] = e
json.dump(traffic, open(os.path.join(out, TAG + "_pmc_traffic.json"), "w"), indent=1)
sq = {}
for sub in ("sq1", "sq2", "sq3"):
    for k, d in counters(sub).items():
        sq.setdefault(k, {}).update(d)
with open(os.path.join(out, TAG + "_sq_counters.txt"), "w") as fh:
    fh.write("rocprofv3 --pmc (three passes) -- python3 scripts/prof_pass.py grid 10   (MI355X, 120k x 120k pair, mean per launch, whole chip;\n"
             "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_INSTS_* count wave instructions)\n\n")
    names = sorted({c for d in sq.values() for c in d})
    fh.write("%-22s" % "" + "".join("%26s" % k for k in sq) + "\n")
    for c in names:
        fh.write("%-22s" % c + "".join("%26.0f" % sq[k].get(c, float("nan")) for k in sq) + "\n")
    fh.write("\n")
    for k, d in sq.items():
        if "SQ_WAVES" in d and "SQ_INSTS_VALU" in d and "SQ_WAVE_CYCLES" in d:
            fh.write("%s: %.0f VALU + %.0f SALU + %.0f LDS + %.0f VMEM-read instructions per wave; wave lifetime %.0f cycles, of which waiting on "
                     "memory counters %.0f %%, VALU issue %.0f %%\n" % (k, d["SQ_INSTS_VALU"] / d["SQ_WAVES"], d.get("SQ_INSTS_SALU", 0) / d["SQ_WAVES"],
                                                                        d.get("SQ_INSTS_LDS", 0) / d["SQ_WAVES"], d.get("SQ_INSTS_VMEM_RD", 0) / d["SQ_WAVES"],
                                                                        4 * d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"], 100 * d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"],
                                                                        100 * d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"]))
json.dump({"note": "rocprofv3 --pmc (three passes) -- python3 scripts/prof_pass.py grid 10; mean per launch, whole chip", "kernels": sq},
          open(os.path.join(out, TAG + "_sq_counters.json"), "w"), indent=1)
print(open(os.path.join(out, TAG + "_sq_counters.txt")).read())
print(json.dumps(traffic["kernels"], indent=1))
for name in ("grid", "brute", "bench"):
    p = os.path.join(out, f"{TAG}_{name}_kernel_stats.csv")
    if os.path.exists(p):
        print(name, "kernel stats:")
        for row in list(csv.DictReader(open(p)))[:6]:
            print("   %-60s calls %4s avg %8.1f us" % (row["Name"].split("(")[0][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
