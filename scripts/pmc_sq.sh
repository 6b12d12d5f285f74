#!/bin/bash
# SQ counter passes of the grid ICP pass (one counter group per run) + kernel-trace stats; outputs under gpurun_out/pmc_sq/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o st -- python3 $R/scripts/prof_pass.py grid 20 > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/p1 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/p2 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/p2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT/p3 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_sq"
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{p}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        if "grid_" in k:
            print(p, k[:40], {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
