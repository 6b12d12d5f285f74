#!/bin/bash
# rocprofv3 kernel stats of the fused global initialisation (scripts/init_fused_check.py's fused leg alone)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/init_fused_prof
rm -rf $OUT && mkdir -p $OUT
PAIRS=${PAIRS:-48} rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o st -- python3 $R/scripts/init_fused_check.py /tmp/init_prof.npz > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/init_fused_prof"
for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6, "calls", sum(int(r["Calls"]) for r in rows))
    for r in rows[:28]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
