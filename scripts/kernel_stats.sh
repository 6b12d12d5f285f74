#!/bin/bash
# quick rocprofv3 kernel-trace stats of N grid ICP passes (prof_pass.py): top kernels by total time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/st && mkdir -p $R/gpurun_out/st
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/st -o st -- python3 $R/scripts/prof_pass.py grid ${1:-30} > $R/gpurun_out/st/log.txt 2>&1
python3 - <<'PY'
import csv,glob,os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/st/**/*kernel_stats.csv",recursive=True):
    for row in list(csv.DictReader(open(f)))[:6]:
        print("%-50s calls %4s avg %8.1f us min %8.1f max %8.1f" % (row["Name"].split("(")[0][:50], row["Calls"], float(row["AverageNs"])/1e3, float(row["MinNs"])/1e3, float(row["MaxNs"])/1e3))
PY
tail -2 $R/gpurun_out/st/log.txt
