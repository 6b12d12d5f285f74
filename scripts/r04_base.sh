#!/bin/bash
# round-4 starting point: stage timings + rocprofv3 kernel stats of the global initialisation and of the config5 1M ICP leg
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_base
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/global_time.py > $O/global_time.txt 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/gprof -o g -- python3 $R/scripts/global_time.py > $O/gprof_log.txt 2>&1 || exit 1
SEQ=20,20,20 rocprofv3 --output-format csv --kernel-trace --stats -d $O/c5prof -o c5 -- python3 $R/scripts/c5_repro.py > $O/c5_log.txt 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,os
for d in ("gprof","c5prof"):
    for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04_base/"+d+"/**/*kernel_stats.csv",recursive=True):
        print("==",d)
        for row in list(csv.DictReader(open(f)))[:25]:
            print("%-60s calls %5s tot %9.1f us avg %8.1f min %8.1f max %8.1f" % (row["Name"].split("(")[0][:60], row["Calls"], float(row["TotalDurationNs"])/1e3, float(row["AverageNs"])/1e3, float(row["MinNs"])/1e3, float(row["MaxNs"])/1e3))
PY
cat $O/global_time.txt; tail -5 $O/c5_log.txt
