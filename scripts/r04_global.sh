#!/bin/bash
# stage timings + rocprofv3 kernel stats of the global initialisation (TAG for the output directory)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r04_global}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/global_time.py > $O/global_time.txt 2>&1 || { tail -20 $O/global_time.txt; exit 1; }
STAGES=0 PAIRS=16 rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof -o g -- python3 $R/scripts/global_time.py > $O/prof_log.txt 2>&1 || { tail -20 $O/prof_log.txt; exit 1; }
python3 - <<'PY'
import csv,glob,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ.get("TAG","r04_global")
for f in glob.glob(O+"/prof/**/*kernel_stats.csv",recursive=True):
    for row in list(csv.DictReader(open(f)))[:22]:
        print("%-60s calls %5s tot %9.1f us avg %8.1f min %8.1f max %8.1f" % (row["Name"].split("(")[0][:60], row["Calls"], float(row["TotalDurationNs"])/1e3, float(row["AverageNs"])/1e3, float(row["MinNs"])/1e3, float(row["MaxNs"])/1e3))
PY
cat $O/global_time.txt
