cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/bp && mkdir -p $R/gpurun_out/bp
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/bp -o st -- python3 $R/scripts/batch_prof.py 12 > $R/gpurun_out/bp/log.txt 2>&1
grep "pairs/s" $R/gpurun_out/bp/log.txt
python3 - <<'PY'
import csv,glob,os
tot=0
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/bp/**/*kernel_stats.csv",recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e3
    print("total kernel time %.0f us over 68 pairs = %.0f us/pair, launches %d = %.0f per pair"%(tot, tot/68, sum(int(r["Calls"]) for r in rows), sum(int(r["Calls"]) for r in rows)/68))
    for row in rows[:12]:
        print("  %-60s calls %5s total %8.0f us avg %6.1f" % (row["Name"].split("(")[0][:60], row["Calls"], float(row["TotalDurationNs"])/1e3, float(row["AverageNs"])/1e3))
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/bp/**/*memory_copy_stats.csv",recursive=True):
    print(open(f).read()[:800])
PY
