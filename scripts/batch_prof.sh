# rocprofv3 kernel + copy stats of BASELINE configs[3] through pcr_icp_batch -> gpurun_out/bp (copy what is to be judged into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
MODE=${MODE:-compat}
rm -rf $R/gpurun_out/bp/st_* $R/gpurun_out/bp/log.txt && mkdir -p $R/gpurun_out/bp   # (the copies named by TAG and MODE stay)
REPS=1 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/bp -o st -- python3 $R/scripts/batch_prof.py 8 256 $MODE > $R/gpurun_out/bp/log.txt 2>&1
grep "pairs/s\|pcr_icp_batch" $R/gpurun_out/bp/log.txt
python3 - <<'PY'
import csv,glob,os
R=os.environ["GRAFT_REPO_ROOT"]; tag=os.environ.get("TAG","r03"); mode=os.environ.get("MODE","compat")
for f in glob.glob(R+"/gpurun_out/bp/**/st_kernel_stats.csv",recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e3
    n=sum(int(r["Calls"]) for r in rows)
    print("total kernel time %.0f us, %d launches over 2 x 256 pairs (warm-up + timed) = %.1f us and %.2f launches per pair"%(tot, n, tot/512, n/512))
    for row in rows[:16]:
        print("  %-60s calls %5s total %8.0f us avg %7.1f" % (row["Name"].split("(")[0][:60], row["Calls"], float(row["TotalDurationNs"])/1e3, float(row["AverageNs"])/1e3))
    os.system("cp %s %s/gpurun_out/bp/%s_batch_%s_kernel_stats.csv" % (f, R, tag, mode))
for f in glob.glob(R+"/gpurun_out/bp/**/st_memory_copy_stats.csv",recursive=True):
    print(open(f).read()[:800])
    os.system("cp %s %s/gpurun_out/bp/%s_batch_%s_memory_copy_stats.csv" % (f, R, tag, mode))
PY
