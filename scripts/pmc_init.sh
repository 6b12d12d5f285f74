#!/bin/bash
# SQ counters of the fused global initialisation's kernels (scripts/init_fused_check.py's fused leg), one group per run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_init
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  PAIRS=${PAIRS:-48} timeout -k 10 300 rocprofv3 --output-format csv --pmc $grp -d $OUT/p$i -o p -- python3 $R/scripts/init_fused_check.py /tmp/init_pmc.npz > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_init"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if any(t in k for t in ("scans_kernel", "jobs_kernel")):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k[:60], {c: float("%.4g" % (sum(v) / len(v))) for c, v in sorted(d.items())})
PY
