#!/bin/bash
# Memory-pipeline counters of the 1 M x 1 M registration's kernels (grid_pass_kernel / grid_drain_kernel), one counter group per run;
# per-dispatch rows of the first call's passes.  Output: gpurun_out/pmc_drain/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_drain
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  STAGE=none SEQ=20 PASSLOG=0 timeout -k 10 300 rocprofv3 --output-format csv --pmc $grp -d $OUT/p$i -o p -- python3 $R/scripts/c5_repro.py > $OUT/p$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_drain"
for p in sorted(glob.glob(out + "/p*/")):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if k in ("grid_pass_kernel", "grid_drain_kernel", "grid_drain4_kernel"):
                rows[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for k, d in rows.items():
        for c, v in d.items():
            v.sort()
            vals = [x[1] for x in v]
            print(f"{k:20s} {c:36s} first {vals[0]:.4g}  second {vals[1] if len(vals) > 1 else 0:.4g}  mean {sum(vals) / len(vals):.4g}  last {vals[-1]:.4g}  n {len(vals)}")
PY
cat $OUT/summary.txt
