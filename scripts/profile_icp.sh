#!/bin/bash
# rocprofv3 evidence for the ICP pass kernels (grid and brute), final code: kernel-trace stats, then HBM-side traffic and
# L2 hit counters in SEPARATE --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass), then the
# SQ issue/wait counters.  The program itself follows `--` (no env/bash hop).  Output: gpurun_out/prof_icp/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_icp
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/grid_stats -o st -- python3 $R/scripts/prof_pass.py grid 50 > $OUT/grid_stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/brute_stats -o st -- python3 $R/scripts/prof_pass.py brute 10 > $OUT/brute_stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/bench_stats -o st -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-batch --in-flight 0 > $OUT/bench_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --output-format csv --pmc $c -d $OUT/pmc_$tag -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/pmc_$tag.log 2>&1
done
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/sq2 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/sq2.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT/sq3 -o p -- python3 $R/scripts/prof_pass.py grid 10 > $OUT/sq3.log 2>&1
python3 $R/scripts/fold_icp.py $OUT
