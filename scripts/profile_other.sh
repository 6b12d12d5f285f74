#!/bin/bash
# rocprofv3 kernel stats + HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes)
# for the kernels outside the ICP pass.  Output: gpurun_out/other/<op>/..., folded into gpurun_out/other/summary.json
# (copy to profiles/r03_other_configs.json).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/other
rm -rf $OUT && mkdir -p $OUT
for op in voxel120k voxel1m iss1m knn120k radius20k normals120k; do
  mkdir -p $OUT/$op
  python3 $R/scripts/other_kernels.py $op 5 > $OUT/$op/wall.json 2> $OUT/$op/wall.err || { echo "$op failed"; tail -3 $OUT/$op/wall.err; continue; }
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/$op/stats -o st -- python3 $R/scripts/other_kernels.py $op 5 > $OUT/$op/stats.log 2>&1
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/$op/fetch -o p -- python3 $R/scripts/other_kernels.py $op 5 > $OUT/$op/fetch.log 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/$op/write -o p -- python3 $R/scripts/other_kernels.py $op 5 > $OUT/$op/write.log 2>&1
  echo "$op done"
done
python3 $R/scripts/fold_other.py $OUT > $OUT/summary.json
cat $OUT/summary.json | head -c 6000
