#!/bin/bash
# hunting the intermittently slow 1 M x 1 M calls: repeated calls with the per-pass device log and the host phases
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${OUT:-r04_slow.txt}
: > $O
for k in $(seq 1 ${PROCS:-5}); do
  echo "== process $k" >> $O
  STAGE=none SEQ=20,20,20,20,20,20 PASSLOG=0 python3 $R/scripts/c5_repro.py >> $O 2>&1 || exit 1
done
echo "calls $(grep -c 'rep ' $O), slowest:"; grep "rep " $O | awk '{print $9}' | sort -n | tail -4 | tr '\n' ' '; echo
