#!/bin/bash
# A/B of the 1 M x 1 M registration (bench config5 leg) and of the 120k pass for the variant libraries in scripts/bin
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_1m_ab.txt
: > $O
ROUNDS=${ROUNDS:-12}
for v in in-tree $(ls $R/scripts/bin/libpcr_*.so 2>/dev/null); do
  if [ $v = in-tree ]; then unset PCR_LIB_PATH; else export PCR_LIB_PATH=$v; fi
  for rounds in $ROUNDS; do
    echo "== $v PCR_WT_ROUNDS=$rounds" >> $O
    PCR_WT_ROUNDS=$rounds STAGE=none SEQ=20,20 python3 $R/scripts/c5_repro.py >> $O 2>&1 || exit 1
  done
  python3 $R/scripts/ab_pass.py >> $O 2>&1 || exit 1
  ITERS=1 REPS=9 python3 $R/scripts/ab_pass.py >> $O 2>&1 || exit 1
done
grep -v "amdgpu.ids" $O
