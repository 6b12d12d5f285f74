#!/bin/bash
# round-4 profile set beyond profile_icp.sh / profile_other.sh: rocprofv3 kernel stats of the global initialisation (120k pair +
# 64-pair batch with initialisation) and of the 1 M x 1 M registration leg (three 20-iteration calls with their per-pass logs)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_profiles
rm -rf $O && mkdir -p $O
python3 $R/scripts/global_time.py > $O/global_time.txt 2>&1
STAGES=0 PAIRS=64 rocprofv3 --output-format csv --kernel-trace --stats -d $O/global -o g -- python3 $R/scripts/global_time.py > $O/global_prof.log 2>&1
STAGE=none SEQ=20,20,20 python3 $R/scripts/c5_repro.py > $O/icp1m_calls.txt 2>&1
STAGE=none SEQ=20,20,20 PASSLOG=0 rocprofv3 --output-format csv --kernel-trace --stats -d $O/icp1m -o c5 -- python3 $R/scripts/c5_repro.py > $O/icp1m_prof.log 2>&1
cp $O/global/*/g_kernel_stats.csv $O/r04_global_kernel_stats.csv 2>/dev/null || cp $O/global/g_kernel_stats.csv $O/r04_global_kernel_stats.csv
cp $O/icp1m/*/c5_kernel_stats.csv $O/r04_icp1m_kernel_stats.csv 2>/dev/null || cp $O/icp1m/c5_kernel_stats.csv $O/r04_icp1m_kernel_stats.csv
grep -v amdgpu $O/global_time.txt > $O/r04_global_time.txt
grep -v amdgpu $O/icp1m_calls.txt | sed 's/np.float64//g' > $O/r04_icp1m_calls.txt
head -12 $O/r04_global_kernel_stats.csv | cut -c1-160; head -6 $O/r04_icp1m_kernel_stats.csv | cut -c1-160; cat $O/r04_global_time.txt
