#!/bin/bash
# final records of the round: fused-initialisation kernel stats, the GPU suite, the three bench lines
R=$GRAFT_REPO_ROOT
cd $R
PAIRS=48 bash scripts/init_fused_prof.sh > gpurun_out/r04_init_fused_kernel_stats.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_tests.log 2>&1; tail -3 gpurun_out/r04_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_args.json 2> gpurun_out/r04_bench_driver_args.err
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python bench.py --gpus 2 --dist-backend gloo --device 0 --steps 20 --warmup 5 --no-configs > gpurun_out/r04_bench_2ranks.json 2> gpurun_out/r04_bench_2ranks.err
ls -la gpurun_out/r04_bench_*.json
