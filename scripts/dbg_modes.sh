#!/bin/bash
# timing experiments: lanes per query and which stage of ring 1 costs what
for sg in 8 4 16; do
for m in 0 1 3 4 16; do
  echo "== PCR_SG=$sg PCR_DEBUG_MODE=$m"
  PCR_SG=$sg PCR_DEBUG_MODE=$m python bench.py --steps 30 --warmup 3 --no-cpu --no-brute 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print({k: round(v,1) for k,v in d['kernel_us'].items()}, round(d['ms_per_step']*1e3,1))"
done
done
