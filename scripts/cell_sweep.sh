#!/bin/bash
for c in 0.12 0.15 0.18 0.2 0.25 0.3 0.4; do
  echo -n "cell $c: "
  python bench.py --steps 40 --warmup 5 --no-cpu --no-brute --cell $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']/1e9,3), 'Gcorr/s', round(d['ms_per_step']*1e3,1), 'us', {k: round(v,1) for k,v in d['kernel_us'].items()})"
done
