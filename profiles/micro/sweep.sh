#!/bin/bash
# usage: gpurun_sweep.sh "VAR=val VAR2=val2" ... ; each argument is one environment for a short bench run
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(round(d['value'],1), round(r['frac'],3), r['by_level_ms'][5:], r['phase_ms_per_iter']['factor'], r['phase_ms_per_iter']['sample'])
" || { echo FAILED; tail -5 gpurun_out/sweep.err; break; }
done
