#!/bin/bash
# Diagnostic: usage  profiles/micro/sweep.sh "VAR=val VAR2=val2" ... ; each argument is one environment for a short bench run
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/sweep.err | python profiles/micro/sweep_fmt.py \
    || { echo FAILED; tail -5 gpurun_out/sweep.err; break; }
done
