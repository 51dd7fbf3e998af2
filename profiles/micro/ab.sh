#!/bin/bash
# Diagnostic: A/B timing of library builds on ONE box (boxes of the pool differ by ~3 %): usage
#   profiles/micro/ab.sh libA.so libB.so [bench args...]   -- alternates A, B, A, B; prints one summary line per run
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in "$A" "$B"; do
    echo "== $L"
    SPAMTREE_LIB=$L timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>gpurun_out/ab.err | python profiles/micro/sweep_fmt.py \
      || { echo FAILED; tail -5 gpurun_out/ab.err; exit 1; }
  done
done
