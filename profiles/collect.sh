#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash profiles/collect.sh TAG'): bench line, rocprofv3 kernel-trace statistics and the
# two HBM counter passes (separate --pmc runs, kernel-trace only) of the same bench command; raw output under
# gpurun_out/ (merged back by gpurun); profiles/summarize.py, run afterwards in the repo, writes profiles/<round>/<TAG>_* (ROUND env, default r03).
TAG=${1:-c_quad}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py --steps 30 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 || exit 1
done
# then, back in the repo (gpurun merges gpurun_out/ back): python profiles/summarize.py $TAG
# SQ counters of the same command (own pass, kernel-trace only): MFMA-busy and wait shares per kernel
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_SQ -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_SQ.log 2>&1 || exit 1
