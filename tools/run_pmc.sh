#!/bin/bash
# usage: [DXR_AMD_LIB=...] [BENCH_ARGS="--steps 40 --warmup 20"] tools/run_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 --pmc pass per argument around a short bench run; one table of the traversal kernels -> gpurun_out/pmc_<tag>.md
TAG=$1; shift
R=$PWD
n=0
mkdir -p gpurun_out
: > gpurun_out/pmc_${TAG}.md
ARGS=${BENCH_ARGS:---steps 40 --warmup 20}
for C in "$@"; do
  n=$((n+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python3 $R/bench.py $ARGS --cpu-seconds 0 --no-roofline --no-frame-by-frame --hbm-frames 0 --no-live-pmc > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 )
  python3 tools/profile_summary.py gpurun_out/pmc_${TAG}_$n "rocprofv3 --pmc $C : python3 bench.py $ARGS (1080p Sponza-class)" | grep -E "^#|^\| (kernel|---|k_trace|k_primary)" >> gpurun_out/pmc_${TAG}.md
  rm -rf gpurun_out/pmc_${TAG}_$n gpurun_out/pmc_${TAG}_$n.log
done
cat gpurun_out/pmc_${TAG}.md
