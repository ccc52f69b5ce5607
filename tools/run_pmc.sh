#!/bin/bash
# usage: tools/run_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
# One rocprofv3 --pmc pass per argument around a short bench run; summaries -> gpurun_out/pmc_<tag>_<n>.md
TAG=$1; shift
R=$PWD
n=0
for C in "$@"; do
  n=$((n+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_${TAG}_$n -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 )
  python3 tools/profile_summary.py gpurun_out/pmc_${TAG}_$n "rocprofv3 --pmc $C : python3 bench.py --steps 3 (1080p Sponza-class)" | grep -E "^#|^\| (kernel|---|k_trace|k_primary|k_shade|k_resolve)" > gpurun_out/pmc_${TAG}_$n.md
  rm -rf gpurun_out/pmc_${TAG}_$n
done
cat gpurun_out/pmc_${TAG}_*.md
