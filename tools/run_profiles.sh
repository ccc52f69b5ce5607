#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (they do not fit one pass on gfx950), each around the same bench command.
# usage: tools/run_profiles.sh <tag>     -> gpurun_out/prof_<tag>_{kt,fetch,write}/
TAG=${1:-r01}
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_${TAG}_fetch -o pmc --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-roofline > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_${TAG}_write -o pmc --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-roofline > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cd $R
for k in kt fetch write; do python3 tools/profile_summary.py gpurun_out/prof_${TAG}_$k "rocprofv3 $k pass: python3 bench.py (1080p Sponza-class)" > gpurun_out/prof_${TAG}_$k.md; rm -f gpurun_out/prof_${TAG}_$k/*/*kernel_trace.csv gpurun_out/prof_${TAG}_$k/*kernel_trace.csv; done
for f in gpurun_out/prof_${TAG}_*.log; do tail -n 1 $f; done
