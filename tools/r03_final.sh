#!/bin/bash
# the round's final measurements: default bench line (twice: no flags, the driver's flags), the kernel trace of that very command
# (c2h), the counter passes of the headline's launches (c2b: sets of 30 frames), of single frames (c2) and of the 10 M-triangle
# scene frame by frame (c5) and in sets of 8 (c5b)
mkdir -p gpurun_out/r03f
python bench.py > gpurun_out/r03f/bench_n1.json 2> gpurun_out/r03f/bench_n1.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03f/bench_n1_driver_args.json 2>/dev/null
PASSES="kt" tools/run_profiles.sh r03f c2h > gpurun_out/r03f/run_c2h.log 2>&1
tools/run_profiles.sh r03f c2 c2b c5 c5b > gpurun_out/r03f/run_rest.log 2>&1
rm -f gpurun_out/r03f/*_*.log
