#!/bin/bash
# the round's final measurements: default bench line, the profile passes of every workload, traffic.json inputs
mkdir -p gpurun_out/r03f
python bench.py > gpurun_out/r03f/bench_n1.json 2> gpurun_out/r03f/bench_n1.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03f/bench_n1_driver_args.json 2>/dev/null
tools/run_profiles.sh r03f c2h > gpurun_out/r03f/run_c2h.log 2>&1
PASSES="kt" tools/run_profiles.sh r03f c2h >> gpurun_out/r03f/run_c2h.log 2>&1
tools/run_profiles.sh r03f c2 c2b > gpurun_out/r03f/run_c2.log 2>&1
