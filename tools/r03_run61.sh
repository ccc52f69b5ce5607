#!/bin/bash
# single frames on the seven-wave (14-row) kernels, now that the shadow cache and the issue priorities are in (RT_SEVEN_WAVES_ALWAYS=1)
mkdir -p gpurun_out/r61
{
for rep in 1 2; do
STEPS=30 WARM=10 BATCH=1 HBM=4 tools/bench_env.sh "RT_SEVEN_WAVES_ALWAYS=0" "RT_SEVEN_WAVES_ALWAYS=1"
done
} > gpurun_out/r61/seven_fbf.txt 2>&1
