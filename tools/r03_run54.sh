#!/bin/bash
# s_setprio around the refill (a wave that refills issues ahead of the others): -DRT_REFILL_PRIO=1 / 3
mkdir -p gpurun_out/r54
{
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libprio1.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libprio3.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libprio3.so"
} > gpurun_out/r54/prio.txt 2>&1
