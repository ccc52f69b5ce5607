#!/bin/bash
# s_setprio around the leaf phase (2), around refill (3) and leaf phase (2 / 1) together
mkdir -p gpurun_out/r55
{
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libprio3.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/liblprio.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libbprio.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libbprio2.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libbprio.so"
} > gpurun_out/r55/prio2.txt 2>&1
