#!/bin/bash
# more s_setprio placements: refill 3 + leaf 3; refill 3 + leaf 1 + node loads 2; node loads 3 alone
mkdir -p gpurun_out/r56
{
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libbprio2.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp333.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpload.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libploadonly.so" "RT_X=default"
} > gpurun_out/r56/prio3.txt 2>&1
