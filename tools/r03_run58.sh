#!/bin/bash
# with the issue priorities in place: straggler exit K = 2, refill threshold 8 / 24 (defaults 1, 16)
mkdir -p gpurun_out/r58
{
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libk2.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librf8.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librf24.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libk2.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librf8.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librf24.so"
} > gpurun_out/r58/knobs.txt 2>&1
