#!/bin/bash
# refill threshold with the shadow cache on (a refill now holds the ray load AND the dependent entry gather): is a refill a stall worth avoiding?
mkdir -p gpurun_out/r47
{
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_X=base16" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librefill4.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librefill8.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librefill28.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librefill40.so" "RT_X=base16"
} > gpurun_out/r47/refill.txt 2>&1
