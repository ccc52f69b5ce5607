#!/bin/bash
# ESTIMATE ONLY (wrong images): what the shadow stage would cost if the point light's (and the directional light's) shadow rays were
# cut to a short interval -- the upper bound of what a conservative per-direction free-distance map around the lights could give
mkdir -p gpurun_out/r41
{
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libshort1.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libshort03.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libshortboth.so"
STEPS=20 WARM=5 BATCH=1 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libshort03.so"
} > gpurun_out/r41/short_estimate.txt 2>&1
