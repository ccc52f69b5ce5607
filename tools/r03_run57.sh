#!/bin/bash
# s_setprio: refill / leaf phase / node loads = 3/3/0, 3/1/2, 3/3/2, 3/3/1, 2/2/1; sets of 30 + 10 M triangles, then frame by frame and the 4096-instance frame for the best
mkdir -p gpurun_out/r57
{
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp333.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpload.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp332.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp331.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp222.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 HBM=4 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp332.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpload.so"
python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libp332.so python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpload.so python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
} > gpurun_out/r57/prio4.txt 2>&1
