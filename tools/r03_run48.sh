#!/bin/bash
# refill threshold of the any-hit (shadow) walks alone: 16 (default) vs 24 / 28 / 32 idle lanes; sets of 30, frame by frame, 10 M triangles, 4096 instances
mkdir -p gpurun_out/r48
{
for rep in 1 2; do
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=base16" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany24.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany28.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany32.so"
done
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=base16" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany24.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany28.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany32.so"
python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libany28.so python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
} > gpurun_out/r48/refill_anyhit.txt 2>&1
