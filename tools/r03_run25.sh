#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/sort_nearest_only_batch.txt
BATCH=16 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnear1.so" "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnear1.so" > $O 2>&1
cat $O
