#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/shadow_cache_clear_on_miss.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libcom.so timeout 900 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "shadow_cache or atrium" 2>&1 | tail -2 > $O
STEPS=60 WARM=30 BATCH=32 HBM=0 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libcom.so" "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libcom.so" >> $O 2>&1
cat $O
