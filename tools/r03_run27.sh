#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/eight_waves.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw8x.so timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 > $O
STEPS=60 WARM=15 BATCH=16 HBM=0 tools/bench_env.sh "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw7.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw8x.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw7.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw8x.so" >> $O 2>&1
BATCH=1 HBM=0 tools/bench_env.sh "RT_PERSISTENT_BLOCKS_PER_CU=6 DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw7.so" "RT_PERSISTENT_BLOCKS_PER_CU=5 DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw7.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw8x.so" >> $O 2>&1
cat $O
