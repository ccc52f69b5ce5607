#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/prefetch_leaf.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libpfl.so timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 > $O
STEPS=60 WARM=30 BATCH=32 HBM=0 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpfl.so" "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpfl.so" >> $O 2>&1
BATCH=1 HBM=6 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpfl.so" >> $O 2>&1
cat $O
