#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/set_chunk_frames.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libcf4.so timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -2 > $O
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libcf16.so timeout 1500 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2 >> $O
for rep in 1 2; do
STEPS=64 WARM=32 BATCH=32 HBM=0 tools/bench_env.sh "RT_X=F1" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libcf4.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libcf8.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libcf16.so" >> $O 2>&1
done
cat $O
