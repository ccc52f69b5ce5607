#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/nt_streams.txt
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 > $O
HBM=6 tools/bench_env.sh "RT_X=nt" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnont.so" "RT_X=nt" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnont.so" >> $O 2>&1
cat $O
