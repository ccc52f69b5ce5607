#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/sort_nearest_only.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libnear1.so timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -3 > $O
HBM=6 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnear1.so" "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnear1.so" >> $O 2>&1
cat $O
