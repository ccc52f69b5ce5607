#!/bin/bash
# ray splitting in the drain: parity (default build = split on, compiler's own occupancy), then A/B
mkdir -p gpurun_out/r03
O=gpurun_out/r03/drain_split.txt
timeout 1200 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_wide_tree.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -15 > $O
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw6.so timeout 1200 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5 >> $O
HBM=6 tools/bench_env.sh "RT_X=split" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw6.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnosplit.so" "RT_X=split" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libw6.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnosplit.so" >> $O 2>&1
cat $O
