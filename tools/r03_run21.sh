#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/pk_fma.txt
timeout 1200 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_wide_tree.py -m gpu -x -q 2>&1 | tail -3 > $O
HBM=6 tools/bench_env.sh "RT_X=pkfma" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnosplit.so" "RT_X=pkfma" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnosplit.so" >> $O 2>&1
cat $O
