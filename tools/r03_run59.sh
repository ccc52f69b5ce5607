#!/bin/bash
# the next chunk's index asked for ahead of time (-DRT_POOL_PREFETCH=n: once n rays or fewer are left of the current chunk)
mkdir -p gpurun_out/r59
{
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf32.so python -m pytest tests/test_gpu_trace.py tests/test_gpu_batch.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -2
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf16.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf32.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf63.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf32.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libpf63.so"
} > gpurun_out/r59/pool_prefetch.txt 2>&1
