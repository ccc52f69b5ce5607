#!/bin/bash
# the compiler's other instruction-scheduling strategies for the whole library (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause)
mkdir -p gpurun_out/r53
{
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libilp.so python -m pytest tests/test_gpu_trace.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libilp.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libmemcl.so" "RT_X=default"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libilp.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libmemcl.so"
} > gpurun_out/r53/sched.txt 2>&1
