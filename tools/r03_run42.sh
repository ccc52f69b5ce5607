#!/bin/bash
# staggered retirement of the waves at the end of a persistent launch (-DRT_RETIRE_PERMILLE=b: two thirds of the workgroups stop taking
# chunks while up to b/1000 chunks per wave of their pool group are left, each at a level of its own): parity, then frame by frame and in sets
mkdir -p gpurun_out/r42
DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libret2000.so python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r42/tests.txt
{
for rep in 1 2; do
STEPS=30 WARM=5 BATCH=1 HBM=4 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libret1000.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libret2000.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libret4000.so"
done
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_X=base" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libret2000.so"
} > gpurun_out/r42/retire.txt 2>&1
