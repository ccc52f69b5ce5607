mkdir -p gpurun_out/r03
PF=$PWD/dxrexperiments_amd/lib/variants/libpf.so
DXR_AMD_LIB=$PF timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -m gpu -q -x 2>&1 | tail -3
HBM=6 tools/bench_env.sh "RT_X=default" "DXR_AMD_LIB=$PF" "RT_X=default" "DXR_AMD_LIB=$PF" 2>&1 | tee gpurun_out/r03/prefetch_pop.txt
