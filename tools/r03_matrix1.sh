mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_wide_tree.py tests/test_gpu_trace.py tests/test_gpu_bvh.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -8
W4=DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw4.so
HBM=3 tools/bench_env.sh "RT_X=w8sah" "RT_WIDE_SAH=0" "RT_SAH_PRIM=0.3" "RT_SAH_PRIM=1.0" "RT_LEAF_MAX=3" "RT_LEAF_MAX=4" "RT_LEAF_MAX=1" \
  "$W4" "$W4 RT_WIDE_SAH=0" "$W4 RT_SAH_PRIM=0.3" "$W4 RT_SAH_PRIM=1.0" "$W4 RT_LEAF_MAX=3" "$W4 RT_LEAF_MAX=4" 2>&1 | tee gpurun_out/r03/matrix1.txt
