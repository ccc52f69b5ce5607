mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
echo "---- eight-wide build: layout + trace + pipeline tests"
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw8.so timeout 900 python -m pytest tests/test_gpu_wide_tree.py tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -m gpu -q -x 2>&1 | tail -5
W8=DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw8.so
HBM=3 tools/bench_env.sh "RT_X=default" "RT_WIDE_SAH=1" "$W8" "$W8 RT_WIDE_SAH=1" 2>&1 | tee gpurun_out/r03/matrix2.txt
python tools/tree_quality.py > gpurun_out/r03/tree_quality_w4.txt 2>&1; RT_WIDE_SAH=1 python tools/tree_quality.py >> gpurun_out/r03/tree_quality_w4.txt 2>&1
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw8.so python tools/tree_quality.py > gpurun_out/r03/tree_quality_w8.txt 2>&1; DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libw8.so RT_WIDE_SAH=1 python tools/tree_quality.py >> gpurun_out/r03/tree_quality_w8.txt 2>&1
cat gpurun_out/r03/tree_quality_w4.txt gpurun_out/r03/tree_quality_w8.txt
