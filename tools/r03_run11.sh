#!/bin/bash
# drain compaction: parity, then A/B against the same source with -DRT_DRAIN_COMPACT=0
mkdir -p gpurun_out/r03
O=gpurun_out/r03/drain_compact.txt
timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_wide_tree.py -m gpu -x -q 2>&1 | tail -5 > $O
HBM=6 tools/bench_env.sh "RT_X=compact" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnodc.so" "RT_X=compact" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnodc.so" >> $O 2>&1
cat $O
