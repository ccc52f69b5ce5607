#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_wide_tree.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03/drain_split_parity.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libtimes.so RT_PERSISTENT_BLOCKS_PER_CU=4 timeout 300 python tools/drain_timeline.py > gpurun_out/r03/drain_timeline_split.txt 2>&1
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libtimes0.so RT_PERSISTENT_BLOCKS_PER_CU=4 timeout 300 python tools/drain_timeline.py > gpurun_out/r03/drain_timeline_nosplit4.txt 2>&1
tail -5 gpurun_out/r03/drain_split_parity.txt; grep -A4 "^launch" gpurun_out/r03/drain_timeline_split.txt gpurun_out/r03/drain_timeline_nosplit4.txt
