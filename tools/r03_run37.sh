#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/shadow_cache_two_level.txt
timeout 2000 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -2 > $O
for c in 0 2048 4096 8192; do echo "== RT_SHADOW_CACHE_RES=$c" >> $O; RT_SHADOW_CACHE_RES=$c timeout 600 python tools/profile_c4.py 8 2>&1 | head -3 >> $O; done
echo "== automatic" >> $O; timeout 600 python tools/profile_c4.py 8 2>&1 | head -3 >> $O
cat $O
