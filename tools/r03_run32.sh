#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/shadow_cache.txt
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3 > $O
for rep in 1 2; do
STEPS=60 WARM=30 BATCH=32 HBM=0 tools/bench_env.sh "RT_SHADOW_CACHE_RES=0" "RT_SHADOW_CACHE_RES=1024" "RT_SHADOW_CACHE_RES=2048" "RT_SHADOW_CACHE_RES=4096" >> $O 2>&1
done
BATCH=1 HBM=0 tools/bench_env.sh "RT_SHADOW_CACHE_RES=0" "RT_SHADOW_CACHE_RES=2048" >> $O 2>&1
cat $O
