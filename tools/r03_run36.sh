#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/primary_cache.txt
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2 > $O
for rep in 1 2; do
STEPS=60 WARM=30 BATCH=32 HBM=0 tools/bench_env.sh "RT_PRIMARY_CACHE=0" "RT_PRIMARY_CACHE=1" >> $O 2>&1
done
BATCH=1 HBM=0 tools/bench_env.sh "RT_PRIMARY_CACHE=0" "RT_PRIMARY_CACHE=1" >> $O 2>&1
cat $O
