#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/primary_persistent.txt
RT_PRIMARY_PERSISTENT=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -3 > $O
STEPS=60 WARM=15 BATCH=16 HBM=0 tools/bench_env.sh "RT_PRIMARY_PERSISTENT=0" "RT_PRIMARY_PERSISTENT=1" "RT_PRIMARY_PERSISTENT=0" "RT_PRIMARY_PERSISTENT=1" >> $O 2>&1
BATCH=1 HBM=0 tools/bench_env.sh "RT_PRIMARY_PERSISTENT=0" "RT_PRIMARY_PERSISTENT=1" >> $O 2>&1
cat $O
