#!/bin/bash
# shadow cache with a coarse level behind the tables (cells 2^s times as large, tested where a ray's own cell has never been written):
# parity suites, then the driver's 20 steps after 5 warm-up frames, sets of 30 after 30, frame by frame, 10 M triangles, 4096 instances
mkdir -p gpurun_out/r51
python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r51/tests.txt
{
for rep in 1 2; do
STEPS=20 WARM=5 BATCH=32 tools/bench_env.sh "RT_SHADOW_CACHE_COARSE_SHIFT=0" "RT_SHADOW_CACHE_COARSE_SHIFT=1" "RT_SHADOW_CACHE_COARSE_SHIFT=2" "RT_SHADOW_CACHE_COARSE_SHIFT=3"
done
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_SHADOW_CACHE_COARSE_SHIFT=0" "RT_SHADOW_CACHE_COARSE_SHIFT=1" "RT_SHADOW_CACHE_COARSE_SHIFT=2" "RT_SHADOW_CACHE_COARSE_SHIFT=3"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_SHADOW_CACHE_COARSE_SHIFT=0" "RT_SHADOW_CACHE_COARSE_SHIFT=2"
RT_SHADOW_CACHE_COARSE_SHIFT=0 python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
RT_SHADOW_CACHE_COARSE_SHIFT=2 python tools/profile_c4.py 8 2>&1 | grep "C4:\|stage ms" | head -2
} > gpurun_out/r51/coarse.txt 2>&1
