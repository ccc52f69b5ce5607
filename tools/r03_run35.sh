#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/shadow_cache_nearest_the_light.txt
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2 > $O
for rep in 1 2; do
STEPS=60 WARM=30 BATCH=32 HBM=0 tools/bench_env.sh "RT_SHADOW_CACHE_RES=0" "RT_SHADOW_CACHE_RES=1024" "RT_SHADOW_CACHE_RES=2048" "RT_SHADOW_CACHE_RES=4096" "RT_SHADOW_CACHE_RES=8192" >> $O 2>&1
done
BATCH=1 HBM=0 tools/bench_env.sh "RT_SHADOW_CACHE_RES=0" "RT_SHADOW_CACHE_RES=4096" >> $O 2>&1
for c in 2048 4096 8192; do RT_SHADOW_CACHE_RES=$c python bench.py --workload c5 --hbm-frames 16 --batch 16 --no-live-pmc --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('c5 sets of 16, cache $c:', round(h['ms_per_frame'],2), {k:round(v['avg_ms'],3) for k,v in h['stages'].items() if isinstance(v,dict)})" >> $O; done
cat $O
