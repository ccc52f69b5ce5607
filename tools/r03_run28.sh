#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/seven_waves_sets.txt
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -3 > $O
STEPS=60 WARM=15 BATCH=16 HBM=0 tools/bench_env.sh "RT_X=now" "RT_X=now" >> $O 2>&1
BATCH=1 HBM=0 tools/bench_env.sh "RT_X=now" >> $O 2>&1
python bench.py --workload c5 --hbm-frames 8 --batch 8 --no-live-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('c5 sets of 8:', round(h['ms_per_frame'],2), {k:round(v['avg_ms'],3) for k,v in h['stages'].items() if isinstance(v,dict)})" >> $O
cat $O
