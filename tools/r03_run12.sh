#!/bin/bash
# early shadow launch beside the secondary launch (RT_OVERLAP_SHADOW0=1): parity, then A/B
mkdir -p gpurun_out/r03
O=gpurun_out/r03/overlap_shadow0.txt
RT_OVERLAP_SHADOW0=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -5 > $O
HBM=6 tools/bench_env.sh "RT_OVERLAP_SHADOW0=0" "RT_OVERLAP_SHADOW0=1" "RT_OVERLAP_SHADOW0=0" "RT_OVERLAP_SHADOW0=1" >> $O 2>&1
for b in 0 1; do RT_OVERLAP_SHADOW0=$b python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('overlap=$b batch8', d.get('sample_batches'))" >> $O; done
cat $O
