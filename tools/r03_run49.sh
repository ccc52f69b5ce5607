#!/bin/bash
# does the 20-step line depend on how warm the chip is when the timed region starts?
mkdir -p gpurun_out/r49
for w in 5 40 5 80; do
python bench.py --steps 20 --warmup $w --hbm-frames 0 --cpu-seconds 0 --no-live-pmc --no-frame-by-frame 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warmup $w', round(d['value'],1), round(d['ms_per_step'],4))"
done > gpurun_out/r49/warm.txt 2>&1
