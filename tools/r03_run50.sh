#!/bin/bash
# the same with the shadow cache off: is it the cache that is still filling after 5 frames, or the chip's clocks?
mkdir -p gpurun_out/r50
for w in 5 40 5 40; do
RT_SHADOW_CACHE_RES=0 python bench.py --steps 20 --warmup $w --hbm-frames 0 --cpu-seconds 0 --no-live-pmc --no-frame-by-frame 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cache off, warmup $w', round(d['value'],1), round(d['ms_per_step'],4), d['roofline'].get('clock_GHz_under_load'))"
done > gpurun_out/r50/warm_nocache.txt 2>&1
