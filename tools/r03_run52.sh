#!/bin/bash
# is the 3 - 6 % between 5 and 40 warm-up frames the chip (clocks) or the pipeline's state?  DXR_BENCH_PREROLL=n renders n frames on ANOTHER
# pipeline of the same scene before the 5 warm-up frames
mkdir -p gpurun_out/r52
for e in "X=0" "DXR_BENCH_PREROLL=40" "X=0" "DXR_BENCH_PREROLL=80"; do
env $e python bench.py --steps 20 --warmup 5 --hbm-frames 0 --cpu-seconds 0 --no-live-pmc --no-frame-by-frame 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e warmup 5', round(d['value'],1), round(d['ms_per_step'],4))"
done > gpurun_out/r52/preroll.txt 2>&1
