#!/bin/bash
# the 10 M-triangle 4K workload in sets of 16 (default) / 24 / 32 frames
mkdir -p gpurun_out/r63
for b in 16 24 32; do
timeout 600 python bench.py --workload c5 --hbm-frames $((2*b)) --batch $b --no-live-pmc --no-roofline 2>gpurun_out/r63/err_$b.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print('sets of $b:', round(h['ms_per_frame'],3), 'ms/frame', round(h['Mrays_per_s']), 'Mrays/s', {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in h['stages'].items()})"
done > gpurun_out/r63/c5_sets.txt 2>&1
