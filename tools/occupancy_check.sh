#!/bin/bash
# blocks per CU of the persistent traversal launches (= waves per SIMD) forced down from the 7 the sets' kernels run with
for rep in 1 2; do for b in 0 6 5 4; do
  RT_DEBUG_OPTIONS=persistent_blocks_per_cu=$b python bench.py --steps 60 --warmup 30 --cpu-seconds 0 --no-live-pmc --hbm-frames 16 --no-frame-by-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['roofline_hbm']
print('blocks_per_cu=$b (0: occupancy API)', round(d['ms_per_step'],4), 'ms/frame in sets', {k:round(v['avg_ms'],4) for k,v in d['stages'].items() if isinstance(v,dict)}, '| c5', round(h['ms_per_frame'],3), {k:round(v['avg_ms'],3) for k,v in h['stages'].items() if isinstance(v,dict)})"
done; done
