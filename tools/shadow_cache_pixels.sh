#!/bin/bash
# per-pixel shadow-cache entries for the primary hits' rays on / off: the long form (30 warm-up frames) and the driver's form
for rep in 1 2; do for v in 0 1; do
  for form in "--steps 60 --warmup 30" "--steps 20 --warmup 5"; do
  RT_DEBUG_OPTIONS=shadow_cache_pixels=$v python bench.py $form --cpu-seconds 0 --no-live-pmc --hbm-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); fb=d['frame_by_frame']
print('pixels=$v', '$form', round(d['ms_per_step'],4), 'ms in sets |', round(fb['ms_per_frame'],4), 'frame by frame', {k:round(v['avg_ms'],4) for k,v in d['stages'].items() if isinstance(v,dict)}, {k:round(v,3) for k,v in fb['stage_ms'].items()})"
  done
done; done
for v in 0 1; do RT_DEBUG_OPTIONS=shadow_cache_pixels=$v python tools/c5_batches.py 16 2>/dev/null | tail -1 | cut -c1-60; RT_DEBUG_OPTIONS=shadow_cache_pixels=$v python tools/profile_c4.py 8 2>/dev/null | head -2 | tr '\n' ' '; echo; done
