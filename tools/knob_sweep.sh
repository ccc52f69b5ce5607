#!/bin/bash
# round 4: the run-time and compile-time knobs of the traversal kernels re-measured on the new kernels (bench scene in sets, 10 M triangles in sets of 16)
run() { python bench.py --steps 60 --warmup 30 --cpu-seconds 0 --no-live-pmc --hbm-frames ${HBM:-0} --no-frame-by-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d.get('roofline_hbm')
print('$1', round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['stages'].items() if isinstance(v,dict)}, ('| c5 %.3f' % h['ms_per_frame']) if h else '')"; }
for rep in 1 2; do
  unset DXR_AMD_LIB
  run "default"
  for l in 1 3; do RT_DEBUG_OPTIONS=leaf_max=$l run "RT_LEAF_MAX=$l"; done
  for r in 2048 8192; do RT_DEBUG_OPTIONS=shadow_cache_res=$r run "RT_SHADOW_CACHE_RES=$r"; done
  for v in c32 c128 t64; do DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so run "lib=$v"; done
done
