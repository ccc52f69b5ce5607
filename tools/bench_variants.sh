#!/bin/bash
# usage (on the GPU box): [REPS=2] [HBM=0] tools/bench_variants.sh name1 name2 ...  -> one line per variant and repetition
# (variants are libraries built by tools/build_variant.sh; "default" = the in-tree library).  Runs are interleaved so that the
# box's drift hits every variant alike.
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  if [ "$v" = default ]; then unset DXR_AMD_LIB; else export DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so; fi
  timeout 300 python bench.py --steps ${STEPS:-60} --warmup ${WARMUP:-30} --cpu-seconds 0 --no-live-pmc --hbm-frames ${HBM:-0} ${EXTRA} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
fb=d.get('frame_by_frame',{})
print('$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],4), 'ms in sets |', round(fb.get('ms_per_frame',0),4), 'ms frame by frame', {k:round(v,3) for k,v in fb.get('stage_ms',{}).items()}, '| sets:', {k:(round(v['avg_ms'],4)) if isinstance(v,dict) else round(v,4) for k,v in d['stages'].items()})
h=d.get('roofline_hbm')
if h: print('   c5', round(h['ms_per_frame'],2), 'ms', {k:(round(v['avg_ms'],3)) if isinstance(v,dict) else round(v,3) for k,v in h['stages'].items()})"
done
done
