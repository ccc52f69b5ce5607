#!/bin/bash
# usage (on the GPU box): tools/bench_variants.sh name1 name2 ...  -> one line per variant
for v in "$@"; do
  DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so timeout 300 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-live-pmc --hbm-frames ${HBM:-0} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms', {k:(round(v['avg_ms'],3), round(v['nodes_global_per_ray']+v['nodes_lds_per_ray'],2), round(v['tris_per_ray'],2)) if isinstance(v,dict) else round(v,3) for k,v in d['stages'].items()})
h=d.get('roofline_hbm')
if h: print('   c5', round(h['ms_per_frame'],2), 'ms', {k:(round(v['avg_ms'],3), round(v['nodes_global_per_ray']+v['nodes_lds_per_ray'],2), round(v['tris_per_ray'],2)) if isinstance(v,dict) else round(v,3) for k,v in h['stages'].items()})"
done
