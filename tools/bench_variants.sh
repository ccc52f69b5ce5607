#!/bin/bash
# usage (on the GPU box): tools/bench_variants.sh name1 name2 ...  -> one line per variant
for v in "$@"; do
  DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms', {k:(round(v['avg_ms'],3)) if isinstance(v,dict) else round(v,3) for k,v in d['stages'].items()})"
done
