#!/bin/bash
# usage (on the GPU box): tools/bench_env.sh "VAR=val ..." "VAR=val ..."   -> one bench line per environment setting
for e in "$@"; do
  env $e timeout 300 python bench.py --steps ${STEPS:-30} --warmup ${WARM:-5} --batch ${BATCH:-1} --cpu-seconds 0 --no-live-pmc --hbm-frames ${HBM:-0} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
f=lambda st:{k:(round(v['avg_ms'],3), round(v['nodes_global_per_ray']+v['nodes_lds_per_ray'],2), round(v['tris_per_ray'],2)) for k,v in st.items() if isinstance(v,dict)}
print('%-28s' % '$e', round(d['ms_per_step'],3), 'ms', f(d['stages']), 'shade/resolve', [round(d['stages'][k],3) for k in ('ms_shade0','ms_shade1','ms_resolve')], 'build', round(d['bvh_rebuild_ms'],2))
h=d.get('roofline_hbm')
if h: print('   c5', round(h['ms_per_frame'],2), 'ms', f(h['stages']), 'shade/resolve', [round(h['stages'][k],3) for k in ('ms_shade0','ms_shade1','ms_resolve')])"
done
