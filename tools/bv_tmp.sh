for v in r6 r12 r25 r50; do for lm in 4 8; do echo -n "$v leaf$lm: "; RT_VERBOSE=1 RT_LEAF_MAX=$lm DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so timeout 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms build', round(d['bvh_build_ms'],1), {k:(round(v['avg_ms'],3)) for k,v in d['stages'].items() if isinstance(v,dict)})"; grep dxr_amd /tmp/err.txt | head -1; done; done
