for rep in 1 2; do for w in 5 10 20 40; do
  python bench.py --steps 20 --warmup $w --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --no-frame-by-frame 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('warmup=$w', round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['stages'].items() if isinstance(v,dict)})"
done; done
