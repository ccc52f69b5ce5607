mkdir -p gpurun_out/r03
for pc in 6 5 4 3 2; do
 for b in 1 8; do
  RT_PERSISTENT_BLOCKS_PER_CU=$pc python bench.py --steps 32 --warmup 8 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks/CU $pc batch $b', round(d['ms_per_step'],3), 'ms/frame', {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in d['stages'].items()})"
 done
done | tee gpurun_out/r03/drain_vs_occupancy.txt
