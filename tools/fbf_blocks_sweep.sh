mkdir -p gpurun_out/r05
echo "# frame by frame (bench.py --steps 40 --warmup 20 --batch 1) with the persistent launches' workgroups per CU forced: RT_DEBUG_OPTIONS=persistent_blocks_per_cu=N"
for n in 0 6 5 4 3 2; do
  RT_DEBUG_OPTIONS=persistent_blocks_per_cu=$n python bench.py --steps 40 --warmup 20 --batch 1 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --no-roofline --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('blocks per CU $n:', round(d['ms_per_step'],4), 'ms per frame', {k:(round(v['avg_ms'],4) if isinstance(v,dict) else round(v,4)) for k,v in d.get('stages',{}).items()})"
done
