for rep in 1 2; do
for v in 0 1; do
  RT_DEBUG_OPTIONS=seven_waves_always=$v python bench.py --steps 60 --warmup 30 --batch 1 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('seven_waves_always=$v', round(d['ms_per_step'],4), 'ms frame by frame', round(d['value'],1))"
done; done
