mkdir -p gpurun_out/r03
for e in "RT_X=lds_blas_tops" "RT_LDS_BLAS=0"; do echo "== $e"; env $e python tools/profile_c4.py 8 2>&1 | tail -7; done | tee gpurun_out/r03/c4_profile.txt
for b in 1 4 8; do
  python bench.py --workload c5 --hbm-frames 16 --batch $b --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline_hbm']
print('c5 batch $b', round(h['ms_per_frame'],2), 'ms', round(h['Mrays_per_s']), 'Mrays/s', {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in h['stages'].items()})"
done | tee gpurun_out/r03/c5_batches.txt
python bench.py --steps 64 --warmup 16 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --batch 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 batch 16', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a gpurun_out/r03/c5_batches.txt
