mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5
python tools/profile_c4.py 8 2>&1 | tail -7 | tee gpurun_out/r03/c4_profile_default.txt
for b in 1 8 16; do
  python bench.py --steps 64 --warmup 16 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 batch $b', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/frame', {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in d['stages'].items()})"
done | tee gpurun_out/r03/batch_sweep2.txt
( time python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","n_gpus","steps")}, d.get("sample_batches",{}).get("ms_per_frame"))
r=d["roofline"]; print({k:r[k] for k in ("bound","frac","achieved","peak","traffic_fallback","counters_source")}, r["memory_path"])
h=d.get("roofline_hbm",{}); print(h.get("ms_per_frame"), h.get("roofline",{}).get("frac"))
PY
