mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
DXR_RECORD_DIGEST=1 timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -s -k digest 2>&1 | grep SHA256
echo "---- C4 (LDS BLAS tops) vs round-2 library"
python tools/profile_c4.py 6 2>&1 | tail -8 | tee gpurun_out/r03/c4_profile.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libr02.so python tools/profile_c4.py 6 2>&1 | tail -8 | tee gpurun_out/r03/c4_profile_r02.txt
echo "---- C5 batches"
for b in 1 2 4; do
  python bench.py --workload c5 --hbm-frames 8 --batch $b --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['roofline_hbm']
print('c5 batch $b', round(h['ms_per_frame'],2), 'ms', round(h['Mrays_per_s']), 'Mrays/s', {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in h['stages'].items()})"
done
echo "---- default bench line"
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; tail -2 gpurun_out/r03/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","n_gpus","steps")}, d.get("sample_batches"))
r=d["roofline"]; print({k:r[k] for k in r if k not in ("definition","pmc","contract_8d_hbm")})
h=d.get("roofline_hbm",{}); print(h.get("ms_per_frame"), h.get("roofline",{}).get("frac"), h.get("roofline",{}).get("traffic_source"))
print(d.get("cpu_baseline"))
PY
