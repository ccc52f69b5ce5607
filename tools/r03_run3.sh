mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_batch.py tests/test_gpu_wide_tree.py -m gpu -q -x 2>&1 | tail -12
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
for b in 1 2 4 8; do
  python bench.py --steps 32 --warmup 8 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 --batch $b > gpurun_out/r03/bench_batch$b.json 2>gpurun_out/r03/bench_batch$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03/bench_batch$b.json").read().strip().splitlines()[-1])
print("batch $b", round(d["value"],1), "Mrays/s", round(d["ms_per_step"],3), "ms/frame", {k:(round(v["avg_ms"],3) if isinstance(v,dict) else round(v,3)) for k,v in d["stages"].items()})
PY
done
