#!/bin/bash
mkdir -p gpurun_out/r03
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03/bench_default20.json 2> gpurun_out/r03/bench_default20.err
( time python bench.py ) > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
python bench.py --batch 1 --no-live-pmc --cpu-seconds 0 --hbm-frames 0 > gpurun_out/r03/bench_batch1.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "bench_two_ranks" 2>&1 | tail -3
python - <<EOF2
import json
for f in ("bench_default20","bench_default","bench_batch1"):
    d=json.loads(open("gpurun_out/r03/%s.json"%f).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(f, round(d["value"]), round(d["ms_per_step"],3), d["config"]["frames_per_launch_set"], d["config"]["launch_sets"], {k:(round(r[k],3) if isinstance(r.get(k),float) else r.get(k)) for k in ("achieved","peak","frac","frac_at_2_cycles_per_instruction","avg_launch_ms","frames_per_launch","counters_source","traffic_fallback")}, d.get("frame_by_frame"), (d.get("cpu_baseline") or {}).get("value"))
EOF2
tail -4 gpurun_out/r03/bench_default20.err gpurun_out/r03/bench_default.err
