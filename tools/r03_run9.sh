mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_wrapper_and_io.py -m gpu -q 2>&1 | tail -4
timeout 1200 python tests/fuzz_parity.py 1500 20261003 2>&1 | tail -2
( time python bench.py --steps 256 --warmup 16 --batch 16 --cpu-seconds 0 --no-live-pmc --hbm-frames 0 ) 2>&1 | python -c "
import sys,json
t=sys.stdin.read()
line=[l for l in t.splitlines() if l.startswith('{')][-1]
d=json.loads(line); print('config 3 on one GPU: 256 frames 1080p, 16 per launch set:', round(d['ms_per_step']*256,1), 'ms =', round(d['ms_per_step'],3), 'ms/frame,', round(d['value']), 'Mrays/s')
print([l for l in t.splitlines() if 'real' in l])" | tee gpurun_out/r03/config3_one_gpu.txt
