#!/bin/bash
# free-sphere test against the oracle, the bounded fuzz with random lights, a longer sweep of it, and the bench line
mkdir -p gpurun_out/r39
python -m pytest tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r39/tests.txt
timeout 420 python tests/fuzz_parity.py 500 2026 2>&1 | tail -5 > gpurun_out/r39/fuzz.txt
python bench.py --steps 20 --warmup 5 --no-roofline --hbm-frames 0 --cpu-seconds 0 --no-live-pmc > gpurun_out/r39/bench20.json 2>/dev/null
