#!/bin/bash
# long randomised parity sweep at the round-3 head (random / moving lights, sets of frames, both pipelines)
mkdir -p gpurun_out/r44
for s in 101 102 103 104; do timeout 900 python tests/fuzz_parity.py 5000 $s 2>&1 | tail -1; done > gpurun_out/r44/fuzz.txt
