#!/bin/bash
# 30,000 more randomised draws at the final head
mkdir -p gpurun_out/r62
for s in 301 302 303; do timeout 1500 python tests/fuzz_parity.py 10000 $s 2>&1 | tail -1; done > gpurun_out/r62/fuzz.txt
