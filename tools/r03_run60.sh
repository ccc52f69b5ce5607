#!/bin/bash
# at the final head (issue priorities in): smoke, and 10,000 more randomised draws
mkdir -p gpurun_out/r60
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200 > gpurun_out/r60/smoke.txt
for s in 201 202; do timeout 900 python tests/fuzz_parity.py 5000 $s 2>&1 | tail -1; done > gpurun_out/r60/fuzz.txt
