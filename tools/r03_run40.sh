#!/bin/bash
# k_resolve of a set: the pixel's running value in registers over the set's frames (default) vs read and written every frame (libnoreg.so)
mkdir -p gpurun_out/r40
{
for rep in 1 2; do
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_X=reg" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libnoreg.so"
done
} > gpurun_out/r40/resolve_reg.txt 2>&1
for s in 31 32 33 34; do timeout 200 python tests/fuzz_parity.py 1500 $s 2>&1 | tail -1; done > gpurun_out/r40/fuzz.txt
