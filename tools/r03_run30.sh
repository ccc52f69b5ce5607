#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/refill_sweep_sets.txt
: > $O
for rep in 1 2; do
STEPS=60 WARM=15 BATCH=16 HBM=0 tools/bench_env.sh "RT_X=base(16,K=1)" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librl8.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librl24.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/librl32.so" "DXR_AMD_LIB=dxrexperiments_amd/lib/variants/libk2.so" >> $O 2>&1
done
cat $O
