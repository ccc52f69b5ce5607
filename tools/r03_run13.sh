#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/drain_compact_stats.txt
: > $O
for v in stats0 stats1 stats2; do echo "== $v" >> $O; DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/lib$v.so timeout 300 python tools/trace_stats.py >> $O 2>&1; done
cat $O
