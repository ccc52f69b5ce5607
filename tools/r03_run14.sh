#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/drain_timeline.txt
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libtimes.so timeout 300 python tools/drain_timeline.py > $O 2>&1
cat $O
DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libtimesc.so timeout 300 python tools/drain_timeline.py > gpurun_out/r03/drain_timeline_compact.txt 2>&1
