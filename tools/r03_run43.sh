#!/bin/bash
# instruction-cache counters of the headline launches (sets of 30)
mkdir -p gpurun_out/r43
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_IFETCH_LEVEL" | sort -u > $R/gpurun_out/r43/names.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/r43/ic -o p --output-format csv -- python3 $R/bench.py --steps 60 --warmup 30 --cpu-seconds 0 --hbm-frames 0 --no-live-pmc --no-frame-by-frame --no-roofline > $R/gpurun_out/r43/ic.log 2>&1
python3 $R/tools/profile_summary.py $R/gpurun_out/r43/ic "icache" > $R/gpurun_out/r43/icache.md 2>&1
rm -rf $R/gpurun_out/r43/ic
