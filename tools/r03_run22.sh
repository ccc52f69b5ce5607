#!/bin/bash
# the VALU issue microbenchmark under rocprofv3: GRBM_GUI_ACTIVE (summed over 8 XCDs) per launch -> cycles per wave64 instruction per SIMD
mkdir -p gpurun_out/r03
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/r03/valu_rate_pmc -o p --output-format csv -- $R/tools/microbench/valu_rate > $R/gpurun_out/r03/valu_rate_pmc.log 2>&1
python3 $R/tools/profile_summary.py $R/gpurun_out/r03/valu_rate_pmc "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -- tools/microbench/valu_rate" > $R/gpurun_out/r03/valu_rate_pmc.md
rm -rf $R/gpurun_out/r03/valu_rate_pmc
cat $R/gpurun_out/r03/valu_rate_pmc.log | tail -30
grep -E "GRBM_GUI_ACTIVE|SQ_INSTS_VALU " $R/gpurun_out/r03/valu_rate_pmc.md | head -60
