#!/bin/bash
# the VALU issue microbenchmark under rocprofv3: GRBM_GUI_ACTIVE (summed over 8 XCDs) per launch -> cycles per wave64 instruction per SIMD
# usage: tools/r03_run22.sh [workgroups per CU = waves per SIMD, default 8]
W=${1:-8}
mkdir -p gpurun_out/r03
R=$PWD
T=valu_rate_pmc$([ $W = 8 ] || echo _w$W)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $R/gpurun_out/r03/$T -o p --output-format csv -- $R/tools/microbench/valu_rate $W > $R/gpurun_out/r03/$T.log 2>&1
python3 $R/tools/profile_summary.py $R/gpurun_out/r03/$T "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -- tools/microbench/valu_rate $W" > $R/gpurun_out/r03/$T.md
rm -rf $R/gpurun_out/r03/$T
python3 - <<EOF2 | tee -a $R/gpurun_out/r03/$T.md
rows={}
for l in open("$R/gpurun_out/r03/$T.md"):
    c=[x.strip() for x in l.strip().strip("|").split("|")]
    if len(c)==4 and c[0].startswith("k<"):
        rows.setdefault(c[0],{})[c[1]]=float(c[3])
names=[l.split("  ")[0].strip() for l in open("$R/gpurun_out/r03/$T.log") if " ns per wave64" in l]
print("\n## cycles per wave64 instruction per SIMD = GRBM_GUI_ACTIVE / 8 / (SQ_INSTS_VALU / 1024), $W waves per SIMD\n")
for i,n in enumerate(names):
    r=rows.get("k<%d>"%i)
    if r: print("%-22s %.2f"%(n, r["GRBM_GUI_ACTIVE"]/8/(r["SQ_INSTS_VALU"]/1024)))
EOF2
