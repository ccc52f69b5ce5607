#!/usr/bin/env python3
"""Compiler view of every kernel in one HIP source: VGPRs, SGPRs, LDS, scratch, occupancy.
usage: tools/kernel_resources.py dxrexperiments_amd/csrc/rt_pipeline.hip"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
       "-Iinclude", *sys.argv[2:], "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark: +(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
print("%-34s %6s %6s %8s %8s %5s" % ("kernel", "VGPR", "SGPR", "LDS", "scratch", "occ"))
for k, r in rows.items():
    print("%-34s %6d %6d %8d %8d %5d" % (k[:34], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("LDS", -1),
                                        r.get("ScratchSize", -1), r.get("Occupancy", -1)))
