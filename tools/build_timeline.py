#!/usr/bin/env python3
"""Where a warm acceleration-structure build spends its time: reads the kernel trace rocprofv3 wrote for tools/build_time.py
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/build_time.py) and prints, for the LAST build in the
trace, every kernel's calls / busy time, the span from the first kernel's start to the last one's end, and the idle gaps
between kernels (launch latency and host round trips)."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]))
rows.sort()
# builds are separated by long idle stretches (the host uploads the mesh); take the last burst that contains k_karras
starts = [i for i, r in enumerate(rows) if r[2].endswith("k_tri_boxes")]
last = rows[starts[-1]:]
end = next((i for i, r in enumerate(last) if i > 0 and r[0] - last[i - 1][1] > 2_000_000), len(last))
last = last[:end]
span = last[-1][1] - last[0][0]
busy = sum(e - s for s, e, _ in last)
print("last build: %d kernels, span %.3f ms, kernels busy %.3f ms, idle %.3f ms" % (len(last), span / 1e6, busy / 1e6, (span - busy) / 1e6))
agg = {}
for s, e, k in last:
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-60s %4d calls %8.1f us" % (k[-60:], c, t / 1e3))
gaps = sorted(((last[i][0] - last[i - 1][1], last[i - 1][2], last[i][2]) for i in range(1, len(last))), reverse=True)[:12]
print("largest gaps:")
for g, a, b in gaps:
    print("  %7.1f us between %s and %s" % (g / 1e3, a[-40:], b[-40:]))
