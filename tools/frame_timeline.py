#!/usr/bin/env python3
"""Where a rendered frame spends its time between kernels: reads the kernel trace rocprofv3 wrote for a bench run
(rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...) and prints, averaged over the last frames
(k_primary ... k_resolve), the span, the time kernels are busy, the idle gaps, and every kernel's share."""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
frames = []
cur = None
for s, e, k in rows:
    if k == "k_primary":
        cur = []
    if cur is not None:
        cur.append((s, e, k))
        if k == "k_resolve":
            frames.append(cur)
            cur = None
frames = frames[-int(sys.argv[2]) if len(sys.argv) > 2 else -8:]
span = sum(f[-1][1] - f[0][0] for f in frames) / len(frames)
busy = sum(sum(e - s for s, e, _ in f) for f in frames) / len(frames)
print("%d frames: span %.1f us, kernels busy %.1f us, idle between kernels %.1f us (%d launches per frame)"
      % (len(frames), span / 1e3, busy / 1e3, (span - busy) / 1e3, len(frames[-1])))
agg = {}
for f in frames:
    for i, (s, e, k) in enumerate(f):
        a = agg.setdefault(k, [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        if i:
            a[2] += s - f[i - 1][1]
for k, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %5.1f calls/frame %8.1f us busy %6.1f us of gaps before it" % (k[-28:], c / len(frames), t / len(frames) / 1e3, g / len(frames) / 1e3))
