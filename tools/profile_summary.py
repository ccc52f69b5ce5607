#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace stats and/or PMC counter rows) into a short
markdown table for profiles/.   usage: profile_summary.py <dir-with-csvs> [title]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[0-9]+>)?)", name)
    base = m.group(1) if m else name
    if "rocprim" in name:
        k = re.search(r"(radix_sort_[a-z_]+|merge_sort_[a-z_]+|block_merge[a-z_]*|block_sort[a-z_]*)", name)
        base = "rocprim::" + (k.group(1) if k else "kernel")
    return base[:60]


def main():
    d = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else d
    print("# %s\n" % title)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        for r in rows:
            a = agg[short(r["Name"])]
            a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
            a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
        tot = sum(a[1] for a in agg.values())
        print("## kernel trace stats (%s)\n" % os.path.basename(f))
        print("| kernel | calls | total ms | avg us | min us | max us | % |")
        print("|---|---:|---:|---:|---:|---:|---:|")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / tot))
        print()
    # the traversal kernels launch by launch (template arguments kept: <LDS stack rows, two-level, frames-per-launch > 1>), so that a
    # figure quoted for ONE kind of launch -- bench.py's roofline.avg_launch_ms -- can be read off even when the run holds several kinds
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            m = re.search(r"(?<![a-z_])(k_primary|k_trace_secondary|k_trace_shadow)<([^>]*)>", r["Kernel_Name"])
            if m:
                per["%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
        if per:
            print("## traversal launches in issue order (%s): duration of every launch, ms\n" % os.path.basename(f))
            print("| kernel<LDS stack rows, two-level, set of frames> | launches | median ms | each launch, ms |")
            print("|---|---:|---:|---|")
            for k in sorted(per):
                ds = [x[1] / 1e6 for x in sorted(per[k])]
                med = sorted(ds)[len(ds) // 2]
                shown = ds if len(ds) <= 24 else ds[:8] + [float("nan")] + ds[-12:]
                print("| %s | %d | %.3f | %s |" % (k, len(ds), med, " ".join("..." if x != x else "%.3f" % x for x in shown)))
            print()
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in rows:
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
        print("## PMC counters (%s): average per dispatch\n" % os.path.basename(f))
        print("| kernel | counter | dispatches | avg value |")
        print("|---|---|---:|---:|")
        for k in sorted(agg):
            for c, a in sorted(agg[k].items()):
                print("| %s | %s | %d | %.1f |" % (k, c, a[0], a[1] / a[0]))
        print()


if __name__ == "__main__":
    main()
