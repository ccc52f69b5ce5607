#!/usr/bin/env python3
"""profiles/<round>/pmc_fetch_size.md + pmc_write_size.md -> profiles/<round>/traffic.json
(HBM-side bytes per launch of the traversal kernels, the `roofline.traffic` figure bench.py reports).

usage: tools/traffic_from_pmc.py profiles/r01
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B read requests at 64 B, so it
is doubled; WRITE_SIZE is exact; rocprofv3 reports both in KiB; Infinity-Cache hits are included."""
import json
import re
import subprocess
import sys


def val(path, kernel, counter):
    for line in open(path):
        m = re.match(r"\| %s \| %s \| (\d+) \| ([\d.]+) \|" % (re.escape(kernel), counter), line)
        if m:
            return int(m.group(1)), float(m.group(2))
    raise KeyError((kernel, counter))


def main():
    d = sys.argv[1].rstrip("/")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes around `python3 bench.py --steps 4 --warmup 1 "
                     "--no-roofline` (%s/pmc_fetch_size.md, pmc_write_size.md; tools/run_profiles.sh)" % d,
           "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> doubled; WRITE_SIZE exact; both KiB; "
                         "Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM section)",
           "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip(),
           "kernels": {}}
    for k in ("k_primary", "k_trace_secondary", "k_trace_shadow"):
        nf, f = val(d + "/pmc_fetch_size.md", k, "FETCH_SIZE")
        _, w = val(d + "/pmc_write_size.md", k, "WRITE_SIZE")
        out["kernels"][k] = {"fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w, "dispatches": nf,
                             "bytes_per_launch": int((2.0 * f + w) * 1024)}
    with open(d + "/traffic.json", "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out["kernels"], indent=1))


main()
