#!/usr/bin/env python3
"""profiles/<round>/{c2,c5}_{kt,fetch,write,tcc,sq}.md -> profiles/<round>/traffic.json: per workload and traversal kernel
the memory-side bytes per launch (`roofline.traffic` in bench.py), the rocprofv3 average duration and the SQ / TCC counters.

usage: tools/traffic_from_pmc.py profiles/r03
gfx950 (MI355X_MICROARCH.md, HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size, exact only for 64-B
requests and half the bytes of wide streaming reads; other access shapes are "uncalibrated", so the read bytes are taken from
the requests BY SIZE (TCC_EA0_RDREQ_32B / _64B / _128B, a pass of their own): bytes_per_launch = 32 n32 + 64 n64 + 128 n128 +
WRITE_SIZE.  The guide's FETCH_SIZE x 2 + WRITE_SIZE is kept as bytes_per_launch_fetch_x2.  Infinity-Cache hits are included
in both (these are the L2's memory-side requests)."""
import json
import os
import re
import subprocess
import sys

KERNELS = ("k_primary", "k_trace_secondary", "k_trace_shadow", "k_denoise_h", "k_denoise_v", "k_shade_emit", "k_resolve")


def rows(path):
    """{(kernel, column-or-counter): value} of a profile_summary.py table file."""
    out = {}
    if not os.path.isfile(path):
        return out
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 4 and re.match(r"^[\d.]+$", c[3] or "x"):             # | kernel | counter | dispatches | avg |
            out[(c[0], c[1])] = float(c[3])
            out[(c[0], c[1] + ":n")] = float(c[2])
        elif len(c) == 7 and re.match(r"^[\d.]+$", c[3] or "x"):           # | kernel | calls | total ms | avg us | ...
            out[(c[0], "avg_us")] = float(c[3])
            out[(c[0], "calls")] = float(c[1])
    return out


def main():
    d = sys.argv[1].rstrip("/")
    out = {"source": "rocprofv3 passes of tools/run_profiles.sh around `python3 bench.py` (c2: --steps 8; c5: --workload c5 --hbm-frames 4); "
                     "one .md per pass in %s" % d,
           "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> doubled; WRITE_SIZE exact; both KiB; "
                         "Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM section)",
           "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip(),
           "workloads": {}}
    for w in ("c2", "c5", "c4", "c2b", "c5b", "c2s"):          # (c2b / c5b: the same workloads at 30 / 16 frames per set of launches)
        kt, fe, wr = rows("%s/%s_kt.md" % (d, w)), rows("%s/%s_fetch.md" % (d, w)), rows("%s/%s_write.md" % (d, w))
        tcc, sq, ea = rows("%s/%s_tcc.md" % (d, w)), rows("%s/%s_sq.md" % (d, w)), rows("%s/%s_ea.md" % (d, w))
        tcp, ta2 = rows("%s/%s_tcp.md" % (d, w)), rows("%s/%s_ta2.md" % (d, w))
        # (round 4) thread cycles over 64 x instruction cycles = the active-lane fraction; its two counters come from ONE pass
        lanes = {(k, c): v for (k, c), v in rows("%s/%s_lanes.md" % (d, w)).items() if c.split(":")[0] in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")}
        lanes = {(k, "LANES_" + c if c.startswith("SQ_ACTIVE_INST_VALU") else c): v for (k, c), v in lanes.items()}
        mix = rows("%s/%s_mix.md" % (d, w))              # vector instructions by class (their SQ_INSTS_VALU kept apart: shares within one pass)
        mix = {(k, "SQ_INSTS_VALU_MIX" if c == "SQ_INSTS_VALU" else c): v for (k, c), v in mix.items() if c != "GRBM_GUI_ACTIVE" and not c.startswith("GRBM_GUI_ACTIVE")}
        if not fe and not ea:
            continue
        ks = {}
        for k in KERNELS:
            if (k, "FETCH_SIZE") not in fe and (k, "TCC_EA0_RDREQ_128B_sum") not in ea:
                continue
            wv = wr.get((k, "WRITE_SIZE"), 0.0)
            if (k, "FETCH_SIZE") in fe:
                f = fe[(k, "FETCH_SIZE")]
                e = {"fetch_size_kib_per_launch": f, "write_size_kib_per_launch": wv, "dispatches": int(fe[(k, "FETCH_SIZE:n")]),
                     "bytes_per_launch_fetch_x2": int((2.0 * f + wv) * 1024)}
            else:                            # (passes limited to the read requests by size)
                e = {"write_size_kib_per_launch": wv, "dispatches": int(ea[(k, "TCC_EA0_RDREQ_128B_sum:n")])}
            # read requests by size (their own pass): the calibrated figure for this access shape; FETCH_SIZE x 2 (the guide's
            # correction for wide streaming reads) is kept beside it
            sized = [ea.get((k, "TCC_EA0_RDREQ_%s_sum" % sz)) for sz in ("32B", "64B", "128B")]
            if all(v is not None for v in sized):
                e["read_bytes_by_request_size"] = int(32 * sized[0] + 64 * sized[1] + 128 * sized[2])
                e["bytes_per_launch"] = int(e["read_bytes_by_request_size"] + wv * 1024)
            else:
                e["bytes_per_launch"] = e["bytes_per_launch_fetch_x2"]
            if w == "c2s":
                e["frames_per_launch"] = 20                                # (sets of 20: the driver's --steps 20)
            if w.endswith("b"):
                e["frames_per_launch"] = 30 if w == "c2b" else 16         # (tools/run_profiles.sh: c2b = the bench default's sets of 30, c5b = --batch 16)
            if (k, "avg_us") in kt:
                e["avg_us"] = kt[(k, "avg_us")]
                e["GBps"] = e["bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9
            for src in (tcc, sq, ea, tcp, ta2, mix, lanes):
                for (kk, c), v in src.items():
                    if kk == k and not c.endswith(":n"):
                        e[c] = v
            # round 3: how busy the SIMDs' vector ALUs were (the gfx9 VALUBusy formula: active VALU quad-cycles x 4 over SIMDs x
            # shader-engine-active cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the L1 -> L2 read-request rate
            if "SQ_ACTIVE_INST_VALU" in e and e.get("GRBM_GUI_ACTIVE"):
                e["valu_busy"] = e["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
            if e.get("LANES_SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in e:
                e["active_lane_fraction"] = e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["LANES_SQ_ACTIVE_INST_VALU"])
            if "TCP_TCC_READ_REQ_sum" in e and "avg_us" in e:
                e["l2_read_requests_per_s"] = e["TCP_TCC_READ_REQ_sum"] / (e["avg_us"] * 1e-6)
            ks[k] = e
        out["workloads"][w] = {"kernels": ks}
    with open(d + "/traffic.json", "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out["workloads"], indent=1))


main()
