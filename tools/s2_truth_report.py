#!/usr/bin/env python3
"""S2 against the float64 geometric truth (oracle/truth64.h), per scene, ray set and kind of search: how often the engine's TraceRay (the oracle;
the kernels equal it bit for bit, tests/test_gpu_s2_truth.py) names another primitive than geometry does.  CPU only.
usage: python tools/s2_truth_report.py [rays per set] [--unsplit] [--images W H] [--write]
  --unsplit   also with no triangle split into references (orc_set_split_refs(0))
  --images    frames of the atrium and the stress scene traced by the truth against the same frames traced by the rule (RMS)
  --write     write tests/golden/s2_bounds.json: the measured counts as the bounds tests/test_s2_truth.py holds the rule to, with the hash of the
              rule's code (a change of the rule without new bounds fails that test)"""
import json
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as orc  # noqa: E402
import s2_truth as S  # noqa: E402

args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else S.BOUND_RAYS
variants = [("rule", True)] + ([("unsplit", False)] if "--unsplit" in args else [])
cores = max(1, len(os.sched_getaffinity(0)))
table = {}
for name in S.SCENES:
    models, instances, aim = S.scene_models(name)
    models = S.load_arrays(orc, models)
    sets = S.ray_sets(models, instances, aim, n, seed=S.BOUND_SEED)
    for vname, split in variants:
        orc.set_split_refs(split)
        sc = S.oracle_scene(orc, models, instances)
        t0 = time.time()
        m = S.measure(lambda O, D, f: sc.trace(O, D, f, mode=1, nthreads=cores), lambda O, D, f: sc.truth64(O, D, f, nthreads=cores), sets)
        if split:
            table[name] = m
        for sname in m:
            for mname, c in m[sname].items():
                print("%-16s %-8s %-7s %-13s rays %6d  tie %4d  lost %4d  phantom %4d" % (name, vname, sname, mname, c["rays"], c.get("tie", 0), c["lost"], c["phantom"]), flush=True)
        print("   (%.1f s)" % (time.time() - t0), flush=True)
orc.set_split_refs(True)
images = {}
if "--images" in args:
    k = args.index("--images")
    W, H = int(args[k + 1]), int(args[k + 2])
    for name in ("atrium", "stadium"):
        t0 = time.time()
        images[name] = S.frame_rms(orc, name, W, H, cores)
        print("frame %-8s %s   (%.0f s)" % (name, images[name], time.time() - t0), flush=True)
if "--write" in args:
    assert n == S.BOUND_RAYS and not images or (images and images["atrium"]["width"] == S.BOUND_FRAME[0])
    path = os.path.join(ROOT, "tests", "golden", "s2_bounds.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    doc = {"what": "S2 (TraceRay) against the float64 geometric truth: measured counts, held as upper bounds by tests/test_s2_truth.py and tests/test_gpu_s2_truth.py; written by tools/s2_truth_report.py --write",
           "rule_hash": S.rule_hash(ROOT), "rays_per_set": n, "seed": S.BOUND_SEED, "tie_tolerance": S.TIE,
           "scenes": table, "frames": images or old.get("frames", {}),
           "earlier_rules": old.get("earlier_rules", {})}
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
