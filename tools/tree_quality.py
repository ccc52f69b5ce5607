#!/usr/bin/env python3
"""Surface-area cost and shape of the production traversal tree of the bench scenes (run on the GPU box).
usage: python tools/tree_quality.py [c2] [c5]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import wide_tree as W  # noqa: E402
from dxrexperiments_amd import capi, scenes  # noqa: E402

ctx = capi.Context(0)
for name in (sys.argv[1:] or ["c2"]):
    v, t = scenes.sponza_class(seed=42) if name == "c2" else scenes.displaced_grid(2236, seed=7)     # bench.py: headline scene / 10 M triangles
    sc = capi.Scene(ctx)
    sc.add_model(capi.Model(ctx, v, t))
    sc.build()
    nodes, root, recs = sc.wide_read(0)
    d = W.decode(nodes)
    used = d["code"] != W.NONE
    leaf = used & (d["code"] < 0)
    cnt = (((~d["code"]) & 7) + 1)[leaf]
    nt, it = W.sah(nodes, root)
    print("%s: %d triangles, %d wide nodes, %.2f children/node, leaves of %s triangles, SAH node term %.2f item term %.2f, build %.2f ms"
          % (name, t.shape[0], nodes.shape[0], used.sum() / nodes.shape[0], np.bincount(cnt)[1:].tolist(), nt, it, sc.build_ms()))
