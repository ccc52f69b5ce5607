#!/usr/bin/env python3
"""Triangles, traversal records (one per reference of a split triangle: DESIGN.md section 2) and four-wide nodes of the stress scene at both sizes.
usage (GPU box): [DXR_AMD_LIB=variant.so] python tools/refs_count.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dxrexperiments_amd import capi, scenes  # noqa: E402

for name, scale in (("stadium", 1.0), ("stadium2m", 8.0)):
    ctx = capi.Context(0)
    v, t = scenes.stadium_class(seed=5, scale=scale)
    s = capi.Scene(ctx)
    s.add_model(capi.Model(ctx, v, t))
    s.build()
    ctx.synchronize()
    nn, nr = s.wide_counts(0)
    print(name, "tris", len(t), "records", nr, "nodes", nn, "build ms %.2f" % s.build_ms())
