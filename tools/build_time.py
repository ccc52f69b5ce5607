#!/usr/bin/env python3
"""Acceleration-structure build time of the bench scene, cold and warm (run on the GPU box).

The first build in a process also pays the one-off loading of the build kernels' code objects; later builds
in the same context are the steady state (temporaries come from the context's build arena)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dxrexperiments_amd import capi, scenes  # noqa: E402

ctx = capi.Context(0)
v, t = scenes.sponza_class(seed=42)
for k in range(3):
    sc = capi.Scene(ctx)
    sc.add_model(capi.Model(ctx, v, t))
    sc.build()
    print("build %d of %d triangles: %.3f ms" % (k, t.shape[0], sc.build_ms()))
