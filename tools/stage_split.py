#!/usr/bin/env python3
"""Where the level-1 stage spends its time (run on the GPU box): the bench scene rendered with both secondary
batches, with the specular batch only (noIndirectDiffuse) and with the diffuse batch only (material type 0)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes
W, H = 1920, 1080
ctx = capi.Context(0)
v, t = scenes.sponza_class(seed=42)
sc = capi.Scene(ctx); sc.add_model(capi.Model(ctx, v, t))
c = scenes.sponza_camera(); cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
for name, opt, mtype in (("both", {}, 1), ("specular only", {"noIndirectDiffuse": 1}, 1), ("diffuse only", {}, 0)):
    p = capi.Pipeline(ctx); p.set_scene(sc)
    m = T.default_material(); m["type"] = mtype
    p.add_material(m); p.set_environment_cube(scenes.sky_cubemap(64)); p.create_output(W, H); p.build_acceleration_structures()
    host = capi.ProgressiveHost(1234)
    for k, val in opt.items(): host.options[k] = val
    p.enable_timing(20)
    for f in range(25):
        p.update(host.update(cam, 0.0, f + 1, W, H)); p.render()
    tot = p.totals(); n = tot["frames"]
    print("%-14s secondary rays/frame %8d  trace %.3f ms  shadow rays %9d trace %.3f ms  frame %.3f ms" % (
        name, tot["rays_secondary"] // 25, tot["ms_trace_secondary"] / n, tot["rays_shadow"] // 25, tot["ms_trace_shadow1"] / n, tot["ms_total"] / n))
