#!/usr/bin/env python3
"""What each component of the stress scene (scenes.stadium_class) costs the production tree: the scene without one component at a time,
and every component alone on the hall: ms per 1080p frame in one set of 16, node steps and triangle tests per ray by stage.
usage: tools/stress_parts.py [scale]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes

ALL = ("hall", "pots", "cables", "slats", "debris")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
W, H, N = 1920, 1080, 16
ctx = capi.Context(0)
if len(sys.argv) > 2:
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        ctx.set_option(k, v)
c = scenes.stadium_camera()
cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
cases = [("all", ALL)] + [("without " + x, tuple(y for y in ALL if y != x)) for x in ALL[1:]] + [("hall + " + x, ("hall", x)) for x in ALL[1:]]
for name, parts in cases:
    v, t = scenes.stadium_class(5, scale, parts)
    scene = capi.Scene(ctx)
    scene.add_model(capi.Model(ctx, v, t))
    pipe = capi.Pipeline(ctx)
    pipe.set_scene(scene)
    pipe.add_material(T.default_material())
    pipe.set_environment_cube(scenes.sky_cubemap(32))
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(1)
    pf = [host.update(cam, 0.0, f + 1, W, H) for f in range(2 * N)]
    pipe.render_batch(pf[:N])
    ctx.synchronize()
    pipe.enable_timing(1)
    pipe.render_batch(pf[N:])
    ctx.synchronize()
    st = pipe.stats()
    pipe.update(pf[-1]); pipe.render()
    walk = pipe.count_walk()
    line = "%-16s %8d triangles  frame %7.3f ms" % (name, t.shape[0], st["ms_total"] / N)
    for stage in ("primary", "secondary", "shadow0"):
        w = walk[stage]
        if w["rays"]:
            line += " | %s %.1f steps %.1f tris" % (stage, (w["nodes_global"] + w["nodes_lds"]) / w["rays"], w["tris"] / w["rays"])
    print(line, flush=True)
    pipe.close(); del pipe, scene
