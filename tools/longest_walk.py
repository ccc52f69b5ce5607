#!/usr/bin/env python3
"""Per frame of the bench scene: stage times and the longest single-ray walk of each traversal stage
(rt_pipeline_count_walk).  A persistent traversal launch ends with its slowest lane, so one ray that walks
thousands of nodes shows up as a millisecond tail.   usage: tools/longest_walk.py [frames] [width height]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dxrexperiments_amd import capi, rtypes as T, scenes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
ctx = capi.Context(0)
verts, tris = scenes.sponza_class(seed=42)
scene = capi.Scene(ctx)
scene.add_model(capi.Model(ctx, verts, tris))
pipe = capi.Pipeline(ctx)
pipe.set_scene(scene)
pipe.add_material(T.default_material())
pipe.set_environment_cube(scenes.sky_cubemap(64))
pipe.create_output(W, H)
pipe.build_acceleration_structures()
host = capi.ProgressiveHost(1234)
cam = scenes.sponza_camera()
cam11 = capi.camera_array(cam["eye"], cam["at"], cam["up"], cam["fov"], W / H)
pipe.enable_timing(1)
for f in range(frames):
    pipe.update(host.update(cam11, 0.0, f + 1, W, H))
    pipe.render()
    st = pipe.stats()
    w = pipe.count_walk()
    print("frame %2d  primary %.3f ms  secondary %.3f ms  shadow %.3f ms | longest walk (steps): " % (f, st["ms_primary"], st["ms_trace_secondary"], st["ms_trace_shadow1"])
          + "  ".join("%s %d" % (k, v["longest_walk"]) for k, v in w.items()))
    if w["secondary"]["longest_walk"] > 400:
        o, d = pipe.secondary_ray(w["secondary"]["longest_walk_ray"])
        print("      ray %d: origin %r tmin %g dir %r tmax %g" % (w["secondary"]["longest_walk_ray"], [float(x) for x in o[:3]], o[3], [float(x) for x in d[:3]], d[3]))
