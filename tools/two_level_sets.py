#!/usr/bin/env python3
"""A two-level PROGRESSIVE scene in sets of frames: 1024 instances of two meshes at 1080p, sets of 16 through rt_pipeline_render_batch -- the one
workload whose primary stage runs the two-level kernel of a set (profiles/r04/two_level_sets.txt).   usage (GPU box): python3 tools/two_level_sets.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from dxrexperiments_amd import capi, rtypes as T, scenes
W, H = 1920, 1080
ctx = capi.Context(0)
sus = capi.Model(ctx, path=os.path.join("tests", "golden", "susanne.obj"))
blob = capi.Model(ctx, *scenes.blob_mesh(level=3))
scene = capi.Scene(ctx)
xf = scenes.instance_grid(32, spacing=3.0)
pipe = capi.Pipeline(ctx)
for k in range(xf.shape[0]):
    scene.add_model(sus if k % 2 == 0 else blob, xf[k])
    pipe.add_material(T.default_material())
pipe.set_scene(scene); pipe.set_environment_cube(scenes.sky_cubemap(32)); pipe.create_output(W, H)
pipe.build_acceleration_structures()
host = capi.ProgressiveHost(4)
cam = capi.camera_array((0.0, 20.0, 60.0), (0.0, 0.0, 0.0), (0, 1, 0), 0.9, W / H)
pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(48)]
pipe.reserve_batch(16)
pipe.render_batch(pfcs[:16]); ctx.synchronize()
pipe.enable_timing(32)
t0 = time.perf_counter()
pipe.render_batch(pfcs[16:32]); pipe.render_batch(pfcs[32:48]); ctx.synchronize()
t = (time.perf_counter() - t0) / 32 * 1e3
st = pipe.stats()
print("two-level progressive, 1024 instances, 1080p, sets of 16: %.3f ms per frame" % t, {k: round(v, 3) for k, v in st.items() if k.startswith("ms_")})
