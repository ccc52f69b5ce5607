#!/usr/bin/env python3
"""Profiling workload C4 (BASELINE configs[3]): 4096 instances of two meshes (TLAS over many BLAS instances), 3840x2160,
RealtimeRaytracingPipeline + DenoiseCompositor, N frames.   usage (under rocprofv3): python3 tools/profile_c4.py [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 3840, 2160
ctx = capi.Context(0)
sus = capi.Model(ctx, path=os.path.join(ROOT, "tests", "golden", "susanne.obj"))
blob = capi.Model(ctx, *scenes.blob_mesh(level=3))
scene = capi.Scene(ctx)
xf = scenes.instance_grid(64, spacing=3.0)
pipe = capi.Pipeline(ctx, capi.PIPELINE_REALTIME)
r = np.random.default_rng(5)
for k in range(xf.shape[0]):
    scene.add_model(sus if k % 2 == 0 else blob, xf[k])
    m = T.default_material()
    m["albedo"][:3] = r.uniform(0.1, 0.9, 3)
    m["type"] = k % 3
    pipe.add_material(m)
pipe.set_scene(scene)
pipe.set_environment_cube(scenes.sky_cubemap(32))
pipe.create_output(W, H)
pipe.build_acceleration_structures()
host = capi.ProgressiveHost(4)
cam = capi.camera_array((0.0, 30.0, 110.0), (0.0, 0.0, 0.0), (0, 1, 0), 0.9, W / H)
dn = capi.Denoiser(ctx)
dn.create_output(W, H)
pipe.enable_timing(frames)
for f in range(frames):
    pipe.update(host.update_realtime(cam, 0.0, f + 1, W, H))
    pipe.render()
    dn.dispatch(pipe.output_device_ptr(0), pipe.output_device_ptr(1))
st = pipe.stats()
rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
w = pipe.count_walk()
print("C4: 4096 instances of 2 BLASes, 4K realtime frame %.2f ms = %.0f Mrays/s, denoise %.3f ms, TLAS+BLAS build %.2f ms"
      % (st["ms_total"], rays / st["ms_total"] / 1e3, dn.last_ms(), scene.build_ms()))
print("   stage ms: primary %.3f shade0 %.3f secondary %.3f shade1 %.3f shadow %.3f resolve %.3f"
      % (st["ms_primary"], st["ms_shade0"], st["ms_trace_secondary"], st["ms_shade1"], st["ms_trace_shadow0"] + st["ms_trace_shadow1"], st["ms_resolve"]))
# what the production walk fetched (rt_pipeline_count_walk): per ray node steps from global memory / from the LDS tops, triangle
# tests, instance entries, distinct 64-B lines gathered -- the figures the single-level workloads report in bench.py
for k, v in w.items():
    n = max(v["rays"], 1)
    print("   %-9s rays %9d  nodes global %.2f  lds %.2f  tris %.2f  instance entries %.2f  lines %.2f  longest walk %d"
          % (k, v["rays"], v["nodes_global"] / n, v["nodes_lds"] / n, v["tris"] / n, v["instance_entries"] / n, v["lines"] / n, v["longest_walk"]))
