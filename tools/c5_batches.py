#!/usr/bin/env python3
"""The 10 M-triangle 4K four-bounce workload (BASELINE configs[4] on one GPU) in sets of 1 / 4 / 8 / 16 / 24 / 32 frames:
ms per frame, rays per second, the stage times and the queue memory a set takes (rt_pipeline_get_queue_memory) -- round 3's
queues were sized for the worst case and sets of 32 did not fit 288 GB; round 4 sizes every level by count when the worst
case is over the budget.     usage (GPU box): python3 tools/c5_batches.py [set sizes ...]   -> profiles/r04/c5_batches.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

sets = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 24, 32]
W, H = 3840, 2160
v, tri = scenes.displaced_grid(2236, seed=7)
ctx = capi.Context(0)
model = capi.Model(ctx, v, tri)
scene = capi.Scene(ctx)
scene.add_model(model)
mat = T.default_material()
mat["type"] = 2
mat["reflectivity"] = 0.6
mat["roughness"] = 0.3
cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
print("# python3 tools/c5_batches.py %s: displaced-grid mesh (seed 7, %d triangles), %dx%d, 4 radiance bounces; every row a fresh pipeline, "
      "one warm-up set, then 2 timed sets (sets of 1: 8 frames)" % (" ".join(map(str, sets)), tri.shape[0], W, H))
for S in sets:
    pipe = capi.Pipeline(ctx)
    pipe.set_scene(scene)
    pipe.add_material(mat)
    pipe.set_depth_limits(4, 2)
    pipe.set_environment_cube(scenes.sky_cubemap(32))
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(3)
    host.options["maxIterations"] = 1 << 20
    pipe.set_deferred(S if S > 1 else 0)
    n_warm, n_timed = (S, 2 * S) if S > 1 else (2, 8)
    f = 0
    try:
        for _ in range(n_warm):
            f += 1
            pipe.update(host.update(cam, 0.0, f, W, H)); pipe.render()
        pipe.flush()
        ctx.synchronize()
        pipe.enable_timing(n_timed)
        pipe.reset_totals()
        t0 = time.perf_counter()
        for _ in range(n_timed):
            f += 1
            pipe.update(host.update(cam, 0.0, f, W, H)); pipe.render()
        pipe.flush()
        ctx.synchronize()
        dt = time.perf_counter() - t0
    except capi.RtError as e:
        print("sets of %2d: %s" % (S, e))
        pipe.close()
        continue
    tot = pipe.totals()
    rays = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"] - tot["rays_shadow_skipped"]
    mem, counted = pipe.queue_memory()
    nf = max(int(tot["frames"]), 1)
    st = {k: round(tot[k] / nf, 3) for k in ("ms_primary", "ms_trace_secondary", "ms_trace_shadow1", "ms_shade0", "ms_shade1", "ms_resolve", "ms_total")}
    st["ms_trace_shadow"] = round((tot["ms_trace_shadow0"] + tot["ms_trace_shadow1"]) / nf, 3)
    del st["ms_trace_shadow1"]
    print("sets of %2d: %6.3f ms/frame %5.0f Mrays/s | queue memory %6.2f GB = %5.2f GB per frame of a set (%s) | %s"
          % (S, dt / n_timed * 1e3, rays / dt / 1e6, mem / 1e9, mem / 1e9 / S, "levels sized by count" if counted else "worst case reserved up front", st))
    pipe.close()
