#!/usr/bin/env python3
"""Host-side cost of enqueueing one frame (run on the GPU box): wall time of update()+render() calls issued
back to back without synchronising, against the GPU time of the same frames."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

ctx = capi.Context(0)
CORNELL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cornell.obj")
for name, model, (W, H), cam in (("cornell 256x256", lambda: capi.Model(ctx, path=CORNELL), (256, 256), scenes.cornell_camera()),
                                 ("sponza-class 1080p", lambda: capi.Model(ctx, *scenes.sponza_class(seed=42)), (1920, 1080), scenes.sponza_camera())):
    sc = capi.Scene(ctx)
    sc.add_model(model())
    p = capi.Pipeline(ctx)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.create_output(W, H)
    p.build_acceleration_structures()
    host = capi.ProgressiveHost(1)
    host.options["maxIterations"] = 100000
    c = capi.camera_array(cam["eye"], cam["at"], cam["up"], cam["fov"], W / H)
    pfcs = [host.update(c, 0.0, f + 1, W, H) for f in range(300)]
    for f in range(20):
        p.update(pfcs[f]); p.render()
    ctx.synchronize()
    t0 = time.perf_counter()
    for f in range(20, 300):
        p.update(pfcs[f]); p.render()
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("%-20s enqueue %.1f us/frame, frames complete at %.1f us/frame (%.0f fps)" % (name, (t1 - t0) / 280 * 1e6, (t2 - t0) / 280 * 1e6, 280 / (t2 - t0)))
