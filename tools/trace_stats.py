#!/usr/bin/env python3
"""SIMD utilisation of the traversal engine on the bench workload (instrumentation build only).

  (the counters live in dxrexperiments_amd/csrc/experiments/r03_traversal_experiments.patch since round 4: apply it first)
  tools/build_variant.sh stats -DRT_TRACE_STATS [other -D flags]
  DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libstats.so python tools/trace_stats.py [instances]

Prints, per stage (primary / secondary / shadow): wave-level node steps, the fraction of the 64 lanes
doing useful work in them, leaf phases and triangle iterations likewise."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes


def main():
    W, H = 1920, 1080
    ctx = capi.Context(0)
    scene = capi.Scene(ctx)
    pipe = capi.Pipeline(ctx)
    if len(sys.argv) > 1 and sys.argv[1] == "instances":        # BASELINE config 4's scene: 4096 instances, two-level walks
        model = capi.Model(ctx, *scenes.blob_mesh(level=3))
        xf = scenes.instance_grid(64)
        for k in range(xf.shape[0]):
            scene.add_model(model, xf[k])
            pipe.add_material(T.default_material())
        c = dict(eye=(0.0, 30.0, 110.0), at=(0.0, 0.0, 0.0), up=(0, 1, 0), fov=0.9)
    else:
        verts, tris = scenes.sponza_class(seed=42)
        scene.add_model(capi.Model(ctx, verts, tris))
        pipe.add_material(T.default_material())
        c = scenes.sponza_camera()
    pipe.set_scene(scene)
    pipe.set_environment_cube(scenes.sky_cubemap(64))
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(1234)
    cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
    lib = capi.lib()
    fn = lib.rt_debug_trace_stats
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * 72)()
    for f in range(3):
        pipe.update(host.update(cam, 0.0, f + 1, W, H))
        pipe.render()
    fn(out)
    pipe.update(host.update(cam, 0.0, 4, W, H))
    pipe.render()
    fn(out)
    st = pipe.stats()
    rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
    names = ["node steps", "leaf phases", "triangle iterations", "outer iterations"]
    print("one frame, all traversal kernels: %d rays" % rays)
    for k, n in enumerate(names):
        w, l = out[2 * k], out[2 * k + 1]
        print("%-20s wave-level %12d  lane-level %14d  lanes active %.1f %%  per ray %.1f" % (n, w, l, 100.0 * l / (64.0 * w) if w else 0.0, l / rays))
    hist = np.array(out[8:68], dtype=np.uint64)
    print("shadow cache: occluded shadow rays whose occluder was their cell's entry %d, found by the walk %d (of %d shadow rays)" % (out[68], out[69], st["rays_shadow"]))
    tot = float(hist.sum())
    top = int(np.nonzero(hist)[0].max())
    cum = np.cumsum(hist) / tot
    print("deepest stack pointer per ray (closest-hit rays; any-hit rays that found a hit are not counted): max %d" % top)
    print("  " + "  ".join("<=%d: %.4f%%" % (d, 100 * cum[d]) for d in range(4, top + 1, 2)))




main()
