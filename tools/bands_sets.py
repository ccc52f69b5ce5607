#!/usr/bin/env python3
"""Tile partition in sets of frames (task 4): the 10 M-triangle 4K four-bounce workload (BASELINE configs[4]) rendered
  (a) whole, in sets of S frames (rt_pipeline_render_batch),
  (b) as R = 8 ranks' interleaved 16-row bands, one rank after the other on ONE device, each rank's bands of S frames per set of
      launches (rt_pipeline_render_bands_batch) -- what each of 8 GPUs would run, eight times over,
  (c) the same bands one frame per set of launches (rt_pipeline_render_bands: round 3's form).
Time per frame of (b) and (c) against (a): the price of partitioning per pixel.   usage (GPU box): python3 tools/bands_sets.py [S] [R]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, band = 3840, 2160, 16
v, tri = scenes.displaced_grid(2236, seed=7)
ctx = capi.Context(0)
scene = capi.Scene(ctx)
scene.add_model(capi.Model(ctx, v, tri))
mat = T.default_material()
mat["type"] = 2
mat["reflectivity"] = 0.6
mat["roughness"] = 0.3
pipe = capi.Pipeline(ctx)
pipe.set_scene(scene)
pipe.add_material(mat)
pipe.set_depth_limits(4, 2)
pipe.set_environment_cube(scenes.sky_cubemap(32))
pipe.create_output(W, H)
pipe.build_acceleration_structures()
host = capi.ProgressiveHost(3)
host.options["maxIterations"] = 1 << 20
cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(2 * S)]


def timed(fn):
    fn(pfcs[:S])                      # warm-up set (sizes the queues)
    ctx.synchronize()
    pipe.clear_output()
    t0 = time.perf_counter()
    fn(pfcs[S:])
    ctx.synchronize()
    return (time.perf_counter() - t0) / S * 1e3, pipe.read_output()


def whole(frames):
    pipe.render_batch(frames)


def bands_sets(frames):
    for r in range(R):
        pipe.render_bands_batch(band, r, R, frames)


def bands_single(frames):
    for c in frames:
        pipe.update(c)
        for r in range(R):
            pipe.render_bands(band, r, R)


a, img_a = timed(whole)
b, img_b = timed(bands_sets)
c, img_c = timed(bands_single)
assert np.array_equal(img_a, img_b) and np.array_equal(img_a, img_c), "the partitioned image differs from the whole frames"
print("# python3 tools/bands_sets.py %d %d: 10 M triangles, %dx%d, 4 bounces, %d timed frames after a warm-up set; images bit-identical" % (S, R, W, H, S))
print("whole frames, sets of %d:                         %7.3f ms per frame" % (S, a))
print("%d ranks' bands in turn, sets of %d frames:        %7.3f ms per frame  (x %.3f)" % (R, S, b, b / a))
print("%d ranks' bands in turn, one frame per launch set: %7.3f ms per frame  (x %.3f)" % (R, c, c / a))
