#!/usr/bin/env python3
"""Upper bound for frames in flight: two contexts (a stream each) render the bench scene side by side, their frames enqueued
alternately from one thread; compared with the same number of frames through one.  If the drains of one stream's persistent
launches are filled by the other's kernels, 2 x K frames take less than twice K frames."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes


def make(W, H):
    ctx = capi.Context(0)
    scene = capi.Scene(ctx)
    pipe = capi.Pipeline(ctx)
    verts, tris = scenes.sponza_class(seed=42)
    scene.add_model(capi.Model(ctx, verts, tris))
    pipe.add_material(T.default_material())
    pipe.set_scene(scene)
    pipe.set_environment_cube(scenes.sky_cubemap(64))
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    return ctx, scene, pipe


def main():
    W, H = 1920, 1080
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sets = [make(W, H) for _ in range(n)]
    c = scenes.sponza_camera()
    cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
    hosts = [capi.ProgressiveHost(1234) for _ in sets]
    frame = [0] * n

    def run(which, K):
        for ctx, _, _ in sets:
            ctx.synchronize()
        t = time.perf_counter()
        for k in range(K):
            for i in which:
                frame[i] += 1
                sets[i][2].update(hosts[i].update(cam, 0.0, frame[i], W, H))
                sets[i][2].render()
        for ctx, _, _ in sets:
            ctx.synchronize()
        return (time.perf_counter() - t) * 1e3

    run(range(n), 5)
    for rep in range(2):
        one = run([0], 40) / 40
        many = run(range(n), 40) / (40 * n)
        print("one stream %.3f ms/frame; %d streams side by side %.3f ms/frame" % (one, n, many))


main()
