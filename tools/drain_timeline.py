#!/usr/bin/env python3
"""When do the waves of a persistent traversal launch run out of queue, and when do they leave?  (instrumentation build)

  (the wave clocks live in dxrexperiments_amd/csrc/experiments/r03_traversal_experiments.patch since round 4: apply it first)
  tools/build_variant.sh times -DRT_TRACE_TIMES
  DXR_AMD_LIB=$PWD/dxrexperiments_amd/lib/variants/libtimes.so python tools/drain_timeline.py

Every wave of the selected launch (closest-hit queue = the secondary rays, any-hit queue = the shadow rays) records the
100-MHz wall clock at its start, when the chunk pool had nothing left for it, and at its exit.  Printed: how many waves are
still resident, and how many of them still draw rays from the queue, every 20 us of the launch."""
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
    fn = capi.lib().rt_debug_wave_times
    fn.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
    out = (C.c_ulonglong * (4 * 8192))()
    pipe.enable_timing(1)
    for f in range(3):
        pipe.update(host.update(cam, 0.0, f + 1, W, H))
        pipe.render()
    for sel, name in ((0, "secondary rays (closest hit)"), (1, "shadow rays (any hit)")):
        fn(sel, None)
        pipe.update(host.update(cam, 0.0, 4, W, H))
        pipe.render()
        fn(sel, out)
        st = pipe.stats()
        ms = st["ms_trace_secondary"] if sel == 0 else st["ms_trace_shadow0"] + st["ms_trace_shadow1"]
        a = np.array(out, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
        # the wall clocks of the 8 XCDs are not synchronised: workgroup b runs on XCD b mod 8, and each XCD's first wave
        # start is taken as its time zero (the dispatcher fills all XCDs within microseconds)
        xcd = (np.arange(8192) // 4) % 8
        keep = a[:, 2] > 0
        t0 = np.array([a[keep & (xcd == x), 0].min() for x in range(8)])[xcd]
        a, t0 = a[keep], t0[keep]
        # (ticks of the clock per us: calibrated on the launch's duration by HIP events)
        per_us = float((a[:, 2] - t0).max()) / (ms * 1000.0)
        start = (a[:, 0] - t0) / per_us          # us
        dry = np.where(a[:, 1] > 0, a[:, 1] - t0, a[:, 2] - t0) / per_us
        leave = (a[:, 2] - t0) / per_us
        print("launch %.3f ms by HIP events = %d clock ticks (%.1f per us)" % (ms, int((a[:, 2] - t0).max()), per_us))
        print("%s: %d waves, launch %.0f us; last wave started at %.0f us" % (name, len(a), leave.max(), start.max()))
        print("  pool dry for a wave at: first %.0f, median %.0f, last %.0f us; lanes alive then: mean %.1f" % (dry.min(), np.median(dry), dry.max(), a[:, 3].mean()))
        print("  wave leaves at: 10%% %.0f, median %.0f, 90%% %.0f, 99%% %.0f, last %.0f us" % tuple(np.percentile(leave, [10, 50, 90, 99, 100])))
        print("  after its pool ran dry a wave stays: mean %.0f, median %.0f, 90%% %.0f, max %.0f us" % ((leave - dry).mean(), np.median(leave - dry), np.percentile(leave - dry, 90), (leave - dry).max()))
        print("  time us : waves resident : of them still fed by the queue")
        for t in np.arange(0.0, leave.max() + 20.0, 20.0):
            print("  %7.0f : %5d : %5d" % (t, int(((start <= t) & (leave > t)).sum()), int(((start <= t) & (dry > t)).sum())))


main()
