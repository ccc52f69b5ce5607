#!/usr/bin/env python3
"""The re-packed any-hit engine (option repack=1) on the bench scene: its own tallies -- lanes per node step, per leaf pass, rays through the queues -- beside
the production engine's counting re-walk.  GPU box.   usage: python tools/repack_stats.py [frames per set]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = 1920, 1080
v, t = scenes.sponza_class(seed=42)
cam = scenes.sponza_camera()
cam = np.array([*cam["eye"], *cam["at"], *cam["up"], cam["fov"], W / H], np.float32)
for on in (0, 1):
    ctx = capi.Context(0)
    ctx.set_option("repack", on)
    sc = capi.Scene(ctx)
    sc.add_model(capi.Model(ctx, v, t))
    p = capi.Pipeline(ctx)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.create_output(W, H)
    p.build_acceleration_structures()
    p.set_deferred(S)
    host = capi.ProgressiveHost(1234)
    for rep in range(3):
        if rep == 2:
            ctx.repack_stats()
            ctx.synchronize()
            t0 = time.perf_counter()
        for f in range(S):
            p.update(host.update(cam, 0.0, rep * S + f + 1, W, H))
            p.render()
        p.flush() if hasattr(p, "flush") else None
        ctx.synchronize()
    ms = (time.perf_counter() - t0) / S * 1e3
    tot = p.totals()
    line = "repack=%d  %.3f ms per frame in sets of %d" % (on, ms, S)
    if on:
        st = ctx.repack_stats()
        rays = tot["rays_shadow"] / max(tot["frames"], 1) * S
        line += " | node steps %.2f M per frame with %.3f of their lanes, leaf passes %.2f M with %.3f; per shadow ray: %.2f trips through the leaf queue, %.2f through the node queue; refills %.2f M; aborts %d" % (
            st["node_steps"] / S / 1e6, st["node_step_lanes"] / max(st["node_steps"], 1) / 64, st["leaf_passes"] / S / 1e6, st["leaf_pass_lanes"] / max(st["leaf_passes"], 1) / 64,
            st["rays_to_leaf_queue"] / rays, st["rays_to_node_queue"] / rays, st["refills"] / S / 1e6, st["watchdog_aborts"])
    print(line, flush=True)
    ctx.close()
