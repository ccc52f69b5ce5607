#!/usr/bin/env python3
"""How much would a per-pixel shadow cache (last frame's occluder tested first) save on the bench scene?

Two consecutive frames of the bench view: primary rays restated in numpy (RayGen, ProgressiveRaytracing.hlsl:11-39), traced
through rt_trace_batch; from each hit point one shadow ray per light as the shaders build them (RaytracingCommon.hlsli:126-147),
traced with ACCEPT_FIRST_HIT.  Printed: the share of the level-0 shadow rays that are occluded, and of those the share whose
first occluder is the same triangle in both frames (a lower bound for the cache's hit rate: another triangle found first may
hide that the cached one occludes too)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes


def normalize(v):
    return v / np.sqrt((v * v).sum(axis=1, keepdims=True))


def main():
    W, H = 1920, 1080
    ctx = capi.Context(0)
    scene = capi.Scene(ctx)
    verts, tris = scenes.sponza_class(seed=42)
    scene.add_model(capi.Model(ctx, verts, tris))
    scene.build()
    c = scenes.sponza_camera()
    cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
    host = capi.ProgressiveHost(1234)
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    dx = ((px.ravel() + 0.5) / W) * 2 - 1
    dy = ((py.ravel() + 0.5) / H) * 2 - 1
    any_flags = T.RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | T.RAY_FLAG_SKIP_CLOSEST_HIT_SHADER
    occl = []
    for f in (1, 2):
        pfc = np.frombuffer(np.asarray(host.update(cam, 0.0, f, W, H)).tobytes(), T.PER_FRAME_CONSTANTS)[0]
        cp = pfc["cameraParams"]
        eye, U, V, Wv = (np.array(cp[k][:3], np.float32) for k in ("worldEyePos", "U", "V", "W"))
        o = eye + np.array([cp["jitters"][0] * 30, cp["jitters"][1] * 30, 0], np.float32)
        d = normalize(dx[:, None] * U - dy[:, None] * V + Wv).astype(np.float32)
        n = d.shape[0]
        O = np.concatenate([np.broadcast_to(o, (n, 3)), np.zeros((n, 1), np.float32)], axis=1).astype(np.float32)
        D = np.concatenate([d, np.full((n, 1), 1e38, np.float32)], axis=1).astype(np.float32)
        h = scene.trace(O, D, flags=T.RAY_FLAG_CULL_BACK_FACING_TRIANGLES)
        hit = h["inst"] != 0xFFFFFFFF
        P = (O[:, :3] + h["t"][:, None] * d).astype(np.float32)
        ld = -np.array(pfc["directionalLight"]["forwardDir"][:3], np.float32)
        ld = ld / np.sqrt((ld * ld).sum())
        lp = np.array(pfc["pointLight"]["worldPos"][:3], np.float32)
        frame = []
        for light in (0, 1):
            if light == 0:
                sd, tmax = np.broadcast_to(ld, (n, 3)), np.full(n, 1e38, np.float32)
            else:
                path = lp - P
                dist = np.sqrt((path * path).sum(axis=1))
                sd, tmax = path / dist[:, None], dist - 1e-4
            SO = np.concatenate([P, np.full((n, 1), 1e-4, np.float32)], axis=1).astype(np.float32)
            SD = np.concatenate([sd, tmax[:, None]], axis=1).astype(np.float32)
            s = scene.trace(SO[hit], SD[hit], flags=any_flags)
            prim = np.full(n, 0xFFFFFFFF, np.uint32)
            prim[hit] = np.where(s["inst"] != 0xFFFFFFFF, s["prim"], 0xFFFFFFFF)
            frame.append((hit, prim, P))
        occl.append(frame)
    for light, name in ((0, "directional light"), (1, "point light")):
        (h1, p1, P1), (h2, p2, P2) = occl[0][light], occl[1][light]
        both = h1 & h2
        o2 = both & (p2 != 0xFFFFFFFF)
        same = o2 & (p1 == p2)
        had = o2 & (p1 != 0xFFFFFFFF)
        print("%s: %d pixels hit in both frames; frame 2: %.1f %% of their shadow rays occluded; of those the frame before was occluded "
              "too for %.1f %% and by the same first triangle for %.1f %%; unoccluded rays that would test a stale entry: %.1f %% of all"
              % (name, both.sum(), 100.0 * o2.sum() / both.sum(), 100.0 * had.sum() / max(o2.sum(), 1), 100.0 * same.sum() / max(o2.sum(), 1),
                 100.0 * (both & (p2 == 0xFFFFFFFF) & (p1 != 0xFFFFFFFF)).sum() / both.sum()))

    # the cache as built: a table of 2^21 entries keyed by a hash of the ray origin's grid cell (and the light), filled by the
    # occluded rays of frame 1, asked by the rays of frame 2
    for cell in (0.02, 0.05, 0.1, 0.2, 0.4):
        out = []
        for light in (0, 1):
            (h1, p1, P1), (h2, p2, P2) = occl[0][light], occl[1][light]

            def key(P):
                q = np.floor(P / cell).astype(np.int64)
                return ((q[:, 0] * 73856093) ^ (q[:, 1] * 19349663) ^ (q[:, 2] * 83492791) ^ (light * 0x9E3779B1)) & ((1 << 21) - 1)
            table = np.full(1 << 21, 0xFFFFFFFF, np.uint32)
            w = h1 & (p1 != 0xFFFFFFFF)
            table[key(P1[w])] = p1[w]
            got = table[key(P2)]
            o2 = h2 & (p2 != 0xFFFFFFFF)
            out.append((100.0 * (o2 & (got == p2)).sum() / max(o2.sum(), 1), 100.0 * (h2 & ~o2 & (got != 0xFFFFFFFF)).sum() / max((h2 & ~o2).sum(), 1)))
        print("cell %.2f: occluded rays whose first occluder is the cached triangle: directional %.1f %%, point %.1f %%; unoccluded rays that "
              "find an entry to test: %.1f %% / %.1f %%" % (cell, out[0][0], out[1][0], out[0][1], out[1][1]))


main()
