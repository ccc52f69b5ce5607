#!/usr/bin/env python3
"""Light-buffer form of the shadow cache: the table is keyed by where the ray sits in LIGHT space (directional light: the 2-D cell
of the origin projected along the light direction; point light: the cube-map texel of the direction from the light), so that
level-0 and level-1 shadow rays share entries.  Frames 1 .. N-1 fill the table (level-0 rays from the primary hits, level-1
rays from one random bounce per pixel), frame N asks.  A hit is counted when the cached triangle IS the first occluder the
ordered traversal reports (a lower bound: any occluding triangle would do)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dxrexperiments_amd import capi, rtypes as T, scenes

NONE = 0xFFFFFFFF


def normalize(v):
    return v / np.sqrt((v * v).sum(axis=1, keepdims=True))


def main():
    W, H = 960, 540
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = capi.Context(0)
    scene = capi.Scene(ctx)
    verts, tris = scenes.sponza_class(seed=42)
    scene.add_model(capi.Model(ctx, verts, tris))
    scene.build()
    lo, hi = verts["position"].min(axis=0), verts["position"].max(axis=0)
    centre, radius = 0.5 * (lo + hi), 0.5 * float(np.linalg.norm(hi - lo))
    c = scenes.sponza_camera()
    cam = capi.camera_array(c["eye"], c["at"], c["up"], c["fov"], W / H)
    host = capi.ProgressiveHost(1234)
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    dx = ((px.ravel() + 0.5) / W) * 2 - 1
    dy = ((py.ravel() + 0.5) / H) * 2 - 1
    any_flags = T.RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | T.RAY_FLAG_SKIP_CLOSEST_HIT_SHADER
    rng = np.random.default_rng(5)

    def shadow(P, light, pfc):
        n = P.shape[0]
        if light == 0:
            ld = -np.array(pfc["directionalLight"]["forwardDir"][:3], np.float32)
            ld = ld / np.sqrt((ld * ld).sum())
            sd, tmax = np.broadcast_to(ld, (n, 3)), np.full(n, 1e38, np.float32)
        else:
            path = np.array(pfc["pointLight"]["worldPos"][:3], np.float32) - P
            dist = np.sqrt((path * path).sum(axis=1))
            sd, tmax = path / dist[:, None], dist - 1e-4
        SO = np.concatenate([P, np.full((n, 1), 1e-4, np.float32)], axis=1).astype(np.float32)
        SD = np.concatenate([sd, tmax[:, None]], axis=1).astype(np.float32)
        s = scene.trace(SO, SD, flags=any_flags)
        return np.where(s["inst"] != NONE, s["prim"], NONE).astype(np.uint32), sd

    def keys(P, sd, light, pfc, res):
        if light == 0:          # 2-D cell of the origin in the plane across the light direction
            d = sd[0].astype(np.float64)
            a = np.cross(d, [0.0, 1.0, 0.0] if abs(d[1]) < 0.9 else [1.0, 0.0, 0.0]); a /= np.linalg.norm(a)
            b = np.cross(d, a)
            u = ((P - centre) @ a / radius * 0.5 + 0.5) * res
            v = ((P - centre) @ b / radius * 0.5 + 0.5) * res
            return (np.clip(u, 0, res - 1).astype(np.int64) * res + np.clip(v, 0, res - 1).astype(np.int64))
        m = -sd                 # from the light towards the point: cube-map texel
        ax = np.abs(m).argmax(axis=1)
        ma = np.take_along_axis(m, ax[:, None], axis=1)[:, 0]
        face = ax * 2 + (ma < 0)
        o1, o2 = (ax + 1) % 3, (ax + 2) % 3
        s = np.take_along_axis(m, o1[:, None], axis=1)[:, 0] / np.abs(ma)
        t = np.take_along_axis(m, o2[:, None], axis=1)[:, 0] / np.abs(ma)
        r2 = res // 2
        return face * r2 * r2 + np.clip((s * 0.5 + 0.5) * r2, 0, r2 - 1).astype(np.int64) * r2 + np.clip((t * 0.5 + 0.5) * r2, 0, r2 - 1).astype(np.int64)

    for res in (1024, 2048, 4096):
        tables = [np.full(res * res * 2, NONE, np.uint32) for _ in range(2)]
        host = capi.ProgressiveHost(1234)
        for f in range(1, frames + 1):
            pfc = np.frombuffer(np.asarray(host.update(cam, 0.0, f, W, H)).tobytes(), T.PER_FRAME_CONSTANTS)[0]
            cp = pfc["cameraParams"]
            eye, U, V, Wv = (np.array(cp[k][:3], np.float32) for k in ("worldEyePos", "U", "V", "W"))
            o = eye + np.array([cp["jitters"][0] * 30, cp["jitters"][1] * 30, 0], np.float32)
            d = normalize(dx[:, None] * U - dy[:, None] * V + Wv).astype(np.float32)
            n = d.shape[0]
            O = np.concatenate([np.broadcast_to(o, (n, 3)), np.zeros((n, 1), np.float32)], axis=1).astype(np.float32)
            D = np.concatenate([d, np.full((n, 1), 1e38, np.float32)], axis=1).astype(np.float32)
            h = scene.trace(O, D, flags=T.RAY_FLAG_CULL_BACK_FACING_TRIANGLES)
            hit = h["inst"] != NONE
            P0 = (O[:, :3] + h["t"][:, None] * d).astype(np.float32)[hit]
            # level 1: one random direction per hit point (any hemisphere: sign flipped towards the incoming ray's side)
            rd = normalize(rng.normal(size=P0.shape)).astype(np.float32)
            rd *= np.where((rd * d[hit]).sum(axis=1, keepdims=True) > 0, -1.0, 1.0)
            O1 = np.concatenate([P0, np.full((P0.shape[0], 1), 1e-4, np.float32)], axis=1).astype(np.float32)
            D1 = np.concatenate([rd, np.full((P0.shape[0], 1), 1e38, np.float32)], axis=1).astype(np.float32)
            h1 = scene.trace(O1, D1, flags=0)
            hit1 = h1["inst"] != NONE
            P1 = (P0 + h1["t"][:, None] * rd).astype(np.float32)[hit1]
            report = []
            for level, P in ((0, P0), (1, P1)):
                for light in (0, 1):
                    prim, sd = shadow(P, light, pfc)
                    k = keys(P, sd, light, pfc, res)
                    occ = prim != NONE
                    if f == frames:
                        got = tables[light][k]
                        report.append((level, light, 100.0 * occ.mean(), 100.0 * (occ & (got == prim)).sum() / max(occ.sum(), 1),
                                       100.0 * (~occ & (got != NONE)).sum() / max((~occ).sum(), 1)))
                    else:
                        tables[light][k[occ]] = prim[occ]
            if f == frames:
                for level, light, o_, hit_, stale in report:
                    print("table %4d^2 x 2, %d frames to fill: level %d, %s light: %.1f %% occluded, cached triangle = first occluder for %.1f %% of them; "
                          "unoccluded rays that find an entry: %.1f %%" % (res, frames - 1, level, ("directional", "point")[light], o_, hit_, stale))


main()
