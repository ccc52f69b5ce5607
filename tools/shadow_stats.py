#!/usr/bin/env python3
"""How much of the shadow stage is spent on rays that end up occluded, and how coherent are the occluders?  Primary hit
points of the bench view in 8x8-tile order, one shadow ray each to the point light and to the sun (the frame's level-0 shadow
rays); any-hit through the production kernel for the occluded fraction and the occluder ids, canonical walk for per-ray work.
usage (GPU box): python tools/shadow_stats.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dxrexperiments_amd import capi, scenes  # noqa: E402
from util import ANY, primary_rays  # noqa: E402

ctx = capi.Context(0)
W, H = 1920, 1080
v, t = scenes.sponza_class(seed=42)
sc = capi.Scene(ctx)
sc.add_model(capi.Model(ctx, v, t))
sc.build()
cam = scenes.sponza_camera()
host = capi.ProgressiveHost(1234)
pf = host.update(capi.camera_array(cam["eye"], cam["at"], cam["up"], cam["fov"], W / H), 0.0, 1, W, H)
o, d = primary_rays(pf, W, H)
ys, xs = np.divmod(np.arange(W * H), W)
order = np.lexsort((xs % 8, ys % 8, xs // 8, ys // 8))
o, d = o[order], d[order]
hit = sc.trace(o, d, flags=0x10)
ok = hit["inst"] != 0xFFFFFFFF
P = o[ok, :3] + d[ok, :3] * hit["t"][ok, None]
n = P.shape[0]
lights = {"point light": None, "sun": None}
pl = np.array(pf["pointLight"]["worldPos"][:3], np.float32)
fw = np.array(pf["directionalLight"]["forwardDir"][:3], np.float32)
for name in lights:
    O = np.zeros((n, 4), np.float32); D = np.zeros((n, 4), np.float32)
    O[:, :3] = P; O[:, 3] = 1e-4
    if name == "sun":
        L = -fw / np.linalg.norm(fw)
        D[:, :3] = L; D[:, 3] = 1e38
    else:
        path = pl - P
        dist = np.linalg.norm(path, axis=1)
        D[:, :3] = path / dist[:, None]; D[:, 3] = dist - 1e-4
    a = sc.trace(O, D, flags=ANY)
    c = sc.trace(O, D, flags=ANY, canonical=True)
    occ = a["inst"] != 0xFFFFFFFF
    nodes = c["nodes"].astype(np.float64)
    # coherence: does the ray 1 / 8 / 64 places earlier in the queue have an occluder, and would that triangle ... (same primitive id)
    prim = np.where(occ, a["prim"].astype(np.int64), -1)
    same1 = (prim[1:] == prim[:-1]) & occ[1:]
    tiles = prim[: n // 64 * 64].reshape(-1, 64)
    occ_t = occ[: n // 64 * 64].reshape(-1, 64)
    distinct = np.array([len(set(r[r >= 0].tolist())) for r in tiles[:20000]])
    print("%-12s %d rays: %.1f %% occluded; canonical node tests per ray: occluded %.1f, unoccluded %.1f (share of all node tests spent on occluded rays %.1f %%)"
          % (name, n, 100 * occ.mean(), nodes[occ].mean(), nodes[~occ].mean(), 100 * nodes[occ].sum() / nodes.sum()))
    print("             previous ray in the queue stopped at the SAME triangle: %.1f %% of occluded rays; distinct occluders per 64 consecutive rays with any: %.1f (of %.1f occluded rays)"
          % (100 * same1.sum() / max(occ[1:].sum(), 1), distinct[distinct > 0].mean(), occ_t[:20000].sum(1)[distinct > 0].mean()))

    # two-pass estimate: every 4th ray is traced in full and leaves its occluder behind; would that triangle stop the three rays
    # after it?  (Moller-Trumbore in float64: an estimate of the hit rate, not the canonical test)
    tri = v["position"][t]                                  # (n_tris, 3, 3)
    lead = (np.arange(n) // 4) * 4
    cand = np.where(occ[lead], a["prim"][lead].astype(np.int64), -1)
    follower = (np.arange(n) % 4) != 0
    sel = follower & (cand >= 0)
    T = tri[cand[sel]].astype(np.float64)
    oo = O[sel, :3].astype(np.float64); dd = D[sel, :3].astype(np.float64)
    e1 = T[:, 1] - T[:, 0]; e2 = T[:, 2] - T[:, 0]
    pv = np.cross(dd, e2); det = np.einsum("ij,ij->i", e1, pv)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / det
        tv = oo - T[:, 0]
        uu = np.einsum("ij,ij->i", tv, pv) * inv
        qv = np.cross(tv, e1)
        vv = np.einsum("ij,ij->i", dd, qv) * inv
        tt = np.einsum("ij,ij->i", e2, qv) * inv
    stopped = (det != 0) & (uu >= 0) & (vv >= 0) & (uu + vv <= 1) & (tt > O[sel, 3]) & (tt < D[sel, 3])
    print("             two passes (every 4th ray first): %.1f %% of ALL rays of this light would end at their leader's occluder (%.1f %% of the occluded followers)"
          % (100 * stopped.sum() / n, 100 * stopped.sum() / max((follower & occ).sum(), 1)))
