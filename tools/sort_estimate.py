#!/usr/bin/env python3
"""What would sorting the diffuse rays buy?  Primary hit points of the bench view (8x8-tile order), one cosine-hemisphere-like
ray per hit; the production closest-hit kernel (rt_trace_batch on device... here host arrays, kernel time from events) on
  (a) the rays in tile order (what the pipeline traces),
  (b) the same rays grouped by direction octant inside windows of W consecutive rays,
  (c) fully sorted by (octant, Morton code of the origin).
usage (GPU box): python tools/sort_estimate.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dxrexperiments_amd import capi, scenes  # noqa: E402
from util import primary_rays  # noqa: E402

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
# 8x8 tile order, as the pipeline's pixel slots
ys, xs = np.divmod(np.arange(W * H), W)
order = np.lexsort((xs % 8, ys % 8, xs // 8, ys // 8))
o, d = o[order], d[order]
hit = sc.trace(o, d, flags=0x10)
ok = hit["inst"] != 0xFFFFFFFF
P = o[ok, :3] + d[ok, :3] * hit["t"][ok, None]
r = np.random.default_rng(1)
dirs = r.normal(size=P.shape).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
# flip into the hemisphere of the (geometric) side the camera ray came from
flip = np.sum(dirs * d[ok, :3], axis=1) > 0
dirs[flip] *= -1
O2 = np.zeros((P.shape[0], 4), np.float32); D2 = np.zeros_like(O2)
O2[:, :3] = P; O2[:, 3] = 1e-4
D2[:, :3] = dirs; D2[:, 3] = 1e38
octant = (dirs[:, 0] > 0).astype(np.int64) | ((dirs[:, 1] > 0).astype(np.int64) << 1) | ((dirs[:, 2] > 0).astype(np.int64) << 2)


def run(name, perm):
    a, b = O2[perm], D2[perm]
    best = 1e9
    for _ in range(4):
        sc.trace(a, b)
        best = min(best, sc.trace_last_ms())
    print("%-44s %7.3f ms  %6.0f Mrays/s" % (name, best, a.shape[0] / best / 1e3))


n = P.shape[0]
idx = np.arange(n)
run("tile order (as traced today)", idx)
for win in (1024, 4096, 65536):
    run("octant groups inside windows of %d" % win, np.lexsort((octant, idx // win)))
lo, hi = P.min(0), P.max(0)
q = np.clip(((P - lo) / (hi - lo) * 1023).astype(np.int64), 0, 1023)


def spread(x):
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


morton = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
run("sorted by (octant, origin Morton code)", np.lexsort((morton, octant)))
run("sorted by (origin Morton code >> 12, octant)", np.lexsort((octant, morton >> 12)))
run("random order", r.permutation(n))
