#!/usr/bin/env python3
"""Does the ORDER of the wide nodes in memory matter on the HBM-bound workload (10 M triangles, 4K, 4 bounces)?  Every
fabric request is 128 B for a 64-B node, so a layout that puts a node next to the child a ray is most likely to visit
next could make the second half of the line useful.  The builder numbers nodes breadth first; this script reads the tree
back, renumbers everything behind the LDS-resident top depth first (a node is followed by its first internal child and
that child's subtree), writes it back through rt_debug_wide_write and times the same frames.
usage (GPU box): python tools/layout_estimate.py [grid side, default 2236]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
W, H, frames = 3840, 2160, 6
ctx = capi.Context(0)
v, tri = scenes.displaced_grid(side, seed=7)
scene = capi.Scene(ctx)
scene.add_model(capi.Model(ctx, v, tri))
pipe = capi.Pipeline(ctx)
pipe.set_scene(scene)
mat = T.default_material()
mat["type"] = 2; mat["reflectivity"] = 0.6; mat["roughness"] = 0.3
pipe.add_material(mat)
pipe.set_depth_limits(4, 2)
pipe.set_environment_cube(scenes.sky_cubemap(32))
pipe.create_output(W, H)
pipe.build_acceleration_structures()
host = capi.ProgressiveHost(3)
host.options["maxIterations"] = 1 << 20
cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
frame_no = [0]


def measure(name):
    for _ in range(2):
        frame_no[0] += 1
        pipe.update(host.update(cam, 0.0, frame_no[0], W, H)); pipe.render()
    pipe.enable_timing(frames)
    for _ in range(frames):
        frame_no[0] += 1
        pipe.update(host.update(cam, 0.0, frame_no[0], W, H)); pipe.render()
    st = pipe.stats()
    print("%-46s frame %.2f ms: primary %.2f secondary %.2f shadow %.2f" % (name, st["ms_total"], st["ms_primary"], st["ms_trace_secondary"], st["ms_trace_shadow1"]))


measure("breadth first (as built)")
nodes, root, _ = scene.wide_read(0)
n = nodes.shape[0]
TOP = 128 if capi.wide_layout()[0] == 4 else 80
CODES = slice(12, 16) if capi.wide_layout()[0] == 4 else slice(16, 24)
code = nodes[:, CODES].view(np.int32)
t0 = time.time()
new = np.full(n, -1, np.int64)
new[:TOP] = np.arange(min(TOP, n))
nxt = min(TOP, n)
kids = code.tolist()
for top in range(min(TOP, n)):
    stack = [c for c in reversed(kids[top]) if c >= TOP]
    while stack:
        x = stack.pop()
        new[x] = nxt
        nxt += 1
        for c in reversed(kids[x]):
            if c >= 0:
                stack.append(c)
assert nxt == n and (new >= 0).all()
print("depth-first numbering of %d nodes in %.1f s" % (n, time.time() - t0))
out = np.empty_like(nodes)
out[new] = nodes
oc = out[:, CODES].view(np.int32)
m = oc >= 0
oc[m] = new[oc[m]]
scene.wide_write(out)
measure("depth first behind the top")
# control: a random permutation behind the top
r = np.random.default_rng(1)
perm = np.arange(n); perm[TOP:] = TOP + r.permutation(n - TOP)
out2 = np.empty_like(nodes)
out2[perm] = nodes
oc2 = out2[:, CODES].view(np.int32)
m = oc2 >= 0
oc2[m] = perm[oc2[m]]
scene.wide_write(out2)
measure("random order behind the top (control)")
