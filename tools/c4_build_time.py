import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from dxrexperiments_amd import capi, rtypes as T, scenes
ctx = capi.Context(0)
sus = capi.Model(ctx, path=os.path.join("tests", "golden", "susanne.obj"))
blob = capi.Model(ctx, *scenes.blob_mesh(level=3))
xf = scenes.instance_grid(64, spacing=3.0)
for k in range(4):
    scene = capi.Scene(ctx)
    for i in range(xf.shape[0]):
        scene.add_model(sus if i % 2 == 0 else blob, xf[i])
    t0 = time.perf_counter()
    scene.build()
    ctx.synchronize()
    t1 = time.perf_counter()
    print("build %d: gpu %.3f ms, wall %.3f ms" % (k, scene.build_ms(), (t1 - t0) * 1e3))
