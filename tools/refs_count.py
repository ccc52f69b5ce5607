import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from dxrexperiments_amd import capi, scenes
for name, scale in (("stadium", 1.0), ("stadium2m", None)):
    ctx = capi.Context(0)
    v, t = scenes.stadium_class(seed=5, scale=1.0) if scale else scenes.stadium_class(seed=5, scale=8.0)
    m = capi.Model(ctx, v, t)
    s = capi.Scene(ctx); s.add_model(m)
    t0 = time.perf_counter(); s.build(); ctx.synchronize(); t1 = time.perf_counter()
    s.build(); ctx.synchronize(); t2 = time.perf_counter()
    nn, nr = s.wide_counts(0)
    print(name, "tris", len(t), "records", nr, "nodes", nn, "rebuild ms %.2f" % ((t2 - t1) * 1e3))
