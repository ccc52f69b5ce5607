import sys, numpy as np
sys.path.insert(0,'.')
from dxrexperiments_amd import capi
ctx=capi.Context(0)
for (W,H) in ((1920,1080),(3840,2160)):
    r=np.random.default_rng(1)
    d=r.uniform(0,1,(H,W,4)).astype(np.float32); i=r.uniform(0,1,(H,W,4)).astype(np.float32)
    dn=capi.Denoiser(ctx); dn.create_output(W,H)
    td,ti=ctx.upload(d),ctx.upload(i)
    ms=[]
    for k in range(12):
        dn.dispatch(td.ptr,ti.ptr); ms.append(dn.last_ms())
    print(W,H,"denoise ms min %.3f median %.3f" % (min(ms), sorted(ms)[len(ms)//2]))
