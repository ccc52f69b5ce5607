"""The re-packed any-hit engine (option repack=1, csrc/rt_trace_repack.h; round 6, VERDICT r5 task 2): rays change lanes at every phase switch --
slot-indexed stack columns in LDS, slot records in global memory, leaf / node / free queues in LDS, three node waves and a leaf wave per workgroup.
Measured slower than the production engine (profiles/r06/repack.txt) and therefore off by default; it stays built and tested: the images it renders
are the production engine's and the oracle's bit for bit, whatever lane steps a ray."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import CORNELL_OBJ, cam_array

pytestmark = pytest.mark.gpu


def frames(capi, ctx, model, W, H, cam, n, deferred, env=None):
    sc = capi.Scene(ctx)
    sc.add_model(capi.Model(ctx, *model))
    p = capi.Pipeline(ctx)
    p.set_scene(sc)
    mat = T.default_material()
    mat["type"] = 1
    mat["roughness"] = 0.4
    p.add_material(mat)
    if env is not None:
        p.set_environment_cube(env)
    p.create_output(W, H)
    p.build_acceleration_structures()
    p.set_deferred(deferred)
    host = capi.ProgressiveHost(11)
    pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(n)]
    for pfc in pfcs:
        p.update(pfc)
        p.render()
    return p.read_output(), p.totals(), mat, pfcs


@pytest.mark.parametrize("deferred", (0, 3))
def test_repacked_shadow_stage_is_bit_exact(capi, oracle, deferred):
    cases = [("cornell", oracle.obj_load(CORNELL_OBJ), 96, 80, cam_array(scenes.cornell_camera(), 96 / 80), None),
             ("atrium", scenes.sponza_class(seed=42, detail=0.5), 480, 270, cam_array(scenes.sponza_camera(), 480 / 270), scenes.sky_cubemap(16))]
    for name, model, W, H, cam, env in cases:
        plain = capi.Context(0)
        want, tot0, mat, pfcs = frames(capi, plain, model, W, H, cam, 3, deferred, env)
        plain.close()
        ctx = capi.Context(0)
        ctx.set_option("repack", 1)
        got, tot1, _, _ = frames(capi, ctx, model, W, H, cam, 3, deferred, env)
        st = ctx.repack_stats()
        ctx.close()
        assert st["watchdog_aborts"] == 0 and st["node_steps"] > 0 and st["rays_to_leaf_queue"] > 0, (name, st)
        assert np.array_equal(got, want), "%s: %d pixels differ" % (name, int((got != want).any(axis=2).sum()))
        for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
            assert tot0[k] == tot1[k], (name, k)
        if name == "cornell":          # ... and the oracle's
            osc = oracle.Scene()
            osc.add_instance(osc.add_model(*model))
            osc.build()
            acc = np.zeros((H, W, 4), np.float32)
            for pfc in pfcs:
                acc, _ = osc.render(mat, pfc, W, H, accum=acc, nthreads=4)
            assert np.array_equal(got, acc)
