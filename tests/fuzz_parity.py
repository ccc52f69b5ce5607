#!/usr/bin/env python3
"""Randomised parity sweep.  tests/test_gpu_fuzz.py runs a bounded number of draws of it under `pytest -m gpu`;
by hand, on a GPU box, for as long as wanted:

    python tests/fuzz_parity.py [iterations] [seed]

Every iteration draws a scene (triangle soups, blobs, displaced grids, Cornell; one identity instance or several
transformed ones), materials, debug options, depth limits, an image size and a tile, renders two frames with the
GPU pipeline (progressive or realtime) and with the CPU oracle, and demands bit-equal images and ray counts; progressive
draws then go on for one to five more frames through shared sets of launches -- rt_pipeline_render_batch, the deferred
pipeline behind update() + render(), or the bands of a tile partition rank after rank, with the queues sized for the worst
case or by count -- against the oracle's frame-by-frame accumulation.
Exits non-zero on the first difference and prints the draw that caused it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from dxrexperiments_amd import capi, rtypes as T, scenes  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from util import CORNELL_OBJ, cam_array, random_xforms, sliver_soup, triangle_soup  # noqa: E402

OPTION_FLAGS = ["cosineHemisphereSampling", "showIndirectDiffuseOnly", "showIndirectSpecularOnly", "showAmbientOcclusionOnly",
                "showGBufferAlbedoOnly", "showDirectLightingOnly", "showFresnelTerm", "noIndirectDiffuse"]


def draw(r):
    kind = r.integers(0, 5)
    if kind == 4:            # (round 5) long thin triangles: held as several references each, validated against their boxes (rt_refs.h)
        models = [sliver_soup(int(r.integers(4, 600)), seed=int(r.integers(1 << 30)))]
    elif kind == 0:
        models = [triangle_soup(int(r.integers(1, 4000)), seed=int(r.integers(1 << 30)), extent=4.0, size=float(r.uniform(0.1, 1.5)))]
    elif kind == 1:
        models = [scenes.blob_mesh(level=int(r.integers(1, 4))), triangle_soup(int(r.integers(10, 500)), seed=int(r.integers(1 << 30)), extent=1.5, size=0.4)]
    elif kind == 2:
        models = [scenes.displaced_grid(int(r.integers(4, 90)), seed=int(r.integers(1 << 30)))]
    else:
        models = [oracle.obj_load(CORNELL_OBJ)]
    if r.random() < 0.5:
        inst = [(0, None)]
    else:
        n = int(r.integers(1, 40))
        xf = random_xforms(n, seed=int(r.integers(1 << 30)), spread=float(r.uniform(1.0, 8.0)))
        inst = [(int(r.integers(len(models))), xf[k]) for k in range(n)]
        if r.random() < 0.3:
            inst.append((0, None))
    mats = []
    for _ in inst:
        m = T.default_material()
        m["albedo"][:3] = r.uniform(0.05, 0.95, 3)
        m["specular"][:3] = r.uniform(0.0, 1.0, 3)
        m["emissive"] = tuple(r.uniform(0, 1, 3)) + (float(r.uniform(0, 2)) if r.random() < 0.3 else 0.0,)
        m["roughness"] = r.uniform(0.0, 1.0)
        m["reflectivity"] = r.uniform(0.0, 1.0) if r.random() < 0.8 else 0.0
        m["type"] = int(r.integers(0, 3))
        mats.append(m)
    return models, inst, mats


def run(iters, seed, ctx, verbose=True):
    """Returns None when every draw was bit-exact, else a description of the first mismatch."""
    r = np.random.default_rng(seed)
    for it in range(iters):
        models, inst, mats = draw(r)
        W, H = int(r.integers(8, 200)), int(r.integers(8, 120))
        realtime = r.random() < 0.3
        depth = (int(r.integers(0, 5)), int(r.integers(0, 5)))
        env = scenes.sky_cubemap(int(r.choice([4, 8, 16]))) if r.random() < 0.5 else None
        seamless = bool(r.random() < 0.7)
        desc = dict(it=it, tris=[int(m[1].shape[0]) for m in models], instances=len(inst), size=(W, H), realtime=realtime, depth=depth,
                    seamless=seamless)
        sc = capi.Scene(ctx)
        gm = [capi.Model(ctx, v, i) for v, i in models]
        osc = oracle.Scene()
        for v, i in models:
            osc.add_model(v, i)
        for mi, x in inst:
            sc.add_model(gm[mi], x)
            osc.add_instance(mi, x)
        osc.build()
        p = capi.Pipeline(ctx, capi.PIPELINE_REALTIME if realtime else capi.PIPELINE_PROGRESSIVE)
        p.set_scene(sc)
        for m in mats:
            p.add_material(m)
        if env is not None:
            p.set_environment_cube(env)
        p.set_environment_filter(seamless)
        oracle.set_cube_seamless(seamless)
        p.create_output(W, H)
        p.build_acceleration_structures()
        p.set_depth_limits(*depth)
        host = capi.ProgressiveHost(int(r.integers(1 << 30)))
        if not realtime:
            for f in OPTION_FLAGS:
                host.options[f] = int(r.random() < 0.25)
            host.options["debug"] = int(r.integers(0, 3))
            host.options["environmentStrength"] = float(r.uniform(0.0, 2.0))
        # lights anywhere -- also inside the geometry -- or the reference's (the shadow cache is keyed by the light, the point
        # light's shadow rays end at its free sphere); in a set they may move from frame to frame
        lights = r.random() < 0.6
        moving = lights and r.random() < 0.4

        def relight(pfc):
            if lights:
                pfc["directionalLight"]["forwardDir"][:3] = r.normal(0, 1, 3)
                pfc["pointLight"]["worldPos"][:3] = r.uniform(-5, 5, 3)
            return pfc
        desc["lights"] = "moving" if moving else "random" if lights else "reference"
        eye = r.uniform(-6, 6, 3) + np.array([0, 2, 8.0])
        cam = cam_array(dict(eye=tuple(eye), at=tuple(r.uniform(-1, 1, 3)), up=(0, 1, 0), fov=float(r.uniform(0.4, 1.2))), W / H)
        omats = np.stack(mats)
        acc = np.zeros((H, W, 4), np.float32)
        # (round 6) the reference's accumulation storage for a fifth of the progressive draws: the running mean rounded to fp16 every frame
        f16 = 0 if realtime or r.random() >= 0.2 else int(r.integers(1, 3))
        if f16:
            p.set_accumulation_storage(T.FORMAT_R16G16B16A16_FLOAT, T.ROUND_NEAREST_EVEN if f16 == 1 else T.ROUND_TOWARD_ZERO)
        desc["accum_f16"] = f16
        for f in range(2):
            pfc = host.update_realtime(cam, 0.0, f + 1, W, H) if realtime else host.update(cam, 0.0, f + 1, W, H)
            if f == 0 or moving:
                lit = relight(pfc.copy())
            pfc["directionalLight"] = lit["directionalLight"]
            pfc["pointLight"] = lit["pointLight"]
            p.update(pfc)
            p.render()
            if realtime:
                d, ind, ost = osc.render_realtime(omats, pfc, W, H, env_faces=env, max_radiance_depth=depth[0], max_shadow_depth=depth[1], nthreads=8)
                ok = np.array_equal(p.read_output(0), d) and np.array_equal(p.read_output(1), ind)
            else:
                acc, ost = osc.render(omats, pfc, W, H, accum=acc, env_faces=env, max_radiance_depth=depth[0], max_shadow_depth=depth[1], nthreads=8, accum_f16=f16)
                ok = np.array_equal(p.read_output(), acc)
            gst = p.stats()
            same = all(gst[k] == ost[k] for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"))
            if not (ok and same):
                oracle.set_cube_seamless(True)
                return "MISMATCH %r frame %d image equal: %s gpu %r oracle %r" % (desc, f, ok, {k: gst[k] for k in ost if k in gst}, ost)
        if not realtime:               # the same accumulation continued by a batch of frames in one set of launches
            more = [host.update(cam, 0.0, 3 + k, W, H) for k in range(int(r.integers(1, 6)))]
            for pfc in more:
                if moving:
                    lit = relight(pfc.copy())
                pfc["directionalLight"] = lit["directionalLight"]
                pfc["pointLight"] = lit["pointLight"]
            # (round 4) ... submitted in one of the ways a set can come about: the explicit call, the reference's per-frame calls with
            # the pipeline in deferred mode (flushed by the read below), the ranks' bands of a tile partition one after the other
            # -- and, for half of the draws, with every radiance level's queues sized by count instead of the worst case
            how = int(r.integers(0, 4))
            counted = bool(r.random() < 0.5)
            desc["set"] = ("render_batch", "deferred", "bands", "deferred, small sets")[how] + (", counted queues" if counted else "")
            p.set_queue_budget(1 if counted else 0)
            if how == 0:
                p.render_batch(more)
            elif how == 2:
                world = int(r.integers(1, 6))
                for rank in range(world):
                    p.render_bands_batch(8, rank, world, more)
            else:
                p.set_deferred(32 if how == 1 else 2)
                for pfc in more:
                    p.update(pfc)
                    p.render()
            for pfc in more:
                acc, ost = osc.render(omats, pfc, W, H, accum=acc, env_faces=env, max_radiance_depth=depth[0], max_shadow_depth=depth[1], nthreads=8, accum_f16=f16)
            if not np.array_equal(p.read_output(), acc):
                oracle.set_cube_seamless(True)
                return "MISMATCH %r after a batch of %d frames" % (desc, len(more))
        if verbose and it % 10 == 0:
            print("ok", desc, flush=True)
    oracle.set_cube_seamless(True)
    return None


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = run(iters, seed, capi.Context(0))
    if bad:
        print(bad)
        sys.exit(1)
    print("fuzz parity: %d iterations, seed %d, all bit-exact" % (iters, seed))


if __name__ == "__main__":
    main()
