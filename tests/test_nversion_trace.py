"""TraceRay stated twice: tests/golden/nversion_trace.py is a numpy float32 restatement written from DESIGN.md section 2.1 (S2.1 - S2.6) alone -- every ray against every
triangle at once -- and the oracle (brute force AND BVH traversal) has to return the same bits: hit or miss, primitive, instance, t, u, v; and S2.5's reference boxes computed a
second time have to be the oracle's, bit for bit.  CPU only.  (The float64 truth of tests/test_s2_truth.py says the definition follows geometry; this says the code follows the
definition.)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import nversion_trace as NV  # noqa: E402
from dxrexperiments_amd import scenes  # noqa: E402
from util import CORNELL_OBJ, random_rays, random_xforms, sliver_soup, triangle_soup  # noqa: E402

CULL, ANY = 0x10, 0x4 | 0x8


def both(oracle, models, instances, O, D, refs=None):
    sc = oracle.Scene()
    for v, t in models:
        sc.add_model(v, t)
    for mi, x in instances:
        sc.add_instance(mi, x)
    sc.build()
    nv_models = [(v["position"], t) for v, t in models]
    for flags in (0, CULL):
        want = NV.trace(nv_models, instances, O, D, flags, refs)
        for mode in (0, 1):                                   # brute force, BVH traversal
            got = sc.trace(O, D, flags, mode=mode, nthreads=4)
            assert np.array_equal(got["inst"], want["inst"]), ("hit / instance", flags, mode, int((got["inst"] != want["inst"]).sum()))
            assert np.array_equal(got["prim"], want["prim"]), ("primitive", flags, mode, int((got["prim"] != want["prim"]).sum()))
            h = want["inst"] != 0xFFFFFFFF
            for k in ("t", "u", "v"):
                a, b = got[k][h], want[k][h]
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b), (k, flags, mode)
    want = NV.trace(nv_models, instances, O, D, ANY, refs)
    got = sc.trace(O, D, ANY, mode=0, nthreads=4)
    assert np.array_equal(got["inst"] != 0xFFFFFFFF, want["inst"] != 0xFFFFFFFF), "any-hit"
    return sc


def test_cornell_and_a_soup(oracle):
    v, t = oracle.obj_load(CORNELL_OBJ)
    O, D = random_rays(4000, 5, [-1, -1, -1], [1, 1, 1])
    D[::7, 1] = 0.0                                           # rays inside axis-aligned planes: the flat boxes of the walls
    O[::14, 1] = -1.0
    both(oracle, [(v, t)], [(0, None)], O, D)
    O, D = random_rays(3000, 6, [-4, -4, -4], [4, 4, 4])
    r = np.random.default_rng(2)
    O[:, 3] = r.uniform(0, 2, O.shape[0])
    D[:, 3] = O[:, 3] + r.uniform(-0.5, 12, O.shape[0])      # windows, some empty
    both(oracle, [triangle_soup(700, seed=3, extent=4.0, size=0.8)], [(0, None)], O, D)


def test_instances(oracle):
    xf = random_xforms(7, seed=4, spread=3.0)
    models = [scenes.blob_mesh(seed=3, level=1), triangle_soup(60, seed=8, extent=1.2, size=0.5)]
    inst = [(k % 2, xf[k]) for k in range(7)] + [(1, None)]
    O, D = random_rays(2500, 9, [-5, -5, -5], [5, 5, 5])
    both(oracle, models, inst, O, D)


def test_split_references_boxes_and_hits(oracle):
    v, t = sliver_soup(120, seed=31)
    sc0 = oracle.Scene()
    sc0.add_instance(sc0.add_model(v, t))
    sc0.build()
    off, boxes = sc0.refs(0, t.shape[0])
    assert off is not None
    off2, boxes2 = NV.reference_boxes(v["position"], t)       # S2.5 a second time
    assert np.array_equal(off, off2), "reference counts differ"
    assert np.array_equal(boxes.view(np.uint32), boxes2.view(np.uint32)) or np.array_equal(boxes, boxes2), int((boxes != boxes2).any(1).sum())
    # rays aimed at the slivers (where the box clause's second case lives) and random ones
    P = v["position"][t[:120]].astype(np.float64)
    r = np.random.default_rng(12)
    pick = r.integers(0, 120, 1500)
    b = r.uniform(0, 1, (1500, 2)); f = b.sum(1) > 1; b[f] = 1 - b[f]
    target = P[pick, 0] + b[:, :1] * (P[pick, 1] - P[pick, 0]) + b[:, 1:] * (P[pick, 2] - P[pick, 0])
    origin = r.uniform(-5, 5, (1500, 3))
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    O = np.concatenate([origin, np.full((1500, 1), 1e-3)], 1).astype(np.float32)
    D = np.concatenate([d, np.full((1500, 1), 1e30)], 1).astype(np.float32)
    O2, D2 = random_rays(1500, 13, [-5, -5, -5], [5, 5, 5])
    both(oracle, [(v, t)], [(0, None)], np.concatenate([O, O2]), np.concatenate([D, D2]), refs={0: (off2, boxes2)})


def test_canonical_lbvh_stated_twice(oracle):
    """the canonical acceleration structure ("LBVH"): Morton keys, the Karras radix tree, exact refit -- tests/golden/nversion_lbvh.py against the oracle's
    arrays, BLAS (triangle boxes) and TLAS (instance world boxes), field for field"""
    import nversion_lbvh as LB
    xf = random_xforms(9, seed=21, spread=4.0)
    for (v, t), inst in ((oracle.obj_load(CORNELL_OBJ), [(0, None)]), (triangle_soup(900, seed=17, extent=3.0, size=0.4), [(0, None)]),
                         (scenes.blob_mesh(seed=3, level=1), [(0, x) for x in xf])):
        sc = oracle.Scene()
        sc.add_model(v, t)
        for mi, x in inst:
            sc.add_instance(mi, x)
        sc.build()
        P = v["position"][np.asarray(t).reshape(-1, 3)]
        for which in (0, -1):
            lo, hi = P.min(1), P.max(1)
            if which == -1:
                if len(inst) < 2:
                    continue
                boxes = [sc.instance_info(i)[0] for i in range(len(inst))]       # (the world boxes are S2.4's: restated in nversion_trace.trace)
                lo, hi = np.array([b[:3] for b in boxes], np.float32), np.array([b[3:] for b in boxes], np.float32)
                for i, (mi, x) in enumerate(inst):                                  # ... and checked here against the transformed vertices
                    wp = NV.xform_point(np.asarray(x, np.float32).reshape(3, 4), v["position"][np.asarray(t).reshape(-1)])
                    assert np.array_equal(wp.min(0), lo[i]) and np.array_equal(wp.max(0), hi[i])
            nodes, keys, parents, _ = sc.bvh(which)
            mine = LB.build(lo, hi)
            assert np.array_equal(keys, mine["keys"])
            assert np.array_equal(nodes["left"], mine["left"]) and np.array_equal(nodes["right"], mine["right"])
            assert np.array_equal(nodes["bmin"], mine["bmin"]) and np.array_equal(nodes["bmax"], mine["bmax"])
            assert np.array_equal(parents, mine["parent"])
