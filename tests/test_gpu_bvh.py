"""GPU-built acceleration structures == the oracle's, index for index."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import CORNELL_OBJ, Pair, nodes_equal, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def check_bvh(pair, which, oracle_model=None):
    """which: -1 = TLAS, else instance index on the GPU side (its BLAS); the oracle indexes BLASes by model."""
    gn, gk, gp, gd = pair.g.bvh(which)
    on, ok, op, od = pair.o.bvh(which if oracle_model is None else oracle_model)
    assert np.array_equal(gk, ok), "sorted morton keys differ"
    assert nodes_equal(gn, on), "node arrays differ"
    assert np.array_equal(gp, op), "parent arrays differ"
    assert gd == od, "max depth differs"


def test_cornell_blas_tlas(gpu, oracle, capi):
    v, i = oracle.obj_load(CORNELL_OBJ)
    m = capi.Model(gpu, path=CORNELL_OBJ)
    gv, gi = m.geometry()
    assert np.array_equal(gv, v) and np.array_equal(gi, i), "OBJ ingestion differs from the oracle"
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    check_bvh(p, 0)
    check_bvh(p, -1)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 1000, 100000])
def test_soup_sizes(gpu, oracle, capi, n):
    p = Pair(oracle, capi, gpu, [triangle_soup(n, seed=n)], [(0, None)])
    check_bvh(p, 0)


def test_degenerate_and_duplicate_triangles(gpu, oracle, capi):
    v, i = triangle_soup(64, seed=5)
    tri = v["position"].reshape(-1, 3, 3)               # view: one row per triangle
    tri[1:32] = tri[0]                                  # 31 clones of triangle 0 (identical morton codes)
    tri[32] = tri[32, 0]                                # zero-area triangle
    tri[33, :, 1] = 0.0                                 # flat (zero-thickness) box
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    check_bvh(p, 0)


def test_instanced_scene(gpu, oracle, capi):
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(500, seed=2, extent=2.0, size=0.4)
    xf = random_xforms(37, seed=3)
    inst = [(k % 2, xf[k]) for k in range(37)] + [(0, None)]
    p = Pair(oracle, capi, gpu, [blob, soup], inst)
    check_bvh(p, -1)
    for k in (0, 1, 37):
        check_bvh(p, k, oracle_model=inst[k][0])
        gb, gi = p.g.instance_info(k)
        ob, oi = p.o.instance_info(k)
        assert np.array_equal(gb, ob) and np.array_equal(gi, oi)


def world_box_of_vertices(v, idx, m):
    """the definition (oracle_bvh.h scene_build): exact box of the triangles' transformed vertices, x' = ((m0 x + m1 y) + m2 z) + m3 in fp32"""
    p = v["position"][np.asarray(idx).reshape(-1)].astype(np.float32)
    m = np.asarray(m, np.float32).reshape(3, 4)
    w = np.stack([((m[r, 0] * p[:, 0] + m[r, 1] * p[:, 1]) + m[r, 2] * p[:, 2]) + m[r, 3] for r in range(3)], axis=1)
    return np.concatenate([w.min(axis=0), w.max(axis=0)])


def test_instance_world_boxes_are_the_boxes_of_the_transformed_vertices(gpu, oracle, capi):
    """k_instance_boxes: meshes of one work item (3840 vertex references), of four (15360) and of a single triangle, plus an unreferenced
    far-away vertex that must not count; GPU == oracle == the numpy statement of the definition, and tighter than the eight corners"""
    small, big = scenes.blob_mesh(level=3), scenes.blob_mesh(seed=5, level=4)
    one = triangle_soup(1, seed=9)
    bv = np.concatenate([big[0], big[0][:1]])
    bv["position"][-1] = (1e6, -1e6, 1e6)
    big = (bv, big[1])
    xf = random_xforms(24, seed=11)
    models = [small, big, one]
    inst = [(k % 3, xf[k]) for k in range(24)] + [(1, None)]
    p = Pair(oracle, capi, gpu, models, inst)
    check_bvh(p, -1)
    looser = 0
    for k in range(24):
        gb, gi = p.g.instance_info(k)
        ob, oi = p.o.instance_info(k)
        want = world_box_of_vertices(*models[k % 3], xf[k])
        assert np.array_equal(gb, ob) and np.array_equal(gi, oi) and np.array_equal(gb, want), k
        lo, hi = models[k % 3][0]["position"][np.asarray(models[k % 3][1]).reshape(-1)].min(axis=0), models[k % 3][0]["position"][np.asarray(models[k % 3][1]).reshape(-1)].max(axis=0)
        corners = np.array([[(hi if c & 1 else lo)[0], (hi if c & 2 else lo)[1], (hi if c & 4 else lo)[2]] for c in range(8)], np.float64)
        m = xf[k].reshape(3, 4).astype(np.float64)
        cw = corners @ m[:, :3].T + m[:, 3]
        assert (gb[:3] >= cw.min(axis=0) - 1e-4).all() and (gb[3:] <= cw.max(axis=0) + 1e-4).all()
        looser += int(np.prod(cw.max(axis=0) - cw.min(axis=0)) > 1.2 * np.prod((gb[3:] - gb[:3]).astype(np.float64)))
    assert looser >= 8              # the blobs: a rotated box's box is much wider than the rotated mesh's
    gb, _ = p.g.instance_info(24)   # the identity instance keeps the box of its BLAS
    nodes = p.o.bvh(1)[0]
    assert np.array_equal(gb[:3], nodes["bmin"][0]) and np.array_equal(gb[3:], nodes["bmax"][0])


def test_sponza_class_full(gpu, oracle, capi):
    v, i = scenes.sponza_class()
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    check_bvh(p, 0)
    assert p.g.build_ms() > 0
