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


def test_sponza_class_full(gpu, oracle, capi):
    v, i = scenes.sponza_class()
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    check_bvh(p, 0)
    assert p.g.build_ms() > 0
