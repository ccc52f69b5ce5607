"""The PRODUCTION traversal layout (four-wide quantised nodes, rt_bvh_wide.hip) read back through the C ABI and checked by
an independent numpy reader (tests/wide_tree.py): every primitive is in exactly one leaf, every decoded child box contains
its subtree (the premise of the exactness rule), breadth-first numbering, power-of-two grids.  The
parity tests show that images do not depend on this tree; these show that the tree is what the design says it is."""
import numpy as np
import pytest

import wide_tree as W
from dxrexperiments_amd import scenes
from util import random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def build(capi, gpu, models, instances):
    sc = capi.Scene(gpu)
    ms = [capi.Model(gpu, v, i) for v, i in models]
    for k, xf in instances:
        sc.add_model(ms[k], xf)
    sc.build()
    return sc


def check_blas(sc, which, v, i):
    nodes, root, recs = sc.wide_read(which)
    lo, hi, prim = W.record_bounds(recs)
    n = i.shape[0]
    assert recs.shape[0] == n and np.array_equal(np.sort(prim), np.arange(n, dtype=np.uint32)), "records are not a permutation of the triangles"
    tri = v["position"][i[prim]]                           # the record of primitive p holds p's vertices, bit for bit
    assert np.array_equal(recs[:, :9].reshape(-1, 3, 3), tri)
    st = W.check(nodes, root, lo, hi, n, blas=True)
    return st, nodes, root


@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 7, 64, 1000, 30000])
def test_soups(gpu, capi, n):
    v, i = triangle_soup(n, seed=100 + n)
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    st, nodes, root = check_blas(sc, 0, v, i)
    if n > 2:
        assert st["nodes"] >= 1 and st["nodes"] <= n - 1


def test_duplicates_and_flat_boxes(gpu, capi):
    v, i = triangle_soup(4096, seed=5)
    tri = v["position"].reshape(-1, 3, 3)
    tri[1:600] = tri[0]                                 # 599 clones: identical boxes, identical keys
    tri[600:700, :, 1] = 0.25                           # a sheet of zero-thickness boxes
    tri[700] = tri[700, 0]                              # a point
    tri[701:710] *= np.float32(1e-30)                   # denormal-sized extents around the origin
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    check_blas(sc, 0, v, i)


def test_bench_scene_and_its_cost(gpu, capi):
    v, i = scenes.sponza_class()
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    st, nodes, root = check_blas(sc, 0, v, i)
    assert st["children_per_node"] > 3.0                # the collapse fills its nodes
    node_term, item_term = W.sah(nodes, root)
    # surface-area cost of the tree the bench walks (round 2: ~37 node steps + ~9 triangle tests for a random ray through the
    # root box); a builder change that doubles it is a bug even if every image stays bit-exact
    assert node_term < 80 and item_term < 30, (node_term, item_term)


def test_tlas_and_instanced_blas(gpu, capi):
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(500, seed=2, extent=2.0, size=0.4)
    xf = random_xforms(301, seed=3)
    inst = [(k % 2, xf[k]) for k in range(301)]
    sc = build(capi, gpu, [blob, soup], inst)
    nodes, root, recs = sc.wide_read(-1)
    assert recs.shape[0] == 0
    boxes = np.stack([sc.instance_info(k)[0] for k in range(301)])
    st = W.check(nodes, root, boxes[:, :3], boxes[:, 3:], 301, blas=False)
    assert st["nodes"] >= 75
    for k in (0, 1):
        check_blas(sc, k, *(blob, soup)[k])


def test_single_instance_tlas(gpu, capi):
    v, i = triangle_soup(10, seed=1)
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    nodes, root, recs = sc.wide_read(-1)
    assert nodes.shape[0] == 0 and root == ~0


@pytest.mark.parametrize("batch", ["1", "3"])
def test_short_build_batches_give_the_same_tree(gpu, capi, batch):
    """The builders launch PLOC rounds and collapse levels in batches sized by an estimate and continue with another batch
    when the device-side state says the estimate was short.  RT_BUILD_BATCH forces batches of 1 / 3 rounds, so every
    build goes through that continuation many times; node numbering comes from prefix sums, so the tree must come out
    bit for bit the same as with one batch."""
    import os
    v, i = triangle_soup(40000, seed=4242)
    want_nodes, want_root, want_recs = build(capi, gpu, [(v, i)], [(0, None)]).wide_read(0)
    os.environ["RT_BUILD_BATCH"] = batch
    try:
        ctx2 = capi.Context(0)
    finally:
        del os.environ["RT_BUILD_BATCH"]
    sc = build(capi, ctx2, [(v, i)], [(0, None)])
    nodes, root, recs = sc.wide_read(0)
    assert root == want_root and np.array_equal(nodes, want_nodes) and np.array_equal(recs.view(np.uint32), want_recs.view(np.uint32))
    tn, tr, _ = sc.wide_read(-1)
    assert tn.shape[0] == 0 and tr == ~0


def test_node_numbers_do_not_matter(gpu, capi):
    """Results depend on the set of boxes and triangles a ray meets, not on where a node lives in the array.
    The tree is read back, renumbered at random behind the LDS-resident top (child codes rewritten accordingly), written
    back through rt_debug_wide_write, and the same rays must return the same hits, bit for bit."""
    from util import ANY, random_rays
    v, i = triangle_soup(20000, seed=77)
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    O, D = random_rays(50000, 5, np.full(3, -10.0), np.full(3, 10.0))
    want = sc.trace(O, D)
    want_any = sc.trace(O, D, flags=ANY)
    nodes, root, _ = sc.wide_read(0)
    n = nodes.shape[0]
    top = 128
    assert n > 4 * top
    r = np.random.default_rng(3)
    perm = np.arange(n)
    perm[top:] = top + r.permutation(n - top)               # old index -> new index
    out = np.empty_like(nodes)
    out[perm] = nodes
    code = out[:, 12:16].view(np.int32)
    m = code >= 0
    code[m] = perm[code[m]]
    sc.wide_write(out)
    got = sc.trace(O, D)
    got_any = sc.trace(O, D, flags=ANY)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got_any["inst"] == 0xFFFFFFFF, want_any["inst"] == 0xFFFFFFFF)
