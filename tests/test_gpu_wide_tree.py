"""The PRODUCTION traversal layout (four-wide quantised nodes; eight-wide from a -DRT_WIDE=8 build, rt_bvh_wide.hip) read back through the C ABI and checked by
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
    off, boxes, rec_boxes = sc.refs(which)
    mark = recs.view(np.uint32)[:, 10]
    if off is None:
        assert recs.shape[0] == n and np.array_equal(np.sort(prim), np.arange(n, dtype=np.uint32)), "records are not a permutation of the triangles"
        assert not mark.any()
    else:
        # split references (round 5, rt_refs.h): a record per reference -- every triangle at least once, a split one once per box -- and a
        # record that is one of several is held by ITS box, which is what the child boxes above it must contain
        cnt = np.diff(off)
        single = recs.shape[0] == n                         # a layout that holds every triangle once (LBVH layout, option split_refs=0)
        assert np.array_equal(np.bincount(prim, minlength=n), np.ones(n, np.int64) if single else cnt), "records do not cover the references"
        assert np.array_equal(mark, np.where(cnt[prim] > 1, 2 if single else 1, 0))
        if not single:
            lo = np.where((mark == 1)[:, None], rec_boxes[:, :3], lo)
            hi = np.where((mark == 1)[:, None], rec_boxes[:, 3:], hi)
    tri = v["position"][i[prim]]                           # the record of primitive p holds p's vertices, bit for bit
    assert np.array_equal(recs[:, :9].reshape(-1, 3, 3), tri)
    st = W.check(nodes, root, lo, hi, recs.shape[0], blas=True)
    return st, nodes, root


@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 7, 64, 1000, 30000])
def test_soups(gpu, capi, n):
    v, i = triangle_soup(n, seed=100 + n)
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    st, nodes, root = check_blas(sc, 0, v, i)
    if n > 2:
        assert st["nodes"] >= 1 and st["nodes"] <= max(n, sc.wide_read(0)[2].shape[0]) - 1          # (records: triangles, or their references)


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
    assert st["children_per_node"] > (2.9 if capi.wide_layout()[0] == 4 else 3.9)        # the collapse fills its nodes
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
    assert st["nodes"] >= (75 if capi.wide_layout()[0] == 4 else 40)
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
    when the device-side state says the estimate was short.  The option build_batch forces batches of 1 / 3 rounds, so every
    build goes through that continuation many times; node numbering comes from prefix sums, so the tree must come out
    bit for bit the same as with one batch."""
    v, i = triangle_soup(40000, seed=4242)
    want_nodes, want_root, want_recs = build(capi, gpu, [(v, i)], [(0, None)]).wide_read(0)
    ctx2 = capi.Context(0)
    ctx2.set_option("build_batch", batch)
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
    top = 128 if capi.wide_layout()[0] == 4 else 80
    assert n > 4 * top
    r = np.random.default_rng(3)
    perm = np.arange(n)
    perm[top:] = top + r.permutation(n - top)               # old index -> new index
    out = np.empty_like(nodes)
    out[perm] = nodes
    code = out[:, W.code_columns(out)].view(np.int32)
    m = code >= 0
    code[m] = perm[code[m]]
    sc.wide_write(out)
    got = sc.trace(O, D)
    got_any = sc.trace(O, D, flags=ANY)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got_any["inst"] == 0xFFFFFFFF, want_any["inst"] == 0xFFFFFFFF)


def _same_hits(sc, O, D):
    """the production walk against the canonical-LBVH kernel on the same device: closest hits bit for bit, any-hit verdicts"""
    from util import ANY, assert_hits_equal
    assert_hits_equal(sc.trace(O, D), sc.trace(O, D, canonical=True), "fast vs canonical")
    assert_hits_equal(sc.trace(O, D, flags=ANY), sc.trace(O, D, flags=ANY, canonical=True), "any-hit", closest=False)


def test_axes_that_cannot_be_quantised_never_cull(gpu, capi):
    """a node whose extent overflows (or is not finite) has no byte grid that contains its children.  The builder
    then marks the axis as not quantised (infinite scale, q = 0: the planes decode to NaN, which every slab test ignores)
    instead of writing boxes that do not contain their subtree.  Huge-but-finite coordinates, +-inf and NaN vertices, and an
    instance whose world box spans +-3e38: the reader's containment check holds and the walk returns the canonical hits."""
    from util import random_rays
    v, i = triangle_soup(4000, seed=31)
    p = v["position"].reshape(-1, 3, 3)
    p[10, 0] = (3.0e38, 0.0, 0.0)                       # hi - lo of the root overflows on x
    p[11, 1] = (-3.0e38, 1.0, 2.0)
    p[12, 2] = (0.0, np.inf, 0.0)
    p[13, 0] = (np.nan, 0.5, 0.5)
    sc = build(capi, gpu, [(v, i)], [(0, None)])
    nodes, root, recs = sc.wide_read(0)
    lo, hi, prim = W.record_bounds(recs)
    pts = recs[:, :9].reshape(-1, 3, 3)
    lo, hi = np.fmin.reduce(pts, axis=1), np.fmax.reduce(pts, axis=1)       # (the builders' min / max ignore a NaN operand)
    with np.errstate(invalid="ignore"):
        st = W.check(nodes, root, lo, hi, i.shape[0], blas=True)
    d = st["decoded"]
    assert np.isinf(d["scale"]).any(), "no axis was marked as not quantised"
    O, D = random_rays(40000, 8, np.full(3, -10.0), np.full(3, 10.0))
    _same_hits(sc, O, D)
    # a TLAS over instances one of which is translated to the end of the float range
    small = triangle_soup(200, seed=32, extent=2.0, size=0.5)
    xf = random_xforms(40, seed=33)
    xf[7, 3] = 3.0e38
    xf[8, 7] = -3.0e38
    sc2 = build(capi, gpu, [small], [(0, xf[k]) for k in range(40)])
    tn, troot, _ = sc2.wide_read(-1)
    boxes = np.stack([sc2.instance_info(k)[0] for k in range(40)])
    with np.errstate(invalid="ignore", over="ignore"):
        W.check(tn, troot, boxes[:, :3], boxes[:, 3:], 40, blas=False)
    _same_hits(sc2, O, D)


def test_surface_area_collapse_option(gpu, capi):
    """The option wide_sah=1 (rt_bvh_wide.hip: the collapse that minimises the surface-area cost, Ylitie et al. 2017, instead of the
    area-greedy default) builds a different tree over the same triangles: same invariants, same hits bit for bit."""
    import os
    from util import random_rays
    v, i = triangle_soup(30000, seed=51)
    want = build(capi, gpu, [(v, i)], [(0, None)])
    O, D = random_rays(30000, 6, np.full(3, -10.0), np.full(3, 10.0))
    h0 = want.trace(O, D)
    ctx2 = capi.Context(0)
    ctx2.set_option("wide_sah", 1)
    sc = build(capi, ctx2, [(v, i)], [(0, None)])
    st, nodes, root = check_blas(sc, 0, v, i)
    n0 = want.wide_read(0)[0]
    assert nodes.shape != n0.shape or not np.array_equal(nodes, n0), "the option did not change the tree"
    from util import assert_hits_equal
    assert_hits_equal(sc.trace(O, D), h0, "SAH collapse vs default")
    _same_hits(sc, O, D)
    blob = scenes.blob_mesh(level=2)
    xf = random_xforms(150, seed=3)
    sc3 = build(capi, ctx2, [blob], [(0, xf[k]) for k in range(150)])
    tn, troot, _ = sc3.wide_read(-1)
    boxes = np.stack([sc3.instance_info(k)[0] for k in range(150)])
    W.check(tn, troot, boxes[:, :3], boxes[:, 3:], 150, blas=False)
    _same_hits(sc3, O, D)


def test_three_blases_two_of_them_lds_resident(gpu, capi):
    """RT_LDS_BLAS=1 (round 3; off by default: measured no faster): two-level kernels keep the tops of the TLAS and of the TWO
    most-instanced BLASes in LDS; a third model's instances walk their BLAS from global memory.  Same hits as the canonical
    kernel for all three, in both usage orders, and the same hits as with the option off."""
    import os
    from util import assert_hits_equal, random_rays
    models = [scenes.blob_mesh(level=2), triangle_soup(600, seed=21, extent=2.0, size=0.5), triangle_soup(3000, seed=22, extent=2.5, size=0.3)]
    xf = random_xforms(90, seed=23, spread=9.0)
    O, D = random_rays(40000, 9, np.full(3, -12.0), np.full(3, 12.0))
    for order in ((0, 1, 2), (2, 0, 1)):                 # which model is the rare one (10 of 90 instances)
        inst = [(order[0] if k % 9 else order[2], xf[k]) if k % 2 else (order[1] if k % 9 else order[2], xf[k]) for k in range(90)]
        plain = build(capi, gpu, models, inst).trace(O, D)
        os.environ["RT_LDS_BLAS"] = "1"                  # (read when the TLAS is built)
        try:
            sc = build(capi, gpu, models, inst)
        finally:
            del os.environ["RT_LDS_BLAS"]
        _same_hits(sc, O, D)
        assert_hits_equal(sc.trace(O, D), plain, "BLAS tops in LDS vs not")
