"""Independent reader of the production traversal layout (rt_scene_wide_read): decodes the wide quantised nodes with
numpy, checks the invariants the traversal relies on, and prices the tree with the surface-area heuristic.

Nothing here is shared with the builder (rt_bvh_wide.hip) or the traversal (rt_trace_wave.h): the layout is taken from
the comment in include/dxr_amd.h.  Used by tests/test_gpu_wide_tree.py and tools/tree_quality.py."""
import numpy as np

NONE = -2 ** 31


def decode(nodes):
    """nodes: uint32[n, 16] (four-wide, 64 B) or uint32[n, 32] (eight-wide, 128 B; a -DRT_WIDE=8 build) ->
    dict(lo float32[n, W, 3], hi float32[n, W, 3] (NaN: the axis bounds nothing), code int32[n, W], scale float32[n, 3],
    width W; eight-wide only: valid bool[n, 8], first_child, internal_mask)."""
    return decode8(nodes) if nodes.shape[1] == 32 else decode4(nodes)


def decode4(nodes):
    f = nodes.view(np.float32)
    origin = f[:, 0:3]                                       # q0.xyz
    scale = np.stack([f[:, 3], f[:, 10], f[:, 11]], axis=1)  # q0.w, q2.z, q2.w
    words = {"lo": (nodes[:, 4], nodes[:, 6], nodes[:, 8]), "hi": (nodes[:, 5], nodes[:, 7], nodes[:, 9])}
    out = {"width": 4}
    for name, (wx, wy, wz) in words.items():
        planes = np.empty((nodes.shape[0], 4, 3), np.float32)
        for a, w in enumerate((wx, wy, wz)):
            for k in range(4):
                q = ((w >> np.uint32(8 * k)) & np.uint32(0xFF)).astype(np.float32)
                # plane = fma(q, scale, origin); q <= 255 times a power of two is exact, so mul + add rounds once as well
                # (an unquantised axis: 0 * inf = NaN)
                with np.errstate(invalid="ignore"):
                    planes[:, k, a] = q * scale[:, a] + origin[:, a]
        out[name] = planes
    out["code"] = nodes[:, 12:16].view(np.int32).copy()
    out["scale"] = scale
    return out


def decode8(nodes):
    W = 8
    f = nodes.view(np.float32)
    origin = f[:, 0:3]
    meta = nodes[:, 3]
    expo = np.stack([meta & 0xFF, (meta >> 8) & 0xFF, (meta >> 16) & 0xFF], axis=1).astype(np.int64)
    with np.errstate(over="ignore"):
        scale = np.where(expo == 255, np.float32(np.inf), np.ldexp(np.float32(1.0), (expo - 127).astype(np.int32))).astype(np.float32)
    valid = ((meta >> 24)[:, None] >> np.arange(W, dtype=np.uint32)[None, :]) & 1 == 1
    out = {"width": 8}
    for name, first in (("lo", 4), ("hi", 6)):
        planes = np.empty((nodes.shape[0], W, 3), np.float32)
        for a in range(3):
            for k in range(W):
                w = nodes[:, first + 4 * a + k // 4]
                q = ((w >> np.uint32(8 * (k % 4))) & np.uint32(0xFF)).astype(np.float32)
                with np.errstate(invalid="ignore"):
                    planes[:, k, a] = q * scale[:, a] + origin[:, a]
        out[name] = planes
    out["code"] = nodes[:, 16:24].view(np.int32).copy()
    out["valid"] = valid
    out["scale"] = scale
    out["first_child"] = nodes[:, 24].astype(np.int64)
    out["internal_mask"] = nodes[:, 25]
    return out


def code_columns(nodes):
    """the slice of node words that holds the child codes"""
    return slice(16, 24) if nodes.shape[1] == 32 else slice(12, 16)


def record_bounds(recs):
    """recs float32[m, 12] (p0 p1 p2 prim pad pad) -> (lo[m, 3], hi[m, 3], prim uint32[m])."""
    p = recs[:, :9].reshape(-1, 3, 3)
    return p.min(axis=1), p.max(axis=1), recs[:, 9].copy().view(np.uint32)


def levels_of(code):
    """breadth-first level of every node (root = node 0), and the parent count of every node."""
    n, WIDE = code.shape
    level = np.full(n, -1, np.int64)
    refs = np.zeros(n, np.int64)
    level[0] = 0
    frontier = np.array([0])
    while frontier.size:
        kids = code[frontier]
        lv = np.repeat(level[frontier], WIDE).reshape(-1, WIDE)
        m = kids >= 0
        ids = kids[m]
        np.add.at(refs, ids, 1)
        level[ids] = lv[m] + 1
        frontier = ids
    return level, refs


def check(nodes, root_code, leaf_lo, leaf_hi, n_leaf_items, blas=True):
    """Invariants of a wide tree.  leaf_lo / leaf_hi: true bounds per record (BLAS) or per instance (TLAS).
    Returns a dict of statistics; raises AssertionError with a description on the first violation."""
    n = nodes.shape[0]
    if root_code < 0:
        assert n == 0, "a one-leaf structure has no nodes"
        code = ~root_code
        if blas:
            first, cnt = code >> 3, (code & 7) + 1
            assert first == 0 and cnt == n_leaf_items, "root leaf does not cover the structure"
        else:
            assert code == 0 and n_leaf_items == 1
        return {"nodes": 0, "levels": 0}
    d = decode(nodes)
    code = d["code"]
    WIDE = d["width"]
    used = code != NONE
    assert np.all(used.sum(axis=1) >= 2), "a node with fewer than two children"
    if WIDE == 4:
        assert np.all(used[:, :-1] | ~used[:, 1:]), "children not packed at the front"
    else:
        assert np.array_equal(used, d["valid"]), "valid mask and child codes disagree"
    internal = used & (code >= 0)
    leaf = used & (code < 0)
    assert np.all(code[internal] < n) and np.all(code[internal] > 0), "child index out of range"
    level, refs = levels_of(code)
    assert np.all(level >= 0), "unreachable node"
    assert refs[0] == 0 and np.all(refs[1:] == 1), "a node with more than one parent"
    # breadth-first numbering: levels are contiguous and ascending (the first nodes are the LDS-resident top)
    assert np.all(np.diff(level) >= 0), "nodes are not in breadth-first order"
    if WIDE == 8:
        # the internal children of a node are consecutive, in slot order (the inspection words say where they start)
        imask = (internal.astype(np.uint32) << np.arange(WIDE, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
        assert np.array_equal(imask, d["internal_mask"]), "internal-slot mask"
        rank = np.cumsum(internal, axis=1) - 1
        has = internal.any(axis=1)
        assert np.all((code == (d["first_child"][:, None] + rank))[internal]), "internal children are not consecutive in slot order"
        assert np.all(d["first_child"][~has] == 0)
    # leaves cover every item exactly once
    covered = np.zeros(n_leaf_items, np.int64)
    lc = ~code[leaf]
    if blas:
        first, cnt = lc >> 3, (lc & 7) + 1
        assert np.all(first + cnt <= n_leaf_items), "leaf range out of bounds"
        for k in range(8):
            m = cnt > k
            np.add.at(covered, first[m] + k, 1)
    else:
        assert np.all(lc < n_leaf_items)
        np.add.at(covered, lc, 1)
    assert np.all(covered == 1), "%d items not covered exactly once" % int((covered != 1).sum())
    # true bounds of every child, bottom up
    inf = np.float32(np.inf)
    tlo = np.full((n, WIDE, 3), inf, np.float32)
    thi = np.full((n, WIDE, 3), -inf, np.float32)
    ni, ki = np.nonzero(leaf)
    lc = ~code[ni, ki]
    if blas:
        first, cnt = lc >> 3, (lc & 7) + 1
        for k in range(8):
            m = cnt > k
            tlo[ni[m], ki[m]] = np.minimum(tlo[ni[m], ki[m]], leaf_lo[first[m] + k])
            thi[ni[m], ki[m]] = np.maximum(thi[ni[m], ki[m]], leaf_hi[first[m] + k])
    else:
        tlo[ni, ki] = leaf_lo[lc]
        thi[ni, ki] = leaf_hi[lc]
    for lv in range(int(level.max()), -1, -1):
        at = np.nonzero(level == lv)[0]
        c = code[at]
        m = c >= 0
        a_i, k_i = np.nonzero(m)
        kid = c[m]
        tlo[at[a_i], k_i] = tlo[kid].min(axis=1)      # unused slots hold +-inf: neutral
        thi[at[a_i], k_i] = thi[kid].max(axis=1)
    # containment: the decoded box of every child holds everything below it (the exactness rule's premise)
    # (a NaN plane belongs to an axis the builder could not quantise: it bounds nothing and the traversal ignores it)
    with np.errstate(invalid="ignore"):
        ok = ((d["lo"] <= tlo) | np.isnan(d["lo"])) & ((d["hi"] >= thi) | np.isnan(d["hi"]))
    bad = used[:, :, None] & ~ok
    assert not bad.any(), "decoded child box does not contain its subtree at node %d" % int(np.nonzero(bad.any(axis=(1, 2)))[0][0])
    # quantisation grid: power-of-two scales
    sc = d["scale"]
    mant, _ = np.frexp(sc[np.isfinite(sc)])
    assert np.all(mant == 0.5), "a scale that is not a power of two"
    return {"nodes": n, "levels": int(level.max()) + 1, "children_per_node": float(used.sum()) / n,
            "leaf_children": int(leaf.sum()), "decoded": d, "true_lo": tlo, "true_hi": thi, "level": level}


def half_area(lo, hi):
    e = np.maximum(hi.astype(np.float64) - lo.astype(np.float64), 0.0)
    return e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]


def sah(nodes, root_code, c_node=1.0, c_item=1.0):
    """Surface-area cost of the tree as the traversal sees it (decoded boxes): expected node steps and item tests of a
    random ray that hits the root box.  Returns (node_term, item_term)."""
    if root_code < 0 or nodes.shape[0] == 0:
        return 0.0, 1.0
    d = decode(nodes)
    code = d["code"]
    used = code != NONE
    area = np.where(used, half_area(d["lo"], d["hi"]), 0.0)
    root = half_area(d["lo"][0][used[0]].min(axis=0), d["hi"][0][used[0]].max(axis=0))
    internal = used & (code >= 0)
    leaf = used & (code < 0)
    cnt = np.where(leaf, ((~code) & 7) + 1, 0)
    node_term = c_node * (1.0 + float(area[internal].sum()) / root)
    item_term = c_item * float((area * cnt).sum()) / root
    return node_term, item_term
