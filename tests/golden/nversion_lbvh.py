"""The canonical LBVH stated a second time, from DESIGN.md section 2 ("LBVH"): exact primitive boxes, 30-bit Morton code of the box centre in the structure's box (10 bits per
axis, x high), key = code << 32 | index, ascending sort, the Karras-2012 radix tree over the keys, bottom-up exact union.  Canonical node array: internal nodes 0 .. N-2 (root 0),
leaves N-1 .. 2N-2 in key order.  Pure Python / numpy float32; tests/test_nversion_trace.py compares it with the oracle's arrays field for field."""
import numpy as np

f32 = np.float32
LEAF = 0xFFFFFFFF


def _expand10(v):
    v &= 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def _quant(c, lo, ext):
    if not ext > 0:
        return 0
    q = ((c - lo) / ext) * f32(1024.0)
    q = min(max(q, f32(0.0)), f32(1023.0))
    return int(q)


def keys_of(lo, hi):
    """lo, hi [n,3] float32 primitive boxes -> sorted uint64 keys, the structure's box"""
    blo, bhi = lo.min(0), hi.max(0)
    ext = bhi - blo
    keys = []
    for i in range(lo.shape[0]):
        c = (lo[i] + hi[i]) * f32(0.5)
        q = [_quant(c[a], blo[a], ext[a]) for a in range(3)]
        m = (_expand10(q[0]) << 2) | (_expand10(q[1]) << 1) | _expand10(q[2])
        keys.append((m << 32) | i)
    return np.array(sorted(keys), np.uint64), blo, bhi


def _delta(k, n, i, j):
    if j < 0 or j >= n:
        return -1
    x = int(k[i]) ^ int(k[j])
    return 64 - x.bit_length()


def build(lo, hi):
    """-> dict(left, right [2n-1], bmin, bmax [2n-1,3], keys [n], parent [2n-1])"""
    lo, hi = np.asarray(lo, f32), np.asarray(hi, f32)
    n = lo.shape[0]
    k, _, _ = keys_of(lo, hi)
    nn = 2 * n - 1
    left = np.zeros(nn, np.uint32); right = np.zeros(nn, np.uint32); parent = np.full(nn, LEAF, np.uint32)
    bmin = np.zeros((nn, 3), f32); bmax = np.zeros((nn, 3), f32)
    leaf0 = n - 1
    for j in range(n):
        p = int(k[j] & 0xFFFFFFFF)
        left[leaf0 + j], right[leaf0 + j] = p, LEAF
        bmin[leaf0 + j], bmax[leaf0 + j] = lo[p], hi[p]
    for i in range(n - 1):
        d = 1 if _delta(k, n, i, i + 1) - _delta(k, n, i, i - 1) > 0 else -1
        dmin = _delta(k, n, i, i - d)
        lmax = 2
        while _delta(k, n, i, i + lmax * d) > dmin:
            lmax *= 2
        l, t = 0, lmax // 2
        while t >= 1:
            if _delta(k, n, i, i + (l + t) * d) > dmin:
                l += t
            t //= 2
        j = i + l * d
        dnode = _delta(k, n, i, j)
        s, t = 0, l
        while True:
            t = (t + 1) >> 1
            if _delta(k, n, i, i + (s + t) * d) > dnode:
                s += t
            if t <= 1:
                break
        gamma = i + s * d + min(d, 0)
        a = leaf0 + gamma if min(i, j) == gamma else gamma
        b = leaf0 + gamma + 1 if max(i, j) == gamma + 1 else gamma + 1
        left[i], right[i] = a, b
        parent[a] = parent[b] = i

    def refit(x):
        if right[x] == LEAF:
            return
        refit(left[x]); refit(right[x])
        bmin[x] = np.minimum(bmin[left[x]], bmin[right[x]])
        bmax[x] = np.maximum(bmax[left[x]], bmax[right[x]])
    import sys
    sys.setrecursionlimit(10000)
    if n > 1:
        refit(0)
    return dict(left=left, right=right, bmin=bmin, bmax=bmax, keys=k, parent=parent)
