"""A second, independent restatement of TraceRay -- written from DESIGN.md section 2.1 (paragraphs S2.1 - S2.6), not from oracle_bvh.h or the kernels: numpy float32, every
ray against every triangle of every instance at once, operation for operation in the order the page gives.  tests/test_nversion_trace.py demands the oracle's brute force and
its BVH traversal return the SAME BITS.  (oracle/truth64.h pins the definition to geometry; this pins the oracle's code to the definition's text.)

Everything is float32: numpy's +, -, *, /, sqrt on float32 arrays are the IEEE single operations, nothing is fused; np.fmin / np.fmax ignore a NaN operand as S2.2 asks."""
import numpy as np

f32 = np.float32
SLACK = f32(1.0000152587890625)            # 1 + 2^-16
NO_HIT = np.uint32(0xFFFFFFFF)
CULL, FIRST = 0x10, 0x4


def _dot(ax, ay, az, bx, by, bz):
    s = ax * bx
    s = s + ay * by
    return s + az * bz


def _cross(ax, ay, az, bx, by, bz):
    return ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx


def slab(o, inv, lo, hi, t0, t1):
    """S2.2: o, inv [N,1,3]; lo, hi [1,M,3] (or [N,M,3]); t0, t1 broadcastable to [N,M]  ->  (passes, entry)"""
    with np.errstate(invalid="ignore", over="ignore"):
        a = (lo - o) * inv
        b = (hi - o) * inv
    near = np.fmin(a, b)
    far = np.fmax(a, b)
    entry = np.fmax(np.fmax(near[..., 0], near[..., 1]), np.fmax(near[..., 2], t0))
    exit_ = np.fmin(np.fmin(far[..., 0], far[..., 1]), np.fmin(far[..., 2], t1))
    with np.errstate(invalid="ignore", over="ignore"):
        return entry <= exit_ * SLACK, entry


def box_clause(o, inv, lo, hi, tmin, tmax, tt):
    """S2.4 for one box per (ray, triangle): -> (candidate exists, distance it stands at)"""
    pa, _ = slab(o, inv, lo, hi, tmin, tt)
    pb, entry = slab(o, inv, lo, hi, tmin, tmax)
    with np.errstate(invalid="ignore"):
        pb = pb & (entry < tmax) & ~pa
    return pa | pb, np.where(pa, tt, entry)


def invert3x4(m):
    """S2.6: the fp32 adjugate inverse of a 3x4 row-major object-to-world matrix, sums left to right"""
    m = np.asarray(m, f32).reshape(3, 4)
    a, b, c, d, e, f, g, h, i = (m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2], m[2, 0], m[2, 1], m[2, 2])
    A, B, C = e * i - f * h, f * g - d * i, d * h - e * g
    det = a * A
    det = det + b * B
    det = det + c * C
    idet = f32(1.0) / det
    o = np.zeros((3, 4), f32)
    o[0, :3] = (A * idet, (c * h - b * i) * idet, (b * f - c * e) * idet)
    o[1, :3] = (B * idet, (a * i - c * g) * idet, (c * d - a * f) * idet)
    o[2, :3] = (C * idet, (b * g - a * h) * idet, (a * e - b * d) * idet)
    for r in range(3):
        s = o[r, 0] * m[0, 3]
        s = s + o[r, 1] * m[1, 3]
        s = s + o[r, 2] * m[2, 3]
        o[r, 3] = -s
    return o


def xform_point(m, p):
    """x' = ((m0 x + m1 y) + m2 z) + m3, p [..., 3]"""
    out = []
    for r in range(3):
        s = m[r, 0] * p[..., 0]
        s = s + m[r, 1] * p[..., 1]
        s = s + m[r, 2] * p[..., 2]
        out.append(s + m[r, 3])
    return np.stack(out, -1)


def xform_dir(m, p):
    out = []
    for r in range(3):
        s = m[r, 0] * p[..., 0]
        s = s + m[r, 1] * p[..., 1]
        out.append(s + m[r, 2] * p[..., 2])
    return np.stack(out, -1)


def trace(models, instances, O, D, flags, ref_boxes=None):
    """models: [(positions [nv,3] f32, tris [nt,3])]; instances: [(model, 3x4 or None)]; O = (origin, tmin) [N,4], D = (direction, tmax) [N,4];
    ref_boxes: {model: (off [nt+1], boxes [n,6])} for models with split triangles (S2.5's boxes; None: every triangle has its AABB only)
    -> t (-1: miss), u, v, prim, inst   (closest hit, or with FIRST: whether any candidate exists -- then only inst tells hit from miss)"""
    O, D = np.asarray(O, f32), np.asarray(D, f32)
    N = O.shape[0]
    wo, tmin, wd, tmax = O[:, None, :3], O[:, 3:4], D[:, None, :3], D[:, 3:4]
    with np.errstate(divide="ignore"):
        winv = f32(1.0) / wd
    best_t = np.full(N, np.inf, f32)
    best = [np.zeros(N, f32), np.zeros(N, f32), np.full(N, NO_HIT, np.uint32), np.full(N, NO_HIT, np.uint32)]
    for ii, (mi, xf) in enumerate(instances):
        pos, tri = models[mi]
        pos = np.asarray(pos, f32)
        tri = np.asarray(tri).reshape(-1, 3)
        identity = xf is None or np.array_equal(np.asarray(xf, f32).reshape(12), np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], f32))
        if identity:
            o, d = wo, wd
        else:
            inv_m = invert3x4(xf)
            o, d = xform_point(inv_m, wo), xform_dir(inv_m, wd)
        with np.errstate(divide="ignore"):
            inv = f32(1.0) / d
        v0, v1, v2 = pos[tri[:, 0]][None], pos[tri[:, 1]][None], pos[tri[:, 2]][None]            # [1,M,3]
        e1, e2 = v1 - v0, v2 - v0
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            # S2.3
            px, py, pz = _cross(d[..., 0], d[..., 1], d[..., 2], e2[..., 0], e2[..., 1], e2[..., 2])
            det = _dot(e1[..., 0], e1[..., 1], e1[..., 2], px, py, pz)
            ok = (det > 0) if flags & CULL else ((det != 0) & (det == det))
            idet = f32(1.0) / det
            tv = o - v0
            u = _dot(tv[..., 0], tv[..., 1], tv[..., 2], px, py, pz) * idet
            ok &= (u >= 0) & ~(u > 1)
            qx, qy, qz = _cross(tv[..., 0], tv[..., 1], tv[..., 2], e1[..., 0], e1[..., 1], e1[..., 2])
            v = _dot(d[..., 0], d[..., 1], d[..., 2], qx, qy, qz) * idet
            ok &= (v >= 0) & (u + v <= 1)
            tt = _dot(e2[..., 0], e2[..., 1], e2[..., 2], qx, qy, qz) * idet
            ok &= (tt > tmin) & (tt < tmax)
        # S2.4: the triangle's box(es)
        M = tri.shape[0]
        refs = ref_boxes.get(mi) if ref_boxes else None
        lo = np.fmin(np.fmin(v0, v1), v2)
        hi = np.fmax(np.fmax(v0, v1), v2)
        has, at = box_clause(o, inv, lo, hi, tmin, tmax, tt)
        if refs is not None and refs[0] is not None:
            off, boxes = refs
            cnt = np.diff(off)
            for p in np.nonzero(cnt > 1)[0]:               # a split triangle: the nearest of what its reference boxes say
                b = boxes[off[p]:off[p + 1]].astype(f32)
                hk, ak = box_clause(o, inv, b[None, :, :3], b[None, :, 3:], tmin, tmax, tt[:, p:p + 1])
                ak = np.where(hk, ak, np.inf)
                has[:, p] = hk.any(1)
                at[:, p] = ak.min(1)
        ok &= has
        if not identity:                                    # ... and the instance's world box, on what the triangle's said
            wp = xform_point(np.asarray(xf, f32).reshape(3, 4), pos[tri.reshape(-1)])
            wlo, whi = wp.min(0)[None, None], wp.max(0)[None, None]
            hw, at = box_clause(wo, winv, wlo, whi, tmin, tmax, at)
            ok &= hw
        at = np.where(ok, at, np.inf)
        # S2.6: smallest t, then smaller (instance, primitive): instances ascend, argmin takes the first (smallest) primitive among equal t
        k = at.argmin(1)
        tk = at[np.arange(N), k]
        better = tk < best_t
        best_t = np.where(better, tk, best_t)
        best[0] = np.where(better, u[np.arange(N), k], best[0])
        best[1] = np.where(better, v[np.arange(N), k], best[1])
        best[2] = np.where(better, k.astype(np.uint32), best[2])
        best[3] = np.where(better, np.uint32(ii), best[3])
    hit = np.isfinite(best_t)
    return dict(t=np.where(hit, best_t, f32(-1)), u=best[0], v=best[1], prim=np.where(hit, best[2], NO_HIT), inst=np.where(hit, best[3], NO_HIT))


# ---- S2.5: the boxes of a split triangle, scalar float32, from the page ----------------------------------------------------------------
def reference_boxes(pos, tri):
    """-> (off [nt+1], boxes [n,6]) or (None, None) when no triangle of the model is split"""
    pos = np.asarray(pos, f32)
    tri = np.asarray(tri).reshape(-1, 3)
    P = pos[tri]                                             # [nt,3,3]
    lo, hi = P.min(1), P.max(1)
    ext = hi - lo
    mlo, mhi = pos[tri.reshape(-1)].min(0), pos[tri.reshape(-1)].max(0)
    min_len = f32(max(mhi - mlo)) * f32(0.001953125)
    off, boxes = [0], []
    any_split = False
    for p in range(tri.shape[0]):
        a = P[p]
        k = 1
        ex, ey, ez = ext[p]
        axis = 0 if (ex >= ey and ex >= ez) else (1 if ey >= ez else 2)
        L = ext[p][axis]
        if np.isfinite(a).all() and L > min_len:
            e1, e2 = a[1] - a[0], a[2] - a[0]
            cx, cy, cz = _cross(e1[0], e1[1], e1[2], e2[0], e2[1], e2[2])
            a2 = np.sqrt((cx * cx + cy * cy) + cz * cz)
            sa = (ex * ey + ey * ez) + ez * ex
            if a2 > 0 and sa > f32(4.0) * a2:
                kf, lf = (sa / a2) * f32(0.5), L / min_len
                k = min(128 if kf >= 128 else int(kf), 128 if lf >= 128 else int(lf))
                k = 1 if k < 2 else k
        if k == 1:
            boxes.append(np.concatenate([lo[p], hi[p]]))
        else:
            any_split = True
            uu, ww = (axis + 1) % 3, (axis + 2) % 3
            l0, h0 = lo[p][axis], hi[p][axis]
            ov = (L / f32(k)) * f32(0.25)
            maxabs = np.abs(a).max()
            pad = maxabs * f32(3.814697265625e-06)
            for j in range(k):
                s0 = l0 if j == 0 else max(l0, (l0 + L * (f32(j) / f32(k))) - ov)
                s1 = h0 if j + 1 == k else min(h0, (l0 + L * (f32(j + 1) / f32(k))) + ov)
                us, ws = [], []
                for i in range(3):
                    if s0 <= a[i][axis] <= s1:
                        us.append(a[i][uu]); ws.append(a[i][ww])
                for e in range(3):
                    pi, pj = a[e], a[(e + 1) % 3]
                    for c in (s0, s1):
                        if (pi[axis] < c and pj[axis] > c) or (pi[axis] > c and pj[axis] < c):
                            t = (c - pi[axis]) / (pj[axis] - pi[axis])
                            us.append(pi[uu] + (pj[uu] - pi[uu]) * t); ws.append(pi[ww] + (pj[ww] - pi[ww]) * t)
                if not us:
                    us, ws = [lo[p][uu], hi[p][uu]], [lo[p][ww], hi[p][ww]]
                b = np.zeros(6, f32)
                b[axis], b[3 + axis] = s0 - pad, s1 + pad
                b[uu], b[3 + uu] = min(us) - pad, max(us) + pad
                b[ww], b[3 + ww] = min(ws) - pad, max(ws) + pad
                boxes.append(b)
        off.append(off[-1] + k)
    if not any_split:
        return None, None
    return np.array(off, np.uint32), np.array(boxes, f32)
