/*
 * oracle_bvh.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Acceleration-structure build and TraceRay semantics.  The reference only
 * CALLS this arithmetic (BuildRaytracingAccelerationStructure at
 * libs/DXRFramework/Helpers/BottomLevelASGenerator.cpp:333 and
 * TopLevelASGenerator.cpp:404, DispatchRays at RtContext.cpp:218-221, HLSL
 * TraceRay at ProgressiveRaytracing.hlsl:34,53 and RaytracingCommon.hlsli:94);
 * the implementation lives in the un-vendored, unpinned submodule
 * externals/D3D12RaytracingFallback (.gitmodules:4-6).  PARITY UNPINNED for
 * node layout, traversal order and intersection arithmetic: what follows is
 * this engine's own deterministic definition, built on the public DXR
 * functional-spec semantics (ray flags, back-face rule, TMin/TMax, hit-group
 * indexing), and it is what the HIP kernels must reproduce bit for bit.
 *
 * Geometry conventions follow RtModel.cpp:33-81 (interleaved 24-B vertices,
 * u32 indices, one geometry per BLAS, OPAQUE) and TopLevelASGenerator.cpp:
 * 344-362 (instance = 3x4 row-major object-to-world, mask 0xFF, flags NONE).
 *
 * Definition summary -- the normative text is DESIGN.md section 2.1, paragraphs S2.1 - S2.8, cited below by number:
 *   BLAS/TLAS   LBVH: 30-bit Morton code of the primitive-AABB centre in the
 *               structure's AABB, key = morton<<32 | index, ascending sort,
 *               Karras-2012 radix tree, bottom-up exact min/max refit.
 *   ray/box     S2.2: slab test with precomputed 1/d, min/max ignoring NaN,
 *               passes iff max(lo, tmin) <= min(hi, tcur) * (1 + 2^-16).
 *   ray/tri     S2.3: Moller-Trumbore, u,v for v1,v2 (RaytracingCommon.hlsli:55),
 *               front face  <=>  det > 0  (clockwise from the origin in a
 *               left-handed frame, DXR spec), TMin < t < TMax exclusive.
 *   candidate   S2.4 (box clause): through a box that holds the triangle -- its AABB, or one of
 *               the reference boxes of a triangle held as several references (S2.5) -- and, for a
 *               transformed instance, through the instance's world AABB: the box passing over
 *               [tmin, t] leaves the candidate at t; failing that but meeting the ray inside
 *               (tmin, tmax) moves it to the box's entry distance (round 6; rounds 1 - 5 dropped
 *               it).  Slab tests are monotone under box inclusion and in their upper limit, so
 *               BVH traversal and the brute-force loop agree exactly (S2.7).
 *   closest     S2.6: smaller t wins; equal t -> smaller (instance, primitive).
 *   truth       S2.8: oracle/truth64.h (float64, no boxes) is what this definition is MEASURED against
 *               (tests/test_s2_truth.py, tests/golden/s2_bounds.json).
 */
#ifndef ORACLE_BVH_H
#define ORACLE_BVH_H

#include <algorithm>
#include <vector>
#include "../include/dxr_amd_types.h"
#include "oracle_math.h"

namespace orc {

struct Box { V3 lo, hi; };

static inline Box box_empty()
{
    float inf = u2f(0x7f800000u);
    Box b = { v3(inf, inf, inf), v3(-inf, -inf, -inf) };
    return b;
}
static inline float min2(float a, float b) { return a < b ? a : b; }
static inline float max2(float a, float b) { return a > b ? a : b; }
static inline void box_grow(Box &b, V3 p)
{
    b.lo = v3(min2(b.lo.x, p.x), min2(b.lo.y, p.y), min2(b.lo.z, p.z));
    b.hi = v3(max2(b.hi.x, p.x), max2(b.hi.y, p.y), max2(b.hi.z, p.z));
}
static inline void box_merge(Box &b, const Box &o) { box_grow(b, o.lo); box_grow(b, o.hi); }

/* ---- LBVH ---------------------------------------------------------------- */

static inline uint32_t expand10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8))  & 0x0300F00Fu;
    v = (v | (v << 4))  & 0x030C30C3u;
    v = (v | (v << 2))  & 0x09249249u;
    return v;
}

static inline uint32_t quant10(float c, float lo, float ext)
{
    if (!(ext > 0.0f)) return 0;
    float n = (c - lo) / ext;
    float q = n * 1024.0f;
    q = fmin_(fmax_(q, 0.0f), 1023.0f);
    return (uint32_t)q;
}

static inline uint64_t morton_key(const Box &prim, const Box &scene, uint32_t index)
{
    V3 c = vscale(vadd(prim.lo, prim.hi), 0.5f);
    V3 ext = vsub(scene.hi, scene.lo);
    uint32_t qx = quant10(c.x, scene.lo.x, ext.x);
    uint32_t qy = quant10(c.y, scene.lo.y, ext.y);
    uint32_t qz = quant10(c.z, scene.lo.z, ext.z);
    uint32_t m = (expand10(qx) << 2) | (expand10(qy) << 1) | expand10(qz);
    return ((uint64_t)m << 32) | index;
}

struct Bvh {
    uint32_t n = 0;                    /* primitives */
    std::vector<rt_bvh_node> nodes;    /* 2n-1 */
    std::vector<uint64_t> keys;        /* sorted */
    std::vector<uint32_t> parent;      /* per node, root = 0xFFFFFFFF */
    uint32_t max_depth = 0;            /* edges on the longest root-leaf path */
    Box bounds;
    uint32_t root() const { return 0; }   /* n==1: node 0 is the single leaf */
};

static inline int delta(const std::vector<uint64_t> &k, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(k[i] ^ k[j]);
}

static inline void lbvh_build(const std::vector<Box> &prims, Bvh &out)
{
    const int n = (int)prims.size();
    out.n = (uint32_t)n;
    out.nodes.assign(n ? 2 * n - 1 : 0, rt_bvh_node());
    out.parent.assign(out.nodes.size(), 0xFFFFFFFFu);
    out.keys.resize(n);
    out.bounds = box_empty();
    out.max_depth = 0;
    if (n == 0) return;
    for (int i = 0; i < n; i++) box_merge(out.bounds, prims[i]);
    for (int i = 0; i < n; i++) out.keys[i] = morton_key(prims[i], out.bounds, (uint32_t)i);
    std::sort(out.keys.begin(), out.keys.end());

    const uint32_t leaf0 = (uint32_t)(n - 1);
    for (int k = 0; k < n; k++) {
        uint32_t prim = (uint32_t)(out.keys[k] & 0xFFFFFFFFu);
        rt_bvh_node &nd = out.nodes[leaf0 + k];
        nd.bmin[0] = prims[prim].lo.x; nd.bmin[1] = prims[prim].lo.y; nd.bmin[2] = prims[prim].lo.z;
        nd.bmax[0] = prims[prim].hi.x; nd.bmax[1] = prims[prim].hi.y; nd.bmax[2] = prims[prim].hi.z;
        nd.left = prim;
        nd.right = RT_LEAF;
    }
    const std::vector<uint64_t> &k = out.keys;
    for (int i = 0; i < n - 1; i++) {
        int d = (delta(k, n, i, i + 1) - delta(k, n, i, i - 1)) > 0 ? 1 : -1;
        int dmin = delta(k, n, i, i - d);
        int lmax = 2;
        while (delta(k, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(k, n, i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta(k, n, i, j);
        int s = 0;
        int t = l;
        do {
            t = (t + 1) >> 1;
            if (delta(k, n, i, i + (s + t) * d) > dnode) s += t;
        } while (t > 1);
        int gamma = i + s * d + std::min(d, 0);
        uint32_t left  = (std::min(i, j) == gamma)     ? leaf0 + (uint32_t)gamma       : (uint32_t)gamma;
        uint32_t right = (std::max(i, j) == gamma + 1) ? leaf0 + (uint32_t)(gamma + 1) : (uint32_t)(gamma + 1);
        out.nodes[i].left = left;
        out.nodes[i].right = right;
        out.parent[left] = (uint32_t)i;
        out.parent[right] = (uint32_t)i;
    }
    /* bottom-up refit: every internal node after both children; walk from leaves */
    std::vector<uint8_t> visits(n > 1 ? n - 1 : 0, 0);
    for (int kk = 0; kk < n; kk++) {
        uint32_t cur = out.parent[leaf0 + kk];
        uint32_t depth = 1;
        while (cur != 0xFFFFFFFFu) {
            if (visits[cur]++ == 0) break;        /* first arrival: sibling not done */
            rt_bvh_node &nd = out.nodes[cur];
            const rt_bvh_node &a = out.nodes[nd.left];
            const rt_bvh_node &b = out.nodes[nd.right];
            for (int c = 0; c < 3; c++) {
                nd.bmin[c] = min2(a.bmin[c], b.bmin[c]);
                nd.bmax[c] = max2(a.bmax[c], b.bmax[c]);
            }
            cur = out.parent[cur];
            depth++;
        }
        (void)depth;
    }
    /* depth of the deepest leaf */
    for (int kk = 0; kk < n; kk++) {
        uint32_t d = 0;
        for (uint32_t cur = out.parent[leaf0 + kk]; cur != 0xFFFFFFFFu; cur = out.parent[cur]) d++;
        out.max_depth = std::max(out.max_depth, d);
    }
}

/* ---- scene --------------------------------------------------------------- */

struct Model {
    std::vector<rt_vertex> verts;
    std::vector<uint32_t> idx;
    uint32_t ntris = 0;
    Bvh blas;
    /* validation boxes (round 5, "split references"): triangle p is validated against refs[ref_off[p] .. ref_off[p + 1]); an unsplit
     * triangle has ONE, its own AABB -- rounds 1 - 4's rule.  Empty when no triangle of the model is split. */
    std::vector<uint32_t> ref_off;
    std::vector<Box> refs;
};

struct Instance {
    uint32_t model;
    float m[12];       /* object-to-world, 3x4 row-major (TopLevelASGenerator.cpp:355-357) */
    float inv[12];     /* world-to-object */
    bool identity;
    Box world;
};

struct Scene {
    std::vector<Model> models;
    std::vector<Instance> inst;
    Bvh tlas;
    bool built = false;
};

static inline void tri_verts(const Model &m, uint32_t prim, V3 &a, V3 &b, V3 &c)
{
    const rt_float3 &p0 = m.verts[m.idx[3 * prim + 0]].position;
    const rt_float3 &p1 = m.verts[m.idx[3 * prim + 1]].position;
    const rt_float3 &p2 = m.verts[m.idx[3 * prim + 2]].position;
    a = v3(p0.x, p0.y, p0.z); b = v3(p1.x, p1.y, p1.z); c = v3(p2.x, p2.y, p2.z);
}

static inline Box tri_box(V3 a, V3 b, V3 c)
{
    Box bx = box_empty();
    box_grow(bx, a); box_grow(bx, b); box_grow(bx, c);
    return bx;
}

/*
 * Split references (round 5).  A long thin triangle that runs diagonally through space has an AABB hundreds of times the size it needs
 * (a 47 m x 2 cm cable: AABB 40 x 5 x 25 m): every ray through that box has to test it.  The production tree therefore holds such a
 * triangle as SEVERAL references, each with the box of the part of the triangle inside one slab of its longest axis, and the candidate
 * rule becomes "accepted only if ONE OF THE TRIANGLE'S REFERENCE BOXES passes the slab test over [tmin, t]" -- for a triangle that is not
 * split (every triangle of every scene of rounds 1 - 4) that is its own AABB, the old rule, bit for bit.  Float slab tests are monotone
 * under box inclusion, so any traversal whose node boxes contain the reference boxes -- the canonical tree over whole triangles, the
 * production tree over references, brute force -- still returns the same bits.  The rule below IS the definition; rt_refs.h restates it
 * for the builder and the kernels, operation for operation (no fused operations, IEEE divide and sqrt).
 *   split iff  every coordinate is finite,  L = longest AABB extent > min_len (= the model's longest extent / 512),
 *              a2 = |cross(v1 - v0, v2 - v0)| > 0  and  sa = (ex ey + ey ez) + ez ex > 4 a2   (AABB surface over triangle area > 8)
 *   pieces     k = min(128, trunc((sa / a2) / 2), trunc(L / min_len)), at least 2; piece j = the slab [lo + L (j / k), lo + L ((j + 1) / k)]
 *              of the longest axis (ties: x before y before z), widened by (L / k) / 4 either side and cut back to [lo, hi]; the first
 *              slab starts at lo, the last ends at hi
 *   its box    on the split axis the slab; on the others min / max over the vertices inside the slab and the points where the edges
 *              0->1, 1->2, 2->0 cross the slab's two planes (y = yi + (yj - yi) * ((c - xi) / (xj - xi))); every side moved out by
 *              pad = 2^-18 x the triangle's largest |coordinate| (rounding of the crossings, of the Moller-Trumbore acceptance itself)
 */
/* S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json) */
static inline bool &split_refs_enabled() { static bool on = true; return on; }      /* (tests: the rule of rounds 1 - 4 beside the new one) */
static inline float absf_(float x) { return x < 0.0f ? -x : x; }
static inline uint32_t ref_pieces(V3 a, V3 b, V3 c, float min_len, int *axis_out)
{
    const float v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
    for (int k = 0; k < 9; k++) if (!(v[k] - v[k] == 0.0f)) return 1;
    const Box bx = tri_box(a, b, c);
    const float ex = bx.hi.x - bx.lo.x, ey = bx.hi.y - bx.lo.y, ez = bx.hi.z - bx.lo.z;
    const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    const float L = axis == 0 ? ex : (axis == 1 ? ey : ez);
    if (axis_out) *axis_out = axis;
    if (!(L > min_len)) return 1;
    const V3 e1 = vsub(b, a), e2 = vsub(c, a);
    const V3 cr = cross3(e1, e2);
    const float a2 = sqrtf((cr.x * cr.x + cr.y * cr.y) + cr.z * cr.z);
    if (!(a2 > 0.0f)) return 1;
    const float sa = (ex * ey + ey * ez) + ez * ex;
    if (!(sa > 4.0f * a2)) return 1;
    const float kf = (sa / a2) * 0.5f, lf = L / min_len;
    uint32_t k = kf >= 128.0f ? 128u : (uint32_t)kf;
    const uint32_t kl = lf >= 128.0f ? 128u : (uint32_t)lf;
    if (kl < k) k = kl;
    return k < 2u ? 1u : k;
}
static inline Box ref_box(V3 a, V3 b, V3 c, int axis, uint32_t k, uint32_t j)
{
    const float p[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
    const int u = (axis + 1) % 3, w = (axis + 2) % 3;
    float lo = min2(min2(p[0][axis], p[1][axis]), p[2][axis]), hi = max2(max2(p[0][axis], p[1][axis]), p[2][axis]);
    const float L = hi - lo;
    /* (a quarter of a slab of overlap either side: Moller-Trumbore's t is only good to a per cent or so on a 1000:1 sliver, and the
     * candidate rule asks the ray to be inside the box AT that t) */
    const float ov = (L / (float)k) * 0.25f;
    const float s0 = j == 0 ? lo : max2(lo, (lo + L * ((float)j / (float)k)) - ov);
    const float s1 = j + 1 == k ? hi : min2(hi, (lo + L * ((float)(j + 1) / (float)k)) + ov);
    const float inf = u2f(0x7f800000u);
    float ulo = inf, uhi = -inf, wlo = inf, whi = -inf, maxabs = 0.0f;
    for (int i = 0; i < 3; i++) {
        for (int q = 0; q < 3; q++) maxabs = max2(maxabs, absf_(p[i][q]));
        if (p[i][axis] >= s0 && p[i][axis] <= s1) {
            ulo = min2(ulo, p[i][u]); uhi = max2(uhi, p[i][u]);
            wlo = min2(wlo, p[i][w]); whi = max2(whi, p[i][w]);
        }
    }
    for (int e = 0; e < 3; e++) {
        const float *pi = p[e], *pj = p[(e + 1) % 3];
        for (int side = 0; side < 2; side++) {
            const float cpl = side == 0 ? s0 : s1;
            if ((pi[axis] < cpl && pj[axis] > cpl) || (pi[axis] > cpl && pj[axis] < cpl)) {
                const float t = (cpl - pi[axis]) / (pj[axis] - pi[axis]);
                const float yu = pi[u] + (pj[u] - pi[u]) * t, yw = pi[w] + (pj[w] - pi[w]) * t;
                ulo = min2(ulo, yu); uhi = max2(uhi, yu);
                wlo = min2(wlo, yw); whi = max2(whi, yw);
            }
        }
    }
    if (!(ulo <= uhi) || !(wlo <= whi)) {          /* (cannot happen for a slab inside the triangle's extent; the whole extent then) */
        ulo = min2(min2(p[0][u], p[1][u]), p[2][u]); uhi = max2(max2(p[0][u], p[1][u]), p[2][u]);
        wlo = min2(min2(p[0][w], p[1][w]), p[2][w]); whi = max2(max2(p[0][w], p[1][w]), p[2][w]);
    }
    const float pad = maxabs * 3.814697265625e-06f;        /* 2^-18 */
    float blo[3], bhi[3];
    blo[axis] = s0 - pad; bhi[axis] = s1 + pad;
    blo[u] = ulo - pad; bhi[u] = uhi + pad;
    blo[w] = wlo - pad; bhi[w] = whi + pad;
    Box r = { v3(blo[0], blo[1], blo[2]), v3(bhi[0], bhi[1], bhi[2]) };
    return r;
}

/* S2-RULE-END */
static inline bool is_identity(const float m[12])
{
    static const float id[12] = {1,0,0,0, 0,1,0,0, 0,0,1,0};
    for (int i = 0; i < 12; i++) if (!(m[i] == id[i])) return false;
    return true;
}

/* world-to-object = inverse of the 3x4 affine (adjugate / determinant, fp32) */
static inline void invert3x4(const float m[12], float o[12])
{
    float a = m[0], b = m[1], c = m[2];
    float d = m[4], e = m[5], f = m[6];
    float g = m[8], h = m[9], i = m[10];
    float A = e * i - f * h;
    float B = f * g - d * i;
    float C = d * h - e * g;
    float det = a * A;
    det = det + b * B;
    det = det + c * C;
    float id = 1.0f / det;
    o[0] = A * id;  o[1] = (c * h - b * i) * id;  o[2]  = (b * f - c * e) * id;
    o[4] = B * id;  o[5] = (a * i - c * g) * id;  o[6]  = (c * d - a * f) * id;
    o[8] = C * id;  o[9] = (b * g - a * h) * id;  o[10] = (a * e - b * d) * id;
    float tx = m[3], ty = m[7], tz = m[11];
    for (int r = 0; r < 3; r++) {
        float s = o[4 * r + 0] * tx;
        s = s + o[4 * r + 1] * ty;
        s = s + o[4 * r + 2] * tz;
        o[4 * r + 3] = -s;
    }
}

static inline V3 xform_point(const float m[12], V3 p)
{
    float x = m[0] * p.x; x = x + m[1] * p.y; x = x + m[2]  * p.z; x = x + m[3];
    float y = m[4] * p.x; y = y + m[5] * p.y; y = y + m[6]  * p.z; y = y + m[7];
    float z = m[8] * p.x; z = z + m[9] * p.y; z = z + m[10] * p.z; z = z + m[11];
    return v3(x, y, z);
}
static inline V3 xform_dir(const float m[12], V3 p)
{
    float x = m[0] * p.x; x = x + m[1] * p.y; x = x + m[2]  * p.z;
    float y = m[4] * p.x; y = y + m[5] * p.y; y = y + m[6]  * p.z;
    float z = m[8] * p.x; z = z + m[9] * p.y; z = z + m[10] * p.z;
    return v3(x, y, z);
}

static inline void scene_build(Scene &s)
{
    for (Model &m : s.models) {
        std::vector<Box> boxes(m.ntris);
        for (uint32_t p = 0; p < m.ntris; p++) {
            V3 a, b, c;
            tri_verts(m, p, a, b, c);
            boxes[p] = tri_box(a, b, c);
        }
        lbvh_build(boxes, m.blas);
        /* the validation boxes of split triangles */
        m.ref_off.clear(); m.refs.clear();
        const Box &mb = m.blas.bounds;
        const float ext = max2(max2(mb.hi.x - mb.lo.x, mb.hi.y - mb.lo.y), mb.hi.z - mb.lo.z);
        const float min_len = ext * 0.001953125f;
        std::vector<uint32_t> cnt(m.ntris, 1u);
        bool any = false;
        for (uint32_t p = 0; p < m.ntris; p++) {
            V3 a, b, c;
            tri_verts(m, p, a, b, c);
            cnt[p] = split_refs_enabled() ? ref_pieces(a, b, c, min_len, nullptr) : 1u;
            any = any || cnt[p] > 1u;
        }
        if (any) {
            m.ref_off.resize((size_t)m.ntris + 1);
            uint32_t at = 0;
            for (uint32_t p = 0; p < m.ntris; p++) { m.ref_off[p] = at; at += cnt[p]; }
            m.ref_off[m.ntris] = at;
            m.refs.resize(at);
            for (uint32_t p = 0; p < m.ntris; p++) {
                V3 a, b, c;
                tri_verts(m, p, a, b, c);
                if (cnt[p] == 1u) { m.refs[m.ref_off[p]] = boxes[p]; continue; }
                int axis = 0;
                (void)ref_pieces(a, b, c, min_len, &axis);
                for (uint32_t j = 0; j < cnt[p]; j++) m.refs[m.ref_off[p] + j] = ref_box(a, b, c, axis, cnt[p], j);
            }
        }
    }
    std::vector<Box> ib(s.inst.size());
    for (size_t i = 0; i < s.inst.size(); i++) {
        Instance &in = s.inst[i];
        const Box &b = s.models[in.model].blas.bounds;
        in.identity = is_identity(in.m);
        if (in.identity) {
            in.world = b;
            for (int k = 0; k < 12; k++) in.inv[k] = in.m[k];
        } else {
            invert3x4(in.m, in.inv);
            /* the instance's world box: the exact box of its triangles' transformed vertices (not the box of the BLAS
             * box's eight transformed corners, which is up to 1.6x wider in footprint for a rotated mesh and makes a
             * third of all instance entries false ones, profiles/r04/tight_boxes.txt); min / max: order free */
            const Model &md = s.models[in.model];
            in.world = box_empty();
            for (uint32_t p = 0; p < md.ntris; p++) {
                V3 a, bb, c;
                tri_verts(md, p, a, bb, c);
                box_grow(in.world, xform_point(in.m, a));
                box_grow(in.world, xform_point(in.m, bb));
                box_grow(in.world, xform_point(in.m, c));
            }
        }
        ib[i] = in.world;
    }
    lbvh_build(ib, s.tlas);
    s.built = true;
}

/* ---- rays ---------------------------------------------------------------- */

struct Ray { V3 o; float tmin; V3 d; float tmax; };

struct Hit {
    float t, u, v;
    uint32_t prim, inst;
};

struct Counters { uint32_t nodes, tris; };

struct RayInv { V3 o, inv; };

static inline RayInv ray_inv(V3 o, V3 d)
{
    RayInv r;
    r.o = o;
    r.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

/* Conservative slab test: true iff max(lo, t0) <= min(hi, t1) * (1 + 2^-16);
 * *entry = max(lo,t0).  The slack absorbs the few-ulp disagreement between a
 * slab distance and the Moller-Trumbore distance to a triangle lying in a
 * flat (zero-thickness) box; multiplying by a positive constant keeps the
 * test monotone under box inclusion, which is what the exactness argument in
 * the header needs.  All distances here are >= 0 because t0 >= 0. */
/* S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json) */
#define ORC_SLAB_SLACK 1.0000152587890625f
static inline bool slab(const RayInv &r, const float bmin[3], const float bmax[3], float t0, float t1, float *entry)
{
    float ax = (bmin[0] - r.o.x) * r.inv.x, bx = (bmax[0] - r.o.x) * r.inv.x;
    float ay = (bmin[1] - r.o.y) * r.inv.y, by = (bmax[1] - r.o.y) * r.inv.y;
    float az = (bmin[2] - r.o.z) * r.inv.z, bz = (bmax[2] - r.o.z) * r.inv.z;
    float lo = fmax_(fmax_(fmin_(ax, bx), fmin_(ay, by)), fmax_(fmin_(az, bz), t0));
    float hi = fmin_(fmin_(fmax_(ax, bx), fmax_(ay, by)), fmin_(fmax_(az, bz), t1));
    *entry = lo;
    return lo <= hi * ORC_SLAB_SLACK;
}
static inline bool slab_box(const RayInv &r, const Box &b, float t0, float t1)
{
    float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}, e;
    return slab(r, lo, hi, t0, t1, &e);
}

/*
 * The candidate rule's box clause (paragraph S2.4; why trees cannot matter: S2.7).  Moller-Trumbore said the ray meets the triangle at tt; `b` is a
 * box that contains (a part of) the triangle.  Float slab tests are monotone under box inclusion and in their upper limit, so a candidate
 * whose box passes over [tmin, tt] is reached by every traversal that still looks for hits at tt or beyond: it stands at tt.  On a sliver
 * the fp32 tt can lie a little BEFORE the ray even enters the box the triangle is in (1000:1 slivers: tt is good to a per mille or so;
 * rounds 1 - 5 rejected such a candidate, and the ray went through the triangle: 13 - 27 of 20,000 rays aimed at slivers,
 * profiles/r06/s2_truth_before.txt).  Since round 6 such a candidate stands at the point where the ray enters the box -- the nearest point
 * of the ray that can lie on the triangle at all -- provided the ray meets the box inside (tmin, tmax): a traversal that still looks for
 * hits at that entry distance or beyond visits the box, so trees, visiting orders and brute force keep agreeing bit for bit.
 * Returns false (no candidate) or the distance the candidate stands at.
 */
static inline bool box_clause(const RayInv &ri, const Box &b, float tmin, float tmax, float tt, float *t_at)
{
    float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}, e;
    if (slab(ri, lo, hi, tmin, tt, &e)) { *t_at = tt; return true; }
    if (!slab(ri, lo, hi, tmin, tmax, &e) || !(e < tmax)) return false;      /* (e = max(entry, tmin) > tt > tmin here) */
    *t_at = e;
    return true;
}

/* Moller-Trumbore + the box clause against the triangle's own AABB / its reference boxes.  o,d in the triangle's space. */
static inline bool tri_candidate(V3 o, V3 d, const RayInv &ri, float tmin, float tmax,
                                 V3 v0, V3 v1, V3 v2, bool cull_back, float *t, float *u, float *v,
                                 const Box *refs = nullptr, uint32_t n_refs = 0)
{
    V3 e1 = vsub(v1, v0);
    V3 e2 = vsub(v2, v0);
    V3 p = cross3(d, e2);
    float det = dot3(e1, p);
    if (cull_back) { if (!(det > 0.0f)) return false; }
    else           { if (det == 0.0f || det != det) return false; }
    float inv = 1.0f / det;
    V3 tv = vsub(o, v0);
    float uu = dot3(tv, p) * inv;
    if (!(uu >= 0.0f) || uu > 1.0f) return false;
    V3 q = cross3(tv, e1);
    float vv = dot3(d, q) * inv;
    if (!(vv >= 0.0f) || !(uu + vv <= 1.0f)) return false;
    float tt = dot3(e2, q) * inv;
    if (!(tt > tmin) || !(tt < tmax)) return false;
    float at;
    if (n_refs > 1u) {              /* a split triangle: the nearest of what its reference boxes say */
        bool ok = false;
        for (uint32_t k = 0; k < n_refs; k++) {
            float a;
            if (!box_clause(ri, refs[k], tmin, tmax, tt, &a)) continue;
            if (!ok || a < at) at = a;
            ok = true;
            if (at == tt) break;    /* (no clause answers less than tt) */
        }
        if (!ok) return false;
    } else {
        Box b = tri_box(v0, v1, v2);
        if (!box_clause(ri, b, tmin, tmax, tt, &at)) return false;
    }
    *t = at; *u = uu; *v = vv;
    return true;
}

static inline bool better(float t, uint32_t inst, uint32_t prim, const Hit &h)
{
    if (t < h.t) return true;
    if (t > h.t) return false;
    if (h.inst == RT_NO_HIT) return false;   /* t == ray.tmax is excluded by t < tmax */
    if (inst != h.inst) return inst < h.inst;
    return prim < h.prim;
}

/* S2-RULE-END */
struct ObjRay { V3 o, d; RayInv ri; };

static inline ObjRay to_object(const Instance &in, const Ray &r)
{
    ObjRay o;
    if (in.identity) { o.o = r.o; o.d = r.d; }
    else { o.o = xform_point(in.inv, r.o); o.d = xform_dir(in.inv, r.d); }
    o.ri = ray_inv(o.o, o.d);
    return o;
}

/* one candidate triangle of one instance against the running best */
static inline bool test_prim(const Scene &s, uint32_t ii, uint32_t prim, const Ray &r, const RayInv &wri,
                             const ObjRay &orr, uint32_t flags, Hit &best)
{
    const Instance &in = s.inst[ii];
    const Model &m = s.models[in.model];
    V3 a, b, c;
    tri_verts(m, prim, a, b, c);
    float t, u, v;
    bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    /* closest: candidates in (tmin, ray.tmax); ordering vs best handled by better() */
    const Box *refs = nullptr;
    uint32_t n_refs = 0;
    if (!m.ref_off.empty()) { refs = &m.refs[m.ref_off[prim]]; n_refs = m.ref_off[prim + 1] - m.ref_off[prim]; }
    if (!tri_candidate(orr.o, orr.d, orr.ri, r.tmin, r.tmax, a, b, c, cull, &t, &u, &v, refs, n_refs)) return false;
    if (!in.identity && !box_clause(wri, in.world, r.tmin, r.tmax, t, &t)) return false;      /* (the instance's world box: the same clause, on what the triangle's said) */
    if (!better(t, ii, prim, best)) return false;
    best.t = t; best.u = u; best.v = v; best.prim = prim; best.inst = ii;
    return true;
}

static inline Hit miss_hit(const Ray &r)
{
    Hit h; h.t = r.tmax; h.u = 0; h.v = 0; h.prim = RT_NO_HIT; h.inst = RT_NO_HIT;
    return h;
}

/* topology-independent truth: every instance x every triangle */
static inline Hit trace_brute(const Scene &s, const Ray &r, uint32_t flags)
{
    Hit best = miss_hit(r);
    RayInv wri = ray_inv(r.o, r.d);
    bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    for (uint32_t ii = 0; ii < s.inst.size(); ii++) {
        ObjRay orr = to_object(s.inst[ii], r);
        uint32_t nt = s.models[s.inst[ii].model].ntris;
        for (uint32_t p = 0; p < nt; p++) {
            if (test_prim(s, ii, p, r, wri, orr, flags, best) && first) return best;
        }
    }
    return best;
}

/*
 * Canonical two-level stack traversal.  Order: test both children; if both
 * are hit descend into the one with the smaller entry distance (tie: left)
 * and push the other.  Counters: nodes = AABBs slab-tested (roots included),
 * tris = triangles handed to the Moller-Trumbore test.
 */
static inline bool traverse_blas(const Scene &s, uint32_t ii, const Ray &r, const RayInv &wri,
                                 uint32_t flags, Hit &best, Counters &cnt)
{
    const Instance &in = s.inst[ii];
    const Bvh &bv = s.models[in.model].blas;
    if (bv.n == 0) return false;
    bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    ObjRay orr = to_object(in, r);
    uint32_t stack[128];
    int sp = 0;
    float e;
    cnt.nodes++;
    if (!slab(orr.ri, bv.nodes[0].bmin, bv.nodes[0].bmax, r.tmin, best.t, &e)) return false;
    uint32_t cur = 0;
    for (;;) {
        const rt_bvh_node &nd = bv.nodes[cur];
        if (nd.right == RT_LEAF) {
            cnt.tris++;
            if (test_prim(s, ii, nd.left, r, wri, orr, flags, best) && first) return true;
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const rt_bvh_node &a = bv.nodes[nd.left];
        const rt_bvh_node &b = bv.nodes[nd.right];
        float ea, eb;
        cnt.nodes += 2;
        bool ha = slab(orr.ri, a.bmin, a.bmax, r.tmin, best.t, &ea);
        bool hb = slab(orr.ri, b.bmin, b.bmax, r.tmin, best.t, &eb);
        if (ha && hb) {
            if (eb < ea) { stack[sp++] = nd.left;  cur = nd.right; }
            else         { stack[sp++] = nd.right; cur = nd.left; }
        } else if (ha) cur = nd.left;
        else if (hb)   cur = nd.right;
        else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return false;
}

static inline Hit trace_bvh(const Scene &s, const Ray &r, uint32_t flags, Counters &cnt)
{
    Hit best = miss_hit(r);
    cnt.nodes = 0; cnt.tris = 0;
    const Bvh &tl = s.tlas;
    if (tl.n == 0) return best;
    bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    RayInv wri = ray_inv(r.o, r.d);
    uint32_t stack[128];
    int sp = 0;
    float e;
    cnt.nodes++;
    if (!slab(wri, tl.nodes[0].bmin, tl.nodes[0].bmax, r.tmin, best.t, &e)) return best;
    uint32_t cur = 0;
    for (;;) {
        const rt_bvh_node &nd = tl.nodes[cur];
        if (nd.right == RT_LEAF) {
            if (traverse_blas(s, nd.left, r, wri, flags, best, cnt) && first) return best;
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const rt_bvh_node &a = tl.nodes[nd.left];
        const rt_bvh_node &b = tl.nodes[nd.right];
        float ea, eb;
        cnt.nodes += 2;
        bool ha = slab(wri, a.bmin, a.bmax, r.tmin, best.t, &ea);
        bool hb = slab(wri, b.bmin, b.bmax, r.tmin, best.t, &eb);
        if (ha && hb) {
            if (eb < ea) { stack[sp++] = nd.left;  cur = nd.right; }
            else         { stack[sp++] = nd.right; cur = nd.left; }
        } else if (ha) cur = nd.left;
        else if (hb)   cur = nd.right;
        else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return best;
}

}  // namespace orc

#endif
