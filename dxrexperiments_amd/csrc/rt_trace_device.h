// rt_trace_device.h -- TraceRay on gfx950: ray/box, ray/triangle and the two
// traversal loops (canonical reference-order loop with counters; production loop
// over 64-B slabs with an LDS-resident per-lane stack).
//
// Semantics restated from the DXR functional spec as the reference uses it
// (TraceRay call sites: assets/shaders/ProgressiveRaytracing.hlsl:34,53 and
// RaytracingCommon.hlsli:94; geometry is OPAQUE, BottomLevelASGenerator.h:127, so
// no any-hit shader ever runs; instance mask 0xFF / flags NONE,
// TopLevelASGenerator.cpp:344-362):
//   - triangle accepted iff TMin < t < TMax (exclusive);
//   - RAY_FLAG_CULL_BACK_FACING_TRIANGLES: front face <=> det > 0 (vertices
//     clockwise from the ray origin in a left-handed frame);
//   - RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH ends at the first accepted hit;
//   - closest hit: smaller t wins, equal t -> smaller (instance, primitive);
//   - barycentrics (u, v) weight v1, v2 (RaytracingCommon.hlsli:55).
// Engine-defined parts (the Fallback Layer's own arithmetic is not in the
// reference checkout): slab box test with slack 1+2^-16, Moller-Trumbore, and
// the candidate re-validation against the triangle's own AABB / the instance's
// world AABB that makes BVH order irrelevant to the result (DESIGN.md "Exactness").
#pragma once

#include "rt_device_math.h"
#include "rt_internal.h"

namespace rtd {

#define RT_SLAB_SLACK 1.0000152587890625f

struct RayD {
    f3 o; float tmin;
    f3 d; float tmax;
};

struct HitD {
    float t, u, v;
    uint32_t prim, inst;
};

struct RayInv { f3 o, inv; };

RT_DEV RayInv make_inv(f3 o, f3 d)
{
    RayInv r;
    r.o = o;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

// slab test of one box; entry = max(lo, t0)
RT_DEV bool slab_hit(const RayInv &r, float lx, float hx, float ly, float hy, float lz, float hz,
                     float t0, float t1, float &entry)
{
    const float ax = (lx - r.o.x) * r.inv.x, bx = (hx - r.o.x) * r.inv.x;
    const float ay = (ly - r.o.y) * r.inv.y, by = (hy - r.o.y) * r.inv.y;
    const float az = (lz - r.o.z) * r.inv.z, bz = (hz - r.o.z) * r.inv.z;
    const float lo = fmax2(fmax2(fmin2(ax, bx), fmin2(ay, by)), fmax2(fmin2(az, bz), t0));
    const float hi = fmin2(fmin2(fmax2(ax, bx), fmax2(ay, by)), fmin2(fmax2(az, bz), t1));
    entry = lo;
    return lo <= hi * RT_SLAB_SLACK;
}

RT_DEV f3 xform_point(const float *m, f3 p)
{
    float x = m[0] * p.x; x += m[1] * p.y; x += m[2] * p.z; x += m[3];
    float y = m[4] * p.x; y += m[5] * p.y; y += m[6] * p.z; y += m[7];
    float z = m[8] * p.x; z += m[9] * p.y; z += m[10] * p.z; z += m[11];
    return mk3(x, y, z);
}
RT_DEV f3 xform_dir(const float *m, f3 p)
{
    float x = m[0] * p.x; x += m[1] * p.y; x += m[2] * p.z;
    float y = m[4] * p.x; y += m[5] * p.y; y += m[6] * p.z;
    float z = m[8] * p.x; z += m[9] * p.y; z += m[10] * p.z;
    return mk3(x, y, z);
}

// Moller-Trumbore + validation against the triangle's own AABB over [tmin, t].
RT_DEV bool tri_candidate(f3 o, f3 d, const RayInv &ri, float tmin, float tmax, f3 v0, f3 v1, f3 v2, bool cull,
                          float &t, float &u, float &v)
{
    const f3 e1 = v1 - v0;
    const f3 e2 = v2 - v0;
    const f3 p = cross(d, e2);
    const float det = dot(e1, p);
    if (cull) { if (!(det > 0.0f)) return false; }
    else { if (det == 0.0f || det != det) return false; }
    const float inv = 1.0f / det;
    const f3 tv = o - v0;
    const float uu = dot(tv, p) * inv;
    if (!(uu >= 0.0f) || uu > 1.0f) return false;
    const f3 q = cross(tv, e1);
    const float vv = dot(d, q) * inv;
    if (!(vv >= 0.0f) || !(uu + vv <= 1.0f)) return false;
    const float tt = dot(e2, q) * inv;
    if (!(tt > tmin) || !(tt < tmax)) return false;
    float e;
    if (!slab_hit(ri, fmin2(fmin2(v0.x, v1.x), v2.x), fmax2(fmax2(v0.x, v1.x), v2.x),
                  fmin2(fmin2(v0.y, v1.y), v2.y), fmax2(fmax2(v0.y, v1.y), v2.y),
                  fmin2(fmin2(v0.z, v1.z), v2.z), fmax2(fmax2(v0.z, v1.z), v2.z), tmin, tt, e))
        return false;
    t = tt; u = uu; v = vv;
    return true;
}

RT_DEV bool hit_better(float t, uint32_t inst, uint32_t prim, const HitD &h)
{
    if (t < h.t) return true;
    if (t > h.t) return false;
    if (h.inst == RT_NO_HIT) return false;
    if (inst != h.inst) return inst < h.inst;
    return prim < h.prim;
}

RT_DEV HitD make_miss(const RayD &r)
{
    HitD h;
    h.t = r.tmax; h.u = 0.0f; h.v = 0.0f; h.prim = RT_NO_HIT; h.inst = RT_NO_HIT;
    return h;
}

struct ObjRay { f3 o, d; RayInv ri; };

RT_DEV ObjRay to_object(const InstanceRec &in, const RayD &r)
{
    ObjRay o;
    if (in.flags & RT_INST_IDENTITY) { o.o = r.o; o.d = r.d; }
    else { o.o = xform_point(in.inv, r.o); o.d = xform_dir(in.inv, r.d); }
    o.ri = make_inv(o.o, o.d);
    return o;
}

// candidate (v0,v1,v2,prim) of instance ii against the running best
RT_DEV bool accept_candidate(const InstanceRec &in, uint32_t ii, uint32_t prim, f3 v0, f3 v1, f3 v2, const RayD &r,
                             const RayInv &wri, const ObjRay &orr, bool cull, HitD &best)
{
    float t, u, v;
    if (!tri_candidate(orr.o, orr.d, orr.ri, r.tmin, r.tmax, v0, v1, v2, cull, t, u, v)) return false;
    if (!(in.flags & RT_INST_IDENTITY)) {
        float e;
        if (!slab_hit(wri, in.wlo[0], in.whi[0], in.wlo[1], in.whi[1], in.wlo[2], in.whi[2], r.tmin, t, e)) return false;
    }
    if (!hit_better(t, ii, prim, best)) return false;
    best.t = t; best.u = u; best.v = v; best.prim = prim; best.inst = ii;
    return true;
}

// ---------------------------------------------------------------------------
// Canonical traversal: the reference-order loop over 32-B canonical nodes.  Its
// node/triangle counters define the ALGORITHMIC bytes of a ray.
// ---------------------------------------------------------------------------

RT_DEV bool node_slab(const RayInv &ri, const rt_bvh_node &n, float t0, float t1, float &e)
{
    return slab_hit(ri, n.bmin[0], n.bmax[0], n.bmin[1], n.bmax[1], n.bmin[2], n.bmax[2], t0, t1, e);
}

RT_DEV bool canonical_blas(const SceneDev &sc, uint32_t ii, const RayD &r, const RayInv &wri, uint32_t flags, HitD &best,
                           uint32_t &cn, uint32_t &ct)
{
    const InstanceRec &in = sc.inst[ii];
    if (in.n_prims == 0) return false;
    const bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    const ObjRay orr = to_object(in, r);
    const rt_bvh_node *nodes = in.cnodes;
    uint32_t stack[128];
    int sp = 0;
    float e;
    cn++;
    if (!node_slab(orr.ri, nodes[0], r.tmin, best.t, e)) return false;
    uint32_t cur = 0;
    for (;;) {
        const rt_bvh_node nd = nodes[cur];
        if (nd.right == RT_LEAF) {
            ct++;
            const uint32_t prim = nd.left;
            const rt_float3 p0 = in.verts[in.indices[3 * prim + 0]].position;
            const rt_float3 p1 = in.verts[in.indices[3 * prim + 1]].position;
            const rt_float3 p2 = in.verts[in.indices[3 * prim + 2]].position;
            if (accept_candidate(in, ii, prim, mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, wri, orr,
                                 cull, best) && first)
                return true;
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const rt_bvh_node a = nodes[nd.left];
        const rt_bvh_node b = nodes[nd.right];
        float ea, eb;
        cn += 2;
        const bool ha = node_slab(orr.ri, a, r.tmin, best.t, ea);
        const bool hb = node_slab(orr.ri, b, r.tmin, best.t, eb);
        if (ha && hb) {
            if (eb < ea) { stack[sp++] = nd.left; cur = nd.right; }
            else { stack[sp++] = nd.right; cur = nd.left; }
        } else if (ha) cur = nd.left;
        else if (hb) cur = nd.right;
        else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return false;
}

RT_DEV HitD trace_canonical(const SceneDev &sc, const RayD &r, uint32_t flags, uint32_t &cn, uint32_t &ct)
{
    HitD best = make_miss(r);
    cn = 0; ct = 0;
    if (sc.n_inst == 0) return best;
    const bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const RayInv wri = make_inv(r.o, r.d);
    const rt_bvh_node *nodes = sc.tlas_cnodes;
    uint32_t stack[128];
    int sp = 0;
    float e;
    cn++;
    if (!node_slab(wri, nodes[0], r.tmin, best.t, e)) return best;
    uint32_t cur = 0;
    for (;;) {
        const rt_bvh_node nd = nodes[cur];
        if (nd.right == RT_LEAF) {
            if (canonical_blas(sc, nd.left, r, wri, flags, best, cn, ct) && first) return best;
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        const rt_bvh_node a = nodes[nd.left];
        const rt_bvh_node b = nodes[nd.right];
        float ea, eb;
        cn += 2;
        const bool ha = node_slab(wri, a, r.tmin, best.t, ea);
        const bool hb = node_slab(wri, b, r.tmin, best.t, eb);
        if (ha && hb) {
            if (eb < ea) { stack[sp++] = nd.left; cur = nd.right; }
            else { stack[sp++] = nd.right; cur = nd.left; }
        } else if (ha) cur = nd.left;
        else if (hb) cur = nd.right;
        else {
            if (sp == 0) break;
            cur = stack[--sp];
        }
    }
    return best;
}

// ---------------------------------------------------------------------------
// Production traversal.  One ray per lane; the traversal stack lives in LDS as
// stack[level][lane-in-block] (one dword per lane per level: bank = lane mod 32,
// conflict free for both halves of the wave).  Internal nodes are 64-B slabs
// holding both child boxes, fetched as four 16-B loads from one half cache line.
// A single loop walks the TLAS and, below an instance leaf, that instance's BLAS
// with the ray transformed to object space; RT_SENTINEL marks the BLAS bottom.
// ---------------------------------------------------------------------------

#define RT_STACK_SENTINEL 0x7FFFFFFF   // pops back out of a BLAS
#define RT_STACK_EMPTY    0x7FFFFFFE

// Traversal stack of STACK entries per lane: the first RT_LDS_STACK levels live in LDS
// (smem[level][lane]: one dword per lane per level, conflict free), deeper levels --
// rarely reached: a level is only used while BOTH children of that many ancestors were
// hit -- spill to a per-lane private array.  Keeping the LDS part short is what lets
// 5+ waves per SIMD stay resident (16 levels x 256 lanes x 4 B = 16 KiB per block).
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 16
#endif

template <int STACK>
struct StackShape {
    static constexpr int LDSN = STACK < RT_LDS_STACK ? STACK : RT_LDS_STACK;
    static constexpr int SPILL = STACK - LDSN;
};

template <int STACK, int BLOCK>
struct LdsStack {
    static constexpr int LDSN = StackShape<STACK>::LDSN;
    static constexpr int SPILL = StackShape<STACK>::SPILL;
    int *base;      // &smem[threadIdx.x]
    int sp;
    int spill[SPILL > 0 ? SPILL : 1];
    RT_DEV void init(int *smem) { base = smem + threadIdx.x; sp = 0; }
    RT_DEV void push(int v)
    {
        if (SPILL == 0 || sp < LDSN) base[sp * BLOCK] = v;
        else spill[sp - LDSN] = v;
        sp++;
    }
    RT_DEV int pop()
    {
        sp--;
        if (SPILL == 0 || sp < LDSN) return base[sp * BLOCK];
        return spill[sp - LDSN];
    }
};

template <int STACK, int BLOCK>
RT_DEV HitD trace_fast(const SceneDev &sc, const RayD &r, uint32_t flags, int *smem)
{
    HitD best = make_miss(r);
    if (sc.n_inst == 0 || !(r.tmax > r.tmin)) return best;
    const bool first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;

    LdsStack<STACK, BLOCK> st;
    st.init(smem);
    st.push(RT_STACK_EMPTY);

    const RayInv wri = make_inv(r.o, r.d);
    ObjRay cur_ray;                 // ray in the space of the structure being walked
    cur_ray.o = r.o; cur_ray.d = r.d; cur_ray.ri = wri;
    const Slab *slabs = sc.tlas_slabs;
    const InstanceRec *in = nullptr;
    uint32_t ii = RT_NO_HIT;
    bool in_blas = false;
    int node = sc.tlas_root_code;
    bool done = false;

    while (!done) {
        // ---- descend through internal nodes
        while (node >= 0 && node < RT_STACK_EMPTY) {
            const Slab *s = slabs + node;
            const float4 q0 = s->q0, q1 = s->q1, q2 = s->q2, q3 = s->q3;
            float e0, e1;
            const bool h0 = slab_hit(cur_ray.ri, q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, r.tmin, best.t, e0);
            const bool h1 = slab_hit(cur_ray.ri, q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, r.tmin, best.t, e1);
            const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool swap = e1 < e0;
                st.push(swap ? c0 : c1);
                node = swap ? c1 : c0;
            } else if (h0) node = c0;
            else if (h1) node = c1;
            else node = st.pop();
        }
        if (node == RT_STACK_EMPTY) break;
        if (node == RT_STACK_SENTINEL) {
            // leave the BLAS: back to the TLAS in world space
            in_blas = false;
            slabs = sc.tlas_slabs;
            cur_ray.o = r.o; cur_ray.d = r.d; cur_ray.ri = wri;
            node = st.pop();
            continue;
        }
        // ---- leaf
        const uint32_t code = (uint32_t)~node;
        if (!in_blas) {
            ii = code;
            in = sc.inst + ii;
            float e;
            // the instance box was tested as a child of its TLAS parent, except when
            // the TLAS is a single leaf: test it here for that case
            bool enter = true;
            if (sc.n_inst == 1)
                enter = slab_hit(wri, in->wlo[0], in->whi[0], in->wlo[1], in->whi[1], in->wlo[2], in->whi[2], r.tmin, best.t, e);
            if (enter) {
                cur_ray = to_object(*in, r);
                slabs = in->slabs;
                in_blas = true;
                st.push(RT_STACK_SENTINEL);
                node = in->root_code;
            } else node = st.pop();
            continue;
        }
        const uint32_t firstTri = code >> 3, cnt = (code & 7u) + 1u;
        for (uint32_t k = 0; k < cnt; k++) {
            const TriRec *tp = in->tris + firstTri + k;
            const float4 a = tp->a, b = tp->b, c = tp->c;
            const uint32_t prim = __float_as_uint(c.y);
            if (accept_candidate(*in, ii, prim, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, wri, cur_ray, cull,
                                 best) && first) {
                done = true;
                break;
            }
        }
        if (done) break;
        node = st.pop();
    }
    return best;
}

}  // namespace rtd
