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
// reference checkout), normative text DESIGN.md section 2.1: slab box test with slack
// 1+2^-16 (S2.2), Moller-Trumbore (S2.3), and the box clause against the triangle's own
// AABB / reference boxes / the instance's world AABB (S2.4) that makes BVH order
// irrelevant to the result (S2.7) and is measured against the test infrastructure's float64 brute force, truth64.h (S2.8).
#pragma once

#include "rt_device_math.h"
#include "rt_internal.h"

namespace rtd {

// S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json)
#define RT_SLAB_SLACK 1.0000152587890625f

// S2-RULE-END
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

// S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json)
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
// (Plane distances as one fma each, t = fma(plane, inv, -o*inv), were built and measured in round 2: 13.5 % fewer VALU
// instructions, no change in stage time -- the walk is bound by distinct cache lines per ray, not by VALU issue -- and
// the rounded -o*inv leaves an ABSOLUTE error that lets rays grazing a tessellated wall pass every box of that wall
// (4 ms tails) unless every axis carries its own margin.  profiles/r02/fma_slab_experiment.md.)

// S2-RULE-END
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

// S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json)
// The candidate rule's box clause (the oracle's box_clause in oracle_bvh.h is its twin).  Moller-Trumbore said the
// ray meets the triangle at tt; the box holds (a part of) the triangle.  Passing over [tmin, tt]: the candidate stands at tt.  Failing that but
// meeting the box inside (tmin, tmax) -- on a sliver the fp32 tt can lie a little before the ray enters the box -- it stands at the entry
// distance (round 6; rounds 1 - 5 rejected it and the ray went through the triangle).  Either way every traversal that still looks for hits at
// that distance or beyond reaches the box: slab tests are monotone under box inclusion and in their upper limit.
RT_DEV bool box_clause(const RayInv &ri, float lx, float hx, float ly, float hy, float lz, float hz, float tmin, float tmax, float tt, float &at)
{
    float e;
    if (slab_hit(ri, lx, hx, ly, hy, lz, hz, tmin, tt, e)) { at = tt; return true; }
    if (!slab_hit(ri, lx, hx, ly, hy, lz, hz, tmin, tmax, e) || !(e < tmax)) return false;
    at = e;
    return true;
}

// Moller-Trumbore + the box clause against the triangle's own AABB -- or, for a triangle the builder holds as several
// references (rt_refs.h; round 5), against ITS REFERENCE BOXES: ref = 1: this visit's own box (own6; the walk visits the others on its
// own and the nearest answer wins), ref = 2: the nearest of what the triangle's n_refs boxes say (refs6).  ref = 0 is every triangle of rounds 1 - 4.
RT_DEV bool tri_candidate(f3 o, f3 d, const RayInv &ri, float tmin, float tmax, f3 v0, f3 v1, f3 v2, bool cull,
                          float &t, float &u, float &v, uint32_t ref = 0u, const float *own6 = nullptr, const float *refs6 = nullptr, uint32_t n_refs = 0u)
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
    float at;
    if (ref == 1u) {
        if (!box_clause(ri, own6[0], own6[3], own6[1], own6[4], own6[2], own6[5], tmin, tmax, tt, at)) return false;
    } else if (ref == 2u) {
        bool ok = false;
        at = tt;
        for (uint32_t k = 0; k < n_refs; k++) {
            const float *b = refs6 + 6 * (size_t)k;
            float a;
            if (!box_clause(ri, b[0], b[3], b[1], b[4], b[2], b[5], tmin, tmax, tt, a)) continue;
            if (!ok || a < at) at = a;
            ok = true;
            if (at == tt) break;
        }
        if (!ok) return false;
    } else if (!box_clause(ri, fmin2(fmin2(v0.x, v1.x), v2.x), fmax2(fmax2(v0.x, v1.x), v2.x),
                           fmin2(fmin2(v0.y, v1.y), v2.y), fmax2(fmax2(v0.y, v1.y), v2.y),
                           fmin2(fmin2(v0.z, v1.z), v2.z), fmax2(fmax2(v0.z, v1.z), v2.z), tmin, tmax, tt, at))
        return false;
    t = at; u = uu; v = vv;
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

// S2-RULE-END
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

// candidate (v0,v1,v2,prim) of instance ii against the running best.  REFS: how split references (rt_refs.h) are looked up --
//   0  the scene has none: the code of rounds 1 - 4 (the traversal kernels sit exactly on their register budgets: the instantiations for
//      scenes without split triangles must not carry the lookup -- with it the seven-wave kernels spilled 20 - 36 more bytes and the bench
//      scene's frame took 9 % longer, profiles/r05/split_refs.txt);
//   1  production walk of a scene with split triangles: the record says (ref = TriRec::c.z bits) whether it is one reference of several
//      (its box: rec_boxes[rec]) or, in a layout that holds a split triangle once, that all its boxes are to be tried (by primitive);
//   2  canonical walk: by primitive.
template <int REFS = 0>
RT_DEV bool accept_candidate(const InstanceRec &in, uint32_t ii, uint32_t prim, f3 v0, f3 v1, f3 v2, const RayD &r,
                             const RayInv &wri, const ObjRay &orr, bool cull, HitD &best, uint32_t ref = 0u, uint32_t rec = RT_NO_HIT)
{
    float t, u, v;
    const float *own6 = nullptr, *refs6 = nullptr;
    uint32_t n_refs = 0u;
    if (REFS == 0) ref = 0u;
    if (REFS == 2) ref = in.ref_off && in.ref_off[prim + 1] - in.ref_off[prim] > 1u ? 2u : 0u;
    if (REFS != 0) {
        if (ref == 1u) own6 = in.rec_boxes + 6 * (size_t)rec;
        else if (ref == 2u) { const uint32_t first = in.ref_off[prim]; refs6 = in.ref_boxes + 6 * (size_t)first; n_refs = in.ref_off[prim + 1] - first; }
    }
    if (!tri_candidate(orr.o, orr.d, orr.ri, r.tmin, r.tmax, v0, v1, v2, cull, t, u, v, ref, own6, refs6, n_refs)) return false;
    if (!(in.flags & RT_INST_IDENTITY)) {        // the instance's world box: the same clause, on what the triangle's said
        if (!box_clause(wri, in.wlo[0], in.whi[0], in.wlo[1], in.whi[1], in.wlo[2], in.whi[2], r.tmin, r.tmax, t, t)) return false;
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
            if (accept_candidate<2>(in, ii, prim, mk3(p0.x, p0.y, p0.z), mk3(p1.x, p1.y, p1.z), mk3(p2.x, p2.y, p2.z), r, wri, orr,
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

}  // namespace rtd
