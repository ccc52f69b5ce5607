/*
 * truth64.h -- TEST INFRASTRUCTURE ONLY: the geometric truth of TraceRay, independent of every box.
 *
 * The reference calls TraceRay with DXR semantics (assets/shaders/ProgressiveRaytracing.hlsl:34,53, RaytracingCommon.hlsli:84-96) on
 * geometry that is OPAQUE (libs/DXRFramework/Helpers/BottomLevelASGenerator.h:127): a ray that geometrically passes through a triangle hits
 * it.  This file says what "geometrically" means, without asking any acceleration structure: float64 Moller-Trumbore over EVERY instance x
 * EVERY triangle -- no AABB, no reference box, no slab test, no BVH, nothing of oracle_bvh.h but the scene's arrays -- with only the rules
 * the DXR functional spec gives:
 *     TMin < t < TMax (exclusive);   u >= 0, v >= 0, u + v <= 1;
 *     RAY_FLAG_CULL_BACK_FACING_TRIANGLES: front face <=> det > 0 in OBJECT space (winding is judged before the instance transform);
 *     closest hit: smallest t, equal t -> smaller (instance, primitive);   ACCEPT_FIRST_HIT: any accepted triangle.
 * Instances: the ray goes to object space through the float64 inverse of the instance's fp32 3x4 matrix (t is the same number in both
 * spaces).  The engine's own TraceRay (oracle_bvh.h, fp32, with its candidate rule) is MEASURED against this, and the measured rates are
 * committed as bounds (tests/golden/s2_bounds.json, tests/test_s2_truth.py): the definition the kernels reproduce bit for bit may not
 * drift away from geometry unnoticed.  PARITY UNPINNED like everything about TraceRay here: the Fallback Layer's arithmetic is not in the
 * checkout; this is the functional spec's statement of it.
 */
#ifndef ORACLE_TRUTH64_H
#define ORACLE_TRUTH64_H

#include <stdint.h>
#include <vector>

namespace truth64 {

struct Hit { double t, u, v; uint32_t prim, inst; };       /* inst == 0xFFFFFFFF: miss */

struct Tri { double v0[3], e1[3], e2[3]; };

struct Model { std::vector<Tri> tris; };

struct Instance { uint32_t model; double inv[12]; };        /* world -> object, float64 */

struct Scene {
    std::vector<Model> models;
    std::vector<Instance> inst;
};

static inline void invert3x4(const float m[12], double o[12])
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
    const double id = 1.0 / (a * A + b * B + c * C);
    o[0] = A * id; o[1] = (c * h - b * i) * id; o[2]  = (b * f - c * e) * id;
    o[4] = B * id; o[5] = (a * i - c * g) * id; o[6]  = (c * d - a * f) * id;
    o[8] = C * id; o[9] = (b * g - a * h) * id; o[10] = (a * e - b * d) * id;
    for (int r = 0; r < 3; r++) o[4 * r + 3] = -(o[4 * r] * (double)m[3] + o[4 * r + 1] * (double)m[7] + o[4 * r + 2] * (double)m[11]);
}

/* positions: nv x (stride floats), the first three of each are x y z; idx: 3 per triangle */
static inline void add_model(Scene &s, const float *positions, size_t stride, const uint32_t *idx, uint32_t ntris)
{
    Model m;
    m.tris.resize(ntris);
    for (uint32_t p = 0; p < ntris; p++) {
        const float *a = positions + stride * idx[3 * p], *b = positions + stride * idx[3 * p + 1], *c = positions + stride * idx[3 * p + 2];
        for (int k = 0; k < 3; k++) {
            m.tris[p].v0[k] = a[k];
            m.tris[p].e1[k] = (double)b[k] - (double)a[k];
            m.tris[p].e2[k] = (double)c[k] - (double)a[k];
        }
    }
    s.models.push_back(std::move(m));
}

static inline void add_instance(Scene &s, uint32_t model, const float m[12])
{
    Instance in;
    in.model = model;
    invert3x4(m, in.inv);
    s.inst.push_back(in);
}

static inline Hit trace(const Scene &s, const float o_[3], float tmin_, const float d_[3], float tmax_, bool cull_back, bool first)
{
    Hit best = { (double)tmax_, 0.0, 0.0, 0xFFFFFFFFu, 0xFFFFFFFFu };
    const double tmin = tmin_, tmax = tmax_;
    for (uint32_t ii = 0; ii < s.inst.size(); ii++) {
        const Instance &in = s.inst[ii];
        const double *M = in.inv;
        const double wo[3] = {o_[0], o_[1], o_[2]}, wd[3] = {d_[0], d_[1], d_[2]};
        double o[3], d[3];
        for (int r = 0; r < 3; r++) {
            o[r] = M[4 * r] * wo[0] + M[4 * r + 1] * wo[1] + M[4 * r + 2] * wo[2] + M[4 * r + 3];
            d[r] = M[4 * r] * wd[0] + M[4 * r + 1] * wd[1] + M[4 * r + 2] * wd[2];
        }
        const std::vector<Tri> &tr = s.models[in.model].tris;
        const uint32_t n = (uint32_t)tr.size();
        for (uint32_t p = 0; p < n; p++) {
            const Tri &T = tr[p];
            const double px = d[1] * T.e2[2] - d[2] * T.e2[1], py = d[2] * T.e2[0] - d[0] * T.e2[2], pz = d[0] * T.e2[1] - d[1] * T.e2[0];
            const double det = T.e1[0] * px + T.e1[1] * py + T.e1[2] * pz;
            if (cull_back ? !(det > 0.0) : (det == 0.0 || det != det)) continue;
            const double inv = 1.0 / det;
            const double tx = o[0] - T.v0[0], ty = o[1] - T.v0[1], tz = o[2] - T.v0[2];
            const double u = (tx * px + ty * py + tz * pz) * inv;
            if (!(u >= 0.0) || u > 1.0) continue;
            const double qx = ty * T.e1[2] - tz * T.e1[1], qy = tz * T.e1[0] - tx * T.e1[2], qz = tx * T.e1[1] - ty * T.e1[0];
            const double v = (d[0] * qx + d[1] * qy + d[2] * qz) * inv;
            if (!(v >= 0.0) || !(u + v <= 1.0)) continue;
            const double t = (T.e2[0] * qx + T.e2[1] * qy + T.e2[2] * qz) * inv;
            if (!(t > tmin) || !(t < tmax)) continue;
            if (best.inst != 0xFFFFFFFFu && !(t < best.t)) continue;      /* (instances and primitives ascend: an equal t keeps the smaller pair) */
            best.t = t; best.u = u; best.v = v; best.prim = p; best.inst = ii;
            if (first) return best;
        }
    }
    return best;
}

}  // namespace truth64

#endif
