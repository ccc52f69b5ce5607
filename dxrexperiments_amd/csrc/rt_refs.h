// rt_refs.h -- split references (round 5): the validation boxes of long thin triangles.  Normative text: DESIGN.md section 2.1, paragraph S2.5 (the rule)
// and S2.4 (what a box is for); this file and the test oracle (oracle_bvh.h) implement it operation for operation.
//
// A 47 m x 2 cm cable that runs diagonally through a hall has an AABB of 40 x 5 x 25 m: every ray through that box has to test the
// triangle, and 12,000 such triangles (4 % of the stress scene, tests/test_gpu_stress_scene.py) cost three times what the other 260,000
// cost together (profiles/r05/stress_parts.txt).  The production tree therefore holds such a triangle as SEVERAL references -- the parts of
// it inside overlapping slabs of its longest axis, each with its own small box -- and the candidate rule of the engine
// reads "a candidate only through ONE OF THE TRIANGLE'S REFERENCE BOXES (S2.4: at t if the box passes over [tmin, t], at the box's entry if the ray only meets it later)".  A triangle that is not split has one
// reference, its own AABB: rounds 1 - 4's rule, bit for bit, and no triangle of any earlier scene is split.  Float slab tests are monotone
// under box inclusion, so the exactness rule stands: the canonical tree over whole triangles, the production tree over references and
// brute force return the same bits.  The rule is DEFINED in the test oracle (oracle_bvh.h) ("Split references"); this file restates it for the
// builder, operation for operation: no fused operations (-ffp-contract=off), IEEE divide and sqrt, min / max that ignore nothing
// (every coordinate is finite or the triangle is not split).
//   split iff  every coordinate is finite,  L = longest AABB extent > min_len (= the model's longest extent / 512),
//              a2 = |cross(v1 - v0, v2 - v0)| > 0  and  sa = (ex ey + ey ez) + ez ex > 4 a2   (AABB surface over triangle area > 8)
//   pieces     k = min(128, trunc((sa / a2) / 2), trunc(L / min_len)), at least 2 (the references' surface then stays within a few times the triangle's;
//              the first form of the rule -- min(32, sqrt(sa / a2), L / (extent / 256)) -- left the 2.2 M-triangle stress scene 2.5x slower: profiles/r05/ref_rule.txt); piece j = the slab [lo + L (j / k), lo + L ((j + 1) / k)]
//              of the longest axis (ties: x before y before z), widened by (L / k) / 4 either side and cut back to [lo, hi]
//   its box    on the split axis the slab; on the others min / max over the vertices inside the slab and the points where the edges
//              0->1, 1->2, 2->0 cross the slab's two planes; every side moved out by pad = 2^-18 x the triangle's largest |coordinate|
// No reference counterpart: the Fallback Layer's builder is not in the checkout (BottomLevelASGenerator.cpp:333 only calls it).
#pragma once

#include "rt_device_math.h"
#include "rt_internal.h"

namespace rtd {

// S2-RULE-BEGIN (tests/test_s2_truth.py hashes the code between the marks: a change here needs new bounds in tests/golden/s2_bounds.json)
#define RT_REF_MAX_PIECES 128u

RT_DEV float ref_min2(float a, float b) { return a < b ? a : b; }
RT_DEV float ref_max2(float a, float b) { return a > b ? a : b; }

// pieces of the triangle (1: not split) and the axis they are cut along
RT_DEV uint32_t ref_pieces(const float p[3][3], float min_len, int &axis)
{
    axis = 0;
    for (int i = 0; i < 3; i++)
        for (int q = 0; q < 3; q++)
            if (!(p[i][q] - p[i][q] == 0.0f)) return 1u;
    float lo[3], hi[3];
    for (int q = 0; q < 3; q++) {
        lo[q] = ref_min2(ref_min2(ref_min2(__uint_as_float(0x7f800000u), p[0][q]), p[1][q]), p[2][q]);      // (box_grow from the empty box, as the oracle's tri_box)
        hi[q] = ref_max2(ref_max2(ref_max2(__uint_as_float(0xff800000u), p[0][q]), p[1][q]), p[2][q]);
    }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    const float L = axis == 0 ? ex : (axis == 1 ? ey : ez);
    if (!(L > min_len)) return 1u;
    const float e1x = p[1][0] - p[0][0], e1y = p[1][1] - p[0][1], e1z = p[1][2] - p[0][2];
    const float e2x = p[2][0] - p[0][0], e2y = p[2][1] - p[0][1], e2z = p[2][2] - p[0][2];
    const float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
    const float a2 = __fsqrt_rn((cx * cx + cy * cy) + cz * cz);
    if (!(a2 > 0.0f)) return 1u;
    const float sa = (ex * ey + ey * ez) + ez * ex;
    if (!(sa > 4.0f * a2)) return 1u;
    const float kf = __fdiv_rn(sa, a2) * 0.5f, lf = __fdiv_rn(L, min_len);
    uint32_t k = kf >= (float)RT_REF_MAX_PIECES ? RT_REF_MAX_PIECES : (uint32_t)kf;
    const uint32_t kl = lf >= (float)RT_REF_MAX_PIECES ? RT_REF_MAX_PIECES : (uint32_t)lf;
    if (kl < k) k = kl;
    return k < 2u ? 1u : k;
}

// box of piece j of k along `axis`: out[0..2] = lo, out[3..5] = hi
RT_DEV void ref_box(const float p[3][3], int axis, uint32_t k, uint32_t j, float out[6])
{
    const int u = (axis + 1) % 3, w = (axis + 2) % 3;
    const float lo = ref_min2(ref_min2(p[0][axis], p[1][axis]), p[2][axis]), hi = ref_max2(ref_max2(p[0][axis], p[1][axis]), p[2][axis]);
    const float L = hi - lo;
    const float ov = __fdiv_rn(L, (float)k) * 0.25f;
    const float s0 = j == 0u ? lo : ref_max2(lo, (lo + L * __fdiv_rn((float)j, (float)k)) - ov);
    const float s1 = j + 1u == k ? hi : ref_min2(hi, (lo + L * __fdiv_rn((float)(j + 1u), (float)k)) + ov);
    const float inf = __uint_as_float(0x7f800000u);
    float ulo = inf, uhi = -inf, wlo = inf, whi = -inf, maxabs = 0.0f;
    for (int i = 0; i < 3; i++) {
        for (int q = 0; q < 3; q++) maxabs = ref_max2(maxabs, p[i][q] < 0.0f ? -p[i][q] : p[i][q]);
        if (p[i][axis] >= s0 && p[i][axis] <= s1) {
            ulo = ref_min2(ulo, p[i][u]); uhi = ref_max2(uhi, p[i][u]);
            wlo = ref_min2(wlo, p[i][w]); whi = ref_max2(whi, p[i][w]);
        }
    }
    for (int e = 0; e < 3; e++) {
        const float *pi = p[e], *pj = p[(e + 1) % 3];
        for (int side = 0; side < 2; side++) {
            const float c = side == 0 ? s0 : s1;
            if ((pi[axis] < c && pj[axis] > c) || (pi[axis] > c && pj[axis] < c)) {
                const float t = __fdiv_rn(c - pi[axis], pj[axis] - pi[axis]);
                const float yu = pi[u] + (pj[u] - pi[u]) * t, yw = pi[w] + (pj[w] - pi[w]) * t;
                ulo = ref_min2(ulo, yu); uhi = ref_max2(uhi, yu);
                wlo = ref_min2(wlo, yw); whi = ref_max2(whi, yw);
            }
        }
    }
    if (!(ulo <= uhi) || !(wlo <= whi)) {
        ulo = ref_min2(ref_min2(p[0][u], p[1][u]), p[2][u]); uhi = ref_max2(ref_max2(p[0][u], p[1][u]), p[2][u]);
        wlo = ref_min2(ref_min2(p[0][w], p[1][w]), p[2][w]); whi = ref_max2(ref_max2(p[0][w], p[1][w]), p[2][w]);
    }
    const float pad = maxabs * 3.814697265625e-06f;        // 2^-18
    out[axis] = s0 - pad; out[3 + axis] = s1 + pad;
    out[u] = ulo - pad; out[3 + u] = uhi + pad;
    out[w] = wlo - pad; out[3 + w] = whi + pad;
}

// S2-RULE-END
}  // namespace rtd
