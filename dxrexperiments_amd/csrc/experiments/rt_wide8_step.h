// experiments/rt_wide8_step.h -- round 3's EXPERIMENT, not part of the default build: the node step for eight-wide 128-B nodes
// (-DRT_WIDE=8; rt_trace_wave.h includes this file only then).  Measured 29 - 39 % SLOWER than the production four-wide step
// (a third fewer steps per ray, twice the instructions per step, two waves per SIMD fewer): profiles/r03/wide8_experiment.md.
// Kept so that the measurement stays reproducible; the builder side of the layout is the RT_WIDE == 8 branch of rt_bvh_wide.hip.
// (Included by rt_trace_wave.h INSIDE namespace rtd, after ldg16 / LaneStack / RayInv are defined.)
#pragma once

// One step on a wide node: slab-test the eight children, enter the hit one that comes first in the ray's octant order,
// push the other hit ones so that the next in that order is on top, pop if none is hit.  The lane's stack pointer may rise
// by seven, so the pure-LDS instantiation (DEEP = false) is only called with sp <= STACK - 7.
//
// Order: the builder has put the children into slots by where they lie in the node (rt_bvh_wide.hip); for a ray whose
// direction is negative on the axes of `oct`, (slot XOR oct) ascending is a front-to-back order.  Nothing is sorted by
// distance: a hit child sets bit (slot ^ oct) of a mask, the lowest bit is entered, and the stack position of every other
// one is the number of hit children that come after it.  (Any-hit rays use the same order: near occluders first.)
//
// The eight boxes are tested in the node's quantised frame: a plane at grid step q lies at origin + q * scale, so its
// distance along the ray is  t(q) = q * A + B  with  A = scale * inv,  B = (origin - o) * inv  per axis -- one cvt and
// one fma per plane instead of decode, subtract, multiply.  This is CULLING arithmetic, not the canonical slab test
// (rt_trace_device.h), so it carries an explicit margin per axis,
//     D = 2^-20 * (|B| + |inv| * (|origin| + 255 * scale)) + 1e-37,
// a bound (with a factor of >8 to spare) on every rounding that separates t(q) from the canonical distance of the
// decoded plane rn(origin + q * scale): the rounding of that plane itself (<= 2^-24 |plane| |inv|), the canonical test's
// own two roundings (<= 2^-23 |t|), and the three roundings here (B twice, the fma once).  Near planes use B - D, far
// planes B + D; which byte is the near plane follows the sign of inv, so no min / max per axis is needed.  Hence
//     canonical test passes on the true child box  =>  it passes on the decoded box (monotone, rt_bvh_wide.hip)
//                                                  =>  this test passes,
// which is all the exactness rule asks of a traversal.  (All reciprocals are finite and at most 2^16 here: steeper rays
// take the exact path inside the step.  An axis the builder could not quantise has an infinite scale and q = 0 planes:
// A is +-inf, t(0) = fma(0, inf, B) is NaN, and max / min ignore a NaN operand -- that axis does not cull.)
template <bool DEEP, bool ANYHIT, int STACK, int BLOCK>
RT_DEV void wide_step(const WNode *nodes, const int *top, uint32_t top_lim, const RayInv &ri, float tmin, float tbest,
                      const LaneStack<STACK, BLOCK> &st, int &node, int &sp)
{
    v4f q0, q1, q2, q3, q4, q5;
    if ((uint32_t)node < top_lim) {
        // the top of the tree is LDS resident: every ray walks it
        const v4f *t = (const v4f *)(top + node * RT_TOP_WORDS);
        q0 = t[0]; q1 = t[1]; q2 = t[2]; q3 = t[3]; q4 = t[4]; q5 = t[5];
    } else {
        // 32-bit byte offset from the (wave-uniform in single-level walks) node base: SGPR base + VGPR offset addressing
        const char *nd = (const char *)nodes + ((uint32_t)node << 7);
        q0 = ldg16(nd, 0); q1 = ldg16(nd, 16); q2 = ldg16(nd, 32); q3 = ldg16(nd, 48); q4 = ldg16(nd, 64); q5 = ldg16(nd, 80);
    }
    const uint32_t meta = __float_as_uint(q0.w);
    const float sx = __uint_as_float((meta & 0xffu) << 23), sy = __uint_as_float((meta & 0xff00u) << 15), sz = __uint_as_float((meta & 0xff0000u) << 7);
    // plane bytes: [axis][lo / hi][slots 0..3 / 4..7]
    const uint32_t lx[2] = {__float_as_uint(q1.x), __float_as_uint(q1.y)}, hx[2] = {__float_as_uint(q1.z), __float_as_uint(q1.w)};
    const uint32_t ly[2] = {__float_as_uint(q2.x), __float_as_uint(q2.y)}, hy[2] = {__float_as_uint(q2.z), __float_as_uint(q2.w)};
    const uint32_t lz[2] = {__float_as_uint(q3.x), __float_as_uint(q3.y)}, hz[2] = {__float_as_uint(q3.z), __float_as_uint(q3.w)};
    const int c[8] = {__float_as_int(q4.x), __float_as_int(q4.y), __float_as_int(q4.z), __float_as_int(q4.w),
                      __float_as_int(q5.x), __float_as_int(q5.y), __float_as_int(q5.z), __float_as_int(q5.w)};
    const uint32_t oct = (__float_as_uint(ri.inv.x) >> 31) | ((__float_as_uint(ri.inv.y) >> 31) << 1) | ((__float_as_uint(ri.inv.z) >> 31) << 2);
    uint32_t hits = 0;          // bit `slot` for every hit child
    // A ray that runs (almost) inside an axis-aligned plane -- a direction component below 2^-16, about one ray in 10^4 --
    // needs that axis resolved exactly: it lies IN a tessellated wall, only the exact plane distance (o is within an ulp
    // of the wall) tells which of the wall's boxes it is in, and with the margin D it would walk all of them (measured:
    // walks of thousands of nodes, a 2 ms tail on a 1 ms stage).  Such a lane decodes the boxes and runs the canonical
    // slab test itself; a zero component (reciprocal +-inf) goes the same way and is treated exactly as the definition says.
    const float steep = fmax2(fmax2(__builtin_fabsf(ri.inv.x), __builtin_fabsf(ri.inv.y)), __builtin_fabsf(ri.inv.z));
    if (!(steep <= 65536.0f)) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int w = k >> 2, sh = 8 * (k & 3);
            // plane = fma(q, scale, origin): the expression rt_bvh_wide.hip verified the containment with
            const float blx = __builtin_fmaf((float)((lx[w] >> sh) & 0xffu), sx, q0.x), bhx = __builtin_fmaf((float)((hx[w] >> sh) & 0xffu), sx, q0.x);
            const float bly = __builtin_fmaf((float)((ly[w] >> sh) & 0xffu), sy, q0.y), bhy = __builtin_fmaf((float)((hy[w] >> sh) & 0xffu), sy, q0.y);
            const float blz = __builtin_fmaf((float)((lz[w] >> sh) & 0xffu), sz, q0.z), bhz = __builtin_fmaf((float)((hz[w] >> sh) & 0xffu), sz, q0.z);
            float e;
            if (slab_hit(ri, blx, bhx, bly, bhy, blz, bhz, tmin, tbest, e)) hits |= 1u << k;
        }
    } else {
        const float ax = sx * ri.inv.x, ay = sy * ri.inv.y, az = sz * ri.inv.z;
        const float bx = (q0.x - ri.o.x) * ri.inv.x, by = (q0.y - ri.o.y) * ri.inv.y, bz = (q0.z - ri.o.z) * ri.inv.z;
        const float k20 = 9.5367431640625e-07f;      // 2^-20
        const float dx = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.x), __builtin_fmaf(255.0f, sx, __builtin_fabsf(q0.x)), __builtin_fabsf(bx)), k20, 1.0e-37f);
        const float dy = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.y), __builtin_fmaf(255.0f, sy, __builtin_fabsf(q0.y)), __builtin_fabsf(by)), k20, 1.0e-37f);
        const float dz = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.z), __builtin_fmaf(255.0f, sz, __builtin_fabsf(q0.z)), __builtin_fabsf(bz)), k20, 1.0e-37f);
        const float bnx = bx - dx, bfx = bx + dx, bny = by - dy, bfy = by + dy, bnz = bz - dz, bfz = bz + dz;
        // near / far plane bytes by the sign of the direction
        const bool ngx = (oct & 1u) != 0u, ngy = (oct & 2u) != 0u, ngz = (oct & 4u) != 0u;
#pragma unroll
        for (int w = 0; w < 2; w++) {
            const uint32_t nx4 = ngx ? hx[w] : lx[w], fx4 = ngx ? lx[w] : hx[w], ny4 = ngy ? hy[w] : ly[w], fy4 = ngy ? ly[w] : hy[w],
                           nz4 = ngz ? hz[w] : lz[w], fz4 = ngz ? lz[w] : hz[w];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float nx = __builtin_fmaf((float)((nx4 >> (8 * j)) & 0xffu), ax, bnx), fx = __builtin_fmaf((float)((fx4 >> (8 * j)) & 0xffu), ax, bfx);
                const float ny = __builtin_fmaf((float)((ny4 >> (8 * j)) & 0xffu), ay, bny), fy = __builtin_fmaf((float)((fy4 >> (8 * j)) & 0xffu), ay, bfy);
                const float nz = __builtin_fmaf((float)((nz4 >> (8 * j)) & 0xffu), az, bnz), fz = __builtin_fmaf((float)((fz4 >> (8 * j)) & 0xffu), az, bfz);
                const float lo = fmax2(fmax2(nx, ny), fmax2(nz, tmin));
                const float hi = fmin2(fmin2(fx, fy), fmin2(fz, tbest));
                if (lo <= hi * RT_SLAB_SLACK) hits |= 1u << (4 * w + j);
            }
        }
    }
    hits &= meta >> 24;         // slots in use
    // the same in octant order: bit p of the permuted mask = bit (p ^ oct) of the slot mask
    hits = (oct & 1u) ? ((hits & 0x55u) << 1) | ((hits & 0xaau) >> 1) : hits;
    hits = (oct & 2u) ? ((hits & 0x33u) << 2) | ((hits & 0xccu) >> 2) : hits;
    hits = (oct & 4u) ? ((hits & 0x0fu) << 4) | ((hits & 0xf0u) >> 4) : hits;
    const int below = sp > 0 ? sp - 1 : 0;
    const int under = DEEP ? st.read(below) : st.lds[below * BLOCK];       // speculative pop (unconditional read)
    if (hits != 0u) {
        const uint32_t enter = (uint32_t)__builtin_ctz(hits) ^ oct;        // slot of the child to enter
        const uint32_t rest = hits & (hits - 1u);                          // the children to push
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (enter == (uint32_t)k) node = c[k];
            const uint32_t above = rest >> ((uint32_t)k ^ oct);            // bit 0: this child is pushed; higher bits: the pushed ones that come after it
            if (above & 1u) {
                const int row = sp + __popc(above >> 1);                   // the later in the order, the deeper in the stack
                if (DEEP) st.write(row, c[k]); else st.lds[row * BLOCK] = c[k];
            }
        }
        sp += __popc(rest);
    } else {
        node = sp > 0 ? under : RT_NODE_EMPTY;
        sp = below;
    }
}
