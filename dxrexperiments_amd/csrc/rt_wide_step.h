// rt_wide_step.h -- one traversal step on a four-wide quantised node (WNode, rt_internal.h) and the per-lane stack it
// pushes to: the arithmetic half of the persistent engine in rt_trace_wave.h, which includes this file.
#pragma once

namespace rtd {

// The traversal stack: STACK rows per lane in LDS (stk[row * BLOCK], one dword per lane per row, bank =
// lane mod 32: conflict free), rows beyond that in global memory (deep[(row - STACK) * threads + thread]).
// The LDS rows are sized for occupancy, not for the deepest possible walk: Sponza-class rays never hold
// more than 15 entries although the tree is 29 levels deep, so the global rows are
// correctness insurance that is rarely or never touched; the hot loop runs only while sp < STACK and
// is pure LDS, a second copy of the step (DEEP) serves the lanes above that.
template <int STACK, int BLOCK>
struct LaneStack {
    int *lds;            // smem + threadIdx.x
    int *deep;           // global rows of this thread (nullptr when the tree cannot need them)
    uint32_t threads;    // threads of the launch: stride between global rows
    RT_DEV int read(int row) const { return row < STACK ? lds[row * BLOCK] : deep[(size_t)(row - STACK) * threads]; }
    RT_DEV void write(int row, int v) const
    {
        if (row < STACK) lds[row * BLOCK] = v;
        else deep[(size_t)(row - STACK) * threads] = v;
    }
};

#if RT_WIDE == 8      // round 3's experiment, not part of the default build
#include "experiments/rt_wide8_step.h"
#else
// One step on a wide node: slab-test the four children, enter the nearest hit one (any-hit: the first in slot order),
// push the other hit ones (farthest first), pop if none is hit.  The lane's stack pointer may rise by three, so the
// pure-LDS instantiation (DEEP = false) is only called with sp <= STACK - 3.
//
// The four boxes are tested in the node's quantised frame: a plane at grid step q lies at origin + q * scale, so its
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
    v4f q0, q1, q2, q3;
#if RT_LOAD_PRIO
    __builtin_amdgcn_s_setprio(RT_LOAD_PRIO);
#endif
    if ((uint32_t)node < top_lim) {
        // the top of the tree is LDS resident: every ray walks it
        const v4f *t = (const v4f *)(top + (node << 4));
        q0 = t[0]; q1 = t[1]; q2 = t[2]; q3 = t[3];
    } else {
        // 32-bit byte offset from the (wave-uniform in single-level walks) node base: SGPR base + VGPR offset addressing
        const char *nd = (const char *)nodes + ((uint32_t)node << 6);
        q0 = ldg16(nd, 0); q1 = ldg16(nd, 16); q2 = ldg16(nd, 32); q3 = ldg16(nd, 48);
    }
#if RT_LOAD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    const uint32_t lx = __float_as_uint(q1.x), hx = __float_as_uint(q1.y), ly = __float_as_uint(q1.z), hy = __float_as_uint(q1.w);
    const uint32_t lz = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
    int c[4] = {__float_as_int(q3.x), __float_as_int(q3.y), __float_as_int(q3.z), __float_as_int(q3.w)};
    float d[4];
    bool h[4];
    // A ray that runs (almost) inside an axis-aligned plane -- a direction component below 2^-16, about one ray in 10^4 --
    // needs that axis resolved exactly: it lies IN a tessellated wall, only the exact plane distance (o is within an ulp
    // of the wall) tells which of the wall's boxes it is in, and with the margin D it would walk all of them (measured:
    // walks of thousands of nodes, a 2 ms tail on a 1 ms stage).  Such a lane decodes the boxes and runs the canonical
    // slab test itself; a zero component (reciprocal +-inf) goes the same way and is treated exactly as the definition says.
    const float steep = fmax2(fmax2(__builtin_fabsf(ri.inv.x), __builtin_fabsf(ri.inv.y)), __builtin_fabsf(ri.inv.z));
    if (!(steep <= 65536.0f)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // plane = fma(q, scale, origin): the expression rt_bvh_wide.hip verified the containment with
            const float blx = __builtin_fmaf((float)((lx >> (8 * k)) & 0xffu), q0.w, q0.x), bhx = __builtin_fmaf((float)((hx >> (8 * k)) & 0xffu), q0.w, q0.x);
            const float bly = __builtin_fmaf((float)((ly >> (8 * k)) & 0xffu), q2.z, q0.y), bhy = __builtin_fmaf((float)((hy >> (8 * k)) & 0xffu), q2.z, q0.y);
            const float blz = __builtin_fmaf((float)((lz >> (8 * k)) & 0xffu), q2.w, q0.z), bhz = __builtin_fmaf((float)((hz >> (8 * k)) & 0xffu), q2.w, q0.z);
            float e;
            h[k] = slab_hit(ri, blx, bhx, bly, bhy, blz, bhz, tmin, tbest, e) && c[k] != RT_NODE_NONE;
            d[k] = h[k] ? e : __uint_as_float(0x7f800000u);
        }
    } else {
        const float ax = q0.w * ri.inv.x, ay = q2.z * ri.inv.y, az = q2.w * ri.inv.z;
        const float bx = (q0.x - ri.o.x) * ri.inv.x, by = (q0.y - ri.o.y) * ri.inv.y, bz = (q0.z - ri.o.z) * ri.inv.z;
        const float k20 = 9.5367431640625e-07f;      // 2^-20
        const float dx = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.x), __builtin_fmaf(255.0f, q0.w, __builtin_fabsf(q0.x)), __builtin_fabsf(bx)), k20, 1.0e-37f);
        const float dy = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.y), __builtin_fmaf(255.0f, q2.z, __builtin_fabsf(q0.y)), __builtin_fabsf(by)), k20, 1.0e-37f);
        const float dz = __builtin_fmaf(__builtin_fmaf(__builtin_fabsf(ri.inv.z), __builtin_fmaf(255.0f, q2.w, __builtin_fabsf(q0.z)), __builtin_fabsf(bz)), k20, 1.0e-37f);
        const float bnx = bx - dx, bfx = bx + dx, bny = by - dy, bfy = by + dy, bnz = bz - dz, bfz = bz + dz;
        // near / far plane bytes by the sign of the direction
        const bool ngx = ri.inv.x < 0.0f, ngy = ri.inv.y < 0.0f, ngz = ri.inv.z < 0.0f;
        const uint32_t nx4 = ngx ? hx : lx, fx4 = ngx ? lx : hx, ny4 = ngy ? hy : ly, fy4 = ngy ? ly : hy, nz4 = ngz ? hz : lz, fz4 = ngz ? lz : hz;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float nx = __builtin_fmaf((float)((nx4 >> (8 * k)) & 0xffu), ax, bnx), fx = __builtin_fmaf((float)((fx4 >> (8 * k)) & 0xffu), ax, bfx);
            const float ny = __builtin_fmaf((float)((ny4 >> (8 * k)) & 0xffu), ay, bny), fy = __builtin_fmaf((float)((fy4 >> (8 * k)) & 0xffu), ay, bfy);
            const float nz = __builtin_fmaf((float)((nz4 >> (8 * k)) & 0xffu), az, bnz), fz = __builtin_fmaf((float)((fz4 >> (8 * k)) & 0xffu), az, bfz);
            const float lo = fmax2(fmax2(nx, ny), fmax2(nz, tmin));
            const float hi = fmin2(fmin2(fx, fy), fmin2(fz, tbest));
            h[k] = lo <= hi * RT_SLAB_SLACK && c[k] != RT_NODE_NONE;
            d[k] = h[k] ? lo : __uint_as_float(0x7f800000u);
        }
    }
    const int below = sp > 0 ? sp - 1 : 0;
    const int under = DEEP ? st.read(below) : st.lds[below * BLOCK];       // speculative pop (unconditional read)
    bool p3, p2, p1, any;
    if (ANYHIT) {
        // no order needed: the first hit ends the ray.  Enter the first hit slot, push every later hit one.
        any = h[0] || h[1] || h[2] || h[3];
        p3 = h[3] && (h[0] || h[1] || h[2]);
        p2 = h[2] && (h[0] || h[1]);
        p1 = h[1] && h[0];
        c[0] = h[0] ? c[0] : (h[1] ? c[1] : (h[2] ? c[2] : c[3]));
    } else {
        // sort the four (entry, code) pairs by entry distance; misses carry +inf and end up last
        // (only bringing the nearest to the front -- three exchanges -- costs 1.3 % more steps and the same time)
#define RT_CE(i, j) { const bool sw = d[j] < d[i]; const float td = sw ? d[j] : d[i]; d[j] = sw ? d[i] : d[j]; d[i] = td; \
                      const int tc = sw ? c[j] : c[i]; c[j] = sw ? c[i] : c[j]; c[i] = tc; }
        RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
        const float inf = __uint_as_float(0x7f800000u);
        any = d[0] < inf; p1 = d[1] < inf; p2 = d[2] < inf; p3 = d[3] < inf;
    }
    if (p3) { if (DEEP) st.write(sp, c[3]); else st.lds[sp * BLOCK] = c[3]; sp++; }
    if (p2) { if (DEEP) st.write(sp, c[2]); else st.lds[sp * BLOCK] = c[2]; sp++; }
    if (p1) { if (DEEP) st.write(sp, c[1]); else st.lds[sp * BLOCK] = c[1]; sp++; }
    if (any) node = c[0];
    else { node = sp > 0 ? under : RT_NODE_EMPTY; sp = below; }
}
#endif

}  // namespace rtd
