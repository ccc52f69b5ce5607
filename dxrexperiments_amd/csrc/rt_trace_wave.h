// rt_trace_wave.h -- production TraceRay for gfx950: a persistent, wave64-wide
// traversal engine.
//
// Design (MI355X-first; nothing here comes from the Fallback Layer, whose source is
// not in the reference checkout):
//   * one ray per lane, 4 waves per workgroup, as many workgroups as stay resident
//     (persistent).  A wave takes 64-ray chunks of its queue -- through one of 32
//     counters per launch, shared by a group of workgroups on one XCD -- and refills
//     its idle lanes from the chunk once REFILL of them are idle (ballot + popcount
//     prefix sum hands out the indices): on wave64, incoherent rays otherwise leave
//     most of a wave's 64 lanes parked while the longest ray finishes;
//   * "while-while" order with a straggler exit: lanes walk internal nodes until they
//     stand on a leaf; once those still walking are fewer than half of those waiting,
//     the wave turns to the leaves, so the (short) triangle code is not serialised
//     against the (long) node code and neither waits for the slowest lane;
//   * internal nodes are 64-B lines holding FOUR quantised child boxes (WNode, rt_internal.h); four
//     16-B loads of one line through explicitly global (address-space 1) pointers; the first
//     RT_TOP_NODES nodes of the structure a ray starts in (breadth first = the top of the tree) are
//     read from LDS;
//   * a step tests the four boxes in the node's quantised frame (one cvt + one fma per plane), enters
//     the nearest hit child and pushes the others farthest first (closest-hit rays: a five-exchange
//     sorting network on (entry distance, code); any-hit rays: slot order, where the builders put
//     the larger children first); the slot below the top is read speculatively (it only counts if
//     no child is hit);
//   * (-DRT_WIDE=8, round 3: eight children in a 128-B record, visited in the ray's octant order
//     without sorting -- a third fewer steps per ray, twice the instructions per step and two waves
//     per SIMD fewer: 30 - 40 % slower, profiles/r03/wide8_experiment.md; kept as a build option)
//   * the stack is LDS resident, stack[row][lane-in-block]: one dword per lane per
//     row, bank = lane mod 32, conflict free for both halves of a wave; a fixed
//     number of rows whatever the tree, the rare deeper walk continues in global rows.
//
// Exactness: culling uses the same monotone slab test as the oracle and candidates
// are validated exactly as the canonical definition prescribes (rt_trace_device.h),
// so the visiting order, leaf collapsing and ray-to-lane assignment used here cannot
// change any result bit.
#pragma once

#include "rt_trace_device.h"

namespace rtd {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f *gptr4;

// 16-B load through a global (not flat) pointer
RT_DEV v4f ldg16(const void *base, size_t byte_off)
{
    return *(gptr4)((const char *)base + byte_off);
}

#define RT_TMAX_SKIPPED  (-2.0f)        // a queue slot whose ray was emitted but need not be traversed (any-hit launches count them)
#define RT_NODE_EMPTY    0x7FFFFFFE     // popped from an empty stack: the ray is finished
#define RT_NODE_SENTINEL 0x7FFFFFFF     // bottom of a BLAS walk: return to the TLAS

#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 16              // refill a wave once this many of its 64 lanes are idle
#endif
// Issue priority of a wave by what it is doing (s_setprio; 0 = the hardware's default, at which the node steps' arithmetic runs): a wave
// that refills its idle lanes, fetches a node or tests a leaf's triangles is about to put lanes back to work or to send a request
// on its way, and goes ahead of the waves that are in the middle of a step's arithmetic.  Measured (profiles/r03/setprio.txt):
// sets of frames -2 %, frame by frame -1.6 %, 10 M triangles -1.5 %; bit-exact by construction (no instruction changes).
#ifndef RT_REFILL_PRIO
#define RT_REFILL_PRIO 3
#endif
#ifndef RT_LEAF_PRIO
#define RT_LEAF_PRIO 1
#endif
#ifndef RT_LOAD_PRIO
#define RT_LOAD_PRIO 2
#endif
#ifndef RT_POOL_CHUNK
#define RT_POOL_CHUNK 64u               // rays per chunk of the queue a wave takes at a time (32: 3.55, 64: 3.44, 128: 3.47, 256: 3.57 ms/frame)
#endif

#ifndef RT_POOL_GROUPS
#define RT_POOL_GROUPS 32u              // chunk counters per traversal launch (8: 3.08, 32: 3.07, 128: 3.09, 512: 3.12 ms; static: 3.21)
#endif
#define RT_POOL_STRIDE 32u              // words between two counters: one 128-B L2 line each
#ifndef RT_DRAIN_SPLIT
#define RT_DRAIN_SPLIT 0                // 1: once a wave's queue is dry, its idle lanes take pending subtrees off the busy lanes' stacks (see trace_wave;
                                        //    round 3's experiment, bit-exact and measured slower: profiles/r03/drain_experiments.md)
#endif
#ifndef RT_SPLIT_MIN_IDLE
#define RT_SPLIT_MIN_IDLE 4             // ... when at least this many lanes are idle
#endif
#ifndef RT_EXIT_K
#define RT_EXIT_K 1                     // leave the node loop once (lanes still on internal nodes) * K < lanes waiting on a leaf
                                        //   (four-wide nodes, ms per frame 1080p / 10 M triangles 4K: K = 0 3.31 / 21.9, 1 2.80 / 15.5, 2 2.86 / 16.5, 3 2.88 / 16.9)
#endif

// (Packed fp32: v_pk_add_f32 / v_pk_mul_f32 issue in twice the time of the scalar forms, so packing (lo, hi) plane pairs buys
// nothing there; v_pk_fma_f32 does issue two fmas in the 4 cycles one v_fma_f32 takes (profiles/r03/valu_rate.txt), but the
// step with its 24 plane fmas as 12 packed ones measured 2 % SLOWER (profiles/r03/pk_fma.txt): what the step waits for is
// its own dependent chain, not issue slots.)

#ifdef RT_TRACE_STATS      /* instrumentation build only (tools/trace_stats.py): SIMD-utilisation counters */
__device__ unsigned long long g_trace_stats[8];
__device__ unsigned long long g_trace_sp_hist[64];      // rays by the deepest stack pointer they reached
#define RT_STAT_WAVE(k) do { if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) st_w[k]++; } while (0)
#define RT_STAT_LANE(k) (st_l[k]++)
#else
#define RT_STAT_WAVE(k) ((void)0)
#define RT_STAT_LANE(k) ((void)0)
#endif
#ifdef RT_TRACE_TIMES      /* instrumentation build only (tools/drain_timeline.py) */
// per wave of the persistent launch selected by g_trace_sel (0 closest-hit queues, 1 any-hit queues; the last such launch wins):
// wall clock (100 MHz) at start, when the pool ran dry for it, at exit; + lanes alive when the pool ran dry
__device__ unsigned long long g_trace_wave_t[4 * 8192];
__device__ int g_trace_sel;
#endif

// -DRT_PREFETCH_POP (round 3's experiment for the HBM-bound scene, VERDICT r2 task 4): more misses in flight per lane.  After
// a step the node that will be POPPED next (the top of the stack) is known many steps before it is needed; its line is
// touched so that the pop finds it in the L2 instead of HBM.  Vector loads return in issue order (s_waitcnt vmcnt counts
// them in order), so a touch issued BEFORE the next step's node loads would have to land before that node could be used;
// it is therefore issued right BEHIND them (the step waits with vmcnt(1), the touch stays in flight during the step's
// arithmetic and has a whole step to land) and its value is "consumed" by an empty asm one step later, which keeps the
// destination register reserved until then.
// -DRT_PREFETCH_LEAF (second half of round 3): the same touch, but only where it is nearly free -- at the start of a leaf phase,
// ahead of the triangle loads, for the node this lane will pop when the leaf is done (known when the step that led here pushed it).
// Measured: +2 % in sets of frames (a live register more than the 72 of seven waves hold), +-0 frame by frame, +0.8 % on the
// 10 M-triangle scene (profiles/r03/prefetch_leaf.txt): off.
#if defined(RT_PREFETCH_LEAF) && !defined(RT_PREFETCH_ANY)
#define RT_PREFETCH_ANY
#endif
#if defined(RT_PREFETCH_POP) && !defined(RT_PREFETCH_ANY)
#define RT_PREFETCH_ANY
#endif
struct PopPrefetch {
#ifdef RT_PREFETCH_ANY
    int code;           // node to touch behind the next node loads (RT_NODE_EMPTY: none)
    float val;          // destination of the touch in flight
#endif
};

// The traversal stack: STACK rows per lane in LDS (stk[row * BLOCK], one dword per lane per row, bank =
// lane mod 32: conflict free), rows beyond that in global memory (deep[(row - STACK) * threads + thread]).
// The LDS rows are sized for occupancy, not for the deepest possible walk: Sponza-class rays never hold
// more than 15 entries although the tree is 29 levels deep (tools/trace_stats.py), so the global rows are
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

#if RT_WIDE == 8      // the round-3 experiment (see the header of this file)
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
                      const LaneStack<STACK, BLOCK> &st, int &node, int &sp, PopPrefetch &)
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
#else       // RT_WIDE == 4: the production step (64-B nodes, hit children sorted by entry distance)
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
// take the exact path inside the step.)
template <bool DEEP, bool ANYHIT, int STACK, int BLOCK>
RT_DEV void wide_step(const WNode *nodes, const int *top, uint32_t top_lim, const RayInv &ri, float tmin, float tbest,
                      const LaneStack<STACK, BLOCK> &st, int &node, int &sp, PopPrefetch &pf)
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
#ifdef RT_PREFETCH_POP
    {
        __builtin_amdgcn_sched_barrier(0);                   // the touch goes BEHIND this node's loads
        asm volatile("" :: "v"(pf.val));                     // the touch of the step before: landed by now (waited for here if not)
        // (issued by every lane, a lane without a target touches node 0 -- always cached: a load inside a branch would give
        // the paths different numbers of loads in flight and the compiler would have to wait for all of them, vmcnt(0))
        const bool want = (uint32_t)pf.code >= top_lim && pf.code >= 0 && pf.code < RT_NODE_EMPTY;
        pf.val = *(const __attribute__((address_space(1))) float *)((const char *)nodes + (want ? (uint32_t)pf.code << 6 : 0u));
        __builtin_amdgcn_sched_barrier(0);
    }
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
#define RT_CE(i, j) { const bool sw = d[j] < d[i]; const float td = sw ? d[j] : d[i]; d[j] = sw ? d[i] : d[j]; d[i] = td; \
                      const int tc = sw ? c[j] : c[i]; c[j] = sw ? c[i] : c[j]; c[i] = tc; }
        // (only bringing the nearest to the front -- three exchanges -- costs 1.3 % more steps and the same time)
#ifdef RT_SORT_NEAREST_ONLY
        RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2)
#else
        RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#endif
#undef RT_CE
        const float inf = __uint_as_float(0x7f800000u);
        any = d[0] < inf; p1 = d[1] < inf; p2 = d[2] < inf; p3 = d[3] < inf;
    }
    if (p3) { if (DEEP) st.write(sp, c[3]); else st.lds[sp * BLOCK] = c[3]; sp++; }
    if (p2) { if (DEEP) st.write(sp, c[2]); else st.lds[sp * BLOCK] = c[2]; sp++; }
    if (p1) { if (DEEP) st.write(sp, c[1]); else st.lds[sp * BLOCK] = c[1]; sp++; }
#ifdef RT_PREFETCH_ANY
    // the new top of the stack, if this step pushed one (else: what was touched before, or unknown after a pop)
    pf.code = p1 ? c[1] : (p2 ? c[2] : (p3 ? c[3] : RT_NODE_EMPTY));
#endif
    if (any) node = c[0];
    else { node = sp > 0 ? under : RT_NODE_EMPTY; sp = below; }
}
#endif

// COUNT builds: how many DISTINCT node records the active lanes of this wave step fetch from global memory (lanes on
// the same node share the record; lanes on the LDS-resident top fetch none); added to the calling leader lane's tally
RT_DEV uint32_t distinct_node_lines(int node, bool from_global)
{
    unsigned long long todo = __ballot(from_global);
    uint32_t n = 0;
    while (todo) {
        const int l = __builtin_ctzll(todo);
        const int v = __builtin_amdgcn_readlane(node, l);
        todo &= ~__ballot(from_global && node == v);
        n++;
    }
    return n;
}

RT_DEV bool node_is_internal(int node) { return node >= 0 && node < RT_NODE_EMPTY; }

// A ray source may keep a cache of first candidates for unordered any-hit searches of single-level scenes (the pipeline's shadow
// cache): uint32_t cached_leaf(ray index, const RayD &, uint32_t &slot, uint32_t &instance) -> index into the instance's sorted triangle
// array or RT_NO_HIT, and where a better answer would go; void remember(slot, index, instance).  The walk parks the slot in the last LDS row of the lane's stack (a walk that ever
// needs that row overwrites it: remember() then finds a number that is not a slot, or is somebody else's -- harmless either way).
template <class S, class = void> struct src_has_cache { static constexpr bool value = false; };
template <class S> struct src_has_cache<S, decltype((void)&S::has_first_candidates)> { static constexpr bool value = true; };

RT_DEV unsigned long long lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63u;
    return (1ull << lane) - 1ull;
}

// Ray sources / hit sinks are small functor structs:
//   struct Src  { uint32_t count() const; bool load(uint32_t i, RayD &r) const;   // false: not to be traced
//                 uint32_t flags() const; };
//   struct Sink { void store(uint32_t i, const HitD &h, bool traced) const; };

// COUNT: the walk-counting instantiation (rt_pipeline_count_walk): the same walk, plus per-lane tallies of what it
// fetches -- 64-B nodes from global memory, nodes from the LDS-resident top, 48-B triangle records, the 96-B traversal prefix of
// instance records -- summed into walk[0..5] = rays, nodes from global memory, nodes from LDS, triangles, instance entries,
// distinct 64-B lines (node lines de-duplicated across the lanes of each wave step + the lines the triangle records span);
// walk[6] = max over rays of (node steps << 32 | ray index), the longest single walk (a tail detector).
// These per-ray numbers depend on the ray and the tree only, not on chunking or lane assignment.
template <int STACK, int BLOCK, bool TWO_LEVEL, uint32_t CHUNK, bool ANYHIT = false, bool COUNT = false, class Src, class Sink>
RT_DEV void trace_wave(const SceneDev &sc, const Src &src, const Sink &sink, uint32_t *pool, int *smem, uint32_t *traced_counter,
                       unsigned long long *walk = nullptr)
{
    uint32_t n_traced = 0;           // rays this WAVE actually traversed (statistics; wave-uniform: a scalar register, not a lane's)
    uint32_t n_skipped = 0;          // any-hit launches: queue slots marked RT_TMAX_SKIPPED (likewise)
    uint32_t wk_glob = 0, wk_top = 0, wk_tri = 0, wk_inst = 0, wk_lines = 0;
    uint32_t wk_ray0 = 0;                        // node steps tallied when the lane's current ray started
    unsigned long long wk_longest = 0;           // (node steps << 32 | ray index) of the lane's longest walk
#ifdef RT_TRACE_STATS
    unsigned long long st_w[4] = {0, 0, 0, 0}, st_l[4] = {0, 0, 0, 0};   // node steps, leaf phases, triangle iterations, outer iterations
    int st_maxsp = 0;
#endif
    const uint32_t total = src.count();
    const uint32_t flags = src.flags();
    // ANYHIT instantiations (unordered walks) are only launched for ACCEPT_FIRST_HIT searches: as a compile-time fact it lets the
    // running best hit -- which then never changes before the ray ends -- out of the registers
    const bool first = ANYHIT ? true : (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    LaneStack<STACK, BLOCK> st;
    st.lds = smem + threadIdx.x;
    st.threads = gridDim.x * BLOCK;
    st.deep = sc.deep_stack ? sc.deep_stack + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;

    // single-level scenes (one identity instance) walk the BLAS directly in world space
    const InstanceRec *in0 = sc.inst;
    const WNode *blas_nodes0 = TWO_LEVEL ? nullptr : in0->wide;
    const TriRec *tris0 = TWO_LEVEL ? nullptr : in0->tris;
    // the LDS-resident top of the tree: smem rows STACK .. STACK + RT_TOP_ROWS - 1 hold nodes 0 .. top_n - 1 (breadth-first
    // numbering) of the structure a ray starts in: the BLAS of a single-level scene, the TLAS of a two-level one
    // -DRT_LDS_BLAS_TOPS (round 3's experiment; measured: 41 % fewer L2 node fetches on the 4096-instance frame, no faster --
    // the stages are issue-bound -- and five registers that cost the two-level primary kernel its fifth wave): the table is
    // split into the top of the TLAS (RT_TOP_TLAS nodes) and the tops of the two BLASes most instances use (RT_TOP_BLAS
    // nodes each; the TLAS build marks those instances, InstanceRec::flags bits 8-9), so that the first levels of a walk
    // INSIDE such an instance come from LDS as well.  Default build: the TLAS has the whole table.
    int *topl = smem + STACK * BLOCK;
    if (sc.top_n != 0) {
        const int *src_top = (const int *)(TWO_LEVEL ? sc.tlas_wide : blas_nodes0);
        for (uint32_t i = threadIdx.x; i < sc.top_n * RT_TOP_WORDS; i += BLOCK) topl[i] = src_top[(i / RT_TOP_WORDS) * (uint32_t)(sizeof(WNode) / 4) + i % RT_TOP_WORDS];
    }
#ifdef RT_LDS_BLAS_TOPS
    if (TWO_LEVEL) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int *src_top = (const int *)sc.blas_top[k];
            int *dst = topl + (RT_TOP_TLAS + k * RT_TOP_BLAS) * RT_TOP_WORDS;
            for (uint32_t i = threadIdx.x; i < sc.blas_top_n[k] * RT_TOP_WORDS; i += BLOCK) dst[i] = src_top[(i / RT_TOP_WORDS) * (uint32_t)(sizeof(WNode) / 4) + i % RT_TOP_WORDS];
        }
    }
    const int *top_cur = topl;                    // LDS table of the structure being walked
#else
    const int *const top_cur = topl;
#endif
    if (sc.top_n != 0) __syncthreads();
    const int root0 = TWO_LEVEL ? sc.tlas_root_code : in0->root_code;
    uint32_t top_lim = sc.top_n;                  // node indices below this are read from LDS (two-level: 0 while inside a BLAS)

    bool alive = false;
    bool exhausted = false;          // wave-uniform: the global pool has nothing left
    // the drain's ray splitting (see the loop): what this lane is to the ray it holds
    constexpr bool SPLIT = RT_DRAIN_SPLIT != 0 && !COUNT;
    constexpr uint32_t META_HOME = 63u;             // thief: the lane that stores the ray
    constexpr uint32_t META_THIEF = 64u;            // this lane walks a part of another lane's ray
    constexpr uint32_t META_PARKED = 128u;          // own part done, parts handed out still on their way
    constexpr uint32_t META_FOUND = 256u;           // any-hit rays: a part has found a hit
    constexpr uint32_t META_PEND1 = 1u << 16, META_PEND = 0xffu << 16;     // parts handed out and not yet back
    uint32_t meta = 0u;
    const bool may_split = ANYHIT || !first;        // an ordered first-hit search depends on the order of the walk
#ifdef RT_TRACE_TIMES
    const bool st_timed = !COUNT && g_trace_sel == (ANYHIT ? 1 : 0) && gridDim.x * (BLOCK / 64) <= 8192u && (threadIdx.x & 63u) == 0u;
    const uint32_t st_wave = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
    if (st_timed) { g_trace_wave_t[4 * st_wave] = wall_clock64(); g_trace_wave_t[4 * st_wave + 1] = 0ull; }
    bool st_noted = false;
#endif
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the chunk of the queue being handed out
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    uint32_t next_chunk = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;   // wave-uniform
    const uint32_t n_groups = gridDim.x < RT_POOL_GROUPS ? gridDim.x : RT_POOL_GROUPS;
    const uint32_t pool_group = blockIdx.x % n_groups;
    uint32_t idx = 0;
    RayD r;
    RayInv wri;
    HitD best;
    int node = RT_NODE_EMPTY;
    int sp = 0;
    PopPrefetch pf;
#ifdef RT_PREFETCH_ANY
    pf.code = RT_NODE_EMPTY; pf.val = 0.0f;
#endif
    // two-level state
    ObjRay cur;                       // ray in the space of the structure being walked
    const WNode *nodes = TWO_LEVEL ? sc.tlas_wide : blas_nodes0;
    const InstanceRec *in = in0;
    const TriRec *tris = tris0;
    uint32_t ii = 0;
    bool in_blas = !TWO_LEVEL;
    r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1); r.tmin = 0; r.tmax = 0;
    wri = make_inv(r.o, r.d);
    cur.o = r.o; cur.d = r.d; cur.ri = wri;
    best = make_miss(r);

    for (;;) {
        // ---- refill idle lanes: the wave owns a chunk [chunk_next, chunk_end) of the ray pool and
        //      only goes to the global counter (one atomic, lane 0) when the chunk is used up ----
        const unsigned long long idle = __ballot(!alive);
        const int n_idle = __popcll(idle);
        if (!exhausted && n_idle >= RT_REFILL_LANES) {
#if RT_REFILL_PRIO
            __builtin_amdgcn_s_setprio(RT_REFILL_PRIO);
#endif
            if (chunk_next >= chunk_end) {
                // Which chunk next?  Per-ray cost varies, so a static share per wave leaves the launch waiting
                // for its unluckiest waves.  ONE global counter is no answer: same-address returning atomics
                // cost ~60 ns each here and serialise (a single-counter pool was 3x slower; even handing out
                // only the last 1/16 of a queue that way cost +43 %).  So the queue is dealt to RT_POOL_GROUPS
                // groups of workgroups (group g owns chunks g, g+G, g+2G, ...) and the waves of a group
                // share one counter: 1/G of the contention, balancing across the group's ~190 waves (-4 % on
                // the frame).  Workgroups are dealt to the XCDs round-robin and G is a multiple of 8, so a
                // group and its counter stay on one XCD.
                uint32_t cidx;
                if (pool) {
                    uint32_t k = 0;
                    if ((threadIdx.x & 63u) == 0u) k = atomicAdd(&pool[pool_group * RT_POOL_STRIDE], 1u);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    // (contiguous bands of the queue per group, so that an XCD's L2 sees four parts of the image instead of all
                    // of it, with dry groups moving on to their neighbours' bands: 2.63 vs 2.56 ms, profiles/r03/pool_contiguous.txt)
                    cidx = pool_group + k * n_groups;
                } else {
                    cidx = next_chunk;           // static: wave w of W owns chunks w, w+W, w+2W, ...
                    next_chunk += n_waves;
                }
                const uint32_t base = cidx < 0x4000000u ? cidx * CHUNK : total;
                chunk_next = base;
                chunk_end = base + CHUNK < total ? base + CHUNK : total;
                if (base >= total) { exhausted = true; chunk_end = chunk_next; }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t rank = (uint32_t)__popcll(idle & lanemask_lt());
            bool started = false, skipped = false;
            if (!alive && rank < avail) {
                const uint32_t my = chunk_next + rank;
                idx = my;
                const bool traced = src.load(my, r);
                best = make_miss(r);
                if (traced && r.tmax > r.tmin && sc.n_inst != 0) {
                    wri = make_inv(r.o, r.d);
                    cur.o = r.o; cur.d = r.d; cur.ri = wri;
                    node = root0;
                    sp = 0;
#ifdef RT_PREFETCH_ANY
                    pf.code = RT_NODE_EMPTY;
#endif
                    if (TWO_LEVEL) {
                        nodes = sc.tlas_wide; in_blas = false; top_lim = sc.top_n;
#ifdef RT_LDS_BLAS_TOPS
                        top_cur = topl;
#endif
                    }
                    if constexpr (src_has_cache<Src>::value && ANYHIT && !COUNT) {
                        // the triangle that answered this question last time goes first: a one-triangle leaf in front of the root
                        uint32_t slot, ci;
                        const uint32_t ct = src.template cached_leaf<TWO_LEVEL>(my, r, slot, ci);
                        st.lds[(STACK - 1) * BLOCK] = (int)slot;
                        // (an entry is only ever tested if it names a triangle that exists: the table is emptied with the scene, but a pair
                        // torn between two writers, or a table handed over by mistake, must not read past an array)
                        if (ct != RT_NO_HIT && (!TWO_LEVEL || (ci < sc.n_inst && ct < sc.inst[ci].n_prims))) {      // (single level: the source checks against its own count)
                            st.lds[0] = root0;
                            sp = 1;
                            if (TWO_LEVEL) {          // ... inside its instance: the leaf, then the sentinel that leads back out, then the TLAS root
                                ii = ci;
                                in = sc.inst + ii;
                                cur = to_object(*in, r);
                                nodes = in->wide;
                                tris = in->tris;
                                in_blas = true;
                                top_lim = 0;
                                st.lds[BLOCK] = RT_NODE_SENTINEL;
                                sp = 2;
                            }
                            node = ~(int)(ct << 3);
                        }
                    }
                    alive = true;
                    started = true;
                    if (COUNT) wk_ray0 = wk_glob + wk_top;
                } else {
                    if (ANYHIT && r.tmax == RT_TMAX_SKIPPED) skipped = true;
                    sink.store(my, best, traced);
                }
            }
            n_traced += (uint32_t)__popcll(__ballot(started));
            if (ANYHIT) n_skipped += (uint32_t)__popcll(__ballot(skipped));
            chunk_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
#if RT_REFILL_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
#ifdef RT_TRACE_TIMES
        if (exhausted && !st_noted) {
            st_noted = true;
            const int na_now = __popcll(__ballot(alive));
            if (st_timed) { g_trace_wave_t[4 * st_wave + 1] = wall_clock64(); g_trace_wave_t[4 * st_wave + 3] = (unsigned long long)na_now; }
        }
#endif
        // ---- the drain: split the rays that are left over the lanes that are free (-DRT_DRAIN_SPLIT=1, off by default) ----
        // Once the queue is dry a wave only finishes the rays it holds, and how long that takes is set by its LONGEST ray
        // while more and more lanes sit idle: a third of a persistent launch at 1080p passes this way
        // (profiles/r03/drain_timeline.txt).  A walk is not one chain, though: every entry on a ray's stack is a subtree that
        // can be walked by itself.  So idle lanes (THIEVES) take the top stack entry of busy ones: a thief copies the ray and
        // its running best from the victim lane, walks that one subtree with a stack of its own, and hands what it found to
        // the ray's HOME lane (the lane that loaded the ray), which merges it with hit_better -- the same total order every
        // candidate goes through anyway, so the result is the one any order of the walk gives (rt_trace_device.h) -- and
        // stores the ray when its own part and all handed-out parts are done.  Everything stays inside the wave: lanes
        // exchange registers (ds_bpermute / readlane), no memory, no waiting.  Unordered any-hit rays hand over one bit;
        // ordered first-hit rays (whose result depends on the order) are never split.
        if (SPLIT && exhausted && may_split) {
            const uint32_t lane = threadIdx.x & 63u;
            // 1. parts that have finished: hand the result to the home lane
            unsigned long long fin = __ballot((meta & META_THIEF) != 0u && !alive);
            while (fin) {
                const int t = __builtin_ctzll(fin);
                fin &= fin - 1ull;
                const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)meta, t);
                const uint32_t h = mt & META_HOME;
                if (ANYHIT) {
                    if (lane == h) {
                        meta = (meta - META_PEND1) | (mt & META_FOUND);
                        if ((mt & META_FOUND) != 0u && alive) { node = RT_NODE_EMPTY; sp = 0; }      // one hit is all an any-hit ray asks for
                    }
                } else {
                    const float bt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best.t), t));
                    const float bu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best.u), t));
                    const float bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best.v), t));
                    const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)best.prim, t);
                    const uint32_t bi = (uint32_t)__builtin_amdgcn_readlane((int)best.inst, t);
                    if (lane == h) {
                        meta -= META_PEND1;
                        if (bi != RT_NO_HIT && hit_better(bt, bi, bp, best)) { best.t = bt; best.u = bu; best.v = bv; best.prim = bp; best.inst = bi; }
                    }
                }
            }
            if ((meta & META_THIEF) != 0u && !alive) meta = 0u;
            // 2. rays whose own part and all handed-out parts are done
            if ((meta & META_PARKED) != 0u && (meta & META_PEND) == 0u) {
                if (ANYHIT) {
                    HitD res = make_miss(r);
                    if ((meta & META_FOUND) != 0u) { res.t = r.tmin; res.prim = 0u; res.inst = 0u; }     // (any-hit sinks only ask whether inst is RT_NO_HIT)
                    sink.store(idx, res, true);
                } else sink.store(idx, best, true);
                meta = 0u;
            }
            // 3. idle lanes take the top stack entry of busy ones
            const unsigned long long free_lanes = __ballot(!alive && meta == 0u);
            const int n_free = __popcll(free_lanes);
            if (n_free >= RT_SPLIT_MIN_IDLE) {
                const int top_row = sp > 0 ? sp - 1 : 0;
                const int top_entry = st.read(top_row);
                const bool can_give = alive && sp >= 1 && !(TWO_LEVEL && (top_entry == RT_NODE_SENTINEL || node == RT_NODE_SENTINEL));    // (a lane about to leave its BLAS: the entries below the sentinel belong to the TLAS)
                unsigned long long victims = __ballot(can_give && sp >= 2);
                if (victims == 0ull) victims = __ballot(can_give);
                const int n_victims = __popcll(victims);
                const int n_pairs = n_free < n_victims ? n_free : n_victims;
                if (n_pairs > 0) {
                    const int my_free_rank = __popcll(free_lanes & lanemask_lt());
                    const bool thief = !alive && meta == 0u && my_free_rank < n_pairs;
                    const bool gives = (victims >> lane) & 1ull ? __popcll(victims & lanemask_lt()) < n_pairs : false;
                    // the lane this thief takes from: the victim of its rank
                    int from = 0;
                    {
                        int k = thief ? my_free_rank : 0;
#pragma unroll
                        for (int w = 32; w >= 1; w >>= 1) {
                            const int c = __popcll((victims >> from) & ((1ull << w) - 1ull));
                            if (k >= c) { k -= c; from += w; }
                        }
                    }
                    const int sel = from << 2;                                     // ds_bpermute addresses bytes
                    // (each value goes straight into the thief's own register: nothing else stays live across the exchange)
#define RT_TAKE_F(x) { const float pulled_ = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(x))); x = thief ? pulled_ : x; }
#define RT_TAKE_U(x) { const uint32_t pulled_ = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)(x)); x = thief ? pulled_ : x; }
                    const int p_entry = __builtin_amdgcn_ds_bpermute(sel, top_entry);
                    const uint32_t p_meta = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)meta);
                    RT_TAKE_F(r.o.x) RT_TAKE_F(r.o.y) RT_TAKE_F(r.o.z) RT_TAKE_F(r.d.x) RT_TAKE_F(r.d.y) RT_TAKE_F(r.d.z)
                    RT_TAKE_F(r.tmin) RT_TAKE_F(r.tmax)
                    RT_TAKE_F(wri.inv.x) RT_TAKE_F(wri.inv.y) RT_TAKE_F(wri.inv.z)
                    if (!ANYHIT) { RT_TAKE_F(best.t) RT_TAKE_F(best.u) RT_TAKE_F(best.v) RT_TAKE_U(best.prim) RT_TAKE_U(best.inst) }
                    if (TWO_LEVEL) {
                        uint32_t state = (in_blas ? 1u : 0u) | (ii << 1);
                        RT_TAKE_U(state)
                        RT_TAKE_F(cur.o.x) RT_TAKE_F(cur.o.y) RT_TAKE_F(cur.o.z) RT_TAKE_F(cur.d.x) RT_TAKE_F(cur.d.y) RT_TAKE_F(cur.d.z)
                        RT_TAKE_F(cur.ri.inv.x) RT_TAKE_F(cur.ri.inv.y) RT_TAKE_F(cur.ri.inv.z)
                        if (thief) {
                            in_blas = (state & 1u) != 0u;
                            ii = state >> 1;
                            cur.ri.o = cur.o;
                            if (in_blas) { in = sc.inst + ii; nodes = in->wide; tris = in->tris; top_lim = 0; }
                            else { nodes = sc.tlas_wide; top_lim = sc.top_n; }
                        }
                    }
#undef RT_TAKE_F
#undef RT_TAKE_U
                    if (gives) sp--;
                    // (for every lane, not only the thieves: these are copies of one another throughout the loop, and saying so in
                    // the same form everywhere lets the compiler keep them in one set of registers)
                    wri.o = r.o;
                    if (!TWO_LEVEL) { cur.o = r.o; cur.d = r.d; cur.ri = wri; }
                    if (thief) {
                        node = p_entry;
                        sp = 0;
                        alive = true;
                        // home: the victim's home if the victim is itself walking a part, else the victim
                        meta = META_THIEF | ((p_meta & META_THIEF) != 0u ? (p_meta & META_HOME) : (uint32_t)from);
#ifdef RT_PREFETCH_ANY
                        pf.code = RT_NODE_EMPTY;
#endif
                    }
                    // every home counts the parts it now waits for
                    unsigned long long fresh = __ballot(thief);
                    while (fresh) {
                        const int t = __builtin_ctzll(fresh);
                        fresh &= fresh - 1ull;
                        const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)meta, t) & META_HOME;
                        if (lane == h) meta += META_PEND1;
                    }
                }
            }
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted && (!SPLIT || __ballot(meta != 0u) == 0ull)) break;
            continue;
        }


        // ---- walk internal nodes until this lane stands on a leaf (or runs dry) ----------
#if RT_EXIT_K > 0
        const int n_alive = __popcll(__ballot(alive));
#endif
        while (alive && node_is_internal(node) && sp <= STACK - (RT_WIDE - 1)) {
            RT_STAT_WAVE(0); RT_STAT_LANE(0);
            if (COUNT) {
                if ((uint32_t)node < top_lim) wk_top++; else wk_glob++;
                const uint32_t dl = distinct_node_lines(node, !((uint32_t)node < top_lim));
                if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) wk_lines += (RT_WIDE == 8 ? 2u : 1u) * dl;       // 64-B lines (96 B of a 128-B record: two)
            }
            wide_step<false, ANYHIT>(nodes, top_cur, top_lim, cur.ri, r.tmin, ANYHIT ? r.tmax : best.t, st, node, sp, pf);
#ifdef RT_TRACE_STATS
            st_maxsp = sp > st_maxsp ? sp : st_maxsp;
#endif
#if RT_EXIT_K > 0
            // stragglers: most of the wave already waits on a leaf -> run the leaf phase now, come back after
            const int walking = __popcll(__ballot(alive && node_is_internal(node)));
            if (walking * RT_EXIT_K < n_alive - walking) break;
#endif
        }
        // lanes whose stack has outgrown the LDS rows walk on with the global rows until it fits again
        while (alive && node_is_internal(node) && sp > STACK - (RT_WIDE - 1)) {
            if (COUNT) {
                if ((uint32_t)node < top_lim) wk_top++; else wk_glob++;
                const uint32_t dl = distinct_node_lines(node, !((uint32_t)node < top_lim));
                if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) wk_lines += (RT_WIDE == 8 ? 2u : 1u) * dl;       // 64-B lines (96 B of a 128-B record: two)
            }
            wide_step<true, ANYHIT>(nodes, top_cur, top_lim, cur.ri, r.tmin, ANYHIT ? r.tmax : best.t, st, node, sp, pf);
#ifdef RT_TRACE_STATS
            st_maxsp = sp > st_maxsp ? sp : st_maxsp;
#endif
        }

        // ---- leaves, instance entry / exit, termination -----------------------------------
        RT_STAT_WAVE(3);
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(RT_LEAF_PRIO);
#endif
        if (alive && !node_is_internal(node)) {
            RT_STAT_WAVE(1); RT_STAT_LANE(1);
            bool pop = true;
            if (node == RT_NODE_EMPTY) {
#ifdef RT_TRACE_STATS
                atomicAdd(&g_trace_sp_hist[st_maxsp < 63 ? st_maxsp : 63], 1ull);
                st_maxsp = 0;
#endif
                if (SPLIT && meta != 0u) {             // a split ray (see the drain): a part goes to the home lane, the home waits for its parts
                    if ((meta & META_THIEF) == 0u) meta |= META_PARKED;
                } else sink.store(idx, ANYHIT ? make_miss(r) : best, true);
                alive = false;
                pop = false;
                if (COUNT) { const unsigned long long w = ((unsigned long long)(wk_glob + wk_top - wk_ray0) << 32) | idx; wk_longest = w > wk_longest ? w : wk_longest; }
            } else if (TWO_LEVEL && node == RT_NODE_SENTINEL) {
                in_blas = false;
                nodes = sc.tlas_wide;
#ifdef RT_PREFETCH_ANY
                pf.code = RT_NODE_EMPTY;                     // (codes of the structure just left)
#endif
                top_lim = sc.top_n;
#ifdef RT_LDS_BLAS_TOPS
                top_cur = topl;
#endif
                cur.o = r.o; cur.d = r.d; cur.ri = wri;
            } else if (TWO_LEVEL && !in_blas) {
                ii = (uint32_t)~node;
                in = sc.inst + ii;
                if (COUNT) wk_inst++;
                bool enter = true;
                if (sc.n_inst == 1) {      // the lone instance box was never tested as somebody's child
                    float e;
                    enter = slab_hit(wri, in->wlo[0], in->whi[0], in->wlo[1], in->whi[1], in->wlo[2], in->whi[2], r.tmin, best.t, e);
                }
                if (enter) {
                    cur = to_object(*in, r);
                    nodes = in->wide;
#ifdef RT_PREFETCH_ANY
                    pf.code = RT_NODE_EMPTY;
#endif
                    tris = in->tris;
                    in_blas = true;
#ifdef RT_LDS_BLAS_TOPS
                    {   // the top of this BLAS is LDS resident if it is one of the two the scene uses most
                        const uint32_t slot = (in->flags >> 8) & 3u;
                        top_lim = slot ? sc.blas_top_n[slot - 1u] : 0u;
                        top_cur = topl + (RT_TOP_TLAS + (slot ? slot - 1u : 0u) * RT_TOP_BLAS) * RT_TOP_WORDS;
                    }
#else
                    top_lim = 0;
#endif
                    st.write(sp, RT_NODE_SENTINEL);
                    sp++;
                    node = in->root_code;
                    pop = false;
                }
            } else {
                const uint32_t code = (uint32_t)~node;
                const uint32_t first_tri = code >> 3, cnt = (code & 7u) + 1u;
#ifdef RT_PREFETCH_LEAF
                // the node this lane pops after the leaf: its line is asked for now, ahead of the triangle records (loads return in
                // order: the wait for the records covers it), so that the pop finds it in the L2 at least
                if (sp > 0 && (uint32_t)pf.code >= top_lim && pf.code >= 0 && pf.code < RT_NODE_EMPTY)
                    pf.val = *(const __attribute__((address_space(1))) float *)((const char *)nodes + ((uint32_t)pf.code << 6));
                pf.code = RT_NODE_EMPTY;
#endif
                for (uint32_t k = 0; k < cnt; k++) {
                    RT_STAT_WAVE(2); RT_STAT_LANE(2);
                    if (COUNT) { wk_tri++; const uint32_t by = (first_tri + k) * 48u; wk_lines += 1u + ((by & 63u) > 16u ? 1u : 0u); }   // a 48-B record spans one or two lines
                    const char *tp = (const char *)(tris + first_tri + k);
                    const v4f a = ldg16(tp, 0), b = ldg16(tp, 16), c = ldg16(tp, 32);
                    const uint32_t prim = __float_as_uint(c.y);
                    HitD found = ANYHIT ? make_miss(r) : best;          // (any-hit: the running best never changes before the ray ends)
                    const bool accepted = accept_candidate(*in, ii, prim, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, wri, cur, cull, found);
                    if (!ANYHIT) best = found;
                    if (accepted && first) {
                        if constexpr (src_has_cache<Src>::value && ANYHIT && !COUNT) src.template remember<TWO_LEVEL>((uint32_t)st.lds[(STACK - 1) * BLOCK], first_tri + k, ii);
                        if (SPLIT && meta != 0u) meta |= (meta & META_THIEF) != 0u ? META_FOUND : META_FOUND | META_PARKED;
                        else sink.store(idx, found, true);
                        alive = false;
                        pop = false;
                        if (COUNT) { const unsigned long long w = ((unsigned long long)(wk_glob + wk_top - wk_ray0) << 32) | idx; wk_longest = w > wk_longest ? w : wk_longest; }
                        break;
                    }
                }
            }
#ifdef RT_PREFETCH_LEAF
            asm volatile("" :: "v"(pf.val));
#endif
            if (pop) {
                if (sp > 0) { sp--; node = st.read(sp); }
                else node = RT_NODE_EMPTY;
            }
        }
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
#ifdef RT_TRACE_TIMES
    if (st_timed) g_trace_wave_t[4 * st_wave + 2] = wall_clock64();
#endif
#ifdef RT_TRACE_STATS
    for (int k = 0; k < 4; k++) {
        if (st_w[k]) atomicAdd(&g_trace_stats[2 * k], st_w[k]);
        if (st_l[k]) atomicAdd(&g_trace_stats[2 * k + 1], st_l[k]);
    }
#endif
    if (COUNT && walk) {
        unsigned long long w5[6] = {(threadIdx.x & 63u) == 0u ? n_traced : 0u, wk_glob, wk_top, wk_tri, wk_inst, wk_lines};
        for (int k = 0; k < 6; k++) {
            unsigned long long v = w5[k];
            for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
            if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&walk[k], v);
        }
        atomicMax(&walk[6], wk_longest);
    }
    if (traced_counter) {            // one no-return atomic per persistent wave
        if ((threadIdx.x & 63u) == 0u && n_traced) atomicAdd(traced_counter, n_traced);
        if (ANYHIT && (threadIdx.x & 63u) == 0u && n_skipped) atomicAdd(traced_counter + 1, n_skipped);       // (the word behind an any-hit launch's ray counter tallies its skipped slots)
    }
}

}  // namespace rtd
