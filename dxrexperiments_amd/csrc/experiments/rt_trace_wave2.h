// experiments/rt_trace_wave2.h -- round 4's EXPERIMENT (-DRT_TWO_RAYS=1), not part of the default build: TWO rays per lane.
//
// The counters say the traversal stages issue near their ceiling with about half their lanes live: in a node
// step 66 % of a wave's lanes stand on an internal node, in a triangle step 37 % on a leaf.  Here every lane holds two rays (two
// "slots": ray state in registers, a stack of its own in LDS) and, step by step, works on whichever of its rays wants that kind of
// step: a lane is live in a node step when EITHER ray stands on a node, 1 - (1 - p)^2.  Same rays, same walks per ray, same
// candidate validation: results are bit-identical whatever the pairing of rays and lanes (rt_trace_device.h).
// Price: the ray state twice (about 90 VGPRs: five waves per SIMD instead of seven), a select in front of every step and a
// write-back behind it, twice the stack rows in LDS.  Single-level scenes only, no counting instantiation.
// (Included by rt_trace_wave.h inside namespace rtd once r04_two_rays_hooks.patch is applied.)
// MEASURED SLOWER: sets of frames 1.50 -> 1.76 ms per 1080p frame, 10 M triangles 9.5 -> 11.8 ms (profiles/r04/two_rays.txt): five waves per SIMD
// instead of seven cost 15 - 19 % (occupancy_check.txt) and the second ray does not win that back.
#pragma once

#ifndef RT_TWO_ROWS
#define RT_TWO_ROWS 12                  // LDS stack rows per ray: 2 x 12 + 8 top rows = 32 KiB per block, five blocks per CU
#endif
#ifndef RT_TWO_REFILL
#define RT_TWO_REFILL 24                // refill once this many of the wave's 128 slots are idle
#endif
#define RT_NODE_IDLE RT_NODE_SENTINEL   // a slot that holds no ray (single-level walks never see the sentinel)

struct Slot2 {
    f3 o, d, inv;
    float tmin, tmax;                   // (closest-hit walks: tmax = the running best t)
    float u, v;                         // closest-hit walks: the running best's barycentrics ...
    uint32_t prim;                      // ... and primitive (RT_NO_HIT: none yet)
    int node, sp;
    uint32_t ticket;
};

#define RT_SEL(a, b) (s1 ? (b) : (a))

template <int STACK, int BLOCK, uint32_t CHUNK, bool ANYHIT, class Src, class Sink>
RT_DEV void trace_wave2(const SceneDev &sc, const Src &src, const Sink &sink, uint32_t *pool, int *smem, uint32_t *traced_counter)
{
    uint32_t n_traced = 0, n_skipped = 0;
    const uint32_t total = src.count();
    const uint32_t flags = src.flags();
    const bool first = ANYHIT ? true : (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    const InstanceRec *in0 = sc.inst;
    const WNode *nodes = in0->wide;
    const TriRec *tris = in0->tris;
    int *lds0 = smem + threadIdx.x, *lds1 = lds0 + STACK * BLOCK;
    int *deep0 = sc.deep_stack ? sc.deep_stack + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;
    const uint32_t threads = gridDim.x * BLOCK;          // (the second slot's global rows follow the first's: rt_scene_dev_for_launch reserves for twice the threads)
    int *topl = smem + 2 * STACK * BLOCK;
    if (sc.top_n != 0) {
        const int *src_top = (const int *)nodes;
        for (uint32_t i = threadIdx.x; i < sc.top_n * RT_TOP_WORDS; i += BLOCK) topl[i] = src_top[(i / RT_TOP_WORDS) * (uint32_t)(sizeof(WNode) / 4) + i % RT_TOP_WORDS];
        __syncthreads();
    }
    const int root0 = in0->root_code;
    const uint32_t top_lim = sc.top_n;

    Slot2 a, b;
    a.o = mk3(0, 0, 0); a.d = mk3(0, 0, 1); a.inv = mk3(0, 0, 1); a.tmin = 0; a.tmax = 0; a.u = a.v = 0; a.prim = RT_NO_HIT; a.node = RT_NODE_IDLE; a.sp = 0; a.ticket = 0;
    b = a;
    bool exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    uint32_t next_chunk = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;
    const uint32_t n_groups = gridDim.x < RT_POOL_GROUPS ? gridDim.x : RT_POOL_GROUPS;
    const uint32_t pool_group = blockIdx.x % n_groups;

    for (;;) {
        // ---- refill: a lane with an idle slot takes one ray (a lane with two idle slots takes the second at the next refill)
        const unsigned long long idle_a = __ballot(a.node == RT_NODE_IDLE), idle_b = __ballot(b.node == RT_NODE_IDLE);
        const unsigned long long wants = idle_a | idle_b;
        if (!exhausted && __popcll(idle_a) + __popcll(idle_b) >= RT_TWO_REFILL) {
#if RT_REFILL_PRIO
            __builtin_amdgcn_s_setprio(RT_REFILL_PRIO);
#endif
            if (chunk_next >= chunk_end) {
                uint32_t cidx;
                if (pool) {
                    uint32_t k = 0;
                    if ((threadIdx.x & 63u) == 0u) k = atomicAdd(&pool[pool_group * RT_POOL_STRIDE], 1u);
                    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                    cidx = pool_group + k * n_groups;
                } else {
                    cidx = next_chunk;
                    next_chunk += n_waves;
                }
                const uint32_t base = cidx < 0x4000000u ? cidx * CHUNK : total;
                chunk_next = base;
                chunk_end = base + CHUNK < total ? base + CHUNK : total;
                if (base >= total) { exhausted = true; chunk_end = chunk_next; }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const int n_want = __popcll(wants);
            const uint32_t rank = (uint32_t)__popcll(wants & lanemask_lt());
            bool started = false, skipped = false;
            if (((wants >> (threadIdx.x & 63u)) & 1ull) && rank < avail) {
                const bool s1 = a.node != RT_NODE_IDLE;              // (the first slot if it is idle)
                const uint32_t my = chunk_next + rank;
                RayD r;
                uint32_t ticket;
                const bool traced = load_ray_of(src, my, r, ticket, 0);
                int node = RT_NODE_IDLE, sp = 0;
                int *lds = RT_SEL(lds0, lds1);
                if (traced && r.tmax > r.tmin && sc.n_inst != 0) {
                    node = root0;
                    if constexpr (src_has_cache<Src>::value && ANYHIT) {
                        uint32_t slot, ci;
                        const uint32_t ct = src.template cached_leaf<false>(my, r, slot, ci);
                        lds[(STACK - 1) * BLOCK] = (int)slot;
                        if (ct != RT_NO_HIT) { lds[0] = root0; sp = 1; node = ~(int)(ct << 3); }
                    }
                    started = true;
                } else {
                    if (ANYHIT && r.tmax == RT_TMAX_SKIPPED) skipped = true;
                    sink.store(ticket, make_miss(r), traced);
                }
                const RayInv ri = make_inv(r.o, r.d);
                if (!s1) { a.o = r.o; a.d = r.d; a.inv = ri.inv; a.tmin = r.tmin; a.tmax = r.tmax; a.u = 0.0f; a.v = 0.0f; a.prim = RT_NO_HIT; a.node = node; a.sp = sp; a.ticket = ticket; }
                else { b.o = r.o; b.d = r.d; b.inv = ri.inv; b.tmin = r.tmin; b.tmax = r.tmax; b.u = 0.0f; b.v = 0.0f; b.prim = RT_NO_HIT; b.node = node; b.sp = sp; b.ticket = ticket; }
            }
            n_traced += (uint32_t)__popcll(__ballot(started));
            if (ANYHIT) n_skipped += (uint32_t)__popcll(__ballot(skipped));
            chunk_next += (uint32_t)n_want < avail ? (uint32_t)n_want : avail;
#if RT_REFILL_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        if (__ballot(a.node != RT_NODE_IDLE || b.node != RT_NODE_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---- node steps: every lane on whichever of its rays stands on an internal node (and fits the LDS rows) --------
        for (;;) {
            const bool fa = node_is_internal(a.node) && a.sp <= STACK - (RT_WIDE - 1), fb = node_is_internal(b.node) && b.sp <= STACK - (RT_WIDE - 1);
            if (__ballot(fa || fb) == 0ull) break;
            if (fa || fb) {
                const bool s1 = !fa;
                RayInv ri;
                ri.o = mk3(RT_SEL(a.o.x, b.o.x), RT_SEL(a.o.y, b.o.y), RT_SEL(a.o.z, b.o.z));
                ri.inv = mk3(RT_SEL(a.inv.x, b.inv.x), RT_SEL(a.inv.y, b.inv.y), RT_SEL(a.inv.z, b.inv.z));
                int node = RT_SEL(a.node, b.node), sp = RT_SEL(a.sp, b.sp);
                LaneStack<STACK, BLOCK> st;
                st.lds = RT_SEL(lds0, lds1); st.deep = nullptr; st.threads = 0;
                wide_step<false, ANYHIT>(nodes, topl, top_lim, ri, RT_SEL(a.tmin, b.tmin), RT_SEL(a.tmax, b.tmax), st, node, sp);
                if (s1) { b.node = node; b.sp = sp; } else { a.node = node; a.sp = sp; }
            }
#if RT_EXIT_K > 0
            // stragglers: lanes with no ray on a node that have one on a leaf wait for the leaf phase
            const bool walks = node_is_internal(a.node) || node_is_internal(b.node);
            const int walking = __popcll(__ballot(walks));
            const int waiting = __popcll(__ballot(!walks && (a.node != RT_NODE_IDLE || b.node != RT_NODE_IDLE)));
            if (walking * RT_EXIT_K < waiting) break;
#endif
        }
        // rays whose stack has outgrown the LDS rows walk on with the global rows until it fits again
        for (;;) {
            const bool da = node_is_internal(a.node) && a.sp > STACK - (RT_WIDE - 1), db = node_is_internal(b.node) && b.sp > STACK - (RT_WIDE - 1);
            if (__ballot(da || db) == 0ull) break;
            if (da || db) {
                const bool s1 = !da;
                RayInv ri;
                ri.o = mk3(RT_SEL(a.o.x, b.o.x), RT_SEL(a.o.y, b.o.y), RT_SEL(a.o.z, b.o.z));
                ri.inv = mk3(RT_SEL(a.inv.x, b.inv.x), RT_SEL(a.inv.y, b.inv.y), RT_SEL(a.inv.z, b.inv.z));
                int node = RT_SEL(a.node, b.node), sp = RT_SEL(a.sp, b.sp);
                LaneStack<STACK, BLOCK> st;
                st.lds = RT_SEL(lds0, lds1); st.deep = deep0 ? deep0 + (s1 ? threads : 0u) : nullptr; st.threads = 2u * threads;
                wide_step<true, ANYHIT>(nodes, topl, top_lim, ri, RT_SEL(a.tmin, b.tmin), RT_SEL(a.tmax, b.tmax), st, node, sp);
                if (s1) { b.node = node; b.sp = sp; } else { a.node = node; a.sp = sp; }
            }
        }

        // ---- leaves and ray ends: every lane on whichever of its rays stands on one ---------------------------------------
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(RT_LEAF_PRIO);
#endif
        {
            const bool la = !node_is_internal(a.node) && a.node != RT_NODE_IDLE, lb = !node_is_internal(b.node) && b.node != RT_NODE_IDLE;
            if (la || lb) {
                const bool s1 = !la;
                RayD r;
                r.o = mk3(RT_SEL(a.o.x, b.o.x), RT_SEL(a.o.y, b.o.y), RT_SEL(a.o.z, b.o.z));
                r.d = mk3(RT_SEL(a.d.x, b.d.x), RT_SEL(a.d.y, b.d.y), RT_SEL(a.d.z, b.d.z));
                r.tmin = RT_SEL(a.tmin, b.tmin);
                RayInv ri;
                ri.o = r.o;
                ri.inv = mk3(RT_SEL(a.inv.x, b.inv.x), RT_SEL(a.inv.y, b.inv.y), RT_SEL(a.inv.z, b.inv.z));
                ObjRay cur;
                cur.o = r.o; cur.d = r.d; cur.ri = ri;
                HitD best;
                best.t = RT_SEL(a.tmax, b.tmax); best.u = RT_SEL(a.u, b.u); best.v = RT_SEL(a.v, b.v); best.prim = RT_SEL(a.prim, b.prim);
                best.inst = best.prim == RT_NO_HIT ? RT_NO_HIT : 0u;
                // The ray's own end: any-hit walks keep it (their best never changes).  Closest-hit walks keep only the running best t;
                // the test "t < TMax" is implied by "t <= best.t" once there is a hit (best.t < TMax then), and is "t < best.t" before
                // the first (best.t == TMax): so TMax is stood in for by best.t before the first hit and by +inf after it -- the same
                // accept / reject for every candidate, ties included (hit_better breaks them by primitive)
                const float inf = __uint_as_float(0x7f800000u);
                r.tmax = (ANYHIT || best.prim == RT_NO_HIT) ? best.t : inf;
                int node = RT_SEL(a.node, b.node), sp = RT_SEL(a.sp, b.sp);
                const uint32_t ticket = RT_SEL(a.ticket, b.ticket);
                int *lds = RT_SEL(lds0, lds1);
                LaneStack<STACK, BLOCK> st;
                st.lds = lds; st.deep = deep0 ? deep0 + (s1 ? threads : 0u) : nullptr; st.threads = 2u * threads;
                bool pop = true;
                if (node == RT_NODE_EMPTY) {
                    if (ANYHIT) { HitD miss; miss.t = best.t; miss.u = 0.0f; miss.v = 0.0f; miss.prim = RT_NO_HIT; miss.inst = RT_NO_HIT; sink.store(ticket, miss, true); }
                    else sink.store(ticket, best, true);
                    node = RT_NODE_IDLE;
                    pop = false;
                } else {
                    const uint32_t code = (uint32_t)~node;
                    const uint32_t first_tri = code >> 3, cnt = (code & 7u) + 1u;
                    for (uint32_t k = 0; k < cnt; k++) {
                        const char *tp = (const char *)(tris + first_tri + k);
                        const v4f ta = ldg16(tp, 0), tb = ldg16(tp, 16), tc = ldg16(tp, 32);
                        const uint32_t prim = __float_as_uint(tc.y);
                        HitD found = best;
                        if (ANYHIT) { found.prim = RT_NO_HIT; found.inst = RT_NO_HIT; }
                        const bool accepted = accept_candidate(*in0, 0u, prim, mk3(ta.x, ta.y, ta.z), mk3(ta.w, tb.x, tb.y), mk3(tb.z, tb.w, tc.x), r, ri, cur, cull, found);
                        if (!ANYHIT) { best = found; r.tmax = best.prim == RT_NO_HIT ? best.t : inf; }
                        if (accepted && first) {
                            if constexpr (src_has_cache<Src>::value && ANYHIT) src.template remember<false>((uint32_t)lds[(STACK - 1) * BLOCK], first_tri + k, 0u);
                            sink.store(ticket, found, true);
                            node = RT_NODE_IDLE;
                            pop = false;
                            break;
                        }
                    }
                }
                if (pop) {
                    if (sp > 0) { sp--; node = st.read(sp); }
                    else node = RT_NODE_EMPTY;
                }
                if (s1) { b.node = node; b.sp = sp; if (!ANYHIT) { b.tmax = best.t; b.u = best.u; b.v = best.v; b.prim = best.prim; } }
                else { a.node = node; a.sp = sp; if (!ANYHIT) { a.tmax = best.t; a.u = best.u; a.v = best.v; a.prim = best.prim; } }
            }
        }
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    if (traced_counter) {
        if ((threadIdx.x & 63u) == 0u && n_traced) atomicAdd(traced_counter, n_traced);
        if (ANYHIT && (threadIdx.x & 63u) == 0u && n_skipped) atomicAdd(traced_counter + 1, n_skipped);
    }
}
#undef RT_SEL
