// rt_trace_repack.h -- round 6: the RE-PACKED any-hit engine (VERDICT r5 task 2): rays change lanes at every phase switch.
//
// rt_trace_wave.h keeps a ray in ONE lane from start to end; a lane whose ray stands on a leaf waits for its wave's leaf phase, and in that
// phase the lanes that stand on internal nodes wait: a node step runs with 0.66 of its lanes, a triangle step with 0.37
// (rt_pipeline_count_walk, profiles/r05).  Here a ray is a SLOT of its workgroup (256 per 256-thread workgroup) and its lane is only where it
// happens to be stepped:
//   * the slot's stack column stays in LDS, stack[row][slot]; the rest of its state -- origin, direction, interval, ticket, node, stack
//     pointer: 48 B -- lives in global memory (slot records, L2 resident: 16 KiB per workgroup) and is read or written only at a switch;
//   * NODE waves (3 of 4) hold (origin, 1 / direction, interval, node, stack pointer) of 64 rays in registers and run nothing but node steps;
//     a lane whose ray reaches a leaf writes (node, stack pointer) back, puts the slot number on the workgroup's LEAF QUEUE (LDS ring) and is
//     free at once: the next refill gives it a ray from the NODE QUEUE (rays that come back from a leaf) or a new one from the launch's pool;
//   * the LEAF wave (1 of 4) takes up to 64 slot numbers from the leaf queue, tests their leaves' triangles, pops the ray's next node from its
//     stack column and sends it to the node queue -- or finishes it;
//   * queues: LDS rings of 16-bit slot numbers (reserve by atomic add, publish by the entry's valid bit), wave-aggregated pushes and pops;
//     free slot numbers in a third ring.  No barrier anywhere: waves only ever wait for entries.
// Exactness: the candidate rule is a function of (ray, triangle); which lane steps a ray, and in what order its
// nodes are visited, cannot change a bit.  Any-hit, single-level scenes only (the shadow stage: the frame's dominant kernel).
// No reference counterpart: TraceRay's scheduling is the Fallback Layer's (absent); semantics as rt_trace_device.h cites them.
#pragma once

#include "rt_trace_wave.h"

namespace rtd {

#ifndef RT_RP_NODE_WAVES
#define RT_RP_NODE_WAVES 3               // of the workgroup's four waves; the others are leaf waves
#endif
#define RT_RP_RING 256u
#define RT_RP_VALID 0x8000u
#define RT_RP_SLOT_BYTES 64u            // slot record: float4 (o, tmin), float4 (d, tmax), uint4 (node, sp, ticket, -)
#define RT_RP_LDS_EXTRA_INTS (3 * 128 + 16)      // three rings of 256 x 16 bit + the counters
#ifndef RT_RP_REFILL
#define RT_RP_REFILL 12                 // a node wave refills once this many of its lanes are free
#endif
#ifndef RT_RP_LEAF_MIN
#define RT_RP_LEAF_MIN 32u              // the leaf wave waits (a little) for this many entries
#endif
#define RT_RP_WATCHDOG (1u << 24)       // polls a wave may spend waiting before it gives the launch up (a bug, not a load, if ever reached)

struct RpShared {                       // the workgroup's counters (LDS ints behind the rings)
    uint32_t lq_tail, lq_head, nq_tail, nq_head, fq_tail, fq_head;
    int in_flight;                      // rays started and not finished
    uint32_t nodes_exited;              // node waves that have left
    uint32_t abort;                     // watchdog
};

RT_DEV uint32_t rp_ld(const volatile uint32_t *p) { return *p; }

// push the slot numbers of the lanes with `pred` (wave-aggregated: one atomic)
RT_DEV void rp_push(volatile uint16_t *ring, uint32_t *tail, bool pred, uint32_t value)
{
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const uint32_t n = (uint32_t)__popcll(m), rank = (uint32_t)__popcll(m & lanemask_lt());
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(tail, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (pred) ring[(base + rank) & (RT_RP_RING - 1u)] = (uint16_t)(value | RT_RP_VALID);
}

// take up to `want` entries: returns how many (wave-uniform); lane `rank` < count gets its entry in `value`
RT_DEV uint32_t rp_pop(volatile uint16_t *ring, uint32_t *head, const uint32_t *tail, uint32_t want, uint32_t rank, uint32_t &value, volatile uint32_t *abort_flag)
{
    uint32_t h = 0, k = 0;
    if ((threadIdx.x & 63u) == 0u && want) {
        for (int tries = 0; tries < 64; tries++) {
            h = rp_ld((const volatile uint32_t *)head);
            const uint32_t avail = rp_ld((const volatile uint32_t *)tail) - h;
            k = avail < want ? avail : want;
            if (k == 0u) break;
            if (atomicCAS(head, h, h + k) == h) break;
            k = 0u;
        }
    }
    h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    if (rank < k) {
        volatile uint16_t *e = ring + ((h + rank) & (RT_RP_RING - 1u));
        uint32_t v = *e, spins = 0;
        while (!(v & RT_RP_VALID)) {                 // reserved by its producer, not yet written
            __builtin_amdgcn_s_sleep(1);
            v = *e;
            if (++spins > RT_RP_WATCHDOG) { *abort_flag = 1u; break; }
        }
        *e = 0;
        value = v & (RT_RP_VALID - 1u);
    }
    return k;
}

template <int STACK, int BLOCK, uint32_t CHUNK, bool REFS, class Src, class Sink>
RT_DEV void trace_wave_repack(const SceneDev &sc, const Src &src, const Sink &sink, uint32_t *pool, int *smem, uint32_t *traced_counter, char *slot_records)
{
    static_assert(BLOCK == 256, "four waves: three node waves and a leaf wave");
    // LDS: [STACK rows of stack][top table][rings + counters]
    int *topl = smem + STACK * BLOCK;
    volatile uint16_t *rings = (volatile uint16_t *)(smem + (STACK + RT_TOP_ROWS(BLOCK)) * BLOCK);
    volatile uint16_t *lq = rings, *nq = rings + RT_RP_RING, *fq = rings + 2 * RT_RP_RING;
    RpShared *sh = (RpShared *)(smem + (STACK + RT_TOP_ROWS(BLOCK)) * BLOCK + 3 * 128);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const InstanceRec *in0 = sc.inst;
    const WNode *nodes = in0->wide;
    const TriRec *tris = in0->tris;
    if (sc.top_n != 0) {
        const int *src_top = (const int *)nodes;
        for (uint32_t i = threadIdx.x; i < sc.top_n * RT_TOP_WORDS; i += BLOCK) topl[i] = src_top[(i / RT_TOP_WORDS) * (uint32_t)(sizeof(WNode) / 4) + i % RT_TOP_WORDS];
    }
    lq[threadIdx.x] = 0; nq[threadIdx.x] = 0;
    fq[threadIdx.x] = (uint16_t)(threadIdx.x | RT_RP_VALID);          // every slot is free
    if (threadIdx.x == 0) {
        sh->lq_tail = sh->lq_head = sh->nq_tail = sh->nq_head = 0u;
        sh->fq_head = 0u; sh->fq_tail = RT_RP_RING;
        sh->in_flight = 0; sh->nodes_exited = 0u; sh->abort = 0u;
    }
    __syncthreads();
    const uint32_t top_lim = sc.top_n;
    const int root0 = in0->root_code;
    const uint32_t flags = src.flags();
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    char *records = slot_records + (size_t)blockIdx.x * BLOCK * RT_RP_SLOT_BYTES;
    // what the waves did, for profiles/r06/repack.txt (rt_debug_repack_stats): [0] node steps issued (one per wave per pass of the node loop), [1] lanes live
    // in them, [2] leaf passes (one per wave per pass of the leaf wave's loop), [3] lanes live in them, [4] rays through the leaf queue, [5] rays
    // through the node queue, [6] refills, [7] watchdog aborts
    unsigned long long *stats = (unsigned long long *)(slot_records + (size_t)gridDim.x * BLOCK * RT_RP_SLOT_BYTES);
    uint32_t s_steps = 0, s_step_lanes = 0, s_passes = 0, s_pass_lanes = 0, s_lq = 0, s_nq = 0, s_refills = 0;
    int *deep0 = sc.deep_stack ? sc.deep_stack + (size_t)blockIdx.x * BLOCK : nullptr;
    LaneStack<STACK, BLOCK> st;
    st.threads = gridDim.x * BLOCK;
    st.lds = smem; st.deep = deep0;
    volatile uint32_t *abort_flag = &sh->abort;
    uint32_t watchdog = 0;

    if (wave < RT_RP_NODE_WAVES) {
        // ================================= node waves =================================
        uint32_t n_traced = 0, n_skipped = 0;
        const uint32_t total = src.count();
        bool exhausted = false;
        uint32_t chunk_next = 0, chunk_end = 0;
        const uint32_t n_waves = gridDim.x * RT_RP_NODE_WAVES;
        uint32_t next_chunk = blockIdx.x * RT_RP_NODE_WAVES + wave;
        const uint32_t n_groups = gridDim.x < RT_POOL_GROUPS ? gridDim.x : RT_POOL_GROUPS;
        const uint32_t pool_group = blockIdx.x % n_groups;
        uint32_t xcd_steal = 0;
        bool alive = false;
        RayInv ri;
        float tmin = 0.0f, tmax = 0.0f;
        int node = RT_NODE_EMPTY, sp = 0;
        uint32_t slot = 0, ticket = RT_NO_HIT;
        ri.o = mk3(0, 0, 0); ri.inv = mk3(1, 1, 1);
        for (;;) {
            const unsigned long long idle = __ballot(!alive);
            const int n_idle = __popcll(idle);
            if (n_idle >= RT_RP_REFILL) {
                s_refills++;
                __builtin_amdgcn_s_setprio(RT_REFILL_PRIO);
                // ---- rays that come back from a leaf ----
                uint32_t rank = (uint32_t)__popcll(idle & lanemask_lt()), got_slot = 0;
                const uint32_t back = rp_pop(nq, &sh->nq_head, &sh->nq_tail, (uint32_t)n_idle, alive ? 0xFFFFu : rank, got_slot, abort_flag);
                if (back) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (!alive && rank < back) {
                    slot = got_slot;
                    const char *rec = records + (size_t)slot * RT_RP_SLOT_BYTES;
                    const v4f a = ldg16(rec, 0), b = ldg16(rec, 16), c = ldg16(rec, 32);
                    ri = make_inv(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z));
                    tmin = a.w; tmax = b.w;
                    node = __float_as_int(c.x); sp = __float_as_int(c.y); ticket = __float_as_uint(c.z);
                    st.lds = smem + slot; st.deep = deep0 ? deep0 + slot : nullptr;
                    alive = true;
                }
                // ---- new rays ----
                const unsigned long long idle2 = __ballot(!alive);
                const int n2 = __popcll(idle2);
                if (!exhausted && n2 > 0) {
                    if (chunk_next >= chunk_end) {
                        uint32_t cidx;
                        if (pool && n_groups == RT_POOL_GROUPS) {          // (the any-hit launch's deal: rt_trace_wave.h, RT_POOL_XCD)
                            const uint32_t chunks = (total + CHUNK - 1u) / CHUNK, per = (chunks + 7u) / 8u;
                            cidx = 0x4000000u;
                            while (xcd_steal < 8u) {
                                const uint32_t x = (pool_group + xcd_steal) & 7u;
                                const uint32_t g = x + (pool_group & ~7u);
                                uint32_t k = 0;
                                if (lane == 0u) k = atomicAdd(&pool[g * RT_POOL_STRIDE], 1u);
                                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                                const uint32_t c = x * per + (pool_group >> 3) + k * (RT_POOL_GROUPS / 8u);
                                if (c < (x + 1u) * per && c < chunks) { cidx = c; break; }
                                xcd_steal++;
                            }
                        } else if (pool) {
                            uint32_t k = 0;
                            if (lane == 0u) k = atomicAdd(&pool[pool_group * RT_POOL_STRIDE], 1u);
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
                    const uint32_t want = (uint32_t)n2 < avail ? (uint32_t)n2 : avail;
                    const uint32_t rank2 = (uint32_t)__popcll(idle2 & lanemask_lt());
                    uint32_t free_slot = 0;
                    const uint32_t f = rp_pop(fq, &sh->fq_head, &sh->fq_tail, want, alive ? 0xFFFFu : rank2, free_slot, abort_flag);
                    bool started = false, skipped = false, give_back = false;
                    if (!alive && rank2 < f) {
                        const uint32_t my = chunk_next + rank2;
                        RayD r;
                        const bool traced = load_ray_of(src, my, r, ticket, 0);
                        if (traced && r.tmax > r.tmin && sc.n_inst != 0) {
                            slot = free_slot;
                            st.lds = smem + slot; st.deep = deep0 ? deep0 + slot : nullptr;
                            ri = make_inv(r.o, r.d);
                            tmin = r.tmin; tmax = r.tmax;
                            node = root0; sp = 0;
                            bool occluded = false;
                            if constexpr (src_has_cache<Src>::value) {
                                // the triangle that answered this question last time is tested HERE, before the ray takes a slot: most rays
                                // have one, and sending each of them through the leaf queue first doubled the leaf waves' load
                                uint32_t cslot, ci;
                                const uint32_t ct = src.template cached_leaf<false>(ticket, r, cslot, ci);
                                st.lds[(STACK - 1) * BLOCK] = (int)cslot;
                                if (ct != RT_NO_HIT) {
                                    const char *tp = (const char *)(tris + ct);
                                    const v4f a = ldg16(tp, 0), b = ldg16(tp, 16), c = ldg16(tp, 32);
                                    HitD found = make_miss(r);
                                    ObjRay cur; cur.o = r.o; cur.d = r.d; cur.ri = ri;
                                    if (accept_candidate<REFS ? 1 : 0>(*in0, 0u, __float_as_uint(c.y), mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, ri, cur, cull, found,
                                                                       REFS ? __float_as_uint(c.z) : 0u, ct)) {
                                        sink.store(ticket, found, true);
                                        occluded = true;
                                    }
                                }
                            }
                            if (occluded) give_back = true;
                            else {
                                char *rec = records + (size_t)slot * RT_RP_SLOT_BYTES;
                                *(v4f *)(rec) = (v4f){r.o.x, r.o.y, r.o.z, r.tmin};
                                *(v4f *)(rec + 16) = (v4f){r.d.x, r.d.y, r.d.z, r.tmax};
                                *(v4f *)(rec + 32) = (v4f){__int_as_float(node), __int_as_float(sp), __uint_as_float(ticket), 0.0f};
                                alive = true;
                            }
                            started = true;
                        } else {
                            if (r.tmax == RT_TMAX_SKIPPED) skipped = true;
                            sink.store(ticket, make_miss(r), traced);
                            give_back = true;
                        }
                    }
                    n_traced += (uint32_t)__popcll(__ballot(started));
                    n_skipped += (uint32_t)__popcll(__ballot(skipped));
                    const uint32_t ns = (uint32_t)__popcll(__ballot(started && !give_back));          // the rays that took a slot
                    if (ns && lane == 0u) atomicAdd(&sh->in_flight, (int)ns);
                    rp_push(fq, &sh->fq_tail, give_back, free_slot);
                    chunk_next += f;
                }
                __builtin_amdgcn_s_setprio(0);
            }
            if (__ballot(alive) == 0ull) {
                if (rp_ld(abort_flag)) break;
                if (exhausted && rp_ld((volatile uint32_t *)&sh->in_flight) == 0u && rp_ld(&sh->nq_tail) == rp_ld(&sh->nq_head)) break;
                if (exhausted || rp_ld(&sh->fq_tail) == rp_ld(&sh->fq_head)) {        // nothing to start: wait for rays to come back or slots to come free
                    __builtin_amdgcn_s_sleep(8);
                    if (++watchdog > RT_RP_WATCHDOG) { *abort_flag = 1u; break; }
                }
                continue;
            }
            // ---- node steps, until enough lanes have left to make a refill worth it ----
            const int walking0 = __popcll(__ballot(alive && node_is_internal(node)));
            while (alive && node_is_internal(node) && sp <= STACK - (RT_WIDE - 1)) {
                s_steps++; s_step_lanes += (uint32_t)__popcll(__builtin_amdgcn_read_exec());
                wide_step<false, true>(nodes, topl, top_lim, ri, tmin, tmax, st, node, sp);
                const int walking = __popcll(__ballot(alive && node_is_internal(node)));
                if (walking + RT_RP_REFILL <= walking0) break;          // enough lanes have left: hand their rays on and refill
            }
            while (alive && node_is_internal(node) && sp > STACK - (RT_WIDE - 1))
                wide_step<true, true>(nodes, topl, top_lim, ri, tmin, tmax, st, node, sp);
            // ---- lanes whose ray stands on a leaf, or has nothing left to visit ----
            __builtin_amdgcn_s_setprio(RT_LEAF_PRIO);
            const bool leaving = alive && !node_is_internal(node);
            bool to_leaf = false, done = false;
            if (leaving) {
                if (node == RT_NODE_EMPTY) {              // nothing between the point and the light
                    RayD r; r.o = ri.o; r.d = ri.o; r.tmin = tmin; r.tmax = tmax;
                    sink.store(ticket, make_miss(r), true);
                    done = true;
                } else {
                    char *rec = records + (size_t)slot * RT_RP_SLOT_BYTES;
                    *(float2 *)(rec + 32) = make_float2(__int_as_float(node), __int_as_float(sp));
                    to_leaf = true;
                }
                alive = false;
            }
            if (__ballot(to_leaf)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            rp_push(lq, &sh->lq_tail, to_leaf, slot);
            s_lq += (uint32_t)__popcll(__ballot(to_leaf));
            const uint32_t nd = (uint32_t)__popcll(__ballot(done));
            rp_push(fq, &sh->fq_tail, done, slot);
            if (nd && lane == 0u) atomicSub(&sh->in_flight, (int)nd);
            __builtin_amdgcn_s_setprio(0);
        }
        if (lane == 0u) atomicAdd(&sh->nodes_exited, 1u);
        if (lane == 0u) { atomicAdd(&stats[0], (unsigned long long)s_steps); atomicAdd(&stats[1], (unsigned long long)s_step_lanes); atomicAdd(&stats[4], (unsigned long long)s_lq);
                          atomicAdd(&stats[6], (unsigned long long)s_refills); if (rp_ld(abort_flag)) atomicAdd(&stats[7], 1ull); }
        if (traced_counter) {
            if (lane == 0u && n_traced) atomicAdd(traced_counter, n_traced);
            if (lane == 0u && n_skipped) atomicAdd(traced_counter + 1, n_skipped);
        }
    } else {
        // ================================= the leaf wave =================================
        uint32_t polls = 0;
        for (;;) {
            if (rp_ld(abort_flag)) break;
            const uint32_t avail = rp_ld(&sh->lq_tail) - rp_ld(&sh->lq_head);
            if (avail < RT_RP_LEAF_MIN && polls < 6u && rp_ld(&sh->nodes_exited) < RT_RP_NODE_WAVES) {       // a fuller pass is a cheaper pass
                __builtin_amdgcn_s_sleep(4);
                polls++;
                continue;
            }
            uint32_t slot = 0;
            const uint32_t k = rp_pop(lq, &sh->lq_head, &sh->lq_tail, 64u, lane, slot, abort_flag);
            if (k == 0u) {
                if (rp_ld(&sh->nodes_exited) >= RT_RP_NODE_WAVES) break;
                __builtin_amdgcn_s_sleep(4);
                if (++watchdog > RT_RP_WATCHDOG) { *abort_flag = 1u; break; }
                continue;
            }
            polls = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool active = lane < k;
            RayD r;
            ObjRay cur;
            uint32_t ticket = RT_NO_HIT;
            int node = RT_NODE_EMPTY, sp = 0;
            r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1); r.tmin = 0; r.tmax = 0;
            char *rec = records + (size_t)slot * RT_RP_SLOT_BYTES;
            if (active) {
                const v4f a = ldg16(rec, 0), b = ldg16(rec, 16), c = ldg16(rec, 32);
                r.o = mk3(a.x, a.y, a.z); r.tmin = a.w; r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
                node = __float_as_int(c.x); sp = __float_as_int(c.y); ticket = __float_as_uint(c.z);
                st.lds = smem + slot; st.deep = deep0 ? deep0 + slot : nullptr;
            }
            cur.o = r.o; cur.d = r.d; cur.ri = make_inv(r.o, r.d);
            bool to_node = false, done = false;
            while (__ballot(active)) {
                s_passes++; s_pass_lanes += (uint32_t)__popcll(__ballot(active));
                if (active) {
                    if (node == RT_NODE_EMPTY) {
                        sink.store(ticket, make_miss(r), true);
                        done = true; active = false;
                    } else if (node_is_internal(node)) {
                        *(float2 *)(rec + 32) = make_float2(__int_as_float(node), __int_as_float(sp));
                        to_node = true; active = false;
                    } else {
                        const uint32_t code = (uint32_t)~node;
                        const uint32_t first_tri = code >> 3, cnt = (code & 7u) + 1u;
                        bool hit = false;
                        for (uint32_t kk = 0; kk < cnt; kk++) {
                            const char *tp = (const char *)(tris + first_tri + kk);
                            const v4f a = ldg16(tp, 0), b = ldg16(tp, 16), c = ldg16(tp, 32);
                            const uint32_t prim = __float_as_uint(c.y);
                            HitD found = make_miss(r);
                            if (accept_candidate<REFS ? 1 : 0>(*in0, 0u, prim, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, cur.ri, cur, cull, found,
                                                               REFS ? __float_as_uint(c.z) : 0u, first_tri + kk)) {
                                if constexpr (src_has_cache<Src>::value) src.template remember<false>((uint32_t)st.lds[(STACK - 1) * BLOCK], first_tri + kk, 0u);
                                sink.store(ticket, found, true);
                                hit = true;
                                break;
                            }
                        }
                        if (hit) { done = true; active = false; }
                        else if (sp > 0) { sp--; node = st.read(sp); }
                        else node = RT_NODE_EMPTY;
                    }
                }
            }
            if (__ballot(to_node)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            rp_push(nq, &sh->nq_tail, to_node, slot);
            s_nq += (uint32_t)__popcll(__ballot(to_node));
            const uint32_t nd = (uint32_t)__popcll(__ballot(done));
            rp_push(fq, &sh->fq_tail, done, slot);
            if (nd && lane == 0u) atomicSub(&sh->in_flight, (int)nd);
        }
        if (lane == 0u) { atomicAdd(&stats[2], (unsigned long long)s_passes); atomicAdd(&stats[3], (unsigned long long)s_pass_lanes); atomicAdd(&stats[5], (unsigned long long)s_nq);
                          if (rp_ld(abort_flag)) atomicAdd(&stats[7], 1ull); }
    }
}

}  // namespace rtd
