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
//     most of a wave's 64 lanes parked while the longest ray finishes.  (Round 6: an any-hit
//     walk over a source with cached first candidates tests a new ray's candidate at the
//     refill, goes round until the wave is full, and -- while its rays are short-lived --
//     waits for 32 / 40 free lanes before it starts: RT_REFILL_TESTS, RT_REFILL_ENTER_TESTS*);
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
//   * the stack is LDS resident, stack[row][lane-in-block]: one dword per lane per
//     row, bank = lane mod 32, conflict free for both halves of a wave; a fixed
//     number of rows whatever the tree, the rare deeper walk continues in global rows.
//
// Exactness: culling uses the same monotone slab test as the oracle and candidates
// are validated exactly as the canonical definition prescribes (rt_trace_device.h),
// so the visiting order, leaf collapsing and ray-to-lane assignment used here cannot
// change any result bit.
//
// What was measured and is NOT here (each bit-exact, each slower or flat; the code of every one is the patch
// dxrexperiments_amd/csrc/experiments/r03_traversal_experiments.patch, the numbers are under profiles/r03/): eight-wide 128-B nodes
// (experiments/rt_wide8_step.h, -DRT_WIDE=8), splitting the last rays of a launch over idle lanes, touching the node
// that will be popped next, the tops of the two most-instanced BLASes in LDS, packed-fp32 plane arithmetic, a
// three-exchange sort, and the instrumentation builds that counted lanes and timed waves from the inside.
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
#ifndef RT_POOL_CHUNK_SETS
#define RT_POOL_CHUNK_SETS 128u         // ... in the long launches of sets of frames (round 4, profiles/r04/pool_chunk.txt: 64 / 128 / 256 rays: 1.512 / 1.467 / 1.466 ms per frame
#endif                                  //     in sets -- and 2.06 / 2.15 / 2.33 ms frame by frame, where the end of a launch is a third of it: single frames keep 64)

#ifndef RT_POOL_CHUNK_SETS_ANYHIT
#define RT_POOL_CHUNK_SETS_ANYHIT 256u  // ... and the any-hit launch of a set 256 (round 6, with the late refills of short-lived rays: any-hit stage 0.505 -> 0.491 ms; the closest-hit
#endif                                  //     launches of the 10 M-triangle scene lose 2 % with it and keep 128: profiles/r06/refill_thresholds.txt)
#ifndef RT_POOL_GROUPS
#define RT_POOL_GROUPS 32u              // chunk counters per traversal launch (8: 3.08, 32: 3.07, 128: 3.09, 512: 3.12 ms; static: 3.21)
#endif
#define RT_POOL_STRIDE 32u              // words between two counters: one 128-B L2 line each
#ifndef RT_SENTINEL_INLINE
#define RT_SENTINEL_INLINE 1            // two-level walks leave a BLAS inside the node loop (round 4: the 4096-instance frame 4.70 -> 4.60 ms, profiles/r04/c4_variants.txt)
#endif
#ifndef RT_POOL_XCD
#define RT_POOL_XCD 1                   // any-hit launches: every XCD works through one contiguous eighth of the queue (then helps its neighbours)
#endif
#ifndef RT_REFILL_RAYS_PER_PASS
#define RT_REFILL_RAYS_PER_PASS 28u     // "loading rays is the work": the wave has loaded more rays than this per pass of its outer loop (refill, node loop, leaf phase)
#endif
#ifndef RT_REFILL_ENTER_TESTS
#define RT_REFILL_ENTER_TESTS 32        // single frames (64-ray chunks)
#endif
#ifndef RT_REFILL_STOP_TESTS
#define RT_REFILL_STOP_TESTS 32
#endif
#ifndef RT_REFILL_ENTER_TESTS_SETS
#define RT_REFILL_ENTER_TESTS_SETS 48   // sets of frames (256-ray chunks)
#endif
#ifndef RT_REFILL_STOP_TESTS_SETS
#define RT_REFILL_STOP_TESTS_SETS 24
#endif
#ifndef RT_REFILL_TESTS_TWO_LEVEL
#define RT_REFILL_TESTS_TWO_LEVEL 1     // ... in two-level walks too (the candidate is tested in its instance's space)
#endif
#ifndef RT_REFILL_TESTS
#define RT_REFILL_TESTS 1               // (round 6) a new shadow ray's cached candidate is tested at the refill, and the refill repeats until the wave is full
#endif
#ifndef RT_EXIT_K
#define RT_EXIT_K 1                     // leave the node loop once (lanes still on internal nodes) * K < lanes waiting on a leaf
                                        //   (four-wide nodes, ms per frame 1080p / 10 M triangles 4K: K = 0 3.31 / 21.9, 1 2.80 / 15.5, 2 2.86 / 16.5, 3 2.88 / 16.9)
#endif

}  // namespace rtd
#include "rt_wide_step.h"        // LaneStack + wide_step: the step on one four-wide node
namespace rtd {

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

// A ray source may keep a cache of first candidates for unordered any-hit searches (the pipeline's shadow
// cache): uint32_t cached_leaf(ticket, const RayD &, uint32_t &slot, uint32_t &instance) -> index into the instance's sorted triangle
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
//   struct Sink { void store(uint32_t ticket, const HitD &h, bool traced) const; };
// A source may hand out a TICKET with the ray -- bool load(uint32_t i, RayD &r, uint32_t &ticket) -- which the walk carries in
// place of i and gives to the sink with the ray's result (the shared shadow queue enumerates its rays in one order and stores
// them in another); without it the ticket is i.
template <class S> RT_DEV auto load_ray_of(const S &src, uint32_t i, RayD &r, uint32_t &ticket, int) -> decltype(src.load(i, r, ticket)) { return src.load(i, r, ticket); }
template <class S> RT_DEV bool load_ray_of(const S &src, uint32_t i, RayD &r, uint32_t &ticket, long) { ticket = i; return src.load(i, r); }

// COUNT: the walk-counting instantiation (rt_pipeline_count_walk): the same walk, plus per-lane tallies of what it
// fetches -- 64-B nodes from global memory, nodes from the LDS-resident top, 48-B triangle records, the 96-B traversal prefix of
// instance records -- summed into walk[0..5] = rays, nodes from global memory, nodes from LDS, triangles, instance entries,
// distinct 64-B lines (node lines de-duplicated across the lanes of each wave step + the lines the triangle records span);
// walk[6] = max over rays of (node steps << 32 | ray index), the longest single walk (a tail detector);
// walk[7..9] = what the WAVES did: node steps issued (one per wave per pass of the node loop), leaf phases, triangle
// iterations of those phases -- with walk[1] + walk[2] (the lanes live in the node steps) and walk[3] (the lanes live in
// the triangle iterations) the lane utilisation of the two halves of the walk; walk[10] = the node part of walk[5].
// The per-ray numbers depend on the ray and the tree only, not on chunking or lane assignment; the per-wave ones on both.
// NO_DEEP (round 5): a launch without rows beyond the LDS ones -- the one-tile-per-wave primary launch, whose one thread per pixel slot made
// those rows 224 MB per 1080p frame of a set.  A lane whose walk would need such a row gives the ray up: src.overflow(ticket) puts it on a
// list that a small persistent launch with rows (k_primary_retry, rt_pipeline.hip) walks from the start; nothing is stored for it here.
#define RT_WALK_WORDS 11
// REFS (round 5): the scene holds triangles that are split into references (rt_refs.h): the leaf phase passes every record's reference mark
// to the candidate test.  Instantiations for scenes without such triangles are the kernels of round 4, instruction for instruction.
template <int STACK, int BLOCK, bool TWO_LEVEL, uint32_t CHUNK, bool ANYHIT = false, bool COUNT = false, bool NO_DEEP = false, bool REFS = false, class Src, class Sink>
RT_DEV void trace_wave(const SceneDev &sc, const Src &src, const Sink &sink, uint32_t *pool, int *smem, uint32_t *traced_counter,
                       unsigned long long *walk = nullptr)
{
    uint32_t rf_loaded = 0, rf_passes = 0;      // (REFILL_TESTS; both only ever changed where the whole wave runs: wave-uniform) rays this wave loaded / passes of its outer loop
    uint32_t n_traced = 0;           // rays this WAVE actually traversed (statistics; wave-uniform: a scalar register, not a lane's)
    uint32_t n_skipped = 0;          // any-hit launches: queue slots marked RT_TMAX_SKIPPED (likewise)
    uint32_t wk_glob = 0, wk_top = 0, wk_tri = 0, wk_inst = 0, wk_lines = 0;
    uint32_t wk_ray0 = 0;                        // node steps tallied when the lane's current ray started
    unsigned long long wk_longest = 0;           // (node steps << 32 | ray index) of the lane's longest walk
    uint32_t wv_steps = 0, wv_leaf = 0, wv_tri = 0;      // what the wave did (COUNT): tallied by the first live lane of each step
    uint32_t wk_node_lines = 0;                  // the node part of wk_lines
    const uint32_t total = src.count();
    const uint32_t flags = src.flags();
    // ANYHIT instantiations (unordered walks) are only launched for ACCEPT_FIRST_HIT searches: as a compile-time fact it lets the
    // running best hit -- which then never changes before the ray ends -- out of the registers
    const bool first = ANYHIT ? true : (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0;
    LaneStack<STACK, BLOCK> st;
    st.lds = smem + threadIdx.x;
    st.threads = gridDim.x * BLOCK;
    st.deep = sc.deep_stack && !NO_DEEP ? sc.deep_stack + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr;

    // single-level scenes (one identity instance) walk the BLAS directly in world space
    const InstanceRec *in0 = sc.inst;
    const WNode *blas_nodes0 = TWO_LEVEL ? nullptr : in0->wide;
    const TriRec *tris0 = TWO_LEVEL ? nullptr : in0->tris;
    // the LDS-resident top of the tree: smem rows STACK .. STACK + RT_TOP_ROWS - 1 hold nodes 0 .. top_n - 1 (breadth-first
    // numbering) of the structure a ray starts in: the BLAS of a single-level scene, the TLAS of a two-level one
    int *topl = smem + STACK * BLOCK;
    if (sc.top_n != 0) {
        const int *src_top = (const int *)(TWO_LEVEL ? sc.tlas_wide : blas_nodes0);
        for (uint32_t i = threadIdx.x; i < sc.top_n * RT_TOP_WORDS; i += BLOCK) topl[i] = src_top[(i / RT_TOP_WORDS) * (uint32_t)(sizeof(WNode) / 4) + i % RT_TOP_WORDS];
    }
    const int *const top_cur = topl;
    if (sc.top_n != 0) __syncthreads();
    const int root0 = TWO_LEVEL ? sc.tlas_root_code : in0->root_code;
    uint32_t top_lim = sc.top_n;                  // node indices below this are read from LDS (two-level: 0 while inside a BLAS)

    bool alive = false;
    bool exhausted = false;          // wave-uniform: the global pool has nothing left
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the chunk of the queue being handed out
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    uint32_t next_chunk = blockIdx.x * (BLOCK / 64) + threadIdx.x / 64;   // wave-uniform
    const uint32_t n_groups = gridDim.x < RT_POOL_GROUPS ? gridDim.x : RT_POOL_GROUPS;
    const uint32_t pool_group = blockIdx.x % n_groups;
#if RT_POOL_XCD
    uint32_t xcd_steal = 0;          // wave-uniform: how many XCDs' eighths of the queue this wave has seen run dry
#endif
    uint32_t idx = 0;                 // the ticket of the lane's ray: what the sink gets with its result
    RayD r;
    RayInv wri;
    HitD best;
    int node = RT_NODE_EMPTY;
    int sp = 0;
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
        // (round 6) REFILL_TESTS: any-hit walks of single-level scenes whose source keeps first candidates (the pipeline's shadow cache) test a new
        // ray's candidate RIGHT HERE -- every lane of the refill has one, so the triangle test runs with the whole refill live -- and a ray it occludes
        // never takes the lane: the refill goes round again until the wave is full (or the pool dry).  Until round 5 the candidate went in front of
        // the root as a one-triangle leaf: its lane sat through the wave's next node loop doing nothing, was tested in the leaf phase, and most of
        // the time (a cached occluder usually still occludes) ended there -- node steps of the shadow stage ran with 0.4 - 0.5 of their lanes by
        // the hardware's count (profiles/r05/c2s_lanes.md) against 0.66 in the walk without the cache.
        constexpr bool REFILL_TESTS = src_has_cache<Src>::value && ANYHIT && !COUNT && (!TWO_LEVEL || RT_REFILL_TESTS_TWO_LEVEL) && RT_REFILL_TESTS;
        unsigned long long idle = __ballot(!alive);
        int n_idle = __popcll(idle);
        // (round 6) A refill pass of a walk that tests candidates is ~500 instructions (ray, light ray, 1 / d, cache cell, candidate: half of the any-hit stage's
        // instructions on the bench scene, where a shadow ray takes 3.7 node steps on average because two in three end at their candidate) and should run with many
        // lanes -- but only where loading rays IS the work: where the rays that walk walk long (the stress scene: 40 steps; a cold cache) waiting for more free
        // lanes only empties the node steps.  So the wave keeps count: while it has loaded more than RT_REFILL_RAYS_PER_PASS rays per pass of this loop (short lives: the
        // bench scene's waves load more; the stress scene's, whose rays walk 40 steps, fewer: 28 keeps the one's gain and the other's time), a refill starts at RT_REFILL_ENTER_TESTS free lanes and goes round until
        // fewer than RT_REFILL_STOP_TESTS are free; otherwise at RT_REFILL_LANES as every other walk.  (Both tallies change only where the whole wave runs -- a tally
        // kept inside the node loop, which lanes leave one by one, differs from lane to lane and sends the lanes of a wave different ways at the refill: a hang.)
        // Sets of frames (CHUNK 128) and single frames have their own pair of thresholds (profiles/r06/refill_thresholds.txt).
        if (REFILL_TESTS) rf_passes++;
        const bool mostly_answered = REFILL_TESTS && rf_loaded >= RT_REFILL_RAYS_PER_PASS * rf_passes && rf_loaded >= 64u;
        const int REFILL_ENTER = mostly_answered ? (CHUNK > 64u ? RT_REFILL_ENTER_TESTS_SETS : RT_REFILL_ENTER_TESTS) : RT_REFILL_LANES;
        const int REFILL_STOP = mostly_answered ? (CHUNK > 64u ? RT_REFILL_STOP_TESTS_SETS : RT_REFILL_STOP_TESTS) : RT_REFILL_LANES;
        for (int refill_round = 0; !exhausted && n_idle >= (refill_round == 0 ? REFILL_ENTER : REFILL_STOP) && (refill_round == 0 || REFILL_TESTS); refill_round++) {
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
#if RT_POOL_XCD
                // Any-hit launches (round 4): every XCD owns one contiguous EIGHTH of the queue -- its four groups deal that eighth among
                // themselves -- and moves on to the next XCD's eighth when its own is used up: the shadow rays of one part of the image
                // (their origins: the compaction keeps tile order) share an L2.  Any-hit stage -2.4 % on both workloads; the closest-hit
                // queues lose what the any-hit queue gains (+0.5 ... +2 %) and keep the round-robin deal (profiles/r04/pool_xcd.txt;
                // round 3's per-GROUP bands lost frame by frame: profiles/r03/pool_contiguous.txt).
                if (ANYHIT && pool && n_groups == RT_POOL_GROUPS) {
                    const uint32_t chunks = (total + CHUNK - 1u) / CHUNK, per = (chunks + 7u) / 8u;
                    cidx = 0x4000000u;
                    while (xcd_steal < 8u) {
                        const uint32_t x = (pool_group + xcd_steal) & 7u;
                        const uint32_t g = x + (pool_group & ~7u);
                        uint32_t k = 0;
                        if ((threadIdx.x & 63u) == 0u) k = atomicAdd(&pool[g * RT_POOL_STRIDE], 1u);
                        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                        const uint32_t c = x * per + (pool_group >> 3) + k * (RT_POOL_GROUPS / 8u);
                        if (c < (x + 1u) * per && c < chunks) { cidx = c; break; }
                        xcd_steal++;
                    }
                } else
#endif
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
                const bool traced = load_ray_of(src, my, r, idx, 0);
                best = make_miss(r);
                if (traced && r.tmax > r.tmin && sc.n_inst != 0) {
                    wri = make_inv(r.o, r.d);
                    cur.o = r.o; cur.d = r.d; cur.ri = wri;
                    node = root0;
                    sp = 0;
                    if (TWO_LEVEL) { nodes = sc.tlas_wide; in_blas = false; top_lim = sc.top_n; }
                    bool occluded_at_once = false;
                    if constexpr (REFILL_TESTS) {
                        uint32_t slot, ci;
                        const uint32_t ct = src.template cached_leaf<TWO_LEVEL>(idx, r, slot, ci);
                        st.lds[(STACK - 1) * BLOCK] = (int)slot;
                        // (two-level: an entry is a (triangle, instance) pair, tested in that instance's space; it is only ever tested if it names a triangle that exists)
                        if (ct != RT_NO_HIT && (!TWO_LEVEL || (ci < sc.n_inst && ct < sc.inst[ci].n_recs))) {
                            const InstanceRec *cin = TWO_LEVEL ? sc.inst + ci : in0;
                            const ObjRay orr = TWO_LEVEL ? to_object(*cin, r) : cur;
                            const char *tp = (const char *)((TWO_LEVEL ? cin->tris : tris0) + ct);
                            const v4f a = ldg16(tp, 0), b = ldg16(tp, 16), c = ldg16(tp, 32);
                            HitD found = best;
                            if (accept_candidate<REFS ? 1 : 0>(*cin, TWO_LEVEL ? ci : 0u, __float_as_uint(c.y), mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, wri, orr, cull, found,
                                                               REFS ? __float_as_uint(c.z) : 0u, ct)) {
                                sink.store(idx, found, true);
                                occluded_at_once = true;
                            }
                        }
                    } else
                    if constexpr (src_has_cache<Src>::value && ANYHIT && !COUNT) {
                        // the triangle that answered this question last time goes first: a one-triangle leaf in front of the root
                        uint32_t slot, ci;
                        const uint32_t ct = src.template cached_leaf<TWO_LEVEL>(idx, r, slot, ci);       // (idx: the ray's ticket)
                        st.lds[(STACK - 1) * BLOCK] = (int)slot;
                        // (an entry is only ever tested if it names a triangle that exists: the table is emptied with the scene, but a pair
                        // torn between two writers, or a table handed over by mistake, must not read past an array)
                        if (ct != RT_NO_HIT && (!TWO_LEVEL || (ci < sc.n_inst && ct < sc.inst[ci].n_recs))) {      // (single level: the source checks against its own count)
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
                    alive = !occluded_at_once;
                    started = true;
                    if (COUNT) wk_ray0 = wk_glob + wk_top;
                } else {
                    if (ANYHIT && r.tmax == RT_TMAX_SKIPPED) skipped = true;
                    sink.store(idx, best, traced);
                }
            }
            n_traced += (uint32_t)__popcll(__ballot(started));
            if (REFILL_TESTS) rf_loaded += (uint32_t)__popcll(__ballot(started));
            if (ANYHIT) n_skipped += (uint32_t)__popcll(__ballot(skipped));
            chunk_next += (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
#if RT_REFILL_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            idle = __ballot(!alive);
            n_idle = __popcll(idle);
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---- walk internal nodes until this lane stands on a leaf (or runs dry) ----------
#if RT_EXIT_K > 0
        const int n_alive = __popcll(__ballot(alive));
#endif
        while (alive && node_is_internal(node) && sp <= STACK - (RT_WIDE - 1)) {
            if (COUNT) {
                if ((uint32_t)node < top_lim) wk_top++; else wk_glob++;
                const uint32_t dl = distinct_node_lines(node, !((uint32_t)node < top_lim));
                if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) { wk_lines += (RT_WIDE == 8 ? 2u : 1u) * dl; wk_node_lines += (RT_WIDE == 8 ? 2u : 1u) * dl; wv_steps++; }       // 64-B lines (96 B of a 128-B record: two)
            }
            wide_step<false, ANYHIT>(nodes, top_cur, top_lim, cur.ri, r.tmin, ANYHIT ? r.tmax : best.t, st, node, sp);
#if RT_SENTINEL_INLINE
            // (two-level walks) a pop that brings up the sentinel ends the walk of a BLAS: the lane returns to the TLAS here, inside the
            // node loop, and pops what lies beneath -- instead of waiting for the wave's next leaf phase to do only that
            if (TWO_LEVEL && node == RT_NODE_SENTINEL) {
                in_blas = false;
                nodes = sc.tlas_wide;
                top_lim = sc.top_n;
                cur.o = r.o; cur.d = r.d; cur.ri = wri;
                if (sp > 0) { sp--; node = st.lds[sp * BLOCK]; }
                else node = RT_NODE_EMPTY;
            }
#endif
#if RT_EXIT_K > 0
            // stragglers: most of the wave already waits on a leaf -> run the leaf phase now, come back after
            const int walking = __popcll(__ballot(alive && node_is_internal(node)));
            if (walking * RT_EXIT_K < n_alive - walking) break;
#endif
        }
        // (NO_DEEP: no rows beyond LDS in this launch -- the ray goes to the retry list, at the end of this pass)
        bool gave_up = NO_DEEP && alive && node_is_internal(node) && sp > STACK - (RT_WIDE - 1);
        // lanes whose stack has outgrown the LDS rows walk on with the global rows until it fits again
        if constexpr (!NO_DEEP) {
            while (alive && node_is_internal(node) && sp > STACK - (RT_WIDE - 1)) {
                if (COUNT) {
                    if ((uint32_t)node < top_lim) wk_top++; else wk_glob++;
                    const uint32_t dl = distinct_node_lines(node, !((uint32_t)node < top_lim));
                    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) { wk_lines += (RT_WIDE == 8 ? 2u : 1u) * dl; wk_node_lines += (RT_WIDE == 8 ? 2u : 1u) * dl; wv_steps++; }       // 64-B lines (96 B of a 128-B record: two)
                }
                wide_step<true, ANYHIT>(nodes, top_cur, top_lim, cur.ri, r.tmin, ANYHIT ? r.tmax : best.t, st, node, sp);
            }
        }

        // ---- leaves, instance entry / exit, termination -----------------------------------
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(RT_LEAF_PRIO);
#endif
        if (alive && !node_is_internal(node)) {
            bool pop = true;
            if (COUNT && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) wv_leaf++;
            if (node == RT_NODE_EMPTY) {
                sink.store(idx, ANYHIT ? make_miss(r) : best, true);
                alive = false;
                pop = false;
                if (COUNT) { const unsigned long long w = ((unsigned long long)(wk_glob + wk_top - wk_ray0) << 32) | idx; wk_longest = w > wk_longest ? w : wk_longest; }
            } else if (TWO_LEVEL && node == RT_NODE_SENTINEL) {
                in_blas = false;
                nodes = sc.tlas_wide;
                top_lim = sc.top_n;
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
                if (NO_DEEP && enter && sp >= STACK) {        // (the sentinel would need a row beyond LDS)
                    gave_up = true;
                    pop = false;
                } else if (enter) {
                    cur = to_object(*in, r);
                    nodes = in->wide;
                    tris = in->tris;
                    in_blas = true;
                    top_lim = 0;
                    if constexpr (NO_DEEP) st.lds[sp * BLOCK] = RT_NODE_SENTINEL;      // (sp < STACK here)
                    else st.write(sp, RT_NODE_SENTINEL);
                    sp++;
                    node = in->root_code;
                    pop = false;
                }
            } else {
                const uint32_t code = (uint32_t)~node;
                const uint32_t first_tri = code >> 3, cnt = (code & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; k++) {
                    if (COUNT) {
                        wk_tri++;
                        const uint32_t by = (first_tri + k) * 48u;
                        wk_lines += 1u + ((by & 63u) > 16u ? 1u : 0u);       // a 48-B record spans one or two lines
                        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_read_exec())) wv_tri++;
                    }
                    const char *tp = (const char *)(tris + first_tri + k);
                    const v4f a = ldg16(tp, 0), b = ldg16(tp, 16), c = ldg16(tp, 32);
                    const uint32_t prim = __float_as_uint(c.y);
                    HitD found = ANYHIT ? make_miss(r) : best;          // (any-hit: the running best never changes before the ray ends)
                    const bool accepted = accept_candidate<REFS ? 1 : 0>(*in, ii, prim, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x), r, wri, cur, cull, found,
                                                                         REFS ? __float_as_uint(c.z) : 0u, first_tri + k);
                    if (!ANYHIT) best = found;
                    if (accepted && first) {
                        if constexpr (src_has_cache<Src>::value && ANYHIT && !COUNT) src.template remember<TWO_LEVEL>((uint32_t)st.lds[(STACK - 1) * BLOCK], first_tri + k, ii);
                        sink.store(idx, found, true);
                        alive = false;
                        pop = false;
                        if (COUNT) { const unsigned long long w = ((unsigned long long)(wk_glob + wk_top - wk_ray0) << 32) | idx; wk_longest = w > wk_longest ? w : wk_longest; }
                        break;
                    }
                }
            }
            if (pop) {
                if (sp > 0) {
                    sp--;
                    if constexpr (NO_DEEP) node = st.lds[sp * BLOCK];      // (no entry ever lies beyond the LDS rows)
                    else node = st.read(sp);
                }
                else node = RT_NODE_EMPTY;
            }
        }
        if constexpr (NO_DEEP) {
            if (gave_up) { src.overflow(idx); alive = false; }
        }
#if RT_LEAF_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    if (COUNT && walk) {
        const unsigned long long w9[10] = {(threadIdx.x & 63u) == 0u ? n_traced : 0u, wk_glob, wk_top, wk_tri, wk_inst, wk_lines, wv_steps, wv_leaf, wv_tri, wk_node_lines};
        for (int k = 0; k < 10; k++) {
            unsigned long long v = w9[k];
            for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
            if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&walk[k < 6 ? k : k + 1], v);
        }
        atomicMax(&walk[6], wk_longest);
    }
    if (traced_counter) {            // one no-return atomic per persistent wave
        if ((threadIdx.x & 63u) == 0u && n_traced) atomicAdd(traced_counter, n_traced);
        if (ANYHIT && (threadIdx.x & 63u) == 0u && n_skipped) atomicAdd(traced_counter + 1, n_skipped);       // (the word behind an any-hit launch's ray counter tallies its skipped slots)
    }
}

}  // namespace rtd
