// rt_bvh_wide.hip -- the eight-wide, quantised traversal layout (WNode, rt_internal.h).
//
// Input: a binary tree over the sorted leaves (the PLOC tree of rt_bvh_ploc.hip, or the canonical LBVH for a TLAS,
// tiny meshes and option fast_bvh=lbvh).  Output: nodes with up to EIGHT children in one 128-B-aligned record of which the
// traversal reads the first 96 B.  Why eight (round 3, profiles/r03/slab_fetch.txt): the L2 line and every fabric / HBM
// request are 128 B, so a 64-B node already moved 128 B per miss (TCC_EA0_RDREQ_128B = 99.7 % of the reads on the 10 M
// triangle scene); from HBM a lane's 128-B block costs exactly what its 64-B half costs (291 vs 292 ns per wave step per
// CU), from the Infinity Cache 96 B of one block cost 1.19x a 64-B line (128 B: 1.57x) -- and a tree of eight-wide nodes
// is a third shallower, so a ray makes a third fewer dependent fetches.  Child boxes are quantised to one byte per plane
// on a power-of-two grid anchored at the node's own box: origin, three scale exponents, 48 plane bytes and eight child
// codes are 96 B.
//
// Collapse: breadth first, two launches per level and no host round trip inside a batch of levels (rt_level_scan.h).  A
// frontier element is a binary node that becomes a wide node: its two children are taken, and while fewer than eight, the
// child with the largest surface that is not a leaf is replaced by its own two children (surface-area greedy, as in Wald
// et al. 2008 / Ylitie et al. 2017).  Node numbers come from prefix sums (deterministic), level after level, so the array
// is in breadth-first order and its first RT_TOP_NODES entries are the LDS-resident top of the traversal kernels; the
// internal children of a node have consecutive numbers in slot order.
//
// Slots: the traversal visits the hit children of a node in the order of (slot XOR direction octant) and never sorts by
// distance, so the builder places the children (Ylitie et al. 2017, section 3.3): slot s stands for the box diagonal with
// sign +1 on the axes whose bit is set in s, the cost of putting child c there is dot(centre(c) - centre(node), diagonal(s)),
// and the (child, slot) pairs are fixed greedily, dearest first.  A ray whose direction is negative on the axes in `oct`
// then meets slot `oct` first and slot `7 - oct` last.
//
// Exactness: plane = fma(q, scale, origin) is evaluated here with the same expression the traversal uses; every lo
// plane is stepped down and every hi plane up until the decoded box CONTAINS the child's true box.  The slab test is
// monotone under box inclusion (DESIGN.md "Exactness rule"), so culling against the decoded boxes never loses a
// candidate the canonical definition accepts.  An axis that cannot be quantised at all (non-finite or overflowing
// extents: huge instance boxes, NaN / inf vertices) gets the scale exponent 255 and q = 0 planes: the decoded planes are
// NaN, which every slab test here ignores, i.e. that axis never culls -- slower, never wrong.
#include "rt_internal.h"

#include "rt_level_scan.h"

#include <cstddef>
#include <cstring>

namespace {

constexpr unsigned WB = 256;
constexpr uint32_t NO_KID = 0xFFFFFFFFu;
constexpr uint32_t RT_WIDE_MAX_LEVELS = 254;            // of wide nodes; a collapsed level spans one to two binary levels

struct Box6 { float lo[3]; float hi[3]; };

__device__ __forceinline__ float box_area(const Box6 &b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// The surface-area-optimal collapse (Ylitie, Karras, Laine 2017, section 3.2), one record per internal binary node:
//   cost[i-1], i = 1 .. RT_WIDE: the cheapest way to represent the subtree as a forest of at most i roots (a root is a
//                                leaf or a wide node), in units of (half) surface area;
//   pick[i-1]:  how -- i = 1: 0 leaf, 1 wide node; i >= 2: 0 = as with i - 1 roots, k >= 1: k roots from the left child
//                                and i - k from the right;
//   spread:     the k of the best split of ALL RT_WIDE slots among the two children (what a wide node made of this
//                                binary node does with its slots).
struct SahRecord { float cost[8]; uint8_t pick[8]; uint32_t spread; uint32_t pad; };      // 48 B
static_assert(sizeof(SahRecord) == 48, "SahRecord layout");

struct TreeView {
    const uint32_t *left, *right;      // indexed by id - n
    const Box6 *box;                   // indexed by id
    const uint32_t *size, *offset;     // indexed by id
    const uint32_t *leaf_prim;         // TLAS: instance of leaf id; nullptr for a BLAS
    const SahRecord *sah;              // indexed by id - n; nullptr: the area-greedy collapse
    uint32_t n, leaf_max;
    __device__ bool may_be_leaf(uint32_t id) const { return id < n || (!leaf_prim && size[id] <= leaf_max); }
    __device__ bool is_leaf(uint32_t id) const
    {
        if (id < n) return true;
        if (sah) return sah[id - n].pick[0] == 0;
        return !leaf_prim && size[id] <= leaf_max;
    }
    __device__ int leaf_code(uint32_t id) const
    {
        return leaf_prim ? ~(int)leaf_prim[id] : ~(int)((offset[id] << 3) | (size[id] - 1u));
    }
};

// what the levels of one collapse share in device memory
struct WideLevel { uint32_t count, base; };                 // frontier size, index of its first wide node
struct WideState {
    uint32_t arrivals, error;
    uint32_t done, pad;                                     // levels below `done` were built by k_wide_top
    WideLevel level[RT_WIDE_MAX_LEVELS + 2];
};

// The roots the subtree of `id` contributes when it may use at most `budget` slots, appended to open[] (SAH collapse)
__device__ __forceinline__ void sah_roots(const TreeView &t, uint32_t first_id, uint32_t first_budget, uint32_t open[RT_WIDE], int &nk)
{
    uint32_t st_id[RT_WIDE], st_b[RT_WIDE];
    int sp = 0;
    st_id[0] = first_id; st_b[0] = first_budget; sp = 1;
    while (sp > 0) {
        sp--;
        const uint32_t id = st_id[sp];
        uint32_t i = st_b[sp];
        if (id < t.n) { open[nk++] = id; continue; }
        const SahRecord &r = t.sah[id - t.n];
        while (i > 1 && r.pick[i - 1] == 0) i--;
        if (i == 1) { open[nk++] = id; continue; }
        const uint32_t k = r.pick[i - 1];
        st_id[sp] = t.right[id - t.n]; st_b[sp] = i - k; sp++;       // (left is expanded first: deterministic order)
        st_id[sp] = t.left[id - t.n]; st_b[sp] = k; sp++;
    }
}

// the (up to RT_WIDE) children of frontier node b in their SLOTS (NO_KID: empty slot); returns how many of them are wide
// nodes themselves.
// (Fetching a child's own children's ids together with its size and box -- one trip less per opened child -- was measured:
// the large levels pay more for the extra loads than the small ones gain.)
__device__ __forceinline__ uint32_t expand_node(const TreeView &t, uint32_t b, uint32_t kid[RT_WIDE])
{
    uint32_t open[RT_WIDE];
    int nk = 0;
    if (t.sah) {
        const uint32_t k = t.sah[b - t.n].spread;
        sah_roots(t, t.left[b - t.n], k, open, nk);
        sah_roots(t, t.right[b - t.n], RT_WIDE - k, open, nk);
    } else {
        open[0] = t.left[b - t.n]; open[1] = t.right[b - t.n];
        nk = 2;
        while (nk < RT_WIDE) {
            int best = -1;
            float best_area = -1.0f;
            for (int k = 0; k < nk; k++) {
                if (t.is_leaf(open[k])) continue;
                const float a = box_area(t.box[open[k]]);
                if (a > best_area || best < 0) { best = k; best_area = a; }
            }
            if (best < 0) break;
            const uint32_t id = open[best];
            open[best] = t.left[id - t.n];
            open[nk++] = t.right[id - t.n];
        }
    }
    // (Filling the slots that are still free with the halves of multi-triangle leaves was measured on the four-wide layout:
    // 3 % fewer triangle tests, but more leaf visits, frame 2.67 -> 2.83 ms.)
    for (int k = 0; k < RT_WIDE; k++) kid[k] = NO_KID;
#if RT_WIDE == 8
    // slot assignment: greedy over (child, slot) pairs, dearest first; ties and NaN centres fall to the first free pair
    const Box6 nb = t.box[b];
    float cx[RT_WIDE], cy[RT_WIDE], cz[RT_WIDE];
    for (int k = 0; k < nk; k++) {
        const Box6 cb = t.box[open[k]];
        cx[k] = (cb.lo[0] + cb.hi[0]) - (nb.lo[0] + nb.hi[0]);
        cy[k] = (cb.lo[1] + cb.hi[1]) - (nb.lo[1] + nb.hi[1]);
        cz[k] = (cb.lo[2] + cb.hi[2]) - (nb.lo[2] + nb.hi[2]);
    }
    uint32_t free_kids = (1u << nk) - 1u, free_slots = (1u << RT_WIDE) - 1u;
    for (int round = 0; round < nk; round++) {
        int bc = -1, bs = -1;
        float best = 0.0f;
        for (int c = 0; c < nk; c++) {
            if (!((free_kids >> c) & 1u)) continue;
            for (int sl = 0; sl < RT_WIDE; sl++) {
                if (!((free_slots >> sl) & 1u)) continue;
                const float cost = ((sl & 1) ? cx[c] : -cx[c]) + ((sl & 2) ? cy[c] : -cy[c]) + ((sl & 4) ? cz[c] : -cz[c]);
                if (bc < 0 || cost > best) { best = cost; bc = c; bs = sl; }       // (a NaN cost never beats anything)
            }
        }
        kid[bs] = open[bc];
        free_kids &= ~(1u << bc);
        free_slots &= ~(1u << bs);
    }
#else
    // packed at the front, larger surface first (insertion sort, stable): any-hit rays walk in slot order
    float ar[RT_WIDE];
    for (int k = 0; k < nk; k++) { ar[k] = box_area(t.box[open[k]]); kid[k] = open[k]; }
    for (int i = 1; i < nk; i++)
        for (int j = i; j > 0 && ar[j] > ar[j - 1]; j--) {
            const float ta = ar[j]; ar[j] = ar[j - 1]; ar[j - 1] = ta;
            const uint32_t tk = kid[j]; kid[j] = kid[j - 1]; kid[j - 1] = tk;
        }
#endif
    uint32_t cnt = 0;
    for (int k = 0; k < RT_WIDE; k++)
        if (kid[k] != NO_KID && !t.is_leaf(kid[k])) cnt++;
    return cnt;
}

// The bottom-up pass of the SAH collapse: a thread starts at every leaf and climbs; at a parent the first arriver stops,
// the second fills the parent's record from its children's (the scheme of k_refit, rt_bvh_build.hip: records are published
// with agent-scope stores that are waited for before the arrival is counted, and read with agent-scope loads -- the per-XCD
// L2s are not coherent).  leaf_cost = cost of testing one primitive (a TLAS: of entering one instance).
__device__ __forceinline__ float leaf_sah(const TreeView &t, uint32_t id, float c_prim)
{
    return box_area(t.box[id]) * c_prim * (float)(t.leaf_prim ? 1u : t.size[id]);
}

__global__ void __launch_bounds__(WB) k_wide_sah(TreeView t, const uint32_t *__restrict__ parent, SahRecord *__restrict__ rec, uint32_t *__restrict__ arrive,
                                                 float c_node, float c_prim)
{
    const uint32_t leaf = blockIdx.x * WB + threadIdx.x;
    if (leaf >= t.n) return;
    uint32_t cur = leaf;
    const float inf = __builtin_inff();
    for (;;) {
        const uint32_t p = parent[cur];
        if (p == 0xFFFFFFFFu) return;                       // cur is the root: its record is complete
        // (a relaxed memory-side atomic: what it orders is done by hand -- the stores above have been acknowledged (s_waitcnt), the
        // loads below bypass the local L2 -- and the signal fences keep the compiler from moving either across it)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const uint32_t arrived = __hip_atomic_fetch_add(&arrive[p - t.n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (arrived == 0u) return;                          // the sibling's climber takes over
        // both children are complete: cost tables of the two (a single primitive costs its leaf whatever the budget)
        float cl[RT_WIDE], cr[RT_WIDE];
        const uint32_t kids[2] = {t.left[p - t.n], t.right[p - t.n]};
        for (int side = 0; side < 2; side++) {
            float *c = side ? cr : cl;
            const uint32_t id = kids[side];
            if (id < t.n) {
                const float v = leaf_sah(t, id, c_prim);
                for (int i = 0; i < RT_WIDE; i++) c[i] = v;
            } else {
                const uint64_t *w = reinterpret_cast<const uint64_t *>(rec[id - t.n].cost);
                for (int i = 0; i < RT_WIDE / 2; i++) {
                    const uint64_t v = __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    c[2 * i] = __uint_as_float((uint32_t)v); c[2 * i + 1] = __uint_as_float((uint32_t)(v >> 32));
                }
            }
        }
        // spread(j) = min over k of cl[k] + cr[j - k]: the subtree as j roots, k from the left and j - k from the right
        float cost[8];
        uint32_t pick[8];
        for (int i = 0; i < 8; i++) { cost[i] = inf; pick[i] = 0; }
        uint32_t spread_all = 1;
        for (int j = 2; j <= RT_WIDE; j++) {
            float best = inf;
            uint32_t bk = 1;
            for (int k = 1; k < j; k++) {
                const float v = cl[k - 1] + cr[j - k - 1];
                if (v < best || k == 1) { best = v; bk = (uint32_t)k; }
            }
            cost[j - 1] = best; pick[j - 1] = bk;            // (made monotone below)
            if (j == RT_WIDE) spread_all = bk;
        }
        const float as_node = box_area(t.box[p]) * c_node + cost[RT_WIDE - 1];
        const float as_leaf = t.may_be_leaf(p) ? leaf_sah(t, p, c_prim) : inf;
        // (a NaN cost -- NaN / inf geometry -- compares false everywhere: such a subtree becomes wide nodes down to its primitives)
        const bool leaf = t.may_be_leaf(p) && as_leaf <= as_node;
        cost[0] = leaf ? as_leaf : as_node;
        pick[0] = leaf ? 0u : 1u;
        for (int i = 2; i <= RT_WIDE; i++)
            if (!(cost[i - 1] < cost[i - 2])) { cost[i - 1] = cost[i - 2]; pick[i - 1] = 0u; }
        // publish, wait for the acknowledgement, go on to the parent
        SahRecord *mine = rec + (p - t.n);
        uint64_t *w = reinterpret_cast<uint64_t *>(mine->cost);
        for (int i = 0; i < 4; i++)
            __hip_atomic_store(&w[i], (uint64_t)__float_as_uint(cost[2 * i]) | ((uint64_t)__float_as_uint(cost[2 * i + 1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t pk = 0;
        for (int i = 0; i < 8; i++) pk |= (uint64_t)(pick[i] & 0xffu) << (8 * i);
        __hip_atomic_store(&w[4], pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&w[5], (uint64_t)spread_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur = p;
    }
}

// one frontier element -> its children; the workgroups' tallies of children that are wide nodes themselves become the
// offsets of the next frontier, and the last workgroup writes the next level's size
__global__ void __launch_bounds__(WB) k_wide_expand(TreeView t, const uint32_t *__restrict__ frontier, WideState *__restrict__ ws, uint32_t lvl,
                                                    uint32_t *__restrict__ kids, uint32_t *__restrict__ tally, uint32_t fcap)
{
    __shared__ uint32_t lds[WB];
    __shared__ uint32_t lds_flag;
    if (lvl < ws->done) return;                         // k_wide_top has been here
    const WideLevel L = ws->level[lvl];
    if (L.count == 0) {                                 // past the last level of this tree: hand the end on
        if (blockIdx.x == 0 && threadIdx.x == 0) ws->level[lvl + 1] = L;
        return;
    }
    const uint32_t nblocks = (L.count + WB - 1) / WB;
    if (blockIdx.x >= nblocks) return;
    const uint32_t f = blockIdx.x * WB + threadIdx.x;
    uint32_t cnt = 0;
    if (f < L.count) {
        uint32_t kid[RT_WIDE];
        cnt = expand_node(t, frontier[f], kid);
        for (int k = 0; k < RT_WIDE; k++) kids[RT_WIDE * (size_t)f + k] = kid[k];
    }
    uint32_t block_total;
    (void)rt_scan::block_exclusive<WB>(cnt, lds, block_total);
    if (!rt_scan::publish_and_arrive(tally, block_total, &ws->arrivals, nblocks, &lds_flag)) return;
    const uint32_t total = rt_scan::scan_tallies<uint32_t, WB>(tally, nblocks, lds);
    if (threadIdx.x == 0) {
        WideLevel nx = {total, L.base + L.count};
        if (lvl + 1 > RT_WIDE_MAX_LEVELS || total > fcap || (size_t)nx.base + total > (size_t)t.n - 1) {
            ws->error = 1 + lvl;                        // the host reports it
            nx.count = 0;
        }
        ws->level[lvl + 1] = nx;
        ws->arrivals = 0;
    }
}

// quantises the planes of one axis of the boxes in the slots of `valid` onto origin + q * scale; returns false if 255
// steps do not reach.  lo8 / hi8: one byte per slot, slot k in bits 8k..8k+7.
__device__ bool quantise_axis(const Box6 *cb, uint32_t valid, int axis, float origin, float scale, unsigned long long &lo8, unsigned long long &hi8)
{
    lo8 = 0; hi8 = 0;
    bool ok = true;
    for (int k = 0; k < RT_WIDE; k++) {
        int ql = 255, qh = 0;                           // unused slot: an inverted interval (never tested: not in the valid mask)
        if ((valid >> k) & 1u) {
            const float l = cb[k].lo[axis], h = cb[k].hi[axis];
            float fl = __builtin_floorf((l - origin) / scale), fh = __builtin_ceilf((h - origin) / scale);
            fl = fl > 0.0f ? (fl < 255.0f ? fl : 255.0f) : 0.0f;            // (NaN -> 0)
            fh = fh > 0.0f ? (fh < 255.0f ? fh : 255.0f) : 0.0f;
            ql = (int)fl; qh = (int)fh;
            while (ql > 0 && __builtin_fmaf((float)ql, scale, origin) > l) ql--;
            while (qh < 255 && __builtin_fmaf((float)qh, scale, origin) < h) qh++;
            if (__builtin_fmaf((float)ql, scale, origin) > l || __builtin_fmaf((float)qh, scale, origin) < h) ok = false;
        }
        lo8 |= (unsigned long long)ql << (8 * k);
        hi8 |= (unsigned long long)qh << (8 * k);
    }
    return ok;
}

// writes wide node `node_index` for frontier node b with the children kid[] in their slots (internal ones: bits of
// `internal`), whose own wide nodes start at index `next_level_base + next` of the array (position `next` of the next frontier)
__device__ __forceinline__ void emit_node(const TreeView &t, uint32_t b, const uint32_t kid[RT_WIDE], uint32_t internal, uint32_t node_index,
                                          uint32_t next_level_base, uint32_t next, uint32_t *__restrict__ next_frontier, WNode *__restrict__ out)
{
    const Box6 nb = t.box[b];
    Box6 cb[RT_WIDE];
    int code[RT_WIDE];
    uint32_t valid = 0;
    const uint32_t child_base = next_level_base + next;
    for (int k = 0; k < RT_WIDE; k++) {
        const uint32_t id = kid[k];
        code[k] = RT_NODE_NONE;
        if (id == NO_KID) continue;
        valid |= 1u << k;
        cb[k] = t.box[id];
        if (!((internal >> k) & 1u)) code[k] = t.leaf_code(id);
        else {
            next_frontier[next] = id;
            code[k] = (int)(next_level_base + next);    // the next level starts right behind this one
            next++;
        }
    }
    uint32_t expo[3];
    float scale[3];
    unsigned long long lo8[3], hi8[3];
    for (int a = 0; a < 3; a++) {
        const float ext = nb.hi[a] - nb.lo[a];
        // smallest power of two s with 255 * s >= ext (a zero or denormal extent gets the smallest normal number)
        int e = 0;
        const float m = __builtin_frexpf(ext / 255.0f, &e);      // ext / 255 = m * 2^e, m in [0.5, 1)
        float s = (ext > 0.0f && ext < 3.0e38f) ? __builtin_ldexpf(1.0f, m == 0.5f ? e - 1 : e) : 1.17549435e-38f;
        if (!(s >= 1.17549435e-38f)) s = 1.17549435e-38f;
        bool ok = false;
        for (int tries = 0; tries < 8 && !ok; tries++) {
            ok = s < 1.0e38f && quantise_axis(cb, valid, a, nb.lo[a], s, lo8[a], hi8[a]);
            if (!ok) s = s * 2.0f;                      // rounding in fma left the last step short: coarser grid
        }
        expo[a] = (__float_as_uint(s) >> 23) & 0xffu;
        scale[a] = s;
        if (!ok) {
            // not quantisable (non-finite or overflowing extents): an infinite scale (exponent 255) with q = 0 planes --
            // fma(0, inf, origin) is NaN, which the slab tests ignore: this axis never culls
            expo[a] = 255u; scale[a] = __builtin_inff(); lo8[a] = 0; hi8[a] = 0;
        }
    }
    WNode w;
#if RT_WIDE == 8
    w.q0 = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], __uint_as_float(expo[0] | (expo[1] << 8) | (expo[2] << 16) | (valid << 24)));
    w.q1 = make_float4(__uint_as_float((uint32_t)lo8[0]), __uint_as_float((uint32_t)(lo8[0] >> 32)), __uint_as_float((uint32_t)hi8[0]), __uint_as_float((uint32_t)(hi8[0] >> 32)));
    w.q2 = make_float4(__uint_as_float((uint32_t)lo8[1]), __uint_as_float((uint32_t)(lo8[1] >> 32)), __uint_as_float((uint32_t)hi8[1]), __uint_as_float((uint32_t)(hi8[1] >> 32)));
    w.q3 = make_float4(__uint_as_float((uint32_t)lo8[2]), __uint_as_float((uint32_t)(lo8[2] >> 32)), __uint_as_float((uint32_t)hi8[2]), __uint_as_float((uint32_t)(hi8[2] >> 32)));
    w.q4 = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]), __int_as_float(code[3]));
    w.q5 = make_float4(__int_as_float(code[4]), __int_as_float(code[5]), __int_as_float(code[6]), __int_as_float(code[7]));
    // not read by the traversal: the first internal child (they are consecutive in slot order), the internal-slot mask, the binary node this came from
    w.q6 = make_float4(__uint_as_float(internal ? child_base : 0u), __uint_as_float(internal), __uint_as_float(b), 0.0f);
    w.q7 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    (void)scale;
#else
    (void)expo; (void)child_base;
    w.q0 = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], scale[0]);
    w.q1 = make_float4(__uint_as_float((uint32_t)lo8[0]), __uint_as_float((uint32_t)hi8[0]), __uint_as_float((uint32_t)lo8[1]), __uint_as_float((uint32_t)hi8[1]));
    w.q2 = make_float4(__uint_as_float((uint32_t)lo8[2]), __uint_as_float((uint32_t)hi8[2]), scale[1], scale[2]);
    w.q3 = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]), __int_as_float(code[3]));
#endif
    out[node_index] = w;
}


__global__ void __launch_bounds__(WB) k_wide_emit(TreeView t, const uint32_t *__restrict__ frontier, const WideState *__restrict__ ws, uint32_t lvl,
                                                  const uint32_t *__restrict__ kids, const uint32_t *__restrict__ tally,
                                                  uint32_t *__restrict__ next_frontier, WNode *__restrict__ out)
{
    __shared__ uint32_t lds[WB / 64];
    if (lvl < ws->done) return;
    const WideLevel L = ws->level[lvl];
    const uint32_t count = L.count, base = L.base;
    if (blockIdx.x * WB >= count) return;
    const uint32_t f = blockIdx.x * WB + threadIdx.x;
    uint32_t kid[RT_WIDE], internal = 0, mine = 0;
    for (int k = 0; k < RT_WIDE; k++) kid[k] = NO_KID;
    if (f < count)
        for (int k = 0; k < RT_WIDE; k++) {
            kid[k] = kids[RT_WIDE * (size_t)f + k];
            if (kid[k] != NO_KID && !t.is_leaf(kid[k])) { internal |= 1u << k; mine++; }
        }
    uint32_t block_total;
    const uint32_t next = tally[blockIdx.x] + rt_scan::block_exclusive<WB>(mine, lds, block_total);
    if (f >= count) return;
    emit_node(t, frontier[f], kid, internal, base + f, base + count, next, next_frontier, out);
}

// The top of the tree in ONE workgroup: while a level has at most TOPW nodes, expand, number and emit it between two
// barriers -- a level is ~6 dependent loads whatever its size, and as two launches each it cost 27 us.  Levels 0 .. 4
// (8^2 = 64 nodes) are certain to be handled here; the per-level launches start behind them and skip whatever else this
// kernel got to (ws->done).  (256 threads: the emit code wants 156 VGPRs, a 1024-thread workgroup would spill.)
constexpr uint32_t TOPW = 256, TOP_SURE = 3;            // levels 0 .. TOP_SURE-1 have at most 8^l <= TOPW nodes
__global__ void __launch_bounds__(TOPW) k_wide_top(TreeView t, uint32_t *__restrict__ frontier0, uint32_t *__restrict__ frontier1,
                                                    WideState *__restrict__ ws, uint32_t root, uint32_t fcap, WNode *__restrict__ out)
{
    __shared__ uint32_t lds[TOPW / 64];
    uint32_t *frontier[2] = {frontier0, frontier1};
    uint32_t count = 1, base = 0, lvl = 0, error = 0;
    if (threadIdx.x == 0) frontier0[0] = root;
    __syncthreads();
    while (count > 0 && count <= TOPW) {
        const uint32_t f = threadIdx.x;
        uint32_t kid[RT_WIDE], internal = 0, mine = 0, b = 0;
        for (int k = 0; k < RT_WIDE; k++) kid[k] = NO_KID;
        if (f < count) {
            b = frontier[lvl & 1u][f];
            mine = expand_node(t, b, kid);
            for (int k = 0; k < RT_WIDE; k++)
                if (kid[k] != NO_KID && !t.is_leaf(kid[k])) internal |= 1u << k;
        }
        uint32_t total;
        const uint32_t next = rt_scan::block_exclusive<TOPW>(mine, lds, total);
        if (f < count) emit_node(t, b, kid, internal, base + f, base + count, next, frontier[(lvl & 1u) ^ 1u], out);
        if (threadIdx.x == 0) ws->level[lvl].count = count, ws->level[lvl].base = base;
        base += count;
        count = total;
        lvl++;
        if (lvl > RT_WIDE_MAX_LEVELS || count > fcap || (size_t)base + count > (size_t)t.n - 1) { error = lvl; count = 0; }
        __syncthreads();                                // the next frontier is in global memory: written above, read below
    }
    if (threadIdx.x == 0) {
        // (a tree that ends before level TOP_SURE -- where the per-level launches start -- has its end handed on to there)
        for (uint32_t l = lvl; l <= (count == 0 && lvl < TOP_SURE ? TOP_SURE : lvl); l++) { ws->level[l].count = count; ws->level[l].base = base; }
        ws->done = lvl;
        ws->arrivals = 0;
        ws->error = error;
    }
}

// canonical LBVH (rt_bvh_node[2n-1] + leaf ranges) -> cluster numbering: leaf k -> id k, internal c -> id n + c
__global__ void __launch_bounds__(WB) k_lbvh_to_tree(const rt_bvh_node *__restrict__ nodes, const uint2 *__restrict__ ranges, uint32_t n,
                                                     uint32_t *__restrict__ left, uint32_t *__restrict__ right, Box6 *__restrict__ box,
                                                     uint32_t *__restrict__ size, uint32_t *__restrict__ offset, uint32_t *__restrict__ leaf_prim,
                                                     uint32_t *__restrict__ parent)
{
    const uint32_t c = blockIdx.x * WB + threadIdx.x;
    if (c >= 2 * n - 1) return;
    const rt_bvh_node nd = nodes[c];
    const bool leaf = c >= n - 1;
    const uint32_t id = leaf ? c - (n - 1) : n + c;
    Box6 b;
    for (int k = 0; k < 3; k++) { b.lo[k] = nd.bmin[k]; b.hi[k] = nd.bmax[k]; }
    box[id] = b;
    if (leaf) {
        size[id] = 1;
        offset[id] = id;
        leaf_prim[id] = nd.left;
    } else {
        const uint2 r = ranges[c];
        size[id] = r.y - r.x + 1;
        offset[id] = r.x;
        const uint32_t l = nd.left >= n - 1 ? nd.left - (n - 1) : n + nd.left;
        const uint32_t r2 = nd.right >= n - 1 ? nd.right - (n - 1) : n + nd.right;
        left[c] = l;
        right[c] = r2;
        parent[l] = id;
        parent[r2] = id;
        if (c == 0) parent[id] = 0xFFFFFFFFu;
    }
}

inline unsigned gr(size_t n) { return (unsigned)((n + WB - 1) / WB); }
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// two frontiers, the children of a frontier, one tally per workgroup of a level, the level table
size_t rt_wide_temp_bytes(uint32_t n)
{
    const size_t f = (size_t)n / 2 + 2;
    return 2 * up256(4 * f) + up256(4 * RT_WIDE * f) + up256(4 * (size_t)gr(f)) + up256(sizeof(WideState)) +
           up256(sizeof(SahRecord) * (size_t)n) + up256(4 * (size_t)n);
}

int rt_build_wide_layout(rt_context *ctx, BvhDev &bv, uint32_t n, uint32_t root, const uint32_t *left, const uint32_t *right, const uint32_t *parent,
                         const float *box6, const uint32_t *size, const uint32_t *offset, const uint32_t *leaf_prim, uint32_t leaf_max,
                         void *tmp, size_t tmp_bytes)
{
    hipStream_t st = ctx->stream;
    TreeView t;
    t.left = left; t.right = right; t.box = (const Box6 *)box6; t.size = size; t.offset = offset; t.leaf_prim = leaf_prim;
    t.sah = nullptr;
    t.n = n; t.leaf_max = leaf_max;
    bv.wide_n = 0;
    // the whole structure is one leaf (one primitive, or a BLAS of <= leaf_max triangles)
    if (!leaf_prim && n <= leaf_max) {
        bv.root_code = ~(int)(n - 1);                  // leaf(first 0, count n)
        bv.fast_depth = 0;
        RT_TRY(bv.wide.reserve(sizeof(WNode)));
        return RT_OK;
    }
    const size_t fcap = (size_t)n / 2 + 2;
    const size_t need = rt_wide_temp_bytes(n);
    DevBuf own;                        // only if the caller's slice is too small
    char *p = (char *)tmp;
    if (need > tmp_bytes) { RT_TRY(own.reserve(need)); p = (char *)own.p; }
    uint32_t *frontier[2] = {(uint32_t *)p, (uint32_t *)(p + up256(4 * fcap))};
    p += 2 * up256(4 * fcap);
    uint32_t *kids = (uint32_t *)p; p += up256(4 * RT_WIDE * fcap);
    uint32_t *tally = (uint32_t *)p; p += up256(4 * (size_t)gr(fcap));
    WideState *ws = (WideState *)p; p += up256(sizeof(WideState));
    SahRecord *sah = (SahRecord *)p; p += up256(sizeof(SahRecord) * (size_t)n);
    uint32_t *arrive = (uint32_t *)p;
    int rc = RT_OK;
    do {
        if ((rc = bv.wide.reserve(sizeof(WNode) * (size_t)(n - 1))) != RT_OK) break;
        if (ctx->wide_sah && parent) {
            // which binary nodes become wide nodes, which leaves, and how a wide node spends its slots: one bottom-up pass
            if (hipMemsetAsync(arrive, 0, 4 * (size_t)(n - 1), st) != hipSuccess) { rt_set_error("wide layout: memset failed"); rc = RT_ERR_HIP; break; }
            k_wide_sah<<<gr(n), WB, 0, st>>>(t, parent, sah, arrive, ctx->sah_node, ctx->sah_prim);
            t.sah = sah;
        }
        k_wide_top<<<1, TOPW, 0, st>>>(t, frontier[0], frontier[1], ws, root, (uint32_t)fcap, bv.wide.as<WNode>());
        // Levels are launched in batches without looking at their sizes: a level has at most eight times the nodes of the
        // one before (and never more than fcap), a level past the end of the tree costs two empty launches.  The first
        // batch is sized for a tree half again as deep as a balanced one; the host reads the level table after each batch.
        // k_wide_top has certainly built levels 0 .. TOP_SURE-1 (a level has at most 8^l nodes, it takes every level of <= TOPW)
        uint32_t lvl = TOP_SURE, levels = 0, wide_n = 0;
        uint64_t bound = 2 * TOPW;                            // upper bound of the frontier at level lvl: 8^TOP_SURE
        uint32_t log8 = 0;
        for (uint32_t m = n; m > 1; m >>= 3) log8++;
        uint32_t batch = 4 + log8 + log8 / 2;
        batch = batch > TOP_SURE + 2 ? batch - TOP_SURE : 2; // (the first levels are k_wide_top's)
        if (ctx->build_batch) batch = ctx->build_batch;      // (tests: force short batches)
        WideState host_state;
        bool done = false;
        while (!done) {
            const uint32_t end = lvl + batch < RT_WIDE_MAX_LEVELS + 1 ? lvl + batch : RT_WIDE_MAX_LEVELS + 1;
            for (; lvl < end; lvl++) {
                const unsigned blocks = gr((size_t)(bound < fcap ? bound : fcap));
                k_wide_expand<<<blocks, WB, 0, st>>>(t, frontier[lvl & 1], ws, lvl, kids, tally, (uint32_t)fcap);
                k_wide_emit<<<blocks, WB, 0, st>>>(t, frontier[lvl & 1], ws, lvl, kids, tally, frontier[(lvl & 1) ^ 1], bv.wide.as<WNode>());
                bound = bound < fcap ? bound * RT_WIDE : fcap;
            }
            const size_t bytes = offsetof(WideState, level) + sizeof(WideLevel) * (lvl + 1);
            WideState *back = (ctx->pinned && bytes <= 64 * sizeof(uint32_t)) ? (WideState *)ctx->pinned : &host_state;      // page-locked: no staging copy
            if (hipMemcpyAsync(back, ws, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
                hipGetLastError() != hipSuccess) {
                rt_set_error("wide layout failed: %s", hipGetErrorString(hipGetLastError()));
                rc = RT_ERR_HIP;
                break;
            }
            const WideLevel *table = back->level;
            if (back->error) { rt_set_error("wide layout: the frontier of level %u does not fit (%u primitives)", back->error, n); rc = RT_ERR_STATE; break; }
            for (uint32_t l = 0; l <= lvl; l++)
                if (table[l].count == 0) { levels = l; wide_n = table[l].base; done = true; break; }
            if (!done && lvl >= RT_WIDE_MAX_LEVELS + 1) { rt_set_error("wide layout: deeper than %u levels", RT_WIDE_MAX_LEVELS); rc = RT_ERR_STATE; break; }
            bound = table[lvl].count;
            batch = ctx->build_batch ? ctx->build_batch : 8;
        }
        if (rc != RT_OK) break;
        bv.wide_n = wide_n;
        bv.root_code = 0;
        bv.fast_depth = (RT_WIDE - 1) * levels;        // a step leaves at most seven siblings behind
    } while (0);
    own.release();
    return rc;
}

// arena bytes rt_build_wide_from_lbvh takes: the tree in cluster numbering + the scratch of the collapse
size_t rt_wide_lbvh_temp_bytes(uint32_t n)
{
    const size_t nn2 = 2 * (size_t)n;
    return 2 * up256(4 * (size_t)n) + up256(sizeof(Box6) * nn2) + 4 * up256(4 * nn2) + rt_wide_temp_bytes(n);
}

int rt_build_wide_from_lbvh(rt_context *ctx, BvhDev &bv, bool tlas, uint32_t leaf_max)
{
    const uint32_t n = bv.n;
    hipStream_t st = ctx->stream;
    if (n == 1) {                                      // one primitive: the root is its leaf (TLAS: instance 0; BLAS: triangle 0)
        bv.wide_n = 0;
        bv.root_code = ~(int)0;
        bv.fast_depth = 0;
        RT_TRY(bv.wide.reserve(sizeof(WNode)));
        return RT_OK;
    }
    const size_t nn2 = 2 * (size_t)n - 1;
    const size_t tree = 2 * up256(4 * (size_t)(n - 1)) + up256(sizeof(Box6) * nn2) + 4 * up256(4 * nn2);
    const size_t wide = rt_wide_temp_bytes(n);
    RT_TRY(ctx->build_arena.reserve(tree + wide));
    char *p = (char *)ctx->build_arena.p;
    uint32_t *left = (uint32_t *)p; p += up256(4 * (size_t)(n - 1));
    uint32_t *right = (uint32_t *)p; p += up256(4 * (size_t)(n - 1));
    Box6 *box = (Box6 *)p; p += up256(sizeof(Box6) * nn2);
    uint32_t *size = (uint32_t *)p; p += up256(4 * nn2);
    uint32_t *offset = (uint32_t *)p; p += up256(4 * nn2);
    uint32_t *leaf_prim = (uint32_t *)p; p += up256(4 * nn2);
    uint32_t *parent = (uint32_t *)p; p += up256(4 * nn2);
    k_lbvh_to_tree<<<gr(nn2), WB, 0, st>>>(bv.nodes.as<rt_bvh_node>(), bv.ranges.as<uint2>(), n, left, right, box, size, offset, leaf_prim, parent);
    HIP_TRY(hipGetLastError());
    return rt_build_wide_layout(ctx, bv, n, n /* canonical root 0 */, left, right, parent, (const float *)box, size, offset, tlas ? leaf_prim : nullptr,
                                tlas ? 1u : leaf_max, p, wide);
}
