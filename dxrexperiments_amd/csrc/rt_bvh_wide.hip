// rt_bvh_wide.hip -- the four-wide, quantised traversal layout (WNode, rt_internal.h).
//
// Input: a binary tree over the sorted leaves (the PLOC tree of rt_bvh_ploc.hip, or the canonical LBVH for a TLAS,
// tiny meshes and RT_FAST_BVH=lbvh).  Output: nodes with up to four children in ONE 64-B line each.  Why: the traversal
// stages run at the chip's rate of distinct 64-B lines per second (profiles/r02/slab_fetch.txt: reading 16, 32 or 64 B of
// a line costs the same, a second line costs as much again), so the lever is lines per ray.  A binary node spends its
// line on two children; round 1's four-wide node with full-precision boxes took two lines and gained nothing.  Here four
// child boxes are quantised to one byte per plane on a power-of-two grid anchored at the node's own box, which fits
// origin, scales, 24 plane bytes and four child codes in 64 B and halves the lines per ray.
//
// Collapse: breadth first, two launches per level and no host round trip inside a batch of levels (rt_level_scan.h).  A frontier element is a binary node that becomes a wide node: its two
// children are taken, and while fewer than four, the child with the largest surface that is not a leaf is replaced by
// its own two children (surface-area greedy, as in Wald et al. 2008 / Ylitie et al. 2017).  Node numbers come from
// prefix sums (deterministic), level after level, so the array is in breadth-first order and its first RT_TOP_NODES
// entries are the LDS-resident top of the traversal kernels.
//
// Exactness: plane = fma(q, scale, origin) is evaluated here with the same expression the traversal uses; every lo
// plane is stepped down and every hi plane up until the decoded box CONTAINS the child's true box.  The slab test is
// monotone under box inclusion (DESIGN.md "Exactness rule"), so culling against the decoded boxes never loses a
// candidate the canonical definition accepts.
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

struct TreeView {
    const uint32_t *left, *right;      // indexed by id - n
    const Box6 *box;                   // indexed by id
    const uint32_t *size, *offset;     // indexed by id
    const uint32_t *leaf_prim;         // TLAS: instance of leaf id; nullptr for a BLAS
    uint32_t n, leaf_max;
    __device__ bool is_leaf(uint32_t id) const { return id < n || (!leaf_prim && size[id] <= leaf_max); }
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

// the (up to four) children of frontier node b, largest surface first; returns how many of them are wide nodes themselves.
// (Fetching a child's own children's ids together with its size and box -- one trip less per opened child -- was measured:
// the large levels pay more for the extra loads than the small ones gain.)
__device__ __forceinline__ uint32_t expand_node(const TreeView &t, uint32_t b, uint32_t kid[4])
{
    kid[0] = t.left[b - t.n]; kid[1] = t.right[b - t.n]; kid[2] = NO_KID; kid[3] = NO_KID;
    int nk = 2;
    while (nk < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int k = 0; k < nk; k++) {
            if (t.is_leaf(kid[k])) continue;
            const float a = box_area(t.box[kid[k]]);
            if (a > best_area || best < 0) { best = k; best_area = a; }
        }
        if (best < 0) break;
        const uint32_t id = kid[best];
        kid[best] = t.left[id - t.n];
        kid[nk++] = t.right[id - t.n];
    }
    // (Filling the slots that are still free with the halves of multi-triangle leaves -- 3.0 -> 3.9 children per node, the
    // step tests four boxes either way -- was measured: 3 % fewer triangle tests, but more leaf visits, frame 2.67 -> 2.83 ms.)
    float ar[4];
    for (int k = 0; k < nk; k++) ar[k] = box_area(t.box[kid[k]]);
    for (int i = 1; i < nk; i++)                       // insertion sort, larger surface first, stable
        for (int j = i; j > 0 && ar[j] > ar[j - 1]; j--) {
            const float ta = ar[j]; ar[j] = ar[j - 1]; ar[j - 1] = ta;
            const uint32_t tk = kid[j]; kid[j] = kid[j - 1]; kid[j - 1] = tk;
        }
    uint32_t cnt = 0;
    for (int k = 0; k < nk; k++)
        if (!t.is_leaf(kid[k])) cnt++;
    return cnt;
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
        uint32_t kid[4];
        cnt = expand_node(t, frontier[f], kid);
        for (int k = 0; k < 4; k++) kids[4 * (size_t)f + k] = kid[k];
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

// quantises the planes of one axis of up to four boxes onto origin + q * scale; returns false if 255 steps do not reach
__device__ bool quantise_axis(const Box6 *cb, int nk, int axis, float origin, float scale, uint32_t &lo4, uint32_t &hi4)
{
    lo4 = 0; hi4 = 0;
    bool ok = true;
    for (int k = 0; k < 4; k++) {
        int ql = 255, qh = 0;                           // unused slot: an inverted interval (never tested: its code is RT_NODE_NONE)
        if (k < nk) {
            const float l = cb[k].lo[axis], h = cb[k].hi[axis];
            float fl = __builtin_floorf((l - origin) / scale), fh = __builtin_ceilf((h - origin) / scale);
            fl = fl > 0.0f ? (fl < 255.0f ? fl : 255.0f) : 0.0f;            // (NaN -> 0)
            fh = fh > 0.0f ? (fh < 255.0f ? fh : 255.0f) : 0.0f;
            ql = (int)fl; qh = (int)fh;
            while (ql > 0 && __builtin_fmaf((float)ql, scale, origin) > l) ql--;
            while (qh < 255 && __builtin_fmaf((float)qh, scale, origin) < h) qh++;
            if (__builtin_fmaf((float)ql, scale, origin) > l || __builtin_fmaf((float)qh, scale, origin) < h) ok = false;
        }
        lo4 |= (uint32_t)ql << (8 * k);
        hi4 |= (uint32_t)qh << (8 * k);
    }
    return ok;
}

// writes wide node `base + f` for frontier node b with children kid[] (internal ones: bits of `internal`), whose own wide
// nodes start at index `first_child` of the array and at position `next` of the next frontier
__device__ __forceinline__ void emit_node(const TreeView &t, uint32_t b, const uint32_t kid[4], uint32_t internal, uint32_t node_index,
                                          uint32_t next_level_base, uint32_t next, uint32_t *__restrict__ next_frontier, WNode *__restrict__ out)
{
    const Box6 nb = t.box[b];
    Box6 cb[4];
    int code[4];
    int nk = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t id = kid[k];
        code[k] = RT_NODE_NONE;
        if (id == NO_KID) continue;
        nk = k + 1;                                     // (kids are packed at the front)
        cb[k] = t.box[id];
        if (!((internal >> k) & 1u)) code[k] = t.leaf_code(id);
        else {
            next_frontier[next] = id;
            code[k] = (int)(next_level_base + next);    // the next level starts right behind this one
            next++;
        }
    }
    float scale[3];
    uint32_t lo4[3], hi4[3];
    for (int a = 0; a < 3; a++) {
        const float ext = nb.hi[a] - nb.lo[a];
        // smallest power of two s with 255 * s >= ext (a zero or denormal extent gets the smallest normal number)
        int e = 0;
        const float m = __builtin_frexpf(ext / 255.0f, &e);      // ext / 255 = m * 2^e, m in [0.5, 1)
        float s = (ext > 0.0f && ext < 3.0e38f) ? __builtin_ldexpf(1.0f, m == 0.5f ? e - 1 : e) : 1.17549435e-38f;
        if (!(s >= 1.17549435e-38f)) s = 1.17549435e-38f;
        for (int tries = 0; tries < 8; tries++) {
            if (quantise_axis(cb, nk, a, nb.lo[a], s, lo4[a], hi4[a])) break;
            s = s * 2.0f;                               // rounding in fma left the last step short: coarser grid
        }
        scale[a] = s;
    }
    WNode w;
    w.q0 = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], scale[0]);
    w.q1 = make_float4(__uint_as_float(lo4[0]), __uint_as_float(hi4[0]), __uint_as_float(lo4[1]), __uint_as_float(hi4[1]));
    w.q2 = make_float4(__uint_as_float(lo4[2]), __uint_as_float(hi4[2]), scale[1], scale[2]);
    w.q3 = make_float4(__int_as_float(code[0]), __int_as_float(code[1]), __int_as_float(code[2]), __int_as_float(code[3]));
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
    uint32_t kid[4] = {NO_KID, NO_KID, NO_KID, NO_KID}, internal = 0, mine = 0;
    if (f < count)
        for (int k = 0; k < 4; k++) {
            kid[k] = kids[4 * (size_t)f + k];
            if (kid[k] != NO_KID && !t.is_leaf(kid[k])) { internal |= 1u << k; mine++; }
        }
    uint32_t block_total;
    const uint32_t next = tally[blockIdx.x] + rt_scan::block_exclusive<WB>(mine, lds, block_total);
    if (f >= count) return;
    emit_node(t, frontier[f], kid, internal, base + f, base + count, next, next_frontier, out);
}

// The top of the tree in ONE workgroup: while a level has at most TOPW nodes, expand, number and emit it between two
// barriers -- a level is ~6 dependent loads whatever its size, and as two launches each it cost 27 us.  Levels 0 .. 4
// (4^4 = 256) are certain to be handled here; the per-level launches start behind them and skip whatever else this
// kernel got to (ws->done).  (256 threads: the emit code wants 156 VGPRs, a 1024-thread workgroup would spill.)
constexpr uint32_t TOPW = 256, TOP_SURE = 5;            // levels 0 .. TOP_SURE-1 have at most 4^l <= TOPW nodes
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
        uint32_t kid[4] = {NO_KID, NO_KID, NO_KID, NO_KID}, internal = 0, mine = 0, b = 0;
        if (f < count) {
            b = frontier[lvl & 1u][f];
            mine = expand_node(t, b, kid);
            for (int k = 0; k < 4; k++)
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
                                                     uint32_t *__restrict__ size, uint32_t *__restrict__ offset, uint32_t *__restrict__ leaf_prim)
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
        left[c] = nd.left >= n - 1 ? nd.left - (n - 1) : n + nd.left;
        right[c] = nd.right >= n - 1 ? nd.right - (n - 1) : n + nd.right;
    }
}

inline unsigned gr(size_t n) { return (unsigned)((n + WB - 1) / WB); }
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// two frontiers, the children of a frontier, one tally per workgroup of a level, the level table
size_t rt_wide_temp_bytes(uint32_t n)
{
    const size_t f = (size_t)n / 2 + 2;
    return 2 * up256(4 * f) + up256(16 * f) + up256(4 * (size_t)gr(f)) + up256(sizeof(WideState));
}

int rt_build_wide_layout(rt_context *ctx, BvhDev &bv, uint32_t n, uint32_t root, const uint32_t *left, const uint32_t *right,
                         const float *box6, const uint32_t *size, const uint32_t *offset, const uint32_t *leaf_prim, uint32_t leaf_max,
                         void *tmp, size_t tmp_bytes)
{
    hipStream_t st = ctx->stream;
    TreeView t;
    t.left = left; t.right = right; t.box = (const Box6 *)box6; t.size = size; t.offset = offset; t.leaf_prim = leaf_prim;
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
    uint32_t *kids = (uint32_t *)p; p += up256(16 * fcap);
    uint32_t *tally = (uint32_t *)p; p += up256(4 * (size_t)gr(fcap));
    WideState *ws = (WideState *)p;
    int rc = RT_OK;
    do {
        if ((rc = bv.wide.reserve(sizeof(WNode) * (size_t)(n - 1))) != RT_OK) break;
        k_wide_top<<<1, TOPW, 0, st>>>(t, frontier[0], frontier[1], ws, root, (uint32_t)fcap, bv.wide.as<WNode>());
        // Levels are launched in batches without looking at their sizes: a level has at most four times the nodes of the
        // one before (and never more than fcap), a level past the end of the tree costs two empty launches.  The first
        // batch is sized for a tree half again as deep as a balanced one; the host reads the level table after each batch.
        // k_wide_top has certainly built levels 0 .. TOP_SURE-1 (a level has at most 4^l nodes, it takes every level of <= TOPW)
        uint32_t lvl = TOP_SURE, levels = 0, wide_n = 0;
        uint64_t bound = 4 * TOPW;                            // upper bound of the frontier at level lvl
        uint32_t batch = 6;
        for (uint32_t m = n; m > 1; m >>= 2) batch++;
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
                bound = bound < fcap ? bound * 4 : fcap;
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
        bv.fast_depth = 3 * levels;        // a step leaves at most three siblings behind
    } while (0);
    own.release();
    return rc;
}

// arena bytes rt_build_wide_from_lbvh takes: the tree in cluster numbering + the scratch of the collapse
size_t rt_wide_lbvh_temp_bytes(uint32_t n)
{
    const size_t nn2 = 2 * (size_t)n;
    return 2 * up256(4 * (size_t)n) + up256(sizeof(Box6) * nn2) + 3 * up256(4 * nn2) + rt_wide_temp_bytes(n);
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
    const size_t tree = 2 * up256(4 * (size_t)(n - 1)) + up256(sizeof(Box6) * nn2) + 3 * up256(4 * nn2);
    const size_t wide = rt_wide_temp_bytes(n);
    RT_TRY(ctx->build_arena.reserve(tree + wide));
    char *p = (char *)ctx->build_arena.p;
    uint32_t *left = (uint32_t *)p; p += up256(4 * (size_t)(n - 1));
    uint32_t *right = (uint32_t *)p; p += up256(4 * (size_t)(n - 1));
    Box6 *box = (Box6 *)p; p += up256(sizeof(Box6) * nn2);
    uint32_t *size = (uint32_t *)p; p += up256(4 * nn2);
    uint32_t *offset = (uint32_t *)p; p += up256(4 * nn2);
    uint32_t *leaf_prim = (uint32_t *)p; p += up256(4 * nn2);
    k_lbvh_to_tree<<<gr(nn2), WB, 0, st>>>(bv.nodes.as<rt_bvh_node>(), bv.ranges.as<uint2>(), n, left, right, box, size, offset, leaf_prim);
    HIP_TRY(hipGetLastError());
    return rt_build_wide_layout(ctx, bv, n, n /* canonical root 0 */, left, right, (const float *)box, size, offset, tlas ? leaf_prim : nullptr,
                                tlas ? 1u : leaf_max, p, wide);
}
