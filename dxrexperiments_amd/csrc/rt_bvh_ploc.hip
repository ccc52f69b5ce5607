// rt_bvh_ploc.hip -- traversal-layout optimiser for a BLAS.
//
// The canonical acceleration structure (rt_bvh_build.hip) is an LBVH: fully determined,
// index-exact against the oracle, but only spatial-median quality (~95 box tests per primary
// ray on the Sponza-class scene).  Results do not depend on the tree that is walked (DESIGN.md
// "Exactness rule"), so the PRODUCTION traversal is free to use a better hierarchy over the
// same triangles.  This file builds one with PLOC (parallel locally-ordered clustering,
// Meister & Bittner 2018): clusters start as the Morton-sorted leaves of the canonical tree;
// every round each cluster finds, inside a window of +-RADIUS neighbours, the partner that
// minimises the surface area of the merged box; mutual pairs merge; the cluster array is
// compacted in order (prefix sums by workgroup index, rt_level_scan.h), until one cluster is left.
// Node ids come from those prefix sums, so the build is run-to-run deterministic.  A round is two
// launches; its cluster count lives in device memory and the host launches a batch of rounds blind.
//
// Output: triangles re-gathered in the depth-first order of the new tree, so that every subtree is a
// contiguous triangle range and subtrees of <= leaf_max triangles collapse to leaves, and the binary
// tree itself, which rt_bvh_wide.hip collapses into the four-wide 64-B nodes the traversal walks.
#include "rt_internal.h"

#include "rt_level_scan.h"

#include <cstring>

namespace {

// Search window.  Measured on the three bench scenes, frame ms (1080p 262 k triangles / 4K 10 M triangles, 4 bounces / 4K 4096
// instances): radius 2: 2.75 / 16.3 / 6.28, 3: 2.70 / 15.9, 4: 2.64 / 15.4 / 6.20, 6: 2.83 / 15.6, 8: 2.76 / 15.4 / 6.24,
// 12: 2.82 / 15.9 / 6.17, 16: 2.87 / 16.3 -- a few per cent of tree-quality noise, no trend towards wide windows once the
// binary tree is collapsed four wide; 4 is also the cheapest to build.
#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 4
#endif
constexpr int PLOC_RADIUS = RT_PLOC_RADIUS;
#ifndef RT_PLOC_TAIL_RADIUS
#define RT_PLOC_TAIL_RADIUS RT_PLOC_RADIUS
#endif
constexpr int PLOC_TAIL_RADIUS = RT_PLOC_TAIL_RADIUS;       // window of the single-workgroup rounds at the top of the tree
constexpr unsigned PB = 256;

struct Box6 { float lo[3]; float hi[3]; };

__device__ __forceinline__ float merged_area(const Box6 &a, const Box6 &b)
{
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// a piece of the build's one temporary allocation
struct View {
    void *p = nullptr;
    template <class T> T *as() const { return (T *)p; }
};

// where the clustering stands before round r (device memory, one entry per round of a batch)
struct PlocRound { uint32_t c, next_node, cur, error; };
constexpr uint32_t PLOC_TAIL = 2048, TAIL_BLOCK = 1024, PLOC_MAX_BATCH = 64;   // (tail at 4096: 0.20 ms in the one workgroup; a multi-kernel round costs ~12 us)

// the arrays the PLOC rounds read and write
struct PlocArrays {
    uint32_t *nn, *flags;
    uint64_t *tally;                   // per workgroup: clusters kept (low word) and nodes created (high word)
    uint32_t *arrivals;
    PlocRound *round;                  // PLOC_MAX_BATCH + 1 entries
    uint32_t *cl_node[2];
    Box6 *cl_box[2];
    uint32_t *left, *right, *size, *parent;
    Box6 *node_box;
};

// clusters start as the canonical leaves, in key order
// (leaf_box6 != nullptr: the leaves are the references of a model with split triangles, rt_refs.h -- 6 floats each, in Morton order)
__global__ void k_ploc_init(const rt_bvh_node *__restrict__ nodes, const float *__restrict__ leaf_box6, uint32_t n, uint32_t *__restrict__ cl_node, Box6 *__restrict__ cl_box,
                            uint32_t *__restrict__ size, uint32_t *__restrict__ parent, PlocRound *__restrict__ round0, uint32_t *__restrict__ arrivals)
{
    const uint32_t k = blockIdx.x * PB + threadIdx.x;
    if (k == 0) { const PlocRound r0 = {n, n, 0u, 0u}; *round0 = r0; *arrivals = 0; }
    if (k >= n) return;
    Box6 b;
    if (leaf_box6) {
        for (int c = 0; c < 3; c++) { b.lo[c] = leaf_box6[6 * (size_t)k + c]; b.hi[c] = leaf_box6[6 * (size_t)k + 3 + c]; }
    } else {
        const rt_bvh_node nd = nodes[n - 1 + k];
        for (int c = 0; c < 3; c++) { b.lo[c] = nd.bmin[c]; b.hi[c] = nd.bmax[c]; }
    }
    cl_node[k] = k;                 // node ids: leaves 0..n-1 (key order), internal n..2n-2 (creation order)
    cl_box[k] = b;
    size[k] = 1;
    parent[k] = 0xFFFFFFFFu;
}

// First half of a round.  Nearest neighbour inside the window (smallest merged area, ties -> lower index) for the
// workgroup's clusters and a halo of RADIUS either side, so that "is the choice mutual" needs no second launch; then the
// flags (bit 0: the cluster survives, bit 1: it creates a node), the workgroup's tally, and -- in the last workgroup to
// arrive -- the offsets of all workgroups and the size of the next round.
__global__ void __launch_bounds__(PB) k_ploc_pair(PlocArrays a, uint32_t r)
{
    constexpr int R = PLOC_RADIUS;
    __shared__ Box6 tile[PB + 4 * R];
    __shared__ uint32_t near[PB + 2 * R];
    __shared__ uint64_t lds64[PB];
    __shared__ uint32_t lds[PB / 64];
    __shared__ uint32_t lds_flag;
    const PlocRound s = a.round[r];
    if (s.c <= PLOC_TAIL || s.error) {                 // the tail kernel's share, or a failed round: hand the state on
        if (blockIdx.x == 0 && threadIdx.x == 0) a.round[r + 1] = s;
        return;
    }
    const uint32_t c = s.c, nblocks = (c + PB - 1) / PB;
    if (blockIdx.x >= nblocks) return;
    const Box6 *cl_box = a.cl_box[s.cur];
    const int base = (int)(blockIdx.x * PB) - 2 * R;    // tile[t] = cluster base + t
    for (int t = threadIdx.x; t < (int)PB + 4 * R; t += PB) {
        const int g = base + t;
        if (g >= 0 && g < (int)c) tile[t] = cl_box[g];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (int)PB + 2 * R; t += PB) {       // near[t] = nn of cluster base + R + t
        const int i = base + R + t;
        if (i < 0 || i >= (int)c) continue;
        const Box6 me = tile[t + R];
        float best = __uint_as_float(0x7f800000u);
        uint32_t arg = (uint32_t)i;
        for (int d = -R; d <= R; d++) {
            const int j = i + d;
            if (d == 0 || j < 0 || j >= (int)c) continue;
            const float ar = merged_area(me, tile[t + R + d]);
            if (ar < best) { best = ar; arg = (uint32_t)j; }
        }
        near[t] = arg;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    uint32_t packed = 0;                                // kept | created << 16
    if (i < c) {
        const uint32_t j = near[threadIdx.x + R];
        const bool mutual = j != i && near[(int)j - base - R] == i;
        const uint32_t mg = (mutual && i < j) ? 1u : 0u;        // the lower index of a mutual pair creates the node
        const uint32_t kp = (mutual && i > j) ? 0u : 1u;        // the higher index disappears
        a.nn[i] = j;
        a.flags[i] = kp | (mg << 1);
        packed = kp | (mg << 16);
    }
    uint32_t block_total;
    (void)rt_scan::block_exclusive<PB>(packed, lds, block_total);
    const uint64_t mine = (uint64_t)(block_total & 0xFFFFu) | ((uint64_t)(block_total >> 16) << 32);
    if (!rt_scan::publish_and_arrive(a.tally, mine, a.arrivals, nblocks, &lds_flag)) return;
    const uint64_t total = rt_scan::scan_tallies<uint64_t, PB>(a.tally, nblocks, lds64);
    if (threadIdx.x == 0) {
        const uint32_t kept = (uint32_t)total, created = (uint32_t)(total >> 32);
        PlocRound nx = {kept, s.next_node + created, s.cur ^ 1u, 0u};
        if (created == 0 || kept != c - created) { nx = s; nx.error = 1; }      // no progress: reported by the host
        a.round[r + 1] = nx;
        *a.arrivals = 0;
    }
}

// Second half: every surviving cluster moves to its place in the other cluster array; the lower index of a mutual pair
// becomes the new node (same arithmetic and numbering as the tail kernel below).
__global__ void __launch_bounds__(PB) k_ploc_apply(PlocArrays a, uint32_t r, uint32_t n)
{
    __shared__ uint32_t lds[PB / 64];
    const PlocRound s = a.round[r];
    if (s.c <= PLOC_TAIL || s.error || a.round[r + 1].error) return;
    const uint32_t c = s.c;
    if (blockIdx.x * PB >= c) return;
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    const uint32_t fl = i < c ? a.flags[i] : 0u;
    uint32_t block_total;
    const uint32_t before = rt_scan::block_exclusive<PB>((fl & 1u) | ((fl >> 1) << 16), lds, block_total);
    if (!(fl & 1u)) return;
    const uint64_t off = a.tally[blockIdx.x];
    const uint32_t pos = (uint32_t)off + (before & 0xFFFFu);
    const uint32_t *cl_node = a.cl_node[s.cur];
    const Box6 *cl_box = a.cl_box[s.cur];
    uint32_t *out_node = a.cl_node[s.cur ^ 1u];
    Box6 *out_box = a.cl_box[s.cur ^ 1u];
    if (fl & 2u) {
        const uint32_t j = a.nn[i];
        const uint32_t m = s.next_node + (uint32_t)(off >> 32) + (before >> 16);
        const uint32_t na = cl_node[i], nb = cl_node[j];
        const Box6 ba = cl_box[i], bb = cl_box[j];
        Box6 u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
        a.left[m - n] = na;
        a.right[m - n] = nb;
        a.node_box[m] = u;
        a.size[m] = a.size[na] + a.size[nb];
        a.parent[na] = m;
        a.parent[nb] = m;
        a.parent[m] = 0xFFFFFFFFu;
        out_node[pos] = m;
        out_box[pos] = u;
    } else {
        out_node[pos] = cl_node[i];
        out_box[pos] = cl_box[i];
    }
}

// The tail of the clustering: once PLOC_TAIL or fewer clusters are left, ONE workgroup runs all the remaining rounds
// (nearest neighbour, mutual-pair flags, the two prefix sums, merge) back to back with barriers in between -- the same
// arithmetic and the same node numbering as the multi-kernel rounds, without ~40 rounds of launches and host round trips
// for a handful of clusters each.  The cluster array (boxes, node ids, sizes) lives in LDS for the whole tail: 2048 x 34 B
// = 68 KiB of the CU's 160; the first version kept it in global memory and spent 0.36 ms of a 2.5 ms build on round trips.
struct TailLds {
    Box6 box[PLOC_TAIL];
    uint32_t node[PLOC_TAIL], size[PLOC_TAIL];
    uint16_t nn[PLOC_TAIL];
    uint32_t scan[TAIL_BLOCK / 64];
};
__global__ void __launch_bounds__(TAIL_BLOCK) k_ploc_tail(PlocArrays a, uint32_t r, uint32_t n, PlocRound *__restrict__ result)
{
    const PlocRound s = a.round[r];
    if (s.c > PLOC_TAIL || s.error) {                  // the batch of rounds fell short (or failed): the host decides
        if (threadIdx.x == 0) *result = s;
        return;
    }
    uint32_t c = s.c, next_node = s.next_node;
    const uint32_t cur = s.cur;
    extern __shared__ __align__(16) unsigned char tail_smem[];
    TailLds &L = *reinterpret_cast<TailLds *>(tail_smem);
    constexpr uint32_t PER = PLOC_TAIL / TAIL_BLOCK;
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < c; i += TAIL_BLOCK) {
        const uint32_t id = a.cl_node[cur][i];
        L.box[i] = a.cl_box[cur][i];
        L.node[i] = id;
        L.size[i] = a.size[id];
    }
    __syncthreads();
    while (c > 1) {
        for (uint32_t i = t; i < c; i += TAIL_BLOCK) {               // nearest neighbour (as k_ploc_pair)
            const Box6 me = L.box[i];
            float best = __uint_as_float(0x7f800000u);
            uint32_t arg = i;
            for (int d = -PLOC_TAIL_RADIUS; d <= PLOC_TAIL_RADIUS; d++) {
                const int j = (int)i + d;
                if (d == 0 || j < 0 || j >= (int)c) continue;
                const float ar = merged_area(me, L.box[j]);
                if (ar < best) { best = ar; arg = (uint32_t)j; }
            }
            L.nn[i] = (uint16_t)arg;
        }
        __syncthreads();
        // flags, PER consecutive clusters per thread, kept in registers; tallies packed kept | created << 16
        uint32_t mine = 0, keep_bits = 0, merge_bits = 0;
        for (uint32_t e = 0; e < PER; e++) {
            const uint32_t i = t * PER + e;
            if (i >= c) break;
            const uint32_t j = L.nn[i];
            const bool mutual = j != i && L.nn[j] == i;
            const uint32_t mg = (mutual && i < j) ? 1u : 0u, kp = (mutual && i > j) ? 0u : 1u;
            merge_bits |= mg << e;
            keep_bits |= kp << e;
            mine += kp | (mg << 16);
        }
        uint32_t total;
        const uint32_t before = rt_scan::block_exclusive<TAIL_BLOCK>(mine, L.scan, total);
        uint32_t k_run = before & 0xFFFFu, m_run = before >> 16;
        const uint32_t kept = total & 0xFFFFu, merged = total >> 16;
        if (merged == 0 || kept != c - merged) break;                 // no progress: reported by the host (block-uniform)
        // what this thread's clusters become, computed into registers before anybody overwrites the array
        Box6 out_box[PER];
        uint32_t out_node[PER], out_size[PER], out_pos[PER];
        for (uint32_t e = 0; e < PER; e++) {
            const uint32_t i = t * PER + e;
            out_pos[e] = 0xFFFFFFFFu;
            if (i >= c || !((keep_bits >> e) & 1u)) continue;
            out_pos[e] = k_run;
            if ((merge_bits >> e) & 1u) {
                const uint32_t j = L.nn[i], m = next_node + m_run;
                const uint32_t na = L.node[i], nb = L.node[j], sz = L.size[i] + L.size[j];
                const Box6 ba = L.box[i], bb = L.box[j];
                Box6 u;
                for (int k = 0; k < 3; k++) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
                a.left[m - n] = na;
                a.right[m - n] = nb;
                a.node_box[m] = u;
                a.size[m] = sz;
                a.parent[na] = m;
                a.parent[nb] = m;
                a.parent[m] = 0xFFFFFFFFu;
                out_node[e] = m;
                out_box[e] = u;
                out_size[e] = sz;
                m_run++;
            } else {
                out_node[e] = L.node[i];
                out_box[e] = L.box[i];
                out_size[e] = L.size[i];
            }
            k_run++;
        }
        __syncthreads();
        for (uint32_t e = 0; e < PER; e++)
            if (out_pos[e] != 0xFFFFFFFFu) { L.box[out_pos[e]] = out_box[e]; L.node[out_pos[e]] = out_node[e]; L.size[out_pos[e]] = out_size[e]; }
        __syncthreads();
        c = kept;
        next_node += merged;
    }
    if (t == 0) { const PlocRound out = {c, next_node, cur, 0u}; *result = out; }
}

// leaf boxes into node_box[0..n-1] so every node id indexes one box array
__global__ void k_ploc_leaf_boxes(const Box6 *__restrict__ cl_box, uint32_t n, Box6 *__restrict__ node_box)
{
    const uint32_t k = blockIdx.x * PB + threadIdx.x;
    if (k < n) node_box[k] = cl_box[k];
}

// depth-first offset of every node = number of leaves before it
__global__ void k_ploc_offsets(const uint32_t *__restrict__ left, const uint32_t *__restrict__ right, const uint32_t *__restrict__ size,
                               const uint32_t *__restrict__ parent, uint32_t n, uint32_t *__restrict__ offset)
{
    const uint32_t id = blockIdx.x * PB + threadIdx.x;
    if (id >= 2 * n - 1) return;
    uint32_t off = 0, cur = id;
    for (uint32_t p = parent[cur]; p != 0xFFFFFFFFu; p = parent[cur]) {
        if (right[p - n] == cur) off += size[left[p - n]];
        cur = p;
    }
    offset[id] = off;
}

// (leaf_prim != nullptr: leaf k is a reference of primitive leaf_prim[k] with the box leaf_box6[k]; the record of a split triangle's reference
// says so and its box goes to rec_boxes beside it)
__global__ void k_ploc_tris(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ leaf_prim, const float *__restrict__ leaf_box6,
                            const uint32_t *__restrict__ ref_off, const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx,
                            const uint32_t *__restrict__ offset, uint32_t n, TriRec *__restrict__ tris, float *__restrict__ rec_boxes)
{
    const uint32_t k = blockIdx.x * PB + threadIdx.x;
    if (k >= n) return;
    const uint32_t prim = leaf_prim ? leaf_prim[k] : (uint32_t)(keys[k] & 0xFFFFFFFFull);
    const rt_float3 p0 = verts[idx[3 * prim + 0]].position;
    const rt_float3 p1 = verts[idx[3 * prim + 1]].position;
    const rt_float3 p2 = verts[idx[3 * prim + 2]].position;
    TriRec t;
    t.a = make_float4(p0.x, p0.y, p0.z, p1.x);
    t.b = make_float4(p1.y, p1.z, p2.x, p2.y);
    const bool split = leaf_prim && ref_off[prim + 1] - ref_off[prim] > 1u;
    t.c = make_float4(p2.z, __uint_as_float(prim), __uint_as_float(split ? 1u : 0u), 0.0f);
    tris[offset[k]] = t;
    if (rec_boxes)
        for (int q = 0; q < 6; q++) rec_boxes[6 * (size_t)offset[k] + q] = leaf_box6[6 * (size_t)k + q];
}

inline unsigned gr(size_t n) { return (unsigned)((n + PB - 1) / PB); }

}  // namespace

// bytes of temporaries rt_build_ploc_layout slices out of the context's build arena (an upper estimate: the
// scan scratch is small next to the rest)
size_t rt_ploc_temp_bytes(uint32_t n)
{
    const size_t nn2 = 2 * (size_t)n;
    return 8 * (size_t)n + 2 * sizeof(Box6) * (size_t)n + 8 * (size_t)n + 8 * ((size_t)n / PB + 1) + 64 * (size_t)n + 8 * (size_t)n + sizeof(Box6) * nn2 + 12 * nn2 +
           sizeof(PlocRound) * (PLOC_MAX_BATCH + 2) + 16 + 16 * 256;
}

// Rebuilds m->tris and m->blas.wide / root_code / fast_depth from a PLOC tree.  The canonical arrays
// (nodes, keys, parents) are left untouched.  Returns RT_OK with *done = false for meshes too small to bother.
int rt_build_ploc_layout(rt_context *ctx, rt_model *m, bool *done, uint32_t n_leaves, const float *leaf_box6, const uint32_t *leaf_prim)
{
    const uint32_t n = leaf_box6 ? n_leaves : m->n_tris;          // leaves of the tree: triangles, or the references of a model with split ones
    *done = false;
    if (n < 2 * ctx->leaf_max + 2) return RT_OK;          // tiny meshes: the LBVH layout is as good as any
    hipStream_t st = ctx->stream;
    // every temporary of the build is carved out of ONE allocation (hipMalloc / hipFree synchronise the
    // device and cost more than the kernels of a small build)
    View cl_node[2], cl_box[2], nn, flags, tally, wide_extra, left, right, node_box, size, parent, offset, state;
    int rc = RT_OK;
    do {
        const size_t nn2 = 2 * (size_t)n - 1;
        struct Want { View *v; size_t bytes; };
        const Want wants[] = {{&cl_node[0], 4 * (size_t)n}, {&cl_node[1], 4 * (size_t)n}, {&cl_box[0], sizeof(Box6) * (size_t)n}, {&cl_box[1], sizeof(Box6) * (size_t)n},
                              {&nn, 4 * (size_t)n}, {&flags, 4 * (size_t)n}, {&tally, 8 * (size_t)gr(n)},
                              {&wide_extra, 64 * (size_t)n},           // (only widens the slice the collapse uses as its scratch: rt_wide_temp_bytes)
                              {&left, 4 * (size_t)(n - 1)}, {&right, 4 * (size_t)(n - 1)},
                              {&node_box, sizeof(Box6) * nn2}, {&size, 4 * nn2}, {&parent, 4 * nn2}, {&offset, 4 * nn2},
                              {&state, sizeof(PlocRound) * (PLOC_MAX_BATCH + 2) + 16}};
        size_t total = 0;
        for (const Want &w : wants) total += (w.bytes + 255) & ~(size_t)255;
        if ((rc = ctx->build_arena.reserve(total)) != RT_OK) break;      // normally already there (rt_ploc_temp_bytes)
        size_t at = 0;
        for (const Want &w : wants) { w.v->p = (char *)ctx->build_arena.p + at; at += (w.bytes + 255) & ~(size_t)255; }

        PlocArrays pa;
        pa.nn = nn.as<uint32_t>(); pa.flags = flags.as<uint32_t>(); pa.tally = tally.as<uint64_t>();
        pa.round = state.as<PlocRound>();
        PlocRound *d_result = pa.round + PLOC_MAX_BATCH + 1;
        pa.arrivals = (uint32_t *)(d_result + 1);
        for (int k = 0; k < 2; k++) { pa.cl_node[k] = cl_node[k].as<uint32_t>(); pa.cl_box[k] = cl_box[k].as<Box6>(); }
        pa.left = left.as<uint32_t>(); pa.right = right.as<uint32_t>(); pa.size = size.as<uint32_t>(); pa.parent = parent.as<uint32_t>();
        pa.node_box = node_box.as<Box6>();
        // 68 KiB of dynamic LDS: above the default 64 KiB limit of a launch (the attribute is per device, set every build)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ploc_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TailLds)) != hipSuccess) {
            rt_set_error("PLOC tail: cannot reserve %zu bytes of LDS", sizeof(TailLds));
            rc = RT_ERR_HIP;
            break;
        }

        if (leaf_box6) {          // records and their boxes for every reference
            if ((rc = m->tris.reserve(sizeof(TriRec) * (size_t)n)) != RT_OK || (rc = m->rec_boxes.reserve(24 * (size_t)n)) != RT_OK) break;
        }
        k_ploc_init<<<gr(n), PB, 0, st>>>(m->blas.nodes.as<rt_bvh_node>(), leaf_box6, n, cl_node[0].as<uint32_t>(), cl_box[0].as<Box6>(),
                                         size.as<uint32_t>(), parent.as<uint32_t>(), pa.round, pa.arrivals);
        k_ploc_leaf_boxes<<<gr(n), PB, 0, st>>>(cl_box[0].as<Box6>(), n, node_box.as<Box6>());
        // Rounds are launched in batches without looking at the cluster count: every launch covers the count the batch
        // started with (workgroups past the live count leave at once), a round past the tail threshold costs two empty
        // launches, and the tail kernel closes the batch.  A round keeps ~0.76 of its clusters on the scenes measured;
        // the estimate below assumes 0.78 and a batch that falls short is simply followed by another.
        PlocRound res = {n, n, 0u, 0u};
        for (int batchno = 0; batchno < 4096; batchno++) {
            uint32_t rounds = 0;
            for (double x = (double)res.c; x > (double)PLOC_TAIL && rounds < PLOC_MAX_BATCH; x *= 0.78) rounds++;
            if (rounds > 0 && rounds < PLOC_MAX_BATCH) rounds++;
            if (ctx->build_batch && rounds > ctx->build_batch) rounds = ctx->build_batch;      // (tests: force short batches)
            for (uint32_t r = 0; r < rounds; r++) {
                k_ploc_pair<<<gr(res.c), PB, 0, st>>>(pa, r);
                k_ploc_apply<<<gr(res.c), PB, 0, st>>>(pa, r, n);
            }
            k_ploc_tail<<<1, TAIL_BLOCK, sizeof(TailLds), st>>>(pa, rounds, n, d_result);
            PlocRound *host = ctx->pinned ? (PlocRound *)ctx->pinned : &res;      // page-locked: no staging copy
            if (hipMemcpyAsync(host, d_result, sizeof(PlocRound), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
                hipGetLastError() != hipSuccess) {
                rt_set_error("PLOC rounds failed: %s", hipGetErrorString(hipGetLastError()));
                rc = RT_ERR_HIP;
                break;
            }
            const uint32_t before = res.c;
            res = *host;
            if (res.error || (res.c > PLOC_TAIL && res.c >= before)) {
                rt_set_error("PLOC made no progress (%u clusters)", res.c);
                rc = RT_ERR_STATE;
                break;
            }
            if (res.c <= PLOC_TAIL) break;
            // short of the tail: the next batch starts from where this one stopped
            if (hipMemcpyAsync(pa.round, d_result, sizeof(PlocRound), hipMemcpyDeviceToDevice, st) != hipSuccess) { rt_set_error("PLOC: state copy failed"); rc = RT_ERR_HIP; break; }
        }
        if (rc != RT_OK) break;
        const uint32_t c = res.c, next_node = res.next_node;
        if (c != 1 || next_node != 2 * n - 1) {
            rt_set_error("PLOC did not converge (%u clusters, %u nodes)", c, next_node);
            rc = RT_ERR_STATE;
            break;
        }
        k_ploc_offsets<<<gr(nn2), PB, 0, st>>>(left.as<uint32_t>(), right.as<uint32_t>(), size.as<uint32_t>(), parent.as<uint32_t>(), n,
                                              offset.as<uint32_t>());
        k_ploc_tris<<<gr(n), PB, 0, st>>>(m->blas.keys.as<uint64_t>(), leaf_prim, leaf_box6, leaf_box6 ? m->ref_off.as<uint32_t>() : nullptr,
                                         m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(), offset.as<uint32_t>(), n, m->tris.as<TriRec>(),
                                         leaf_box6 ? m->rec_boxes.as<float>() : nullptr);
        if (hipGetLastError() != hipSuccess) { rt_set_error("PLOC layout kernels failed"); rc = RT_ERR_HIP; break; }
        // four-wide nodes from the binary tree (root = the last node created); the cluster arrays of the rounds are free now
        // and serve as its scratch
        if ((rc = rt_build_wide_layout(ctx, m->blas, n, 2 * n - 2, left.as<uint32_t>(), right.as<uint32_t>(), parent.as<uint32_t>(), (const float *)node_box.p,
                                       size.as<uint32_t>(), offset.as<uint32_t>(), nullptr, ctx->leaf_max, cl_node[0].p,
                                       (size_t)((char *)left.p - (char *)cl_node[0].p))) != RT_OK) break;
        if (leaf_box6) m->n_recs = n;
        *done = true;
    } while (0);
    return rc;
}
