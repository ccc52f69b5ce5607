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
// compacted in order (scans, no atomics), until one cluster is left.  Node ids come from
// prefix sums, so the build is run-to-run deterministic.
//
// Output: triangles re-gathered in the depth-first order of the new tree, so that every subtree is a
// contiguous triangle range and subtrees of <= leaf_max triangles collapse to leaves, and the binary
// tree itself, which rt_bvh_wide.hip collapses into the four-wide 64-B nodes the traversal walks.
#include "rt_internal.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 8
#endif
constexpr int PLOC_RADIUS = RT_PLOC_RADIUS;
constexpr unsigned PB = 256;

struct Box6 { float lo[3]; float hi[3]; };

__device__ __forceinline__ float merged_area(const Box6 &a, const Box6 &b)
{
    const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// clusters start as the canonical leaves, in key order
__global__ void k_ploc_init(const rt_bvh_node *__restrict__ nodes, uint32_t n, uint32_t *__restrict__ cl_node, Box6 *__restrict__ cl_box,
                            uint32_t *__restrict__ size, uint32_t *__restrict__ parent)
{
    const uint32_t k = blockIdx.x * PB + threadIdx.x;
    if (k >= n) return;
    const rt_bvh_node nd = nodes[n - 1 + k];
    Box6 b;
    for (int c = 0; c < 3; c++) { b.lo[c] = nd.bmin[c]; b.hi[c] = nd.bmax[c]; }
    cl_node[k] = k;                 // node ids: leaves 0..n-1 (key order), internal n..2n-2 (creation order)
    cl_box[k] = b;
    size[k] = 1;
    parent[k] = 0xFFFFFFFFu;
}

// nearest neighbour inside the window, smallest merged area, ties -> lower index
__global__ void __launch_bounds__(PB) k_ploc_nn(const Box6 *__restrict__ cl_box, uint32_t c, uint32_t *__restrict__ nn)
{
    __shared__ Box6 tile[PB + 2 * PLOC_RADIUS];
    const int base = (int)(blockIdx.x * PB) - PLOC_RADIUS;
    for (int t = threadIdx.x; t < (int)PB + 2 * PLOC_RADIUS; t += PB) {
        const int g = base + t;
        if (g >= 0 && g < (int)c) tile[t] = cl_box[g];
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= c) return;
    const Box6 me = tile[threadIdx.x + PLOC_RADIUS];
    float best = __uint_as_float(0x7f800000u);
    uint32_t arg = i;
    for (int d = -PLOC_RADIUS; d <= PLOC_RADIUS; d++) {
        const int j = (int)i + d;
        if (d == 0 || j < 0 || j >= (int)c) continue;
        const float a = merged_area(me, tile[threadIdx.x + PLOC_RADIUS + d]);
        if (a < best) { best = a; arg = (uint32_t)j; }
    }
    nn[i] = arg;
}

__global__ void k_ploc_flags(const uint32_t *__restrict__ nn, uint32_t c, uint32_t *__restrict__ keep, uint32_t *__restrict__ merge)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= c) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    merge[i] = (mutual && i < j) ? 1u : 0u;        // the lower index of a mutual pair creates the node
    keep[i] = (mutual && i > j) ? 0u : 1u;         // the higher index disappears
}

// a piece of the build's one temporary allocation
struct View {
    void *p = nullptr;
    template <class T> T *as() const { return (T *)p; }
};

// the arrays one PLOC round reads and writes
struct PlocArrays {
    uint32_t *nn, *keep, *merge, *keep_pos, *merge_pos;
    uint32_t *cl_node[2];
    Box6 *cl_box[2];
    uint32_t *left, *right, *size, *parent;
    Box6 *node_box;
};

__device__ __forceinline__ void ploc_apply_one(uint32_t i, const uint32_t *nn, const uint32_t *keep, const uint32_t *merge,
                                               const uint32_t *keep_pos, const uint32_t *merge_pos, uint32_t n, uint32_t next_node,
                                               const uint32_t *cl_node, const Box6 *cl_box, uint32_t *out_node, Box6 *out_box,
                                               uint32_t *left, uint32_t *right, Box6 *node_box, uint32_t *size, uint32_t *parent)
{
    if (!keep[i]) return;
    const uint32_t pos = keep_pos[i];
    if (merge[i]) {
        const uint32_t j = nn[i];
        const uint32_t m = next_node + merge_pos[i];
        const uint32_t a = cl_node[i], b = cl_node[j];
        const Box6 ba = cl_box[i], bb = cl_box[j];
        Box6 u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
        left[m - n] = a;
        right[m - n] = b;
        node_box[m] = u;
        size[m] = size[a] + size[b];
        parent[a] = m;
        parent[b] = m;
        parent[m] = 0xFFFFFFFFu;
        out_node[pos] = m;
        out_box[pos] = u;
    } else {
        out_node[pos] = cl_node[i];
        out_box[pos] = cl_box[i];
    }
}

__global__ void k_ploc_apply(const uint32_t *__restrict__ nn, const uint32_t *__restrict__ keep, const uint32_t *__restrict__ merge,
                             const uint32_t *__restrict__ keep_pos, const uint32_t *__restrict__ merge_pos, uint32_t c, uint32_t n,
                             uint32_t next_node, const uint32_t *__restrict__ cl_node, const Box6 *__restrict__ cl_box,
                             uint32_t *__restrict__ out_node, Box6 *__restrict__ out_box, uint32_t *__restrict__ left,
                             uint32_t *__restrict__ right, Box6 *__restrict__ node_box, uint32_t *__restrict__ size,
                             uint32_t *__restrict__ parent)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= c) return;
    ploc_apply_one(i, nn, keep, merge, keep_pos, merge_pos, n, next_node, cl_node, cl_box, out_node, out_box, left, right, node_box, size, parent);
}

// The tail of the clustering: once PLOC_TAIL or fewer clusters are left, ONE workgroup runs all the
// remaining rounds (nearest neighbour, mutual-pair flags, the two prefix sums, merge) back to back with
// barriers in between -- the same arithmetic and the same node numbering as the multi-kernel rounds,
// without ~40 rounds of launches and host round trips for a handful of clusters each.
constexpr uint32_t PLOC_TAIL = 4096, TAIL_BLOCK = 1024;
__global__ void __launch_bounds__(TAIL_BLOCK) k_ploc_tail(PlocArrays a, uint32_t c, uint32_t n, uint32_t next_node, int cur, uint32_t *__restrict__ result)
{
    __shared__ uint32_t part_keep[TAIL_BLOCK], part_merge[TAIL_BLOCK];
    constexpr uint32_t PER = PLOC_TAIL / TAIL_BLOCK;
    const uint32_t t = threadIdx.x;
    while (c > 1) {
        const Box6 *box = a.cl_box[cur];
        for (uint32_t i = t; i < c; i += TAIL_BLOCK) {               // nearest neighbour (k_ploc_nn)
            const Box6 me = box[i];
            float best = __uint_as_float(0x7f800000u);
            uint32_t arg = i;
            for (int d = -PLOC_RADIUS; d <= PLOC_RADIUS; d++) {
                const int j = (int)i + d;
                if (d == 0 || j < 0 || j >= (int)c) continue;
                const float ar = merged_area(me, box[j]);
                if (ar < best) { best = ar; arg = (uint32_t)j; }
            }
            a.nn[i] = arg;
        }
        __syncthreads();
        uint32_t k_sum = 0, m_sum = 0;
        for (uint32_t e = 0; e < PER; e++) {                          // flags (k_ploc_flags), PER consecutive clusters per thread
            const uint32_t i = t * PER + e;
            if (i >= c) break;
            const uint32_t j = a.nn[i];
            const bool mutual = j != i && a.nn[j] == i;
            const uint32_t mg = (mutual && i < j) ? 1u : 0u, kp = (mutual && i > j) ? 0u : 1u;
            a.merge[i] = mg;
            a.keep[i] = kp;
            k_sum += kp;
            m_sum += mg;
        }
        part_keep[t] = k_sum;
        part_merge[t] = m_sum;
        __syncthreads();
        for (uint32_t off = 1; off < TAIL_BLOCK; off <<= 1) {         // inclusive scan of the per-thread sums
            const uint32_t pk = t >= off ? part_keep[t - off] : 0u, pm = t >= off ? part_merge[t - off] : 0u;
            __syncthreads();
            part_keep[t] += pk;
            part_merge[t] += pm;
            __syncthreads();
        }
        uint32_t k_run = part_keep[t] - k_sum, m_run = part_merge[t] - m_sum;      // exclusive prefix of this thread's run
        for (uint32_t e = 0; e < PER; e++) {
            const uint32_t i = t * PER + e;
            if (i >= c) break;
            a.keep_pos[i] = k_run;
            a.merge_pos[i] = m_run;
            k_run += a.keep[i];
            m_run += a.merge[i];
        }
        const uint32_t kept = part_keep[TAIL_BLOCK - 1], merged = part_merge[TAIL_BLOCK - 1];
        __syncthreads();
        if (merged == 0 || kept != c - merged) break;                 // no progress: reported by the host
        for (uint32_t i = t; i < c; i += TAIL_BLOCK)
            ploc_apply_one(i, a.nn, a.keep, a.merge, a.keep_pos, a.merge_pos, n, next_node, a.cl_node[cur], a.cl_box[cur], a.cl_node[cur ^ 1],
                           a.cl_box[cur ^ 1], a.left, a.right, a.node_box, a.size, a.parent);
        __syncthreads();
        c = kept;
        next_node += merged;
        cur ^= 1;
    }
    if (t == 0) { result[0] = c; result[1] = next_node; }
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

__global__ void k_ploc_tris(const uint64_t *__restrict__ keys, const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx,
                            const uint32_t *__restrict__ offset, uint32_t n, TriRec *__restrict__ tris)
{
    const uint32_t k = blockIdx.x * PB + threadIdx.x;
    if (k >= n) return;
    const uint32_t prim = (uint32_t)(keys[k] & 0xFFFFFFFFull);
    const rt_float3 p0 = verts[idx[3 * prim + 0]].position;
    const rt_float3 p1 = verts[idx[3 * prim + 1]].position;
    const rt_float3 p2 = verts[idx[3 * prim + 2]].position;
    TriRec t;
    t.a = make_float4(p0.x, p0.y, p0.z, p1.x);
    t.b = make_float4(p1.y, p1.z, p2.x, p2.y);
    t.c = make_float4(p2.z, __uint_as_float(prim), 0.0f, 0.0f);
    tris[offset[k]] = t;
}

inline unsigned gr(size_t n) { return (unsigned)((n + PB - 1) / PB); }

}  // namespace

// bytes of temporaries rt_build_ploc_layout slices out of the context's build arena (an upper estimate: the
// scan scratch is small next to the rest)
size_t rt_ploc_temp_bytes(uint32_t n)
{
    const size_t nn2 = 2 * (size_t)n;
    return 8 * (size_t)n + 2 * sizeof(Box6) * (size_t)n + 20 * ((size_t)n + 1) + 8 * (size_t)n + sizeof(Box6) * nn2 + 12 * nn2 + ((size_t)1 << 20) + 18 * 256;
}

// Rebuilds m->tris and m->blas.wide / root_code / fast_depth from a PLOC tree.  The canonical arrays
// (nodes, keys, parents) are left untouched.  Returns RT_OK with *done = false for meshes too small to bother.
int rt_build_ploc_layout(rt_context *ctx, rt_model *m, bool *done)
{
    const uint32_t n = m->n_tris;
    *done = false;
    if (n < 2 * ctx->leaf_max + 2) return RT_OK;          // tiny meshes: the LBVH layout is as good as any
    hipStream_t st = ctx->stream;
    // every temporary of the build is carved out of ONE allocation (hipMalloc / hipFree synchronise the
    // device and cost more than the kernels of a small build)
    View cl_node[2], cl_box[2], nn, keep, merge, keep_pos, merge_pos, left, right, node_box, size, parent, offset, scan_tmp, depth;
    int rc = RT_OK;
    do {
        const size_t nn2 = 2 * (size_t)n - 1;
        size_t tmp_bytes = 0;
        if (rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st) != hipSuccess) {
            rt_set_error("rocprim::exclusive_scan sizing failed");
            rc = RT_ERR_HIP;
            break;
        }
        struct Want { View *v; size_t bytes; };
        const Want wants[] = {{&cl_node[0], 4 * (size_t)n}, {&cl_node[1], 4 * (size_t)n}, {&cl_box[0], sizeof(Box6) * (size_t)n}, {&cl_box[1], sizeof(Box6) * (size_t)n},
                              {&nn, 4 * ((size_t)n + 1)}, {&keep, 4 * ((size_t)n + 1)}, {&merge, 4 * ((size_t)n + 1)}, {&keep_pos, 4 * ((size_t)n + 1)},
                              {&merge_pos, 4 * ((size_t)n + 1)}, {&left, 4 * (size_t)(n - 1)}, {&right, 4 * (size_t)(n - 1)}, {&node_box, sizeof(Box6) * nn2},
                              {&size, 4 * nn2}, {&parent, 4 * nn2}, {&offset, 4 * nn2}, {&scan_tmp, tmp_bytes}, {&depth, 8}};
        size_t total = 0;
        for (const Want &w : wants) total += (w.bytes + 255) & ~(size_t)255;
        if ((rc = ctx->build_arena.reserve(total)) != RT_OK) break;      // normally already there (rt_ploc_temp_bytes)
        size_t at = 0;
        for (const Want &w : wants) { w.v->p = (char *)ctx->build_arena.p + at; at += (w.bytes + 255) & ~(size_t)255; }

        k_ploc_init<<<gr(n), PB, 0, st>>>(m->blas.nodes.as<rt_bvh_node>(), n, cl_node[0].as<uint32_t>(), cl_box[0].as<Box6>(),
                                         size.as<uint32_t>(), parent.as<uint32_t>());
        k_ploc_leaf_boxes<<<gr(n), PB, 0, st>>>(cl_box[0].as<Box6>(), n, node_box.as<Box6>());
        uint32_t c = n, next_node = n;
        int cur = 0;
        for (int round = 0; c > PLOC_TAIL && round < 4096; round++) {
            k_ploc_nn<<<gr(c), PB, 0, st>>>(cl_box[cur].as<Box6>(), c, nn.as<uint32_t>());
            k_ploc_flags<<<gr(c), PB, 0, st>>>(nn.as<uint32_t>(), c, keep.as<uint32_t>(), merge.as<uint32_t>());
            // scans run over c+1 elements so that element c holds the totals (its input flag is garbage-free: set to 0)
            (void)hipMemsetAsync(keep.as<uint32_t>() + c, 0, 4, st);
            (void)hipMemsetAsync(merge.as<uint32_t>() + c, 0, 4, st);
            size_t tb = tmp_bytes;
            (void)rocprim::exclusive_scan(scan_tmp.p, tb, keep.as<uint32_t>(), keep_pos.as<uint32_t>(), 0u, (size_t)c + 1, rocprim::plus<uint32_t>(), st);
            tb = tmp_bytes;
            (void)rocprim::exclusive_scan(scan_tmp.p, tb, merge.as<uint32_t>(), merge_pos.as<uint32_t>(), 0u, (size_t)c + 1, rocprim::plus<uint32_t>(), st);
            k_ploc_apply<<<gr(c), PB, 0, st>>>(nn.as<uint32_t>(), keep.as<uint32_t>(), merge.as<uint32_t>(), keep_pos.as<uint32_t>(),
                                              merge_pos.as<uint32_t>(), c, n, next_node, cl_node[cur].as<uint32_t>(), cl_box[cur].as<Box6>(),
                                              cl_node[cur ^ 1].as<uint32_t>(), cl_box[cur ^ 1].as<Box6>(), left.as<uint32_t>(),
                                              right.as<uint32_t>(), node_box.as<Box6>(), size.as<uint32_t>(), parent.as<uint32_t>());
            uint32_t stack_totals[2];
            uint32_t *totals = ctx->pinned ? ctx->pinned : stack_totals;      // page-locked: no staging copy
            if (hipMemcpyAsync(&totals[0], keep_pos.as<uint32_t>() + c, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&totals[1], merge_pos.as<uint32_t>() + c, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) {
                rt_set_error("PLOC round %d failed: %s", round, hipGetErrorString(hipGetLastError()));
                rc = RT_ERR_HIP;
                break;
            }
            if (totals[1] == 0 || totals[0] != c - totals[1]) {
                rt_set_error("PLOC round %d made no progress (%u clusters, %u merges)", round, c, totals[1]);
                rc = RT_ERR_STATE;
                break;
            }
            c = totals[0];
            next_node += totals[1];
            cur ^= 1;
        }
        if (rc != RT_OK) break;
        if (c > 1) {
            PlocArrays pa;
            pa.nn = nn.as<uint32_t>(); pa.keep = keep.as<uint32_t>(); pa.merge = merge.as<uint32_t>();
            pa.keep_pos = keep_pos.as<uint32_t>(); pa.merge_pos = merge_pos.as<uint32_t>();
            for (int k = 0; k < 2; k++) { pa.cl_node[k] = cl_node[k].as<uint32_t>(); pa.cl_box[k] = cl_box[k].as<Box6>(); }
            pa.left = left.as<uint32_t>(); pa.right = right.as<uint32_t>(); pa.size = size.as<uint32_t>(); pa.parent = parent.as<uint32_t>();
            pa.node_box = node_box.as<Box6>();
            uint32_t *d_res = depth.as<uint32_t>();              // (two words: reserved below)
            k_ploc_tail<<<1, TAIL_BLOCK, 0, st>>>(pa, c, n, next_node, cur, d_res);
            uint32_t res[2] = {0, 0};
            if (hipMemcpyAsync(res, d_res, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rt_set_error("PLOC tail failed: %s", hipGetErrorString(hipGetLastError()));
                rc = RT_ERR_HIP;
                break;
            }
            c = res[0];
            next_node = res[1];
        }
        if (c != 1 || next_node != 2 * n - 1) {
            rt_set_error("PLOC did not converge (%u clusters, %u nodes)", c, next_node);
            rc = RT_ERR_STATE;
            break;
        }
        k_ploc_offsets<<<gr(nn2), PB, 0, st>>>(left.as<uint32_t>(), right.as<uint32_t>(), size.as<uint32_t>(), parent.as<uint32_t>(), n,
                                              offset.as<uint32_t>());
        k_ploc_tris<<<gr(n), PB, 0, st>>>(m->blas.keys.as<uint64_t>(), m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(),
                                         offset.as<uint32_t>(), n, m->tris.as<TriRec>());
        if (hipGetLastError() != hipSuccess) { rt_set_error("PLOC layout kernels failed"); rc = RT_ERR_HIP; break; }
        // four-wide nodes from the binary tree (root = the last node created); the cluster arrays of the rounds are free now
        // and serve as its scratch
        if ((rc = rt_build_wide_layout(ctx, m->blas, n, 2 * n - 2, left.as<uint32_t>(), right.as<uint32_t>(), (const float *)node_box.p,
                                       size.as<uint32_t>(), offset.as<uint32_t>(), nullptr, ctx->leaf_max, cl_node[0].p,
                                       (size_t)((char *)left.p - (char *)cl_node[0].p))) != RT_OK) break;
        *done = true;
    } while (0);
    return rc;
}
