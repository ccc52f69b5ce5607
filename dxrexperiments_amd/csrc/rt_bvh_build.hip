// rt_bvh_build.hip -- BLAS / TLAS construction on the GPU.
//
// Stands in for what the reference enqueues through the Fallback Layer:
//   RtModel::build  -> BottomLevelASGenerator::Generate -> BuildRaytracingAccelerationStructure
//       (libs/DXRFramework/RtModel.cpp:86-118, Helpers/BottomLevelASGenerator.cpp:279-343)
//   RtScene::build  -> TopLevelASGenerator::Generate    -> BuildRaytracingAccelerationStructure
//       (libs/DXRFramework/RtScene.cpp:18-52, Helpers/TopLevelASGenerator.cpp:309-414)
// The builder itself (source absent from the reference checkout) is this engine's
// own: an LBVH whose every integer step is fully determined, so the CPU oracle and
// these kernels produce identical node arrays:
//   1. primitive AABBs + structure AABB           (exact float min/max)
//   2. key = morton30(centre of AABB) << 32 | primitive index   (unique keys)
//   3. ascending radix sort of the 62-bit keys
//   4. Karras 2012 binary radix tree over the sorted keys (integer only)
//   5. bottom-up AABB union (exact float min/max; order independent)
//   6. traversal layout (rt_bvh_ploc.hip, rt_bvh_wide.hip): a better binary tree over the same leaves,
//      collapsed into four-wide 64-B nodes; subtrees of <= leaf_max primitives become leaves.
#include "rt_internal.h"

#include <chrono>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "rt_refs.h"

namespace {

struct Box6 { float lo[3]; float hi[3]; };

// order-preserving float <-> uint map for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f_enc(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f_dec(uint32_t k)
{
    uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}
#define ENC_POS_INF 0xFF800000u
#define ENC_NEG_INF 0x007FFFFFu

__device__ __forceinline__ float wave_min(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void k_init_bounds(uint32_t *enc)
{
    if (threadIdx.x < 3) enc[threadIdx.x] = ENC_POS_INF;
    else if (threadIdx.x < 6) enc[threadIdx.x] = ENC_NEG_INF;
}

// Structure bounds: wave reduction, then the block's waves meet in LDS and ONE wave issues the six atomics (same-address
// returning atomics serialise at ~11 ns each: six per WAVE cost 0.28 ms on 262 k triangles, six per 1024-thread block 0.02).
constexpr unsigned BOUNDS_BLOCK = 1024;
__device__ __forceinline__ void reduce_bounds(const Box6 &b, bool valid, uint32_t *enc)
{
    __shared__ float part[BOUNDS_BLOCK / 64][6];
    const float inf = __uint_as_float(0x7f800000u);
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, n_waves = (blockDim.x + 63u) >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = wave_min(valid ? b.lo[c] : inf);
        const float hi = wave_max(valid ? b.hi[c] : -inf);
        if (lane == 0) { part[wave][c] = lo; part[wave][3 + c] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const bool is_lo = threadIdx.x < 3;
        float v = part[0][threadIdx.x];
        for (unsigned w = 1; w < n_waves; w++) v = is_lo ? fminf(v, part[w][threadIdx.x]) : fmaxf(v, part[w][threadIdx.x]);
        if (is_lo) atomicMin(&enc[threadIdx.x], f_enc(v));
        else atomicMax(&enc[threadIdx.x], f_enc(v));
    }
}

__global__ void k_tri_boxes(const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t n,
                            Box6 *__restrict__ boxes, uint32_t *enc)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Box6 b;
    bool valid = i < n;
    if (valid) {
        rt_float3 p0 = verts[idx[3 * i + 0]].position;
        rt_float3 p1 = verts[idx[3 * i + 1]].position;
        rt_float3 p2 = verts[idx[3 * i + 2]].position;
        b.lo[0] = fminf(fminf(p0.x, p1.x), p2.x); b.hi[0] = fmaxf(fmaxf(p0.x, p1.x), p2.x);
        b.lo[1] = fminf(fminf(p0.y, p1.y), p2.y); b.hi[1] = fmaxf(fmaxf(p0.y, p1.y), p2.y);
        b.lo[2] = fminf(fminf(p0.z, p1.z), p2.z); b.hi[2] = fmaxf(fmaxf(p0.z, p1.z), p2.z);
        boxes[i] = b;
    }
    reduce_bounds(b, valid, enc);
}

__global__ void k_box_bounds(const Box6 *__restrict__ boxes, uint32_t n, uint32_t *enc)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Box6 b;
    bool valid = i < n;
    if (valid) b = boxes[i];
    reduce_bounds(b, valid, enc);
}

__global__ void k_decode_bounds(const uint32_t *enc, float *out)
{
    if (threadIdx.x < 6) out[threadIdx.x] = f_dec(enc[threadIdx.x]);
}

__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ uint32_t quant10(float c, float lo, float ext)
{
    if (!(ext > 0.0f)) return 0;
    float q = (c - lo) / ext * 1024.0f;
    q = fminf(fmaxf(q, 0.0f), 1023.0f);
    return (uint32_t)q;
}

__global__ void k_morton(const Box6 *__restrict__ boxes, uint32_t n, const float *__restrict__ bounds,
                         uint64_t *__restrict__ keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Box6 b = boxes[i];
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float ctr = (b.lo[c] + b.hi[c]) * 0.5f;
        float ext = bounds[3 + c] - bounds[c];
        m |= expand10(quant10(ctr, bounds[c], ext)) << (2 - c);
    }
    keys[i] = ((uint64_t)m << 32) | i;
}

__global__ void k_leaves(const uint64_t *__restrict__ keys, const Box6 *__restrict__ boxes, uint32_t n,
                         rt_bvh_node *__restrict__ nodes, uint32_t *__restrict__ parents)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t prim = (uint32_t)(keys[k] & 0xFFFFFFFFull);
    Box6 b = boxes[prim];
    rt_bvh_node nd;
    nd.bmin[0] = b.lo[0]; nd.bmin[1] = b.lo[1]; nd.bmin[2] = b.lo[2];
    nd.bmax[0] = b.hi[0]; nd.bmax[1] = b.hi[1]; nd.bmax[2] = b.hi[2];
    nd.left = prim;
    nd.right = RT_LEAF;
    nodes[n - 1 + k] = nd;
    if (n == 1) parents[0] = 0xFFFFFFFFu;
}

__device__ __forceinline__ int key_delta(const uint64_t *__restrict__ k, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(k[i] ^ k[j]));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", HPG 2012.
__global__ void k_karras(const uint64_t *__restrict__ keys, int n, rt_bvh_node *__restrict__ nodes,
                         uint32_t *__restrict__ parents, uint2 *__restrict__ ranges)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (key_delta(keys, n, i, i + 1) - key_delta(keys, n, i, i - 1)) > 0 ? 1 : -1;
    int dmin = key_delta(keys, n, i, i - d);
    int lmax = 2;
    while (key_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (key_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = key_delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (key_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int first = min(i, j), last = max(i, j);
    uint32_t leaf0 = (uint32_t)(n - 1);
    uint32_t left = (first == gamma) ? leaf0 + (uint32_t)gamma : (uint32_t)gamma;
    uint32_t right = (last == gamma + 1) ? leaf0 + (uint32_t)(gamma + 1) : (uint32_t)(gamma + 1);
    nodes[i].left = left;
    nodes[i].right = right;
    parents[left] = (uint32_t)i;
    parents[right] = (uint32_t)i;
    if (i == 0) parents[0] = 0xFFFFFFFFu;
    ranges[i] = make_uint2((uint32_t)first, (uint32_t)last);
}

// Bottom-up union, every internal node visited ONCE (Karras 2012): a thread starts at its leaf and climbs; at a parent the
// first arriver stops, the second unites the two child boxes and goes on.  The per-XCD L2s of this chip are not coherent
// with each other, so everything two threads exchange is stored and loaded at AGENT scope (write-through / read-through
// past the local L2): a thread publishes its node's box and depth with four 64-bit stores, waits until they are
// acknowledged, only then bumps the parent's arrival counter (a device-scope atomic, executed at the memory side), and the
// second arriver -- ordered behind the first by that counter -- reads the sibling with four agent-scope loads.  One
// read-modify-write per node; the first round-2 version exchanged every word with one (14 per node, 0.15 ms for 262 k
// triangles), round 1's min/max climb issued ~180 per leaf.  min / max are exact and order independent, so the boxes are
// those of the oracle bit for bit.
// scratch per node: 6 box words + depth + arrival counter = 8 words (four 64-bit pairs), zero-initialised.
__global__ void k_refit(const rt_bvh_node *__restrict__ nodes, const uint32_t *__restrict__ parents, uint32_t n,
                        uint32_t *__restrict__ scratch, uint32_t *__restrict__ max_depth)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const rt_bvh_node nd = nodes[n - 1 + k];
    float lo[3] = {nd.bmin[0], nd.bmin[1], nd.bmin[2]}, hi[3] = {nd.bmax[0], nd.bmax[1], nd.bmax[2]};
    uint32_t depth = 0, cur = n - 1 + k;
    auto pack = [](uint32_t a, uint32_t b) { return (uint64_t)a | ((uint64_t)b << 32); };
    for (;;) {
        const uint32_t parent = parents[cur];
        if (parent == 0xFFFFFFFFu) { atomicMax(max_depth, depth); return; }       // cur is the root
        // publish this node (leaves too: the sibling's climber reads them the same way), then arrive
        uint64_t *mine = reinterpret_cast<uint64_t *>(scratch + (size_t)cur * 8);
        __hip_atomic_store(&mine[0], pack(__float_as_uint(lo[0]), __float_as_uint(lo[1])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine[1], pack(__float_as_uint(lo[2]), __float_as_uint(hi[0])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine[2], pack(__float_as_uint(hi[1]), __float_as_uint(hi[2])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (word 7 of a node is its arrival counter: the depth is published as a 32-bit store so that it is left alone)
        __hip_atomic_store(scratch + (size_t)cur * 8 + 6, depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every store must have been ACKNOWLEDGED before the arrival is counted; inline asm, because the compiler may drop a
        // wait it believes redundant (MI355X_MICROARCH.md, "Compiler hazard")
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (a relaxed memory-side atomic: the ordering it needs is made by hand -- stores acknowledged above, sc1 loads below --
        // and the two signal fences keep the COMPILER from moving any of them across it: ADVICE r2)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const uint32_t arrived = __hip_atomic_fetch_add(&scratch[(size_t)parent * 8 + 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (arrived == 0u) return;                       // the sibling's climber will take over from here
        const rt_bvh_node pn = nodes[parent];
        const uint32_t sib = pn.left == cur ? pn.right : pn.left;
        const uint64_t *other = reinterpret_cast<const uint64_t *>(scratch + (size_t)sib * 8);
        const uint64_t o0 = __hip_atomic_load(&other[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t o1 = __hip_atomic_load(&other[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t o2 = __hip_atomic_load(&other[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t od = __hip_atomic_load(scratch + (size_t)sib * 8 + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo[0] = fminf(lo[0], __uint_as_float((uint32_t)o0));
        lo[1] = fminf(lo[1], __uint_as_float((uint32_t)(o0 >> 32)));
        lo[2] = fminf(lo[2], __uint_as_float((uint32_t)o1));
        hi[0] = fmaxf(hi[0], __uint_as_float((uint32_t)(o1 >> 32)));
        hi[1] = fmaxf(hi[1], __uint_as_float((uint32_t)o2));
        hi[2] = fmaxf(hi[2], __uint_as_float((uint32_t)(o2 >> 32)));
        depth = (depth > od ? depth : od) + 1u;
        cur = parent;
    }
}

// the boxes the climbers left in the scratch -> the internal nodes (the root's, which nobody published, is the union of
// its children's)
__global__ void k_refit_decode(const uint32_t *__restrict__ scratch, rt_bvh_node *__restrict__ nodes, uint32_t n_internal)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    if (i == 0) {
        const uint32_t *a = scratch + (size_t)nodes[0].left * 8, *b = scratch + (size_t)nodes[0].right * 8;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            nodes[0].bmin[c] = fminf(__uint_as_float(a[c]), __uint_as_float(b[c]));
            nodes[0].bmax[c] = fmaxf(__uint_as_float(a[3 + c]), __uint_as_float(b[3 + c]));
        }
        return;
    }
    const uint32_t *p = scratch + (size_t)i * 8;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        nodes[i].bmin[c] = __uint_as_float(p[c]);
        nodes[i].bmax[c] = __uint_as_float(p[3 + c]);
    }
}

__global__ void k_gather_tris(const uint64_t *__restrict__ keys, const rt_vertex *__restrict__ verts,
                              const uint32_t *__restrict__ idx, uint32_t n, TriRec *__restrict__ tris)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t prim = (uint32_t)(keys[k] & 0xFFFFFFFFull);
    rt_float3 p0 = verts[idx[3 * prim + 0]].position;
    rt_float3 p1 = verts[idx[3 * prim + 1]].position;
    rt_float3 p2 = verts[idx[3 * prim + 2]].position;
    TriRec t;
    t.a = make_float4(p0.x, p0.y, p0.z, p1.x);
    t.b = make_float4(p1.y, p1.z, p2.x, p2.y);
    t.c = make_float4(p2.z, __uint_as_float(prim), 0.0f, 0.0f);
    tris[k] = t;
}

// the vertex normals of every primitive, gathered once per build into one record per primitive (shading reads them per hit)
__global__ void k_normal_records(const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t n, TriRec *__restrict__ out)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const rt_float3 n0 = verts[idx[3 * p + 0]].normal;
    const rt_float3 n1 = verts[idx[3 * p + 1]].normal;
    const rt_float3 n2 = verts[idx[3 * p + 2]].normal;
    TriRec t;
    t.a = make_float4(n0.x, n0.y, n0.z, n1.x);
    t.b = make_float4(n1.y, n1.z, n2.x, n2.y);
    t.c = make_float4(n2.z, 0.0f, 0.0f, 0.0f);
    out[p] = t;
}

inline unsigned grid_for(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// 64-bit sum of n 32-bit counts (grid-stride; one atomic per wave)
__global__ void k_sum64(const uint32_t *__restrict__ v, uint32_t n, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += v[i];
    for (int o = 32; o > 0; o >>= 1) s += (unsigned long long)__shfl_xor((long long)s, o, 64);
    if ((threadIdx.x & 63u) == 0u && s) atomicAdd(out, s);
}

// ---- split references (rt_refs.h): count the pieces of every triangle, then (after a scan) write their boxes ----
__device__ __forceinline__ void ref_load_tri(const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t i, float p[3][3])
{
    for (int k = 0; k < 3; k++) {
        const rt_float3 v = verts[idx[3 * i + k]].position;
        p[k][0] = v.x; p[k][1] = v.y; p[k][2] = v.z;
    }
}
__device__ __forceinline__ float ref_min_len(const float *__restrict__ bounds)
{
    const float ex = bounds[3] - bounds[0], ey = bounds[4] - bounds[1], ez = bounds[5] - bounds[2];
    return rtd::ref_max2(rtd::ref_max2(ex, ey), ez) * 0.001953125f;          // the model's longest extent / 512
}
__global__ void k_ref_count(const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t n, const float *__restrict__ bounds,
                            uint32_t *__restrict__ count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { count[n] = 0u; return; }           // (the scan's last element: the total)
    float p[3][3];
    ref_load_tri(verts, idx, i, p);
    int axis;
    count[i] = rtd::ref_pieces(p, ref_min_len(bounds), axis);
}
__global__ void k_ref_emit(const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ idx, uint32_t n, const float *__restrict__ bounds,
                           const uint32_t *__restrict__ off, float *__restrict__ ref_boxes, uint32_t *__restrict__ ref_prim)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p[3][3];
    ref_load_tri(verts, idx, i, p);
    const uint32_t first = off[i], k = off[i + 1] - first;
    if (k == 1u) {                                     // not split: its own box (what k_tri_boxes wrote: exact min / max)
        float *b = ref_boxes + 6 * (size_t)first;
        for (int q = 0; q < 3; q++) {
            b[q] = fminf(fminf(p[0][q], p[1][q]), p[2][q]);
            b[3 + q] = fmaxf(fmaxf(p[0][q], p[1][q]), p[2][q]);
        }
        ref_prim[first] = i;
        return;
    }
    int axis;
    (void)rtd::ref_pieces(p, ref_min_len(bounds), axis);
    for (uint32_t j = 0; j < k; j++) {
        float b[6];
        rtd::ref_box(p, axis, k, j, b);
        for (int q = 0; q < 6; q++) ref_boxes[6 * (size_t)(first + j) + q] = b[q];
        ref_prim[first + j] = i;
    }
}
// Morton keys of the references (the LBVH's codes, over the same bounds) and, after the sort, the leaves PLOC starts from
__global__ void k_ref_keys(const float *__restrict__ ref_boxes, uint32_t m, const float *__restrict__ bounds, uint64_t *__restrict__ keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float *b = ref_boxes + 6 * (size_t)i;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float ctr = (b[c] + b[3 + c]) * 0.5f;
        code |= expand10(quant10(ctr, bounds[c], bounds[3 + c] - bounds[c])) << (2 - c);
    }
    keys[i] = ((uint64_t)code << 32) | i;
}
__global__ void k_ref_leaves(const uint64_t *__restrict__ keys, uint32_t m, const float *__restrict__ ref_boxes, const uint32_t *__restrict__ ref_prim,
                             float *__restrict__ leaf_box, uint32_t *__restrict__ leaf_prim)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint32_t r = (uint32_t)(keys[k] & 0xFFFFFFFFull);
    for (int q = 0; q < 6; q++) leaf_box[6 * (size_t)k + q] = ref_boxes[6 * (size_t)r + q];
    leaf_prim[k] = ref_prim[r];
}
// a model with split triangles in a layout that holds every triangle once (LBVH layout): its records say "validate by primitive"
__global__ void k_ref_mark_records(TriRec *__restrict__ tris, uint32_t n, const uint32_t *__restrict__ off)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t prim = __float_as_uint(tris[k].c.y);
    if (off[prim + 1] - off[prim] > 1u) tris[k].c.z = __uint_as_float(2u);
}

// Temporaries of a build over n primitives as slices of the context's build arena (sized for the larger of
// the LBVH and the PLOC phase, which run one after the other).  A slice that turns out too small makes
// its DevBuf allocate on its own (DevBuf::reserve), so the sizes here are a fast path, not a contract.
struct BuildTemps { DevBuf boxes, enc, bounds, tkeys, tsort, tenc, tdepth, extra; };
int take_build_temps(rt_context *ctx, uint32_t n, BuildTemps &t, size_t extra_bytes = 0)
{
    const size_t A = 256;
    auto up = [&](size_t b) { return (b + A - 1) & ~(A - 1); };
    const size_t want[8] = {up(sizeof(Box6) * (size_t)n), A, A, up(8 * (size_t)n), up(16 * (size_t)n + (4u << 20)), up(64 * (size_t)n), A, up(extra_bytes)};
    size_t lbvh = 0;
    for (size_t w : want) lbvh += w;
    const size_t ploc = rt_ploc_temp_bytes(n), wide = rt_wide_lbvh_temp_bytes(n);
    size_t most = lbvh > ploc ? lbvh : ploc;
    most = most > wide ? most : wide;
    RT_TRY(ctx->build_arena.reserve(most));
    DevBuf *bufs[8] = {&t.boxes, &t.enc, &t.bounds, &t.tkeys, &t.tsort, &t.tenc, &t.tdepth, &t.extra};
    size_t at = 0;
    for (int k = 0; k < 8; k++) { bufs[k]->adopt((char *)ctx->build_arena.p + at, want[k]); at += want[k]; }
    return RT_OK;
}
void drop_build_arena_if_large(rt_context *ctx)
{
    if (ctx->build_arena.bytes > ((size_t)256 << 20)) ctx->build_arena.release();      // keep small arenas for the next build
}

// Steps 2..6 for a structure whose primitive boxes and bounds are already on the device.
int lbvh_from_boxes(rt_context *ctx, BvhDev &bv, const Box6 *boxes, uint32_t n, const float *d_bounds, bool tlas,
                    DevBuf &tmp_keys, DevBuf &tmp_sort, DevBuf &tmp_enc, DevBuf &tmp_depth)
{
    hipStream_t st = ctx->stream;
    const unsigned B = 256;
    // the traversal addresses a node as base + (index << RT_NODE_SHIFT) with a 32-bit byte offset (rt_trace_wave.h): 64-B nodes
    // leave room for 2^26 of them
    constexpr uint32_t kMaxPrims = 1u << (32 - RT_NODE_SHIFT);
    if (n > kMaxPrims) { rt_set_error("acceleration structure over %u primitives: the limit is 2^%d (%u)", n, 32 - RT_NODE_SHIFT, kMaxPrims); return RT_ERR_UNSUPPORTED; }
    bv.n = n;
    RT_TRY(bv.nodes.reserve(sizeof(rt_bvh_node) * (2 * (size_t)n - 1)));
    RT_TRY(bv.keys.reserve(sizeof(uint64_t) * n));
    RT_TRY(bv.parents.reserve(sizeof(uint32_t) * (2 * (size_t)n - 1)));
    RT_TRY(bv.ranges.reserve(sizeof(uint2) * (n > 1 ? n - 1 : 1)));
    RT_TRY(tmp_keys.reserve(sizeof(uint64_t) * n));
    RT_TRY(tmp_depth.reserve(sizeof(uint32_t)));

    k_morton<<<grid_for(n, B), B, 0, st>>>(boxes, n, d_bounds, tmp_keys.as<uint64_t>());
    // keys are (30-bit Morton code << 32) | index with the indices ascending on input: a STABLE sort of the code bits alone
    // gives the order of the full 64-bit keys (the oracle's) with half the radix passes
    size_t sort_bytes = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, sort_bytes, tmp_keys.as<uint64_t>(), bv.keys.as<uint64_t>(), n, 32, 62, st));
    RT_TRY(tmp_sort.reserve(sort_bytes));
    HIP_TRY(rocprim::radix_sort_keys(tmp_sort.p, sort_bytes, tmp_keys.as<uint64_t>(), bv.keys.as<uint64_t>(), n, 32, 62, st));

    rt_bvh_node *nodes = bv.nodes.as<rt_bvh_node>();
    uint32_t *parents = bv.parents.as<uint32_t>();
    k_leaves<<<grid_for(n, B), B, 0, st>>>(bv.keys.as<uint64_t>(), boxes, n, nodes, parents);
    HIP_TRY(hipMemsetAsync(tmp_depth.p, 0, sizeof(uint32_t), st));
    if (n > 1) {
        k_karras<<<grid_for(n - 1, B), B, 0, st>>>(bv.keys.as<uint64_t>(), (int)n, nodes, parents, bv.ranges.as<uint2>());
        RT_TRY(tmp_enc.reserve(sizeof(uint32_t) * 8 * (2 * (size_t)n - 1)));
        HIP_TRY(hipMemsetAsync(tmp_enc.p, 0, sizeof(uint32_t) * 8 * (2 * (size_t)n - 1), st));
        k_refit<<<grid_for(n, B), B, 0, st>>>(nodes, parents, n, tmp_enc.as<uint32_t>(), tmp_depth.as<uint32_t>());
        k_refit_decode<<<grid_for(n - 1, B), B, 0, st>>>(tmp_enc.as<uint32_t>(), nodes, n - 1);
    }
    HIP_TRY(hipGetLastError());
    // depth and bounds come back through page-locked memory and are picked up by lbvh_collect() once the caller has queued
    // the rest of the build: no host round trip in the middle of it
    uint32_t *back = ctx->pinned ? ctx->pinned + RT_PINNED_LBVH : nullptr;
    HIP_TRY(hipMemcpyAsync(back ? (void *)back : (void *)&bv.max_depth, tmp_depth.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(back ? (void *)(back + 1) : (void *)bv.bounds, d_bounds, 6 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (!back) HIP_TRY(hipStreamSynchronize(st));
    (void)tlas;
    return RT_OK;
}

// joins the stream and takes the depth and bounds lbvh_from_boxes queued for read-back
int lbvh_collect(rt_context *ctx, BvhDev &bv)
{
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->pinned) {
        bv.max_depth = ctx->pinned[RT_PINNED_LBVH];
        memcpy(bv.bounds, ctx->pinned + RT_PINNED_LBVH + 1, 6 * sizeof(float));
    }
    return RT_OK;
}

}  // namespace

int rt_build_blas(rt_context *ctx, rt_model *m)
{
    if (m->built) return RT_OK;
    hipStream_t st = ctx->stream;
    const unsigned B = 256;
    const uint32_t n = m->n_tris;
    BuildTemps bt;
    DevBuf &boxes = bt.boxes, &enc = bt.enc, &bounds = bt.bounds, &tkeys = bt.tkeys, &tsort = bt.tsort, &tenc = bt.tenc, &tdepth = bt.tdepth;
    int rc = RT_OK;
    const bool verbose = ctx->verbose;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {          // RT_VERBOSE: wall time of each build phase (synchronises: diagnostics only)
        if (!verbose) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[dxr_amd]   BLAS %-18s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    do {
        // (+ the reference count of every triangle, its scan and the scan's scratch: slices of the arena, no allocation of their own)
        const size_t ref_words = ((size_t)n + 1 + 63) & ~(size_t)63;
        if ((rc = take_build_temps(ctx, n, bt, 8 * ref_words + ((size_t)1 << 20))) != RT_OK) break;
        mark("arena");
        if ((rc = boxes.reserve(sizeof(Box6) * (size_t)n)) != RT_OK) break;
        if ((rc = enc.reserve(6 * sizeof(uint32_t))) != RT_OK) break;
        if ((rc = bounds.reserve(6 * sizeof(float))) != RT_OK) break;
        if ((rc = m->tris.reserve(sizeof(TriRec) * (size_t)n)) != RT_OK) break;
        if ((rc = m->normals.reserve(sizeof(TriRec) * (size_t)n)) != RT_OK) break;
        mark("tris alloc");
        k_init_bounds<<<1, 64, 0, st>>>(enc.as<uint32_t>());
        k_tri_boxes<<<grid_for(n, BOUNDS_BLOCK), BOUNDS_BLOCK, 0, st>>>(m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(), n, boxes.as<Box6>(),
                                                                        enc.as<uint32_t>());
        k_decode_bounds<<<1, 64, 0, st>>>(enc.as<uint32_t>(), bounds.as<float>());
        mark("alloc + boxes");
        if ((rc = lbvh_from_boxes(ctx, m->blas, boxes.as<Box6>(), n, bounds.as<float>(), false, tkeys, tsort, tenc, tdepth)) != RT_OK) break;
        mark("LBVH");
        k_gather_tris<<<grid_for(n, B), B, 0, st>>>(m->blas.keys.as<uint64_t>(), m->d_verts.as<rt_vertex>(),
                                                    m->d_idx.as<uint32_t>(), n, m->tris.as<TriRec>());
        k_normal_records<<<grid_for(n, B), B, 0, st>>>(m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(), n, m->normals.as<TriRec>());
        if (hipGetLastError() != hipSuccess) {
            rt_set_error("BLAS build kernels failed");
            rc = RT_ERR_HIP;
            break;
        }
        // Split references (rt_refs.h): how many boxes every triangle is validated against.  One each -- every scene of rounds 1 - 4 --
        // leaves the model as it was; otherwise the boxes by primitive (the canonical walk validates against them) and, below, the
        // references as the leaves the production tree is built from.
        m->n_recs = n;
        m->rec_boxes.release(); m->ref_off.release(); m->ref_boxes.release();
        DevBuf r_scan, r_prim, r_keys, r_sorted, r_sort_tmp, r_leaf_box, r_leaf_prim;
        uint32_t n_refs = n;
        uint32_t *d_count = bt.extra.as<uint32_t>(), *d_off = d_count + ref_words;
        {
            void *scan_tmp = (void *)(d_off + ref_words);
            const size_t scan_room = (size_t)1 << 20;
            k_ref_count<<<grid_for(n + 1, B), B, 0, st>>>(m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(), n, bounds.as<float>(), d_count);
            size_t scan_bytes = 0;
            if (rocprim::exclusive_scan(nullptr, scan_bytes, d_count, d_off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st) != hipSuccess) { rt_set_error("reference scan failed"); rc = RT_ERR_HIP; break; }
            if (scan_bytes > scan_room) { if ((rc = r_scan.reserve(scan_bytes)) != RT_OK) break; scan_tmp = r_scan.p; }
            if (rocprim::exclusive_scan(scan_tmp, scan_bytes, d_count, d_off, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), st) != hipSuccess) { rt_set_error("reference scan failed"); rc = RT_ERR_HIP; break; }
            uint32_t *back = ctx->pinned ? ctx->pinned : &n_refs;
            if (hipMemcpyAsync(back, d_off + n, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rt_set_error("reference count read-back failed");
                rc = RT_ERR_HIP;
                break;
            }
            n_refs = *back;
            r_scan.release();
            // up to RT_REF_MAX_PIECES references per triangle: beyond 2^32 / 128 triangles the 32-bit scan can wrap past a plausible
            // total, so such a mesh has its counts summed again in 64 bits
            if ((uint64_t)n * RT_REF_MAX_PIECES > 0xFFFFFFFFull) {
                DevBuf sum64;
                unsigned long long total64 = 0;
                if ((rc = sum64.reserve(8)) != RT_OK) break;
                if (hipMemsetAsync(sum64.p, 0, 8, st) != hipSuccess) { rt_set_error("reference count (64-bit) failed"); rc = RT_ERR_HIP; break; }
                k_sum64<<<1024, B, 0, st>>>(d_count, n, sum64.as<unsigned long long>());
                if (hipMemcpyAsync(&total64, sum64.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                    rt_set_error("reference count (64-bit) failed");
                    rc = RT_ERR_HIP;
                    break;
                }
                if (total64 != (unsigned long long)n_refs) { rt_set_error("split references: %llu for %u triangles", total64, n); rc = RT_ERR_UNSUPPORTED; break; }
            }
        }
        if (n_refs != n) {
            if (n_refs < n || n_refs > (1u << (32 - RT_NODE_SHIFT))) { rt_set_error("split references: %u for %u triangles", n_refs, n); rc = RT_ERR_UNSUPPORTED; break; }
            if ((rc = m->ref_off.reserve(4 * ((size_t)n + 1))) != RT_OK) break;
            if (hipMemcpyAsync(m->ref_off.p, d_off, 4 * ((size_t)n + 1), hipMemcpyDeviceToDevice, st) != hipSuccess) { rt_set_error("reference offsets copy failed"); rc = RT_ERR_HIP; break; }
            if ((rc = m->ref_boxes.reserve(24 * (size_t)n_refs)) != RT_OK || (rc = r_prim.reserve(4 * (size_t)n_refs)) != RT_OK) break;
            k_ref_emit<<<grid_for(n, B), B, 0, st>>>(m->d_verts.as<rt_vertex>(), m->d_idx.as<uint32_t>(), n, bounds.as<float>(), m->ref_off.as<uint32_t>(),
                                                    m->ref_boxes.as<float>(), r_prim.as<uint32_t>());
            if (ctx->verbose) fprintf(stderr, "[dxr_amd]   BLAS %u triangles -> %u references\n", n, n_refs);
        }
        const bool split_layout = n_refs > n && ctx->opt_split_refs && ctx->use_ploc;
        if (split_layout) {
            // the references in Morton order: the leaves of the production tree
            if ((rc = r_keys.reserve(8 * (size_t)n_refs)) != RT_OK || (rc = r_sorted.reserve(8 * (size_t)n_refs)) != RT_OK ||
                (rc = r_leaf_box.reserve(24 * (size_t)n_refs)) != RT_OK || (rc = r_leaf_prim.reserve(4 * (size_t)n_refs)) != RT_OK) break;
            k_ref_keys<<<grid_for(n_refs, B), B, 0, st>>>(m->ref_boxes.as<float>(), n_refs, bounds.as<float>(), r_keys.as<uint64_t>());
            size_t sort_bytes = 0;
            if (rocprim::radix_sort_keys(nullptr, sort_bytes, r_keys.as<uint64_t>(), r_sorted.as<uint64_t>(), n_refs, 32, 62, st) != hipSuccess ||
                (rc = r_sort_tmp.reserve(sort_bytes)) != RT_OK ||
                rocprim::radix_sort_keys(r_sort_tmp.p, sort_bytes, r_keys.as<uint64_t>(), r_sorted.as<uint64_t>(), n_refs, 32, 62, st) != hipSuccess) {
                if (rc == RT_OK) { rt_set_error("reference sort failed"); rc = RT_ERR_HIP; }
                break;
            }
            k_ref_leaves<<<grid_for(n_refs, B), B, 0, st>>>(r_sorted.as<uint64_t>(), n_refs, m->ref_boxes.as<float>(), r_prim.as<uint32_t>(),
                                                           r_leaf_box.as<float>(), r_leaf_prim.as<uint32_t>());
        }
        // production traversal layout: re-cluster the same leaves with PLOC (rt_bvh_ploc.hip), then collapse the binary
        // tree into wide quantised nodes (rt_bvh_wide.hip); tiny meshes and option fast_bvh=lbvh collapse the LBVH itself
        mark("gather");
        bool ploc_done = false;
        if (ctx->use_ploc) {
            rc = split_layout ? rt_build_ploc_layout(ctx, m, &ploc_done, n_refs, r_leaf_box.as<float>(), r_leaf_prim.as<uint32_t>())
                              : rt_build_ploc_layout(ctx, m, &ploc_done);
            // PLOC's nearest-neighbour rounds make no progress on boxes whose surface is not finite (NaN / inf vertices,
            // extents that overflow): such a mesh keeps the LBVH as its traversal layout (m->tris is still in LBVH order)
            if (rc == RT_ERR_STATE || (rc == RT_OK && ctx->opt_fail_ploc_rounds)) {      // (the option: tests force this path)
                rc = RT_OK;
                ploc_done = false;
                if (split_layout) {
                    // the split-reference path had already grown m->tris for one record per REFERENCE (DevBuf::reserve keeps no contents):
                    // back to one LBVH-ordered record per triangle, which is what the collapse below and k_ref_mark_records index
                    m->rec_boxes.release();
                    m->n_recs = n;
                    if ((rc = m->tris.reserve(sizeof(TriRec) * (size_t)n)) != RT_OK) break;
                    k_gather_tris<<<grid_for(n, B), B, 0, st>>>(m->blas.keys.as<uint64_t>(), m->d_verts.as<rt_vertex>(),
                                                                m->d_idx.as<uint32_t>(), n, m->tris.as<TriRec>());
                }
            }
            if (rc != RT_OK) break;
        }
        mark("PLOC + wide layout");
        if (!ploc_done && (rc = rt_build_wide_from_lbvh(ctx, m->blas, false, ctx->leaf_max)) != RT_OK) break;
        if (n_refs > n && !(ploc_done && split_layout)) {
            // a layout with one record per triangle (LBVH; PLOC with the option split_refs=0): split triangles are validated by primitive
            m->n_recs = n;
            k_ref_mark_records<<<grid_for(n, B), B, 0, st>>>(m->tris.as<TriRec>(), n, m->ref_off.as<uint32_t>());
        }
        r_prim.release(); r_keys.release(); r_sorted.release(); r_sort_tmp.release(); r_leaf_box.release(); r_leaf_prim.release();
        mark("wide layout (LBVH)");
        if ((rc = lbvh_collect(ctx, m->blas)) != RT_OK) break;
        m->built = true;
    } while (0);
    boxes.release(); enc.release(); bounds.release(); tkeys.release(); tsort.release(); tenc.release(); tdepth.release();
    drop_build_arena_if_large(ctx);
    mark("free");
    return rc;
}

// world-to-object = inverse of the affine 3x4 (adjugate / determinant in fp32,
// operation order fixed: see DESIGN.md "Instances")
static void invert3x4(const float m[12], float o[12])
{
    float a = m[0], b = m[1], c = m[2];
    float d = m[4], e = m[5], f = m[6];
    float g = m[8], h = m[9], i = m[10];
    float A = e * i - f * h;
    float B = f * g - d * i;
    float C = d * h - e * g;
    float det = a * A;
    det = det + b * B;
    det = det + c * C;
    float id = 1.0f / det;
    o[0] = A * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[4] = B * id; o[5] = (a * i - c * g) * id; o[6] = (c * d - a * f) * id;
    o[8] = C * id; o[9] = (b * g - a * h) * id; o[10] = (a * e - b * d) * id;
    for (int r = 0; r < 3; r++) {
        float s = o[4 * r + 0] * m[3];
        s = s + o[4 * r + 1] * m[7];
        s = s + o[4 * r + 2] * m[11];
        o[4 * r + 3] = -s;
    }
}

// The world box of a transformed instance: the exact box of its triangles' transformed vertices (oracle_bvh.h
// scene_build; the box of the BLAS box's eight transformed corners is up to 1.6x wider in footprint for a rotated mesh, and a third
// of all instance entries of the 4096-instance frame were false ones, profiles/r04/tight_boxes.txt).  One block per work item =
// (instance, chunk of INST_BOX_REFS vertex references); x' = ((m0 x + m1 y) + m2 z) + m3 as the oracle's xform_point; min / max
// are order free, so the atomics cannot change a bit.
constexpr unsigned INST_BOX_REFS = 4096;
__global__ void __launch_bounds__(BOUNDS_BLOCK) k_instance_boxes(const InstanceRec *__restrict__ inst, const float *__restrict__ xf,
                                                                const uint2 *__restrict__ items, uint32_t *__restrict__ enc)
{
    const uint2 it = items[blockIdx.x];
    const InstanceRec &in = inst[it.x];
    const float *m = xf + 12u * (size_t)it.x;
    const uint32_t refs = 3u * in.n_prims;
    const float inf = __uint_as_float(0x7f800000u);
    Box6 b;
    for (int c = 0; c < 3; c++) { b.lo[c] = inf; b.hi[c] = -inf; }
    for (uint32_t k = it.y * INST_BOX_REFS + threadIdx.x; k < refs && k < (it.y + 1u) * INST_BOX_REFS; k += BOUNDS_BLOCK) {
        const rt_float3 p = in.verts[in.indices[k]].position;
        for (int r = 0; r < 3; r++) {
            float w = m[4 * r + 0] * p.x;
            w = w + m[4 * r + 1] * p.y;
            w = w + m[4 * r + 2] * p.z;
            w = w + m[4 * r + 3];
            b.lo[r] = fminf(b.lo[r], w);
            b.hi[r] = fmaxf(b.hi[r], w);
        }
    }
    reduce_bounds(b, true, enc + 6u * (size_t)it.x);
}
__global__ void k_instance_boxes_init(uint32_t *enc, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6u * n) enc[i] = (i % 6u) < 3u ? ENC_POS_INF : ENC_NEG_INF;
}
// ... into the instance records and the TLAS builder's leaf boxes (identity instances keep the box of their BLAS)
__global__ void k_instance_boxes_finish(InstanceRec *__restrict__ inst, const uint32_t *__restrict__ enc, Box6 *__restrict__ boxes, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    InstanceRec &r = inst[i];
    if (!(r.flags & RT_INST_IDENTITY))
        for (int c = 0; c < 3; c++) { r.wlo[c] = f_dec(enc[6u * i + c]); r.whi[c] = f_dec(enc[6u * i + 3 + c]); }
    for (int c = 0; c < 3; c++) { boxes[i].lo[c] = r.wlo[c]; boxes[i].hi[c] = r.whi[c]; }
}

int rt_build_tlas(rt_context *ctx, rt_scene *s)
{
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)s->inst.size();
    static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    s->h_inst.assign(n, InstanceRec());
    std::vector<Box6> hb(n);
    std::vector<uint2> items;          // (instance, chunk of its vertex references): the work list of k_instance_boxes
    std::vector<float> xf;             // the forward transforms, for the same kernel (alive until the build's last synchronisation)
    uint32_t deepest = 0;
    for (uint32_t i = 0; i < n; i++) {
        rt_model *m = s->inst[i].model;
        InstanceRec &r = s->h_inst[i];
        const float *x = s->inst[i].xform;
        bool identity = true;
        for (int k = 0; k < 12; k++) identity = identity && (x[k] == ident[k]);
        const float *bb = m->blas.bounds;
        if (identity) {
            memcpy(r.inv, x, sizeof r.inv);
            for (int c = 0; c < 3; c++) { r.wlo[c] = bb[c]; r.whi[c] = bb[3 + c]; }
        } else {
            invert3x4(x, r.inv);
            const float inf = __builtin_inff();
            for (int c = 0; c < 3; c++) { r.wlo[c] = inf; r.whi[c] = -inf; }
            // (the world box of a transformed instance comes from k_instance_boxes below: empty until then)
        }
        r.root_code = m->blas.root_code;
        r.flags = identity ? RT_INST_IDENTITY : 0u;
        r.wide = m->blas.wide.as<WNode>();
        r.tris = m->tris.as<TriRec>();
        r.cnodes = m->blas.nodes.as<rt_bvh_node>();
        r.verts = m->d_verts.as<rt_vertex>();
        r.indices = m->d_idx.as<uint32_t>();
        r.normals = m->normals.as<TriRec>();
        r.n_prims = m->n_tris;
        r.n_recs = m->n_recs;
        r.pad_ = 0;
        r.material = i;
        r.rec_boxes = m->rec_boxes.p ? m->rec_boxes.as<float>() : nullptr;
        r.ref_off = m->ref_off.p ? m->ref_off.as<uint32_t>() : nullptr;
        r.ref_boxes = m->ref_boxes.p ? m->ref_boxes.as<float>() : nullptr;
        if (i == 0) s->has_refs = false;
        s->has_refs = s->has_refs || m->ref_off.p != nullptr;
        deepest = m->blas.fast_depth > deepest ? m->blas.fast_depth : deepest;
        if (!identity)
            for (uint32_t c = 0; c * INST_BOX_REFS < 3u * m->n_tris; c++) items.push_back(make_uint2(i, c));
    }
    BuildTemps bt;
    DevBuf &boxes = bt.boxes, &enc = bt.enc, &bounds = bt.bounds, &tkeys = bt.tkeys, &tsort = bt.tsort, &tenc = bt.tenc, &tdepth = bt.tdepth;
    // (k_instance_boxes' work memory, a slice of the build arena: per-instance encoded bounds, the forward transforms, the work list)
    const size_t ienc_bytes = (6 * sizeof(uint32_t) * (size_t)n + 255) & ~(size_t)255, ixf_bytes = (12 * sizeof(float) * (size_t)n + 255) & ~(size_t)255;
    int rc = RT_OK;
    do {
        if ((rc = take_build_temps(ctx, n, bt, ienc_bytes + ixf_bytes + items.size() * sizeof(uint2))) != RT_OK) break;
        uint32_t *const ienc = bt.extra.as<uint32_t>();
        float *const ixf = (float *)((char *)bt.extra.p + ienc_bytes);
        uint2 *const iitems = (uint2 *)((char *)bt.extra.p + ienc_bytes + ixf_bytes);
        if ((rc = s->d_inst.reserve(sizeof(InstanceRec) * (size_t)n)) != RT_OK) break;
        if ((rc = boxes.reserve(sizeof(Box6) * (size_t)n)) != RT_OK) break;
        if ((rc = enc.reserve(6 * sizeof(uint32_t))) != RT_OK) break;
        if ((rc = bounds.reserve(6 * sizeof(float))) != RT_OK) break;
        if (hipMemcpyAsync(s->d_inst.p, s->h_inst.data(), sizeof(InstanceRec) * n, hipMemcpyHostToDevice, st) != hipSuccess) {
            rt_set_error("instance upload failed");
            rc = RT_ERR_HIP;
            break;
        }
        // world boxes: of the transformed instances from their vertices, then all of them into the records and the leaf boxes
        k_instance_boxes_init<<<grid_for(6 * n, 256), 256, 0, st>>>(ienc, n);        // (a transformed instance without triangles keeps the empty box)
        if (!items.empty()) {
            xf.resize(12 * (size_t)n);
            for (uint32_t i = 0; i < n; i++) memcpy(&xf[12 * (size_t)i], s->inst[i].xform, 12 * sizeof(float));
            if (hipMemcpyAsync(ixf, xf.data(), xf.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(iitems, items.data(), items.size() * sizeof(uint2), hipMemcpyHostToDevice, st) != hipSuccess) {
                rt_set_error("instance upload failed");
                rc = RT_ERR_HIP;
                break;
            }
            k_instance_boxes<<<(uint32_t)items.size(), BOUNDS_BLOCK, 0, st>>>(s->d_inst.as<InstanceRec>(), ixf, iitems, ienc);
        }
        k_instance_boxes_finish<<<grid_for(n, 256), 256, 0, st>>>(s->d_inst.as<InstanceRec>(), ienc, boxes.as<Box6>(), n);
        if (hipMemcpyAsync(hb.data(), boxes.p, sizeof(Box6) * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rt_set_error("instance box read-back failed");
            rc = RT_ERR_HIP;
            break;
        }
        for (uint32_t i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) { s->h_inst[i].wlo[c] = hb[i].lo[c]; s->h_inst[i].whi[c] = hb[i].hi[c]; }
        k_init_bounds<<<1, 64, 0, st>>>(enc.as<uint32_t>());
        k_box_bounds<<<grid_for(n, BOUNDS_BLOCK), BOUNDS_BLOCK, 0, st>>>(boxes.as<Box6>(), n, enc.as<uint32_t>());
        k_decode_bounds<<<1, 64, 0, st>>>(enc.as<uint32_t>(), bounds.as<float>());
        if ((rc = lbvh_from_boxes(ctx, s->tlas, boxes.as<Box6>(), n, bounds.as<float>(), true, tkeys, tsort, tenc, tdepth)) != RT_OK) break;
        // the TLAS is walked in the same four-wide layout (a single instance: the root is the leaf of instance 0)
        if ((rc = rt_build_wide_from_lbvh(ctx, s->tlas, true, 1)) != RT_OK) break;
        if ((rc = lbvh_collect(ctx, s->tlas)) != RT_OK) break;
        // a step leaves at most three siblings behind; two-level walks add the TLAS path and the sentinel that marks the
        // bottom of a BLAS walk
        s->two_level = !(n == 1 && (s->h_inst[0].flags & RT_INST_IDENTITY));
        s->stack_need = s->two_level ? s->tlas.fast_depth + 1 + deepest : deepest;
        // the canonical traversal (parity / counting kernels and the deep-stack path of the fast one) keeps
        // 128-entry private stacks per structure
        uint32_t canon = s->tlas.max_depth;
        for (uint32_t i = 0; i < n; i++) canon = s->inst[i].model->blas.max_depth > canon ? s->inst[i].model->blas.max_depth : canon;
        if (canon >= 127) { rt_set_error("acceleration structure %u levels deep: the limit is 126", canon); rc = RT_ERR_UNSUPPORTED; break; }
        if (ctx->verbose)
            fprintf(stderr, "[dxr_amd] TLAS %u instances depth %u; deepest BLAS layout depth %u (%s); stack need %u; %s walk\n", n,
                    s->tlas.max_depth, deepest, ctx->use_ploc ? "PLOC" : "LBVH", s->stack_need, s->two_level ? "two-level" : "single-level");
    } while (0);
    boxes.release(); enc.release(); bounds.release(); tkeys.release(); tsort.release(); tenc.release(); tdepth.release();
    bt.extra.release();
    return rc;
}
