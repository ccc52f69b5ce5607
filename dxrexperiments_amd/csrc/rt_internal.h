// rt_internal.h -- objects behind the opaque C-ABI handles (include/dxr_amd.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/dxr_amd.h"

void rt_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            rt_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define RT_TRY(expr)              \
    do {                          \
        int r_ = (expr);          \
        if (r_ != RT_OK) return r_; \
    } while (0)

#define RT_REQUIRE(cond, msg)                        \
    do {                                             \
        if (!(cond)) {                               \
            rt_set_error("%s: %s", __func__, msg);   \
            return RT_ERR_INVALID_ARG;               \
        }                                            \
    } while (0)

// Test hook (rt_debug_set_alloc_limit): device allocations above this many bytes fail with RT_ERR_OOM as if the device were
// full, so that the out-of-memory paths can be exercised on a 288 GB part.  Default: no limit.
size_t &rt_alloc_limit_ref();
static inline size_t rt_alloc_limit() { return rt_alloc_limit_ref(); }

// Owning device allocation; grows on demand, never shrinks.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool borrowed = false;       // p points into somebody else's allocation (adopt): never freed here
    // use a slice of another allocation; a later reserve() beyond it falls back to an allocation of its own
    void adopt(void *ptr, size_t n) { release(); p = ptr; bytes = n; borrowed = ptr != nullptr; }
    // Grows to at least n bytes (contents are NOT kept).  The new block is allocated BEFORE the old one is freed, so a
    // growth that fails leaves the buffer as it was; only when old + new do not fit together is the old block given up
    // first -- and then a second failure leaves an EMPTY buffer (p = nullptr, bytes = 0), never a stale pointer.  Callers
    // keep no capacities of their own: what a buffer can hold is `bytes`.
    int reserve(size_t n)
    {
        if (n <= bytes) return RT_OK;
        const size_t want = n ? n : 16;
        void *q = nullptr;
        hipError_t e = want > rt_alloc_limit() ? hipErrorOutOfMemory : hipMalloc(&q, want);
        if (e == hipErrorOutOfMemory && p && !borrowed && want <= rt_alloc_limit()) {
            (void)hipGetLastError();
            (void)hipFree(p);
            p = nullptr; bytes = 0;
            q = nullptr;
            e = hipMalloc(&q, want);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            rt_set_error("hipMalloc(%zu): %s", n, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP;
        }
        if (p && !borrowed) (void)hipFree(p);
        p = q; bytes = want; borrowed = false;
        return RT_OK;
    }
    void release()
    {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        borrowed = false;
    }
    template <class T> T *as() const { return (T *)p; }
};

// ---- device-visible records -------------------------------------------------

// One traversal node = FOUR children in one 64-B line (fetched as 4 x dwordx4).  Child boxes are quantised to 8 bits per plane
// on a power-of-two grid anchored at the node's own box: plane = fma(q, scale, origin), the builder rounds lo planes down and
// hi planes up WITH THIS SAME EXPRESSION until the decoded box contains the child's true box, so culling against it is
// conservative and the candidate validation of rt_trace_device.h keeps every result bit-identical to the canonical definition.
//   q0 = origin.x origin.y origin.z scale.x
//   q1 = lo.x[4] hi.x[4] lo.y[4] hi.y[4]        (one byte per child, child k in bits 8k..8k+7)
//   q2 = lo.z[4] hi.z[4] scale.y scale.z
//   q3 = code[4]
// Child code: >= 0 index of an internal node of the same array; < 0 leaf: ~code = (first << 3) | (count-1) for a BLAS
// (triangles first..first+count-1 of the sorted triangle array), or the instance index for the TLAS; RT_NODE_NONE for an
// unused slot.  Nodes are numbered breadth first, so the first top_n nodes ARE the top of the tree (LDS resident in the
// traversal kernels).  Children are packed at the front, larger surface first (any-hit rays walk unordered and try the
// likelier occluder first); closest-hit rays sort the hit children by entry distance.  A scale of +inf (with q = 0 planes)
// marks an axis the builder could not quantise: its planes decode to NaN and never cull.
//
// RT_WIDE = 8 (build option -DRT_WIDE=8; round 3's experiment, measured SLOWER and kept reproducible: DESIGN.md section 4,
// profiles/r03/wide8_experiment.md): EIGHT children in one 128-B-aligned record of which the traversal reads 96 B,
//   q0 = origin.xyz meta      meta = ex | ey << 8 | ez << 16 | valid << 24: biased exponents of the three scales (255: not
//                             quantised) and the mask of the slots in use
//   q1 = lo.x[0..3] lo.x[4..7] hi.x[0..3] hi.x[4..7]    q2, q3 = the same for y, z    q4 = code[0..3]    q5 = code[4..7]
//   q6 = first internal child, internal-slot mask, source binary node, 0 (inspection only)    q7 = 0
// The internal children of a node are consecutive in slot order.  Slots: the builder puts the child that lies towards
// the (+,+,+) corner of the node in slot 7, towards (-,-,-) in slot 0, ... so that (slot XOR direction octant) ascending is
// a front-to-back order and the traversal never sorts by distance (Ylitie, Karras, Laine 2017).
#ifndef RT_WIDE
#define RT_WIDE 4
#endif
#if RT_WIDE == 8
struct __attribute__((aligned(128))) WNode { float4 q0, q1, q2, q3, q4, q5, q6, q7; };
#define RT_NODE_SHIFT 7               // log2(sizeof(WNode))
#define RT_TOP_WORDS 24               // ints of a node kept in LDS (the part the traversal reads)
#else
struct WNode { float4 q0, q1, q2, q3; };
#define RT_NODE_SHIFT 6
#define RT_TOP_WORDS 16
#endif
#define RT_NODE_NONE ((int)0x80000000)

// One triangle in leaf order: the three ORIGINAL vertex positions (so that the
// Moller-Trumbore edges and the triangle's own AABB are recomputed from the same
// floats the canonical BVH was built from) plus the primitive id.  48 B.
struct TriRec { float4 a, b, c; };   // a = v0.xyz v1.x ; b = v1.yz v2.xy ; c = v2.z prim ref 0
                                     // ref (bits of c.z, round 5, rt_refs.h): 0 = the triangle is validated against its own AABB (every triangle of rounds 1 - 4);
                                     //   1 = this record is ONE reference of a split triangle, its box is InstanceRec::rec_boxes[record index];
                                     //   2 = a split triangle in a layout that holds it once (LBVH layout): validated against all its boxes, by primitive
                                     // (padding a record to one 64-B line so that none straddles two: measured, no change)

#define RT_INST_IDENTITY 1u

struct InstanceRec {
    float inv[12];              // world-to-object, 3x4 row-major
    float wlo[3]; int root_code;
    float whi[3]; uint32_t flags;
    const WNode *wide;
    const TriRec *tris;
    const rt_bvh_node *cnodes;  // canonical nodes of the BLAS
    const rt_vertex *verts;
    const uint32_t *indices;
    const TriRec *normals;      // the three vertex normals of primitive p in ONE 48-B record (n0.xyz n1.x | n1.yz n2.xy | n2.z):
                                //   a shaded hit fetches one record instead of three indices and three scattered vertices
    uint32_t n_prims;           // triangles of the model
    uint32_t material;
    // split references (rt_refs.h): nullptr when no triangle of the model is split
    const float *rec_boxes;     // 6 floats per record of `tris` (meaningful where the record says ref = 1)
    const uint32_t *ref_off;    // per primitive: its boxes are ref_boxes[6 * ref_off[p] .. 6 * ref_off[p + 1])
    const float *ref_boxes;
    uint32_t n_recs;            // records of `tris`: n_prims, or more when some triangles are held as several references
    uint32_t pad_;
};

struct SceneDev {
    const InstanceRec *inst;
    const WNode *tlas_wide;
    const rt_bvh_node *tlas_cnodes;
    uint32_t n_inst;
    int tlas_root_code;
    int *deep_stack;             // global stack rows beyond the LDS rows (rt_trace_wave.h); nullptr: never needed
    uint32_t top_n;              // nodes 0..top_n-1 of the structure a ray starts in (single-level: the BLAS, two-level: the
                                 //   TLAS) are copied into LDS by every traversal workgroup
};

#ifndef RT_TOP_NODES
#if RT_WIDE == 8
#define RT_TOP_NODES  80              // nodes of the LDS-resident top, 96 B each (the part of a node the traversal reads): 7.5 KiB of
                                      //   LDS per 256-thread block; levels 0 .. 2 of an eight-wide tree have at most 73 nodes
#else
#define RT_TOP_NODES  128             // 8 KiB (with four-wide nodes the size hardly matters: 32 .. 192 nodes all within 1 %)
#endif
#endif
#define RT_TOP_ROWS(BLOCK) ((RT_TOP_NODES * RT_TOP_WORDS + (BLOCK) - 1) / (BLOCK))      // LDS rows (of BLOCK ints) the top table takes


// ---- host objects --------------------------------------------------------------

#define RT_PINNED_WORDS 128
#define RT_PINNED_LBVH 64
struct rt_context {
    int refs = 1;                // handle + every model / scene / pipeline created on it
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_trace_ms = 0.0f;
    uint32_t leaf_max = 2;       // triangles per collapsed leaf in the traversal layout (option leaf_max; 2 measured best: 1 4.05, 2 3.69, 3 3.80, 4 3.87, 8 4.23 ms/frame)
    bool use_ploc = true;        // option fast_bvh=lbvh keeps the canonical LBVH as the traversal layout
    bool wide_sah = false;       // option wide_sah=1: the collapse into wide nodes minimises the surface-area cost (rt_bvh_wide.hip); default: area-greedy
                                 //   (measured, profiles/r03/wide8_experiment.md: no faster on either layout)
    float sah_node = 1.0f, sah_prim = 0.5f;     // cost of one wide-node step / one triangle test (options sah_node, sah_prim)
    uint32_t cu_count = 256;     // compute units of the device
    size_t device_mem_total = 0; // bytes of device memory (asked once, when a pipeline first sizes its queues)
    uint32_t blocks_per_cu_override = 0;    // option persistent_blocks_per_cu: 0 = ask the occupancy API per kernel
    bool lds_top = true;                    // option lds_top=0: every node comes from global memory
    uint32_t lds_stack_rows = 0;            // option lds_stack_rows=6: the small-stack instantiation (tests of the deep-stack path)
    DevBuf pool;                 // ray-pool counters of rt_trace_batch
    DevBuf deep_stack;           // global stack rows of the traversal kernels (rt_scene_dev_for_launch)
    uint32_t build_batch = 0;    // test hook (option build_batch): PLOC rounds / collapse levels launched between two looks at the
                                 //   device-side state; 0 = the builders' own estimates
    uint32_t *pinned = nullptr;  // RT_PINNED_WORDS of page-locked host memory: small device-to-host read-backs without staging
                                 // (words 0..63: whoever synchronises next; RT_PINNED_LBVH..+6: depth and bounds of an LBVH in flight)
    DevBuf build_arena;          // temporaries of the acceleration-structure builds: one allocation, sliced (hipMalloc
                                 // and hipFree synchronise the device and cost more than the kernels of a small build)
    DevBuf scratch[8];           // staging for host-pointer batch calls
    // Experiment / test knobs (rt_debug_set_option, include/dxr_amd.h; the one environment variable RT_DEBUG_OPTIONS="name=value,..." is
    // applied once, when the context is created).  Nothing else in the library reads the environment for its behaviour.
    bool verbose = false;                   // verbose=1: build phases and structure depths on stderr
    int opt_shadow_cache_res = -1;          // shadow_cache_res: cells per side for pipelines that have not been told (-1: by triangle count, 0: off)
    int opt_shadow_cache_pixels = -1;       // shadow_cache_pixels: per-pixel entries (-1: two-level scenes only)
    int opt_primary_persistent = -1;        // primary_persistent: the primary stage as a persistent launch (-1: two-level scenes only)
    bool opt_seven_waves_always = false;    // seven_waves_always=1: single frames on the sets' kernels
    bool opt_free_radius = true;            // free_radius=0: no free sphere around the point light
    uint32_t opt_batch_max = 0;             // batch_max: frames per set of launches (0: RT_MAX_BATCH)
    bool opt_repack = false;                // repack=1 (round 6, rt_trace_repack.h): the shadow stage of single-level scenes on the re-packed engine
    unsigned rp_grid = 0;                   // ... the grid they were sized for
    DevBuf rp_records;                      // ... and its slot records (64 B x 256 slots per resident workgroup)
    bool opt_fail_ploc_rounds = false;      // fail_ploc_rounds=1 (tests): the PLOC layout is thrown away as if its rounds had made no progress (non-finite boxes): the LBVH fallback
    bool opt_split_refs = true;             // split_refs=0: no triangle is held as several references (the builder of rounds 1 - 4; the CANDIDATE RULE still
                                            //   follows rt_refs.h -- results do not depend on this option)
    uint32_t opt_primary_retry_cap = 0;     // primary_retry_cap: entries of the primary launch's retry list (0: 2^20; tests: a few, so that the list overflows)
    size_t opt_queue_budget_mb = 0;         // queue_budget_mb: worst-case queue bytes a set may reserve up front (0: a quarter of the device)
    double opt_dist_check_seconds = 5.0;    // dist_check_seconds: how long rt_dist_create waits for the other ranks' device ids
    std::vector<struct rt_pipeline *> deferred;      // pipelines holding frames that render() has accepted and not rendered yet
                                 //   (rt_pipeline_set_deferred): whatever changes what those frames would see flushes them first
};

struct BvhDev {
    uint32_t n = 0;              // primitives
    uint32_t max_depth = 0;
    float bounds[6] = {0, 0, 0, 0, 0, 0};
    DevBuf nodes;                // rt_bvh_node[2n-1]   canonical
    DevBuf keys;                 // uint64[n]           sorted morton keys
    DevBuf parents;              // uint32[2n-1]
    DevBuf ranges;               // uint2[n-1]          leaf range of every internal node
    DevBuf wide;                 // WNode[wide_n]       traversal layout, breadth first
    uint32_t wide_n = 0;
    int root_code = -1;
    uint32_t fast_depth = 0;     // stack entries the traversal layout can need
    void release() { nodes.release(); keys.release(); parents.release(); ranges.release(); wide.release(); }
};

struct rt_model {
    rt_context *ctx = nullptr;
    int refs = 1;
    uint32_t n_verts = 0, n_tris = 0;
    std::vector<rt_vertex> h_verts;
    std::vector<uint32_t> h_idx;
    DevBuf d_verts, d_idx;
    DevBuf tris;                 // TriRec[n_recs] in sorted order
    DevBuf normals;              // TriRec[n_tris] in primitive order: the vertex normals (InstanceRec::normals)
    uint32_t n_recs = 0;         // records of `tris`: n_tris, or more when long thin triangles are held as several references (rt_refs.h)
    DevBuf rec_boxes;            // float[6 n_recs]: the reference box of every record (split models only)
    DevBuf ref_off, ref_boxes;   // uint32[n_tris + 1], float[6 refs]: the same boxes by primitive (canonical walk, LBVH layout)
    BvhDev blas;
    bool built = false;
};

struct SceneInstance {
    rt_model *model;
    float xform[12];
};

struct rt_scene {
    rt_context *ctx = nullptr;
    int refs = 1;                // handle + every pipeline it is set on
    std::vector<SceneInstance> inst;
    std::vector<InstanceRec> h_inst;
    DevBuf d_inst;
    BvhDev tlas;
    bool built = false;
    uint32_t generation = 0;     // bumped by every add_model / build: device pointers cached from an older one are stale
    float build_ms = 0.0f;
    uint32_t stack_need = 0;     // traversal stack entries a ray can hold at once
    bool two_level = true;       // false: one identity instance, rays walk its BLAS directly
    bool has_refs = false;       // some model of the scene holds triangles as several references (rt_refs.h)
    SceneDev dev() const
    {
        SceneDev s;
        s.inst = d_inst.as<InstanceRec>();
        s.tlas_wide = tlas.wide.as<WNode>();
        s.tlas_cnodes = tlas.nodes.as<rt_bvh_node>();
        s.n_inst = (uint32_t)inst.size();
        s.tlas_root_code = tlas.root_code;
        s.deep_stack = nullptr;
        s.top_n = 0;
        return s;
    }
};

// ---- internal entry points shared between translation units ---------------------

// rt_bvh_build.hip
int rt_build_blas(rt_context *ctx, rt_model *m);
int rt_build_tlas(rt_context *ctx, rt_scene *s);

// rt_bvh_ploc.hip
// n_leaves / leaf_box6 / leaf_prim: the leaves to cluster in Morton order -- nullptr: the canonical leaves (one per triangle); else the
// references of a model with split triangles (rt_refs.h): 6 floats and a primitive id per leaf
int rt_build_ploc_layout(rt_context *ctx, rt_model *m, bool *done, uint32_t n_leaves = 0, const float *leaf_box6 = nullptr, const uint32_t *leaf_prim = nullptr);

// rt_bvh_wide.hip: collapses a binary tree into the four-wide traversal layout (bv.wide, wide_n, fast_depth, root_code).
// The binary tree is given in "cluster" numbering: leaves 0..n-1 in sorted-key order, internal nodes n..2n-2;
// left / right are indexed by id - n; parent by id (0xFFFFFFFF for the root); box6 = {lo[3], hi[3]} per id; size = leaves below; offset = leaves before (depth
// first); leaf_prim (TLAS only) maps a leaf to its instance.  Temporaries come from the arena slice [tmp, tmp + tmp_bytes).
size_t rt_wide_temp_bytes(uint32_t n);
size_t rt_wide_lbvh_temp_bytes(uint32_t n);
int rt_build_wide_layout(rt_context *ctx, BvhDev &bv, uint32_t n, uint32_t root, const uint32_t *left, const uint32_t *right, const uint32_t *parent,
                         const float *box6, const uint32_t *size, const uint32_t *offset, const uint32_t *leaf_prim, uint32_t leaf_max,
                         void *tmp, size_t tmp_bytes);
// the same from the canonical LBVH arrays of bv (TLAS, tiny meshes, option fast_bvh=lbvh)
int rt_build_wide_from_lbvh(rt_context *ctx, BvhDev &bv, bool tlas, uint32_t leaf_max);

// rt_trace.hip
struct TraceOut {
    float *t, *u, *v;
    uint32_t *prim, *inst, *cnt_nodes, *cnt_tris;
};
int rt_launch_trace(rt_context *ctx, const rt_scene *s, const float4 *origin_tmin, const float4 *dir_tmax, size_t n,
                    uint32_t ray_flags, uint32_t kernel, const TraceOut &out);

// Grid of a persistent traversal kernel: exactly the blocks that are resident at once (the static
// chunk interleaving gives every launched wave its share of the queue, so a block that has to wait
// for a slot would run its share as a serial tail).
#define RT_RESIDENT_BLOCKS_PER_CU 8
template <class K>
static inline unsigned rt_persistent_grid(const rt_context *ctx, K kernel, int block, size_t rays)
{
    int per_cu = (int)ctx->blocks_per_cu_override;
    if (per_cu <= 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    // the global stack rows behind the LDS ones are sized for RT_RESIDENT_BLOCKS_PER_CU 256-thread workgroups per CU (or the option's count, if larger:
    // scene_for_set): a launch may not have more threads than that, whatever the occupancy query says for a kernel with little LDS
    const int row_cap = ((int)ctx->blocks_per_cu_override > RT_RESIDENT_BLOCKS_PER_CU ? (int)ctx->blocks_per_cu_override : RT_RESIDENT_BLOCKS_PER_CU) * 256 / block;
    if (per_cu > row_cap) per_cu = row_cap;
    const size_t want = (rays + (size_t)block - 1) / (size_t)block;
    const size_t cap = (size_t)ctx->cu_count * (size_t)per_cu;
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

#ifndef RT_LDS_STACK_ROWS
#define RT_LDS_STACK_ROWS 18            // LDS stack rows of the traversal kernels; with the 8-row top table (top of the BLAS for
#endif                                  // single-level walks, top of the TLAS for two-level ones) 26 KiB per 256-thread block = 6 blocks per CU.
                                        // Bench-scene rays: 98.1 % never hold more than 8 entries, 99.97 % not more than 12, none more than 17.
// The traversal kernels' STACK template argument is the number of LDS stack rows, + RT_STACK_REFS for the instantiations that walk scenes
// with split references (rt_refs.h; trace_wave's REFS): one argument, so that every launch site picks both at once
#define RT_STACK_REFS 100
#define RT_ROWS(S) ((S) % RT_STACK_REFS)
#define RT_REFS(S) ((S) >= RT_STACK_REFS)
#define RT_LDS_STACK_ROWS_TEST 6        // second instantiation (option lds_stack_rows=6): tests force rays onto the global rows
// Sets of frames (rt_pipeline_render_batch) run long launches whose ramp and drain no longer matter, and there a seventh wave
// per SIMD pays (-3.5 % on the bench scene; frame by frame it costs 2 %, the drain grows with the resident waves:
// profiles/r03/seven_waves.txt): their single-level kernels keep 14 stack rows in LDS (22 KiB per block = 7 blocks per CU) and
// are compiled for seven waves (72 VGPRs; the 74 - 78 of the 18-row kernels shed two to six into scratch, off the step).
#ifndef RT_LDS_STACK_ROWS_SETS
#define RT_LDS_STACK_ROWS_SETS 14
#endif

// The scene as the traversal kernels see it, with the global stack rows a PERSISTENT launch of `threads` threads whose
// kernels keep `lds_rows` rows in LDS may need: the tree bounds a walk to stack_need (+1 speculative) entries.  (Round 5: the only launch
// that is not persistent, the one-tile-per-wave primary stage -- one thread per pixel slot: 224 MB of rows per 1080p frame of a set in
// rounds 1 - 4 -- keeps none: rt_trace_wave.h NO_DEEP.)
static inline int rt_scene_dev_for_launch(rt_context *ctx, const rt_scene *s, uint32_t lds_rows, size_t threads, SceneDev *out)
{
    *out = s->dev();
    if (ctx->lds_top) {          // one identity instance: rays walk its BLAS directly; otherwise the top of the TLAS
        const BvhDev &bv = s->two_level ? s->tlas : s->inst[0].model->blas;
        out->top_n = bv.wide_n < RT_TOP_NODES ? bv.wide_n : RT_TOP_NODES;
    }
    const uint32_t bound = s->stack_need + 2;
    if (bound <= lds_rows) return RT_OK;
    RT_TRY(ctx->deep_stack.reserve((size_t)(bound - lds_rows) * threads * sizeof(int)));
    out->deep_stack = ctx->deep_stack.as<int>();
    return RT_OK;
}
static inline uint32_t rt_lds_stack_rows(const rt_context *ctx)
{
    return ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST ? RT_LDS_STACK_ROWS_TEST : RT_LDS_STACK_ROWS;
}

// rt_bvh_ploc.hip
size_t rt_ploc_temp_bytes(uint32_t n);

// rt_pipeline.hip: renders the frames every deferred pipeline of the context still holds (rt_pipeline_set_deferred)
int rt_context_flush_deferred(rt_context *ctx);

// rt_api.hip
void rt_context_retain(rt_context *ctx);
void rt_context_release(rt_context *ctx);
void rt_scene_retain(rt_scene *s);

// rt_obj.cpp, rt_fbx.cpp
int rt_obj_parse(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx);
int rt_fbx_parse(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx);
