/*
 * dxr_amd_types.h -- plain-old-data records that cross the C-ABI boundary.
 *
 * These restate, byte for byte, the host/device shared records of the
 * reference (assets/shaders/RaytracingHlslCompat.h:25-96) so that host code
 * written against the reference's PerFrameConstants / MaterialParams / Vertex
 * keeps working, plus the records this engine adds for acceleration-structure
 * inspection and statistics.  C99 and C++ both compile this header; the HIP
 * kernels and the CPU oracle include it too (the reference plays the same
 * host/device-header trick, RaytracingHlslCompat.h:15-22).
 */
#ifndef DXR_AMD_TYPES_H
#define DXR_AMD_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_float2 { float x, y; } rt_float2;
typedef struct rt_float3 { float x, y, z; } rt_float3;
typedef struct rt_float4 { float x, y, z, w; } rt_float4;

/* RaytracingHlslCompat.h:35-39 (dup RtModel.cpp:13-17). 24 B, interleaved. */
typedef struct rt_vertex {
    rt_float3 position;
    rt_float3 normal;
} rt_vertex;

/* RaytracingHlslCompat.h:41-50. 80 B.  U/V/W .w must be 0. */
typedef struct rt_camera_params {
    rt_float4 worldEyePos;
    rt_float4 U;
    rt_float4 V;
    rt_float4 W;
    rt_float2 jitters;
    uint32_t  frameCount;
    uint32_t  accumCount;
} rt_camera_params;

/* RaytracingHlslCompat.h:52-56. */
typedef struct rt_directional_light_params {
    rt_float4 forwardDir;
    rt_float4 color;      /* .w = intensity multiplier (RaytracingCommon.hlsli:133) */
} rt_directional_light_params;

/* RaytracingHlslCompat.h:58-62. */
typedef struct rt_point_light_params {
    rt_float4 worldPos;
    rt_float4 color;      /* .w = intensity multiplier (RaytracingCommon.hlsli:146) */
} rt_point_light_params;

/* RaytracingHlslCompat.h:64-77. 44 B. */
typedef struct rt_debug_options {
    uint32_t maxIterations;
    uint32_t cosineHemisphereSampling;
    uint32_t showIndirectDiffuseOnly;
    uint32_t showIndirectSpecularOnly;
    uint32_t showAmbientOcclusionOnly;
    uint32_t showGBufferAlbedoOnly;
    uint32_t showDirectLightingOnly;
    uint32_t showFresnelTerm;
    uint32_t noIndirectDiffuse;
    float    environmentStrength;
    uint32_t debug;
} rt_debug_options;

/* RaytracingHlslCompat.h:79-85. 188 B, the b0 constant buffer. */
typedef struct rt_per_frame_constants {
    rt_camera_params            cameraParams;
    rt_directional_light_params directionalLight;
    rt_point_light_params       pointLight;
    rt_debug_options            options;
} rt_per_frame_constants;

/* RaytracingHlslCompat.h:87-96. 64 B, one per instance. */
typedef struct rt_material_params {
    rt_float4 albedo;
    rt_float4 specular;
    rt_float4 emissive;
    float     reflectivity;
    float     roughness;
    float     IoR;
    uint32_t  type;       /* 0 diffuse, 1 glossy, 2 specular (glass) */
} rt_material_params;

/*
 * Canonical BVH2 node (this engine's own definition; the Fallback Layer's
 * layout is not in the reference checkout).  A tree over N primitives has
 * 2N-1 nodes: internal nodes 0..N-2 (Karras numbering, root = 0), leaves
 * N-1..2N-2 in sorted-key order.  Internal: left/right are node indices.
 * Leaf: left = primitive (BLAS) or instance (TLAS) index, right = RT_LEAF.
 */
typedef struct rt_bvh_node {
    float    bmin[3];
    uint32_t left;
    float    bmax[3];
    uint32_t right;
} rt_bvh_node;

#define RT_LEAF      0xFFFFFFFFu
#define RT_NO_HIT    0xFFFFFFFFu

/* Ray flags: numeric values of D3D12_RAY_FLAG_* used by the reference
 * (ProgressiveRaytracing.hlsl:34,53; RaytracingCommon.hlsli:94). */
#define RT_RAY_FLAG_NONE                             0x00u
#define RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH  0x04u
#define RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER          0x08u
#define RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES       0x10u

/* Output formats for createOutputResource (DXGI_FORMAT stand-ins). */
#define RT_FORMAT_R32G32B32A32_FLOAT  2u   /* DXGI_FORMAT_R32G32B32A32_FLOAT */
#define RT_FORMAT_R16G16B16A16_FLOAT  10u  /* DXGI_FORMAT_R16G16B16A16_FLOAT */

/* Cube-map filtering of the environment (sampler s0: MIN_MAG_LINEAR_MIP_POINT, ProgressiveRaytracingPipeline.cpp:48-55). */
#define RT_CUBE_SEAMLESS   0u     /* D3D10+ behaviour: bilinear taps cross face edges (default) */
#define RT_CUBE_FACE_CLAMP 1u     /* taps clamped to the selected face */

/* Accumulation modes. */
#define RT_ACCUM_RUNNING_MEAN 0u  /* (n*prev+cur)/(n+1), ProgressiveRaytracing.hlsl:36-38 */
#define RT_ACCUM_SUM          1u  /* prev+cur; caller divides by count (multi-GPU shards) */
/* rounding of the fp32 running mean when the accumulation is STORED as RGBA16F (rt_pipeline_set_accumulation_storage) */
#define RT_ROUND_NEAREST_EVEN 0u
#define RT_ROUND_TOWARD_ZERO  1u

/* Render statistics: rt_pipeline_get_stats = the most recent render call (frames = 1);
 * rt_pipeline_get_totals = sums since rt_pipeline_reset_totals. */
typedef struct rt_stats {
    uint64_t rays_primary;
    uint64_t rays_secondary;     /* closest-hit radiance rays beyond primary */
    uint64_t rays_shadow;
    uint64_t primary_hits;
    uint64_t secondary_hits;
    float    ms_primary;         /* raygen + primary traversal            */
    float    ms_shade0;          /* first-hit shading / ray emission      */
    float    ms_trace_secondary; /* closest-hit traversal of levels >= 1   */
    float    ms_trace_shadow0;   /* 0: every shadow ray of a frame runs in one launch, timed below */
    float    ms_shade1;          /* shading / ray emission of levels >= 1  */
    float    ms_trace_shadow1;   /* shadow traversal, all levels           */
    float    ms_resolve;         /* final shade + accumulate              */
    float    ms_total;
    uint64_t frames;             /* frames the counts / times above cover  */
    uint64_t rays_shadow_skipped;/* of rays_shadow: emitted for a light with N.L == 0, whose visibility is multiplied by zero,
                                    and therefore not traversed (rt_pipeline_set_skip_unlit_shadow_rays); rays_shadow itself counts
                                    every shadow ray the reference's shaders trace                                              */
} rt_stats;

/* Algorithmic work of one traversal stage of the last rendered frame, counted by
 * the canonical (reference-order) traversal: see SURVEY.md 8(d), DESIGN.md. */
typedef struct rt_stage_work {
    uint64_t rays;               /* rays actually traced in the stage        */
    uint64_t nodes;              /* AABBs slab-tested by the canonical loop  */
    uint64_t tris;               /* triangles handed to the triangle test    */
} rt_stage_work;
#define RT_STAGE_PRIMARY   0
#define RT_STAGE_SECONDARY 1
#define RT_STAGE_SHADOW0   2
#define RT_STAGE_SHADOW1   3
#define RT_STAGE_COUNT     4

/* What the PRODUCTION traversal kernels fetch for one stage of the last rendered frame, tallied per lane by a
 * counting instantiation of the same walk (rt_pipeline_count_walk): these, not the canonical counters above,
 * are the bytes the timed kernels really request, and the input of bench.py's L2-bound roofline. */
typedef struct rt_stage_walk {
    uint64_t rays;               /* rays traversed                                              */
    uint64_t nodes_global;       /* 64-B four-wide nodes loaded from global memory, per lane      */
    uint64_t nodes_lds;          /* nodes read from the LDS-resident top of the tree              */
    uint64_t tris;               /* 48-B triangle records loaded                                */
    uint64_t instance_entries;   /* instance records entered (two-level scenes; 96 B read each) */
    uint64_t lines;              /* distinct 64-B lines fetched: node lines de-duplicated over the lanes of each wave step +
                                    the one or two lines each triangle record spans                                   */
    uint64_t longest_walk;       /* node steps of the stage's longest single ray (persistent kernels end with their slowest lane) */
    uint64_t longest_walk_ray;   /* ... and that ray's index in its queue (last launch of the stage) */
    /* what the WAVES of the walk did (round 4): with the per-lane tallies above, the lane utilisation of the walk's two halves,
     * (nodes_global + nodes_lds) / (64 * wave_node_steps) and tris / (64 * wave_tri_steps)                             */
    uint64_t wave_node_steps;    /* node steps issued: one per wave per pass of the node loop                        */
    uint64_t wave_leaf_phases;   /* leaf phases (triangles, instance entry / exit, ray end) entered by a wave         */
    uint64_t wave_tri_steps;     /* triangle-test iterations issued by the waves' leaf phases                         */
    uint64_t node_lines;         /* the node part of `lines`: distinct 64-B node lines per wave step, summed             */
} rt_stage_walk;

/* Status codes returned by every export. */
#define RT_OK                 0
#define RT_ERR_INVALID_ARG   -1
#define RT_ERR_HIP           -2
#define RT_ERR_IO            -3
#define RT_ERR_STATE         -4
#define RT_ERR_OOM           -5
#define RT_ERR_UNSUPPORTED   -6

#ifdef __cplusplus
}  /* extern "C" */

static_assert(sizeof(rt_vertex) == 24, "Vertex is 24 B");
static_assert(sizeof(rt_camera_params) == 80, "CameraParams is 80 B");
static_assert(sizeof(rt_directional_light_params) == 32, "light is 32 B");
static_assert(sizeof(rt_point_light_params) == 32, "light is 32 B");
static_assert(sizeof(rt_debug_options) == 44, "DebugOptions is 44 B");
static_assert(sizeof(rt_per_frame_constants) == 188, "PerFrameConstants is 188 B");
static_assert(sizeof(rt_material_params) == 64, "MaterialParams is 64 B");
static_assert(sizeof(rt_bvh_node) == 32, "BVH node is 32 B");
#endif

#endif /* DXR_AMD_TYPES_H */
