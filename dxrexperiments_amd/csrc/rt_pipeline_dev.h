// rt_pipeline_dev.h -- what the translation units of the pipeline share: the stage constants, the device-visible view of a
// frame (PipeDev: what every kernel of the wavefront DAG takes as its argument), and the host object behind rt_pipeline.
// Kernels and launches: rt_pipeline.hip.  Device shading functions: rt_shade.h.  Outputs, checkpoints, statistics:
// rt_pipeline_host.hip.
#pragma once

#include "rt_trace_wave.h"

namespace rtp {

using namespace rtd;


constexpr int PBLOCK = 256;
#ifndef RT_SHADOW_UNORDERED
#define RT_SHADOW_UNORDERED 1
#endif
// (only the any-hit instantiation of the walk tallies the queue slots marked "emitted but not traversed": with ordered shadow
// walks rt_stats.rays_shadow would under-count when rt_pipeline_set_skip_unlit_shadow_rays is on)
static_assert(RT_SHADOW_UNORDERED != 0, "the shadow stage counts skipped slots in the ANYHIT walk only");

#define RAY_MAX_T 1.0e+38f      // RaytracingCommon.hlsli:8
#define RAY_EPSILON 0.0001f     // RaytracingCommon.hlsli:9
#define HLSL_PI 3.1415927f      // RaytracingUtils.hlsli:22

#define HIT_MISS -1.0f
#define HIT_UNTRACED -2.0f

// Radiance rays are traced level by level: level 0 = primary rays, level L = rays spawned by the hits of
// level L-1.  MAX_RADIANCE_RAY_DEPTH (RaytracingCommon.hlsli:11) is 1 in the reference; the DAG below is
// generic up to MAXD so that BASELINE config 5 ("4-bounce") runs.  Because indirect DIFFUSE is only sampled
// at depth 0 (ProgressiveRaytracing.hlsl:107), every pixel owns at most two chains of specular bounces.
constexpr int MAXD = 4;
enum { C_NHIT = 0,                       // [0..MAXD] compacted hits per level
       C_SECONDARY = MAXD + 1, C_SHADOW = MAXD + 2,
       C_SHADOW_SKIPPED = MAXD + 3,      // shadow rays whose result cannot matter (N.L == 0): emitted, counted (by the shadow
                                         //   launch: traced_counter[1], rt_trace_wave.h), not traversed
       C_COUNT = MAXD + 4 };

// Level L: the rays at radiance depth L (L >= 1; primary rays are generated, not stored), their hit
// records, the compaction of the hits, and the shadow-ray queue of those hits.
//   slots of level 1: w*rstride + k  (w = 0 diffuse / 1 specular batch, k = compact index of the primary hit)
//   slots of level L >= 2: j     (compact index of the level L-1 hit that spawned the ray)
// The strides are the capacities the level's buffers were sized for: the worst case (rstride = hstride(0) = cap pixel slots,
// hstride(L >= 1) = 2 cap) or, for sets of frames whose worst case would not fit the device, the counts the compaction of the
// level before has just produced (launch_frame, "counted" queues).
// The SHADOW rays of all levels share one queue (PipeDev::sh_*, round 4).
struct LevelDev {
    uint32_t rstride;           // level 1: slots between its two batches (>= the hits of level 0); other levels: unused
    uint32_t hstride;           // hits this level's buffers have room for (host side: grids, the shared shadow queue's size)
    float4 *O, *D;              // ray queue (unused at level 0)
    float4 *hit; uint32_t *inst;   // hit records, indexed by slot (level 0: by pixel slot q)
    uint32_t *slot_j, *jlist;   // slot -> compact hit index (RT_NO_HIT if none), compact index -> slot
    uint32_t *pix;              // slot -> pixel slot q (unused at level 0)
    float4 *color;              // deep paths (more than one radiance level): the shaded colour of every hit of this level, by slot
};

// the frame's two light rays as the shadow-queue loader rebuilds them (QueueSrc::load)
struct LightRays {
    uint32_t on;
    float dir_to_light[3];      // normalize(-directionalLight.forwardDir), computed once per frame on the host with the
                                //   device's expression (IEEE sqrt and division, left-to-right sums, no contraction)
    float point_pos[3];         // pointLight.worldPos
    float point_free;           // no geometry lies nearer to the point light than this (0: unknown): its shadow rays end there
                                //   (free_radius(): a lower bound found by a device pass over the triangles' boxes)
};

// The shadow cache (round 3): a light buffer of occluders.  An any-hit search only asks WHETHER something lies between a point and
// a light, and in a progressive render the same question comes back frame after frame: the triangle that answered it last time
// for rays in the same place in LIGHT space -- the 2-D cell of the origin projected along the directional light, the cube-map
// texel of the direction from the point light -- is tested first (as a one-triangle leaf in front of the root on the ray's stack).
// If it occludes, the search is over after one triangle test; if not, the walk starts as always.  Any triangle is a legal first
// candidate, so entries may be stale or collide without touching the result; what the table holds only moves the time.
struct ShadowCacheDev {
    uint32_t *table;            // [res * res] directional cells, then [6 * (res / 2)^2] cube texels; nullptr: off.  0xFFFFFFFF: no entry
    float ua[4], va[4];         // directional light: cell (u, v) = (dot(o, ua.xyz) + ua.w, dot(o, va.xyz) + va.w), in [0, res)
    float lp[3];                // point light position
    float res_f;
    uint32_t res;
    uint32_t two_level;         // entries are (triangle, instance) pairs of 8 B: the triangle index counts inside that instance's BLAS
    uint32_t n_tris;            // single level: triangles of the one model (an entry at or above it is not tested)
    // Per-PIXEL entries for the shadow rays of the primary hits (round 4): the primary hit of a pixel is the same point, up to the
    // frame's sub-pixel jitter, frame after frame, so "the triangle that occluded THIS pixel's ray to this light last time" is a
    // better first candidate than the light-space cell's, and its table is read in slot order (coalesced).  Entries
    // px_base + 2 * (pixel slot in the frame) + (0: sun, 1: point light); px_base = 0: off.
    uint32_t px_base, entries;  // first per-pixel entry; entries in all (remember() checks its slot against it)
    const uint32_t *jlist0;     // compact primary hit -> pixel slot (LevelDev::jlist of level 0)
    uint32_t hstride0;          // storage stride of level 0's shadow rays (ray number < 2 * hstride0: a primary hit's ray)
    uint32_t n_frames, frames_magic;      // frames of the set; floor(2^32 / n_frames) + 1: chunk / n_frames by one mul_hi
    uint32_t px_slots;          // pixel slots of one frame (an index at or above it falls back to the light-space entry)
};

#define RT_MAX_BATCH 32u                // frames one set of launches renders (rt_pipeline_render_batch)

struct PipeDev {
    SceneDev sc;
    rt_per_frame_constants pfc;         // the frame's constants (a batch: of its first frame; kernels take pfcs[frame])
    // a BATCH of frames in one set of launches (config 3, rt_pipeline_render_batch): frame f owns the pixel slots
    // [f * fcap, (f + 1) * fcap), every queue is n_frames times as long, a slot's frame selects constants and lights
    uint32_t n_frames, fcap;            // single frame: 1, cap
    const rt_per_frame_constants *pfcs; // device array [n_frames] (batches only)
    const LightRays *frame_lights;      // device array [n_frames] (batches only)
    const rt_material_params *mats;
    uint32_t nmats;
    const float4 *env;
    uint32_t env_size;
    uint32_t env_filter;                // RT_CUBE_SEAMLESS / RT_CUBE_FACE_CLAMP
    float env_const[3];
    uint32_t width, height;
    uint32_t x0, y0, tw, th, cap;       // tile rectangle; cap = n_frames * tiles_x * tiles_y * 64 pixel slots
    uint32_t tiles_x;
    uint32_t band_rows, band_rank, band_world;      // band_rows != 0: the rectangle's rows are interleaved bands of the image
    uint32_t n_pixels;                  // pixels of the image this launch covers
    uint32_t max_rad, max_shadow;
    uint32_t accum_mode;
    uint32_t accum_f16;            // 0: the running mean stays fp32; 1 / 2: rounded to fp16 every frame, to nearest even / toward zero (rt_pipeline_set_accumulation_storage)
    uint32_t skip_unlit;                // do not traverse shadow rays of lights with N.L == 0 (their visibility is multiplied by 0)
    uint32_t shadow_compact;            // shadow queues hold ONE float4 per shaded hit (QueueSrc, "light rays")
    uint32_t kind;                      // RT_PIPELINE_PROGRESSIVE / RT_PIPELINE_REALTIME
    uint32_t primary_persistent;        // the primary stage is a persistent launch with a chunk pool of its own (experiment: RT_PRIMARY_PERSISTENT=1)
    float4 *accum;
    float4 *aov_direct, *aov_indirect;  // realtime pipeline outputs (RealtimeRaytracing.hlsl:3-4)
    uint32_t *counters;
    // The one-tile-per-wave primary launch has no stack rows beyond LDS (round 5): retry[0] counts the pixel slots whose ray would have
    // needed one, retry[2 + k], k < retry_cap, lists them (a count above retry_cap: the list is incomplete and k_primary_retry walks every
    // slot again).  Cleared by the compaction behind the retry launch.  nullptr: the tree fits the LDS rows / the primary stage is persistent.
    uint32_t *retry;
    uint32_t retry_cap;
    unsigned long long *totals;         // running sums over frames (rt_pipeline_get_totals); updated by the frame's last kernel
    float point_free;           // LightRays::point_free of pfc's point light
    // ONE shadow queue for the hits of every level (round 4; before: a queue per level, five loaders and five sinks unrolled into
    // the any-hit kernel, a division and a remainder per ray loaded and per result stored, and so many kernel arguments that the
    // scalar registers spilled into vector lanes: the any-hit stage -15 % in sets of frames, -24 % frame by frame).
    // Storage: level L's hits take the entries sh_cbase[L] + idx (idx = compact hit index; sh_cbase = the room of the levels
    // before it, lv[].hstride each: host constants, so the shading passes address their entries without reading a counter);
    // shadow ray s of that hit is ray number (sh_cbase[L] << sh_log2) + s * lv[L].hstride + idx -- per level all rays to the
    // directional light, then all rays to the point light: 64-ray chunks hold one kind of ray of consecutive hits.
    //   compact form (both light rays start at the hit point and are rebuilt by the loader): sh_hits[entry] = point + bits
    //   explicit form (the ambient-occlusion view's four random rays): sh_O / sh_D [ray number]
    //   results: sh_vis[ray number]
    // The any-hit launch enumerates only what is there (ShadowSrcN::load: round64(hits) rays per kind and level).
    float4 *sh_hits, *sh_O, *sh_D;
    uint32_t *sh_vis;
    uint32_t sh_log2;           // log2 of the shadow rays per hit: 1 (the two lights) or 2 (the ambient-occlusion view)
    uint32_t sh_levels;         // levels 0 .. sh_levels - 1 cast shadow rays (level 0 always has its entries)
    uint32_t sh_cbase[MAXD + 1];
    uint32_t *pools;            // chunk counters of the persistent launches: [1 + MAXD][RT_POOL_GROUPS], 128 B apart
    LevelDev lv[MAXD + 1];
};

constexpr size_t POOL_BYTES = (size_t)(1 + MAXD) * RT_POOL_GROUPS * RT_POOL_STRIDE * 4;      // shadow launch, levels 1..MAXD
constexpr size_t PRIMARY_POOL_WORDS = (size_t)RT_POOL_GROUPS * RT_POOL_STRIDE;      // the primary stage's own pool, behind the others (cleared by a fill in front of the stage)
constexpr size_t POOL_OFFSET_WORDS = 64;      // the pools start on a 256-B boundary after the scalar counters

// ray number of shadow ray s of hit idx of level L (storage order: see PipeDev::sh_*)
RT_DEV size_t sh_ray(const PipeDev &pd, int L, uint32_t idx, uint32_t s)
{
    return ((size_t)pd.sh_cbase[L] << pd.sh_log2) + (size_t)s * pd.lv[L].hstride + idx;
}
// the any-hit launch's answer for shadow ray number n (ShadowSinkN): bit n & 31 of word n >> 5, 1 = the ray reached its light
RT_DEV uint32_t shadow_bit(const PipeDev &pd, size_t n)
{
    return (pd.sh_vis[n >> 5] >> (uint32_t)(n & 31u)) & 1u;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + PBLOCK - 1) / PBLOCK); }

}  // namespace rtp

using namespace rtp;

// ---- host object ------------------------------------------------------------------------

struct rt_pipeline {
    rt_context *ctx = nullptr;
    uint32_t kind = RT_PIPELINE_PROGRESSIVE;
    DevBuf aov_own;                    // realtime: second output (indirect specular); the first lives in accum_own
    rt_scene *scene = nullptr;
    std::vector<rt_material_params> mats;
    DevBuf d_mats;
    bool mats_dirty = true;
    DevBuf d_env;
    uint32_t env_size = 0;
    uint32_t env_filter = RT_CUBE_SEAMLESS;
    float env_const[3] = {0.5f, 0.5f, 0.5f};
    uint32_t width = 0, height = 0, format = RT_FORMAT_R32G32B32A32_FLOAT;
    DevBuf accum_own;
    float4 *accum = nullptr;
    rt_per_frame_constants pfc;
    bool have_pfc = false;
    uint32_t max_rad = 1, max_shadow = 2, accum_mode = RT_ACCUM_RUNNING_MEAN, accum_f16 = 0;
    uint32_t skip_unlit = 0;           // off by default: every shadow ray the reference traces is traversed (rt_pipeline_set_skip_unlit_shadow_rays)
    // queues: every render call asks each buffer for what its launches need (DevBuf::reserve keeps what it has when that is
    // enough), so there is no second book of capacities that could disagree with the allocations after a failed growth
    size_t queue_budget = 0;           // worst-case queue bytes a set of launches may reserve up front; above it the levels are sized by
                                       //   count (0: a quarter of the device's memory, or the option queue_budget_mb)
    bool counted_queues = false;       // what the last set of launches did
    // Deferred rendering (rt_pipeline_set_deferred, progressive pipeline): render() only records the frame's constants; the
    // frames go through ONE set of launches when `deferred_max` of them have gathered or when anything asks for -- or changes --
    // what they produce.  Bit for bit the image of immediate rendering (rt_pipeline_render_batch's guarantee).
    uint32_t deferred_max = 0;         // 0 / 1: render() renders
    std::vector<rt_per_frame_constants> pending;
    // A flush that some OTHER call set off (rt_scene_add_model, rt_context_synchronize, ...: rt_context_flush_deferred) and that failed
    // does not abort that call: the frames are lost, the error waits here and is returned -- once -- by the next call on THIS
    // pipeline that renders, flushes or reads (round 5)
    int deferred_error = RT_OK;
    std::string deferred_error_msg;
    struct LevelBuf { DevBuf O, D, hit, inst, slot_j, jlist, pix, color; } lv[MAXD + 1];
    DevBuf sh_hits, sh_O, sh_D, sh_vis;      // the shared shadow queue (PipeDev::sh_*)
    DevBuf counters;
    DevBuf retry;                      // PipeDev::retry (2 + capacity words)
    DevBuf half_out;
    std::vector<hipEvent_t> ring;      // EV_COUNT events per remembered frame
    std::vector<uint8_t> ring_levels;  // radiance levels each remembered frame ran
    std::vector<uint8_t> ring_nframes; // frames each remembered entry covers (a batch is one entry)
    DevBuf batch_consts;               // per-frame constants and light rays of a batch (rt_pipeline_render_batch)
    int ring_frames = 0;               // 0 = timing off
    // the shadow cache (ShadowCacheDev): table, the scene it was filled from, the bounds its directional cells span
    DevBuf shadow_cache;
    int shadow_cache_res = -1;         // rt_pipeline_set_shadow_cache: -1 automatic (the option shadow_cache_res, else by triangle count), 0 off, n cells per side
    uint32_t shadow_cache_gen = 0xffffffffu;
    float shadow_cache_centre[3] = {0, 0, 0}, shadow_cache_radius = 1.0f;
    ShadowCacheDev shadow_cache_dev = {};      // what the next shadow launches get (table == nullptr: off)
    // the free sphere around the point light (LightRays::point_free): the least distance from the light to the box of any triangle
    // (instances: to any instance's world box), found by a device pass over them when the scene or the light has changed and read
    // back without a host round trip: page-locked word + event, used from the first render call that finds the event complete
    struct FreeSphere {
        DevBuf d_min;                      // one float (bits): running minimum of the pass in flight
        float *h_min = nullptr;            // page-locked landing place
        hipEvent_t landed = nullptr;
        uint32_t asked_gen = 0xffffffffu, known_gen = 0xffffffffu;
        float asked_lp[3] = {0, 0, 0}, known_lp[3] = {0, 0, 0}, known_radius = 0.0f;
        float asked_size = 0.0f;           // the largest coordinate of the scene's bounds when the pass was queued
        bool in_flight = false;
    } free_sphere;
    // How the primary stage runs on a single-level scene (round 5): one tile per wave without stack rows beyond LDS + the retry launch, unless
    // the retry list says that this scene's primary rays outgrow the LDS rows in numbers (the 10 M-triangle mesh: every retried ray is
    // walked twice, the second time out of order -- primary stage 1.03 -> 2.2 ms); then as a persistent launch with rows, like a two-level
    // scene's.  The list's count of a set of launches comes back through a page-locked word behind the set (never waited for); the first
    // sets of a scene are sampled, the decision holds until the scene changes.  Same image either way.
    struct PrimaryMode {
        uint32_t *h_count = nullptr;       // page-locked landing place of PipeDev::retry[0]
        hipEvent_t landed = nullptr;
        bool in_flight = false;
        uint32_t asked_slots = 0;          // pixel slots of the launch whose count is in flight
        uint32_t gen = 0xffffffffu;        // scene generation the samples belong to
        int samples = 0;                   // sets sampled so far for this scene
        bool persistent = false;           // the decision
    } primary_mode;
    uint64_t ring_pos = 0;             // frames recorded since enable / reset
    DevBuf totals, work;
    PipeDev last_pd;
    rt_stats stats;
    uint32_t last_tile[4] = {0, 0, 0, 0};
    uint32_t last_pixels = 0;
    bool rendered = false;
    uint32_t last_scene_gen = 0;       // generation of the scene last_pd was filled from
};

// renders the frames a deferred pipeline holds (rt_pipeline.hip); every entry point that reads or changes what they see calls it first
int rt_pipeline_flush_pending(rt_pipeline *p);

namespace rtp {


// events of one frame: start | primary | shade 0 | (trace l, shade l) for l = 1..MAXD | shadow | resolve
constexpr int EV_COUNT = 5 + 2 * MAXD;
constexpr int EV_SHADOW = 3 + 2 * MAXD, EV_RESOLVE = 4 + 2 * MAXD;

}  // namespace rtp
