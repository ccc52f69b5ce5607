// rt_pipeline_queues.h -- host-side helpers the two translation units of the pipeline share (rt_pipeline.hip: the frame's kernels and
// launches; rt_pipeline_render.hip: the render calls): the frame's light rays as the host computes them, and the queue memory -- what a
// level's buffers hold, the worst case and the budget, reservation by count, binding a level's buffers into the kernels' argument.
#pragma once

#include "rt_pipeline_dev.h"

namespace rtp {

inline LightRays light_rays(uint32_t shadow_compact, const rt_per_frame_constants &pfc)
{
    LightRays l;
    l.on = shadow_compact;
    const rt_float4 f = pfc.directionalLight.forwardDir, w = pfc.pointLight.worldPos;
    const float x = -f.x, y = -f.y, z = -f.z;              // normalize(): v * (1 / sqrt(dot(v, v))), dot summed left to right
    float d = x * x;
    d += y * y;
    d += z * z;
    const float inv = 1.0f / sqrtf(d);
    l.dir_to_light[0] = x * inv; l.dir_to_light[1] = y * inv; l.dir_to_light[2] = z * inv;
    l.point_pos[0] = w.x; l.point_pos[1] = w.y; l.point_pos[2] = w.z;
    l.point_free = 0.0f;
    return l;
}
inline LightRays light_rays(const PipeDev &pd) { LightRays l = light_rays(pd.shadow_compact, pd.pfc); l.point_free = pd.point_free; return l; }
inline LightRays no_light_rays()
{
    LightRays l;
    memset(&l, 0, sizeof l);
    return l;
}


// ---- queue memory ------------------------------------------------------------------------------------------------------
// Level l keeps, per RAY slot (l = 0: per pixel slot; no ray is stored there), the ray (32 B + 4 B pixel slot), its hit record
// (16 + 4 B) and the two compaction maps (4 + 4 B); per HIT of the level its shadow rays -- compact form: ONE float4 per hit +
// one BIT of visibility per shadow ray (round 6; 4 B until then); explicit form (the ambient-occlusion view): 32 B + a bit per shadow ray; all levels in ONE queue -- and, for paths of more
// than one bounce, 16 B of colour per ray slot.  How many slots a level needs is only known once the level before has been
// compacted: the worst case is 2 rays per pixel at level 1 and as many rays as slots at every deeper level, 228 B per pixel and
// frame for the reference's depth limits and 884 B with four bounces -- times up to 32 frames per set of launches.  So:
//   * when the worst case of the whole set fits the budget (a quarter of the device's memory by default) everything is
//     reserved up front and a set is enqueued without the host ever looking at the device (single frames always go this way);
//   * above it the levels are sized BY COUNT: after the compaction of level l the host reads that one counter (a stream
//     synchronisation: ~20 us against the tens of milliseconds of a set) and sizes the shadow queue of level l and the ray
//     queue of level l + 1 for what is really there.  A 4K four-bounce frame of the 10 M-triangle scene needs 2.7 instead of
//     7.3 GB that way, and sets of 32 fit.
// Buffers only grow (with an eighth to spare, so that the next set's slightly different counts do not reallocate).
inline size_t round64(size_t n) { return (n + 63u) & ~(size_t)63u; }
inline int grow(DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return RT_OK;
    return b.reserve(bytes + bytes / 8);
}
// ray queue + hit records of level l for `slots` ray slots (level 0: pixel slots)
inline int reserve_level_rays(rt_pipeline *p, uint32_t l, size_t slots, bool deep)
{
    rt_pipeline::LevelBuf &b = p->lv[l];
    if (l > 0) { RT_TRY(grow(b.O, slots * 16)); RT_TRY(grow(b.D, slots * 16)); RT_TRY(grow(b.pix, slots * 4)); }
    RT_TRY(grow(b.hit, slots * 16)); RT_TRY(grow(b.inst, slots * 4));
    RT_TRY(grow(b.slot_j, slots * 4)); RT_TRY(grow(b.jlist, slots * 4));
    if (l > 0 && deep) RT_TRY(grow(b.color, slots * 16));          // deep paths only (k_shade_level)
    return RT_OK;
}
// the shared shadow queue for `hits` hits (of all levels together) with 1 << log2 rays each; the first `keep` hits' entries
// survive a growth (counted queues: earlier levels have written theirs when a later level turns out to need more room)
inline int grow_keep(DevBuf &b, size_t bytes, size_t keep_bytes, hipStream_t st)
{
    if (bytes <= b.bytes) return RT_OK;
    DevBuf bigger;
    RT_TRY(bigger.reserve(bytes + bytes / 8));
    if (keep_bytes && b.p) {
        if (hipMemcpyAsync(bigger.p, b.p, keep_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rt_set_error("shadow queue: copy into the grown buffer failed: %s", hipGetErrorString(hipGetLastError()));
            bigger.release();
            return RT_ERR_HIP;
        }
    }
    b.release();
    b = bigger;
    return RT_OK;
}
inline int reserve_shadows(rt_pipeline *p, size_t hits, uint32_t log2, bool compact, size_t keep_hits)
{
    hipStream_t st = p->ctx->stream;
    if (compact) RT_TRY(grow_keep(p->sh_hits, hits * 16, keep_hits * 16, st));
    else { RT_TRY(grow_keep(p->sh_O, (hits << log2) * 16, (keep_hits << log2) * 16, st)); RT_TRY(grow_keep(p->sh_D, (hits << log2) * 16, (keep_hits << log2) * 16, st)); }
    return grow_keep(p->sh_vis, (((hits << log2) + 31) / 32) * 4 + 64, 0, st);         // (results, one bit per ray: nothing is in there before the shadow launch)
}
// levels 0 .. n - 1 cast shadow rays: level 0 always has its entries (their masks are empty when no shadow ray is allowed at all)
inline uint32_t shadow_levels(uint32_t levels, uint32_t max_shadow) { return 1u + (max_shadow > 1u ? (levels < max_shadow - 1u ? levels : max_shadow - 1u) : 0u); }
inline size_t level_ray_bytes(uint32_t l, bool deep) { return (l > 0 ? 36u : 0u) + 28u + (l > 0 && deep ? 16u : 0u); }
inline size_t level_shadow_bytes(uint32_t shadow_slots, bool compact) { return (compact ? 16u : 32u * shadow_slots) + 1u; }      // (+ one bit of visibility per ray: a byte per hit at most)
inline size_t worst_case_queue_bytes(size_t cap, uint32_t levels, uint32_t max_shadow, uint32_t shadow_slots0, bool compact)
{
    const bool deep = levels > 1;
    size_t total = cap * level_ray_bytes(0, deep);
    for (uint32_t l = 1; l <= levels; l++) total += 2 * cap * level_ray_bytes(l, deep);
    return total + (cap + 2 * cap * (shadow_levels(levels, max_shadow) - 1u)) * level_shadow_bytes(shadow_slots0, compact);
}
inline size_t queue_budget(rt_pipeline *p)
{
    if (p->queue_budget) return p->queue_budget;
    if (p->ctx->opt_queue_budget_mb) return p->ctx->opt_queue_budget_mb << 20;
    if (p->ctx->device_mem_total == 0) {         // asked once per context: hipMemGetInfo is a driver round trip, this runs per frame
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
        p->ctx->device_mem_total = total_b;
    }
    return p->ctx->device_mem_total / 4;
}
// everything a set of `cap` pixel slots can need at most, reserved now; the strides that go with it
inline int reserve_worst_case(rt_pipeline *p, size_t cap, uint32_t levels, uint32_t max_shadow, uint32_t shadow_slots0, bool compact)
{
    const bool deep = levels > 1;
    RT_TRY(reserve_level_rays(p, 0, cap, deep));
    for (uint32_t l = 1; l <= levels; l++) RT_TRY(reserve_level_rays(p, l, 2 * cap, deep));
    return reserve_shadows(p, cap + 2 * cap * (shadow_levels(levels, max_shadow) - 1u), shadow_slots0 > 2u ? 2u : 1u, compact, 0);
}
inline void bind_level(const rt_pipeline *p, PipeDev &pd, int l)
{
    const rt_pipeline::LevelBuf &b = p->lv[l];
    LevelDev &d = pd.lv[l];
    d.O = b.O.as<float4>(); d.D = b.D.as<float4>(); d.hit = b.hit.as<float4>(); d.inst = b.inst.as<uint32_t>();
    d.slot_j = b.slot_j.as<uint32_t>(); d.jlist = b.jlist.as<uint32_t>(); d.pix = b.pix.as<uint32_t>();
    d.color = b.color.as<float4>();
    pd.sh_hits = p->sh_hits.as<float4>(); pd.sh_O = p->sh_O.as<float4>(); pd.sh_D = p->sh_D.as<float4>(); pd.sh_vis = p->sh_vis.as<uint32_t>();
}

// radiance levels a frame traces: level l exists when hits of depth l-1 may spawn rays
inline uint32_t frame_levels(const rt_pipeline *p) { return p->max_rad < (uint32_t)MAXD ? p->max_rad : (uint32_t)MAXD; }


}  // namespace rtp

// ---- what rt_pipeline.hip (kernels, launches) offers rt_pipeline_render.hip (the render calls) ----
// the launches of one frame or one set of frames (set_rows: the sets' seven-wave single-level kernels); pd is updated as levels are bound
int rt_frame_launch(rt_pipeline *p, rtp::PipeDev &pd, uint32_t shadow_slots, bool counted, bool set_rows);
// the counting re-walks over the last frame's queues (p->last_pd): launches only, results in w[RT_STAGE_COUNT][RT_WALK_WORDS] / [3]
int rt_frame_count_walk(rt_pipeline *p, unsigned long long *w);
int rt_frame_count_work(rt_pipeline *p, unsigned long long *w);
int rt_frame_debug_cube(hipStream_t st, const rtp::PipeDev &pd, const float *dirs, float *out, size_t n);
