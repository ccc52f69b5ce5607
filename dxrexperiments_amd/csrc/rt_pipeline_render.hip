// rt_pipeline_render.hip -- the render calls of the pipeline: one frame over a region (tile, bands), sets of frames
// (rt_pipeline_render_batch, rt_pipeline_render_bands_batch), deferred mode behind update() + render(), queue-memory
// reservation and reporting, the host side of the shadow cache and of the free sphere around the point light, and the exports that
// replay the last frame's queues (work counting, primary hits).  The kernels and their launch sequence: rt_pipeline.hip.
#include <hip/hip_fp16.h>

#include <array>
#include <new>
#include <utility>

#include "rt_pipeline_queues.h"

using namespace rtd;

namespace {

// How far from a point light the scene certainly is empty: its shadow rays may stop that far short of the light (QueueSrc::load_lit) --
// nothing can occlude them inside that sphere, and all of them would otherwise walk the nodes around the light, where they converge.
// A lower bound is enough: the least distance from the light to the bounding box of any triangle (single-level scenes; to the world box
// of any instance otherwise), times 0.99, minus RAY_EPSILON.  One device pass over the triangle records per (scene, light position),
// queued behind the frame that first sees the pair and read back through page-locked memory: the frames until it has landed run
// with no sphere (a scene that changes every frame never gets one, and never waits for one).  Bench scene: 3.3 of a 32 x 11 x 14
// atrium, shadow stage -3 % (profiles/r03/free_radius.txt).
__global__ void __launch_bounds__(PBLOCK) k_free_sphere(SceneDev sc, uint32_t two_level, uint32_t n, float lx, float ly, float lz, uint32_t *out_bits)
{
    const uint32_t i = blockIdx.x * PBLOCK + threadIdx.x;
    float d2 = __uint_as_float(0x7f800000u);
    if (i < n) {
        float lo[3], hi[3];
        if (two_level) {
            const InstanceRec &in = sc.inst[i];
            for (int c = 0; c < 3; c++) { lo[c] = in.wlo[c]; hi[c] = in.whi[c]; }
        } else {
            const TriRec t = sc.inst[0].tris[i];
            const float vx[3] = {t.a.x, t.a.w, t.b.z}, vy[3] = {t.a.y, t.b.x, t.b.w}, vz[3] = {t.a.z, t.b.y, t.c.x};
            lo[0] = fminf(vx[0], fminf(vx[1], vx[2])); hi[0] = fmaxf(vx[0], fmaxf(vx[1], vx[2]));
            lo[1] = fminf(vy[0], fminf(vy[1], vy[2])); hi[1] = fmaxf(vy[0], fmaxf(vy[1], vy[2]));
            lo[2] = fminf(vz[0], fminf(vz[1], vz[2])); hi[2] = fmaxf(vz[0], fmaxf(vz[1], vz[2]));
        }
        const float l[3] = {lx, ly, lz};
        float s = 0.0f;
        bool known = true;
        for (int c = 0; c < 3; c++) {
            const float d = l[c] < lo[c] ? lo[c] - l[c] : l[c] > hi[c] ? l[c] - hi[c] : 0.0f;
            known = known && lo[c] == lo[c] && hi[c] == hi[c];
            s += d * d;
        }
        // (a box with a NaN in it says nothing about where its triangle is: the sphere has no room then.  A triangle with a NaN
        // CORNER meets no ray, but fminf / fmaxf may have hidden the NaN: not worth telling apart)
        d2 = known ? s : 0.0f;
    }
    for (int o = 32; o > 0; o >>= 1) d2 = fminf(d2, __shfl_xor(d2, o, 64));
    if ((threadIdx.x & 63u) == 0u) atomicMin(out_bits, __float_as_uint(d2));          // (non-negative floats order like their bits)
}

// a pass that has landed becomes what is known
void free_sphere_poll(rt_pipeline::FreeSphere &f)
{
    if (f.in_flight && hipEventQuery(f.landed) == hipSuccess) {
        f.in_flight = false;
        f.known_gen = f.asked_gen;
        memcpy(f.known_lp, f.asked_lp, sizeof f.known_lp);
        // (0.99 and the term in the coordinates' size -- 32 units in the last place of the largest one -- cover the rounding of
        // the rays' own arithmetic: a ray's parameter at a triangle is exact to a few ulps of the coordinates, not of the radius)
        double size = 0.0;
        for (int c = 0; c < 3; c++) size = fmax(size, fmax(fabs((double)f.asked_lp[c]), (double)f.asked_size));
        const double r = sqrt((double)*f.h_min) * 0.99 - (double)RAY_EPSILON - size * 4e-6;
        f.known_radius = r > 0.0 && r < 1e30 ? (float)r : 0.0f;
    }
}

float free_radius(rt_pipeline *p, const PipeDev &pd, const float lp[3])
{
    rt_pipeline::FreeSphere &f = p->free_sphere;
    const rt_scene *s = p->scene;
    hipStream_t st = p->ctx->stream;
    if (!(lp[0] == lp[0] && lp[1] == lp[1] && lp[2] == lp[2])) return 0.0f;
    if (!f.h_min) {
        if (hipHostMalloc((void **)&f.h_min, 64, hipHostMallocDefault) != hipSuccess) { f.h_min = nullptr; return 0.0f; }
        if (hipEventCreateWithFlags(&f.landed, hipEventDisableTiming) != hipSuccess || f.d_min.reserve(64) != RT_OK) return 0.0f;
    }
    free_sphere_poll(f);
    const bool known = f.known_gen == s->generation && memcmp(f.known_lp, lp, sizeof f.known_lp) == 0;
    const bool asked = f.in_flight && f.asked_gen == s->generation && memcmp(f.asked_lp, lp, sizeof f.asked_lp) == 0;
    if (!known && !asked && !f.in_flight) {
        const uint32_t n = s->two_level ? (uint32_t)s->inst.size() : (s->inst.empty() || !s->inst[0].model ? 0u : s->inst[0].model->n_recs);
        if (n == 0u || !f.landed) return 0.0f;
        if (hipMemsetD32Async((hipDeviceptr_t)f.d_min.p, 0x7f800000, 1, st) != hipSuccess) return 0.0f;      // +inf
        k_free_sphere<<<(n + PBLOCK - 1) / PBLOCK, PBLOCK, 0, st>>>(pd.sc, s->two_level ? 1u : 0u, n, lp[0], lp[1], lp[2], f.d_min.as<uint32_t>());
        if (hipMemcpyAsync(f.h_min, f.d_min.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(f.landed, st) != hipSuccess) return 0.0f;
        f.in_flight = true;
        f.asked_gen = s->generation;
        const float *bb = s->two_level ? s->tlas.bounds : s->inst[0].model->blas.bounds;
        f.asked_size = 0.0f;
        for (int c = 0; c < 6; c++) f.asked_size = bb[c] == bb[c] ? fmaxf(f.asked_size, fabsf(bb[c])) : __builtin_inff();
        memcpy(f.asked_lp, lp, sizeof f.asked_lp);
    }
    return known ? f.known_radius : 0.0f;
}

// The shadow cache of the coming launches (ShadowCacheDev): single-level scenes only (an entry is an index into the one sorted
// triangle array); the table is cleared when the scene has changed (an index must stay inside the array), its directional cells
// span the bounding sphere of the model, its basis follows the frame's light (entries of another direction are merely stale).
int prepare_shadow_cache(rt_pipeline *p, const rt_per_frame_constants &pfc, const LightRays &lr)
{
    p->shadow_cache_dev = ShadowCacheDev{};
    const rt_scene *s = p->scene;
    if (p->shadow_cache_res == 0 || s->inst.empty()) return RT_OK;
    size_t n_tris_all = 0;
    for (const SceneInstance &si : s->inst) {
        if (!si.model || si.model->n_tris >= (1u << 28)) return RT_OK;
        n_tris_all += si.model->n_tris;
    }
    int want = p->shadow_cache_res;
    if (want < 0) {                         // not set through the API: the context's option, else by the size of the triangles
        want = p->ctx->opt_shadow_cache_res;
        if (want == 0) return RT_OK;
    }
    if (want < 0) {
        // cells well below the triangles' size: 8 per sqrt(triangle count) across the scene, a power of two in [1024, 8192]
        // (bench scene, 262 k triangles: 1024 / 2048 / 4096 / 8192 cells -> 1.80 / 1.74 / 1.70 / 1.72 ms per frame, off 1.87;
        //  10 M triangles: 2048 -> 11.9, 8192 -> 10.9 ms, off 12.0; profiles/r03/shadow_cache*.txt)
        const double target = 8.0 * sqrt((double)n_tris_all);
        want = 1024;
        while (want < 8192 && (double)want < target) want *= 2;
    }
    if (want > 8192) want = 8192;
    if (want < 16) want = 16;
    const uint32_t res = (uint32_t)want & ~1u;
    // (+ two entries per pixel slot of the output for the primary hits' rays: ShadowCacheDev::px_base.  Measured, profiles/r04/
    // shadow_cache_pixels.txt: two-level scenes -7 % on the any-hit stage (4096 instances: 4.52 -> 4.40 ms); single-level scenes +3 % --
    // their primary hits' rays no longer seed the light-space cells the secondary hits' rays read.  So: on for two-level scenes;
    // the option shadow_cache_pixels = 0 / 1 overrides)
    const int per_pixel_opt = p->ctx->opt_shadow_cache_pixels;
    const bool per_pixel = per_pixel_opt < 0 ? s->two_level : per_pixel_opt != 0;
    const size_t cells = (size_t)res * res + 6u * (size_t)(res / 2) * (res / 2);
    const size_t px_slots = per_pixel ? (size_t)((p->width + 7u) / 8u) * ((p->height + 7u) / 8u) * 64u : 0;
    const size_t entries = cells + 2 * px_slots, entry_bytes = s->two_level ? 8 : 4;
    if (entries >= 0xffffffffull) return RT_OK;
    if (p->shadow_cache.bytes < entries * entry_bytes) { RT_TRY(p->shadow_cache.reserve(entries * entry_bytes)); p->shadow_cache_gen = 0xffffffffu; }
    if (lr.on == 0xffffffffu) return RT_OK;            // (rt_pipeline_reserve_batch: the allocation only)
    if (p->shadow_cache_gen != s->generation) {
        HIP_TRY(hipMemsetAsync(p->shadow_cache.p, 0xff, entries * entry_bytes, p->ctx->stream));
        // the world bounds: what the builds brought back (the model's own box for one identity instance, the TLAS's otherwise)
        const float *bb = s->two_level ? s->tlas.bounds : s->inst[0].model->blas.bounds;
        float lo[3], hi[3];
        for (int c = 0; c < 3; c++) {
            lo[c] = bb[c]; hi[c] = bb[3 + c];
            if (!(lo[c] > -1e30f && hi[c] < 1e30f && lo[c] <= hi[c])) { lo[c] = -1.0f; hi[c] = 1.0f; }      // (any cell size is legal)
        }
        float r2 = 0.0f;
        for (int c = 0; c < 3; c++) { p->shadow_cache_centre[c] = 0.5f * (lo[c] + hi[c]); const float h = 0.5f * (hi[c] - lo[c]); r2 += h * h; }
        p->shadow_cache_radius = r2 > 0.0f && r2 < 1e30f ? sqrtf(r2) : 1.0f;
        p->shadow_cache_gen = s->generation;
    }
    ShadowCacheDev c = {};
    c.table = p->shadow_cache.as<uint32_t>();
    c.res = res; c.res_f = (float)res;
    c.two_level = s->two_level ? 1u : 0u;
    c.n_tris = s->two_level ? 0u : s->inst[0].model->n_recs;
    c.entries = (uint32_t)entries;
    c.px_base = px_slots ? (uint32_t)cells : 0u;
    c.px_slots = (uint32_t)px_slots;                 // (launch_frame fills in what belongs to the launch: jlist0, hstride0, the set's frames)
    // two unit vectors across the direction to the light
    const float d[3] = {lr.dir_to_light[0], lr.dir_to_light[1], lr.dir_to_light[2]};
    const float ref[3] = {fabsf(d[1]) < 0.9f ? 0.0f : 1.0f, fabsf(d[1]) < 0.9f ? 1.0f : 0.0f, 0.0f};
    float a[3] = {d[1] * ref[2] - d[2] * ref[1], d[2] * ref[0] - d[0] * ref[2], d[0] * ref[1] - d[1] * ref[0]};
    const float al = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (!(al > 1e-6f)) { a[0] = 1.0f; a[1] = 0.0f; a[2] = 0.0f; } else { a[0] /= al; a[1] /= al; a[2] /= al; }
    const float b[3] = {d[1] * a[2] - d[2] * a[1], d[2] * a[0] - d[0] * a[2], d[0] * a[1] - d[1] * a[0]};
    const float k = 0.5f * (float)res / p->shadow_cache_radius;
    c.ua[3] = 0.5f * (float)res; c.va[3] = 0.5f * (float)res;
    for (int i = 0; i < 3; i++) {
        c.ua[i] = a[i] * k; c.va[i] = b[i] * k;
        c.ua[3] -= p->shadow_cache_centre[i] * c.ua[i];
        c.va[3] -= p->shadow_cache_centre[i] * c.va[i];
        c.lp[i] = lr.point_pos[i];
    }
    (void)pfc;
    p->shadow_cache_dev = c;
    return RT_OK;
}

// The scene as a set of n_frames frames over `cap` pixel slots walks it: which kernels (LDS stack rows) and the global stack rows
// behind them -- (tree bound - LDS rows) x 4 B per thread of the largest launch, and the primary stage of a single-level scene runs
// one thread per pixel slot: 4.3 GB for 20 frames of 1080p.  Insurance that is never touched on the bench scene, but an allocation:
// rt_pipeline_reserve_batch makes it too (round 4: on some boxes that hipMalloc took 126 ms of the first set's render() call).
int scene_for_set(rt_pipeline *p, uint32_t n_frames, size_t cap, SceneDev *out, bool *sets_kernels = nullptr)
{
    rt_context *ctx = p->ctx;
    const bool seven_waves_always = ctx->opt_seven_waves_always;      // (experiment: single frames on the sets' kernels)
    // (a scene with split triangles keeps the 18-row kernels in sets too: rt_frame_launch)
    const bool set_rows = (n_frames > 1 || seven_waves_always) && !p->scene->two_level && !p->scene->has_refs && ctx->lds_stack_rows != RT_LDS_STACK_ROWS_TEST && RT_LDS_STACK_ROWS_SETS != RT_LDS_STACK_ROWS;
    // (round 5: rows for the threads of the PERSISTENT launches only -- at most eight 256-thread workgroups per CU fit their LDS, sixteen
    // is the option's limit; the one-tile-per-wave primary launch, one thread per pixel slot, keeps none: PipeDev::retry)
    const size_t resident = (size_t)ctx->cu_count * (ctx->blocks_per_cu_override > (uint32_t)RT_RESIDENT_BLOCKS_PER_CU ? ctx->blocks_per_cu_override : (uint32_t)RT_RESIDENT_BLOCKS_PER_CU) * PBLOCK;      // (rt_persistent_grid clamps its launches to the same count)
    (void)cap;
    if (sets_kernels) *sets_kernels = set_rows;
    return rt_scene_dev_for_launch(ctx, p->scene, set_rows ? RT_LDS_STACK_ROWS_SETS : rt_lds_stack_rows(ctx), resident, out);
}

}  // namespace

extern "C" {

// one frame over the rectangle [x0,x1) x [y0,y1); band_rows != 0: over the interleaved bands {b : b mod band_world == band_rank}
// of band_rows rows each (the rectangle then spans the full width and the rank's rows, packed)
// frames / n_frames: the constants of the frames this set of launches renders (one: the last rt_pipeline_update)
static int render_region(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                         uint32_t band_rows, uint32_t band_rank, uint32_t band_world, const rt_per_frame_constants *frames = nullptr, uint32_t n_frames = 1)
{
    RT_REQUIRE(p, "null pipeline");
    if (!p->scene || !p->scene->built) { rt_set_error("render: acceleration structures not built"); return RT_ERR_STATE; }
    if (!p->accum) { rt_set_error("render: no output resource"); return RT_ERR_STATE; }
    if (!frames && !p->have_pfc) { rt_set_error("render: update() has not been called"); return RT_ERR_STATE; }
    if (!frames) { frames = &p->pfc; n_frames = 1; }
    RT_REQUIRE(n_frames >= 1 && n_frames <= RT_MAX_BATCH, "batch size");
    if (p->mats.empty()) { rt_set_error("render: no material"); return RT_ERR_STATE; }
    RT_REQUIRE(width == p->width && height == p->height, "width/height differ from the output resource");
    if (x1 > width) x1 = width;
    if (!band_rows && y1 > height) y1 = height;      // (band mode: y counts the rank's packed rows, checked per pixel)
    RT_REQUIRE(x0 < x1 && y0 < y1, "empty tile");
    rt_context *ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    p->rendered = false;
    // RayGen early-out (ProgressiveRaytracing.hlsl:14-16): nothing is traced or written (batches: the caller has dropped such frames)
    if (p->kind == RT_PIPELINE_PROGRESSIVE && frames[0].cameraParams.accumCount >= frames[0].options.maxIterations) {
        memset(&p->stats, 0, sizeof p->stats);
        return RT_OK;
    }
    if (p->mats_dirty) {
        RT_TRY(p->d_mats.reserve(sizeof(rt_material_params) * p->mats.size()));
        HIP_TRY(hipMemcpyAsync(p->d_mats.p, p->mats.data(), sizeof(rt_material_params) * p->mats.size(), hipMemcpyHostToDevice, st));
        p->mats_dirty = false;
    }
    const uint32_t tw = x1 - x0, th = y1 - y0;
    const uint32_t tiles_x = (tw + 7u) / 8u, fcap = tiles_x * ((th + 7u) / 8u) * 64u;
    RT_REQUIRE((uint64_t)fcap * n_frames < 0x40000000ull, "batch: more than 2^30 pixel slots in one set of launches");
    const uint32_t cap = fcap * n_frames;
    const bool ao_view = p->kind == RT_PIPELINE_PROGRESSIVE && frames[0].options.showAmbientOcclusionOnly;
    const uint32_t shadow_slots = ao_view ? 4u : 2u;
    // queue memory: the worst case up front when it fits the budget, else level by level as the counts come in (launch_frame)
    const uint32_t levels_now = frame_levels(p);
    const bool counted = worst_case_queue_bytes(cap, levels_now, p->max_shadow, shadow_slots, !ao_view) > queue_budget(p);
    if (p->counters.bytes < POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4) {
        RT_TRY(p->counters.reserve(POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4));
        HIP_TRY(hipMemsetAsync(p->counters.p, 0, p->counters.bytes, st));
    }
    if (counted) RT_TRY(reserve_level_rays(p, 0, cap, levels_now > 1));
    else RT_TRY(reserve_worst_case(p, cap, levels_now, p->max_shadow, shadow_slots, !ao_view));
    p->counted_queues = counted;
    if (!p->totals.p) {
        RT_TRY(p->totals.reserve(8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), st));
    }
    PipeDev pd;
    // (threads of the largest launch: the primary stage runs one thread per pixel slot, the persistent stages fewer)
    bool set_rows = false;
    RT_TRY(scene_for_set(p, n_frames, cap, &pd.sc, &set_rows));
    pd.pfc = frames[0];
    pd.n_frames = n_frames; pd.fcap = fcap;
    pd.pfcs = nullptr; pd.frame_lights = nullptr;
    pd.shadow_compact = ao_view ? 0u : 1u;          // the AO view's four rays have random directions
    // (the light buffer is keyed by the two lights: the AO view's random rays do not use it)
    const bool free_on = ctx->opt_free_radius;
    {
        const float lp0[3] = {frames[0].pointLight.worldPos.x, frames[0].pointLight.worldPos.y, frames[0].pointLight.worldPos.z};
        pd.point_free = free_on ? free_radius(p, pd, lp0) : 0.0f;
    }
    if (ao_view) p->shadow_cache_dev = ShadowCacheDev{};
    else RT_TRY(prepare_shadow_cache(p, frames[0], light_rays(1u, frames[0])));
    if (n_frames > 1) {
        // the batch's constants and light rays go to device memory: kernels index them by the frame of a slot
        const size_t cb = sizeof(rt_per_frame_constants) * RT_MAX_BATCH, lb = sizeof(LightRays) * RT_MAX_BATCH;
        RT_TRY(p->batch_consts.reserve(cb + lb));
        std::vector<unsigned char> stage(cb + lb, 0);
        for (uint32_t f = 0; f < n_frames; f++) {
            memcpy(&stage[sizeof(rt_per_frame_constants) * f], &frames[f], sizeof(rt_per_frame_constants));
            LightRays lr = light_rays(pd.shadow_compact, frames[f]);
            // (frames whose point light is where the first frame's is share its sphere; a frame with another light gets none)
            lr.point_free = free_on && memcmp(lr.point_pos, &frames[0].pointLight.worldPos, 3 * sizeof(float)) == 0 ? pd.point_free : 0.0f;
            memcpy(&stage[cb + sizeof(LightRays) * f], &lr, sizeof lr);
        }
        HIP_TRY(hipMemcpyAsync(p->batch_consts.p, stage.data(), cb + lb, hipMemcpyHostToDevice, st));     // (pageable source: staged before the call returns)
        pd.pfcs = p->batch_consts.as<rt_per_frame_constants>();
        pd.frame_lights = (const LightRays *)((const char *)p->batch_consts.p + cb);
    }
    pd.mats = p->d_mats.as<rt_material_params>();
    pd.nmats = (uint32_t)p->mats.size();
    pd.env = p->d_env.as<float4>();
    pd.env_size = p->env_size;
    pd.env_filter = p->env_filter;
    for (int k = 0; k < 3; k++) pd.env_const[k] = p->env_const[k];
    pd.width = width; pd.height = height;
    pd.x0 = x0; pd.y0 = y0; pd.tw = tw; pd.th = th; pd.cap = cap; pd.tiles_x = tiles_x;
    pd.band_rows = band_rows; pd.band_rank = band_rank; pd.band_world = band_world;
    uint32_t owned_rows = th;
    if (band_rows) {            // rows of the rank's bands that lie inside the image (the last band may be short)
        owned_rows = 0;
        for (uint32_t b = band_rank; (uint64_t)b * band_rows < height; b += band_world)
            owned_rows += (uint64_t)(b + 1) * band_rows <= height ? band_rows : height - b * band_rows;
    }
    pd.n_pixels = tw * owned_rows;                  // (per frame)
    pd.max_rad = p->max_rad; pd.max_shadow = p->max_shadow;
    pd.sh_log2 = ao_view ? 2u : 1u;
    pd.sh_levels = shadow_levels(levels_now, p->max_shadow);
    for (int k = 0; k <= MAXD; k++) pd.sh_cbase[k] = 0;
    pd.accum_mode = p->accum_mode;
    pd.accum_f16 = p->accum_f16;
    pd.skip_unlit = p->skip_unlit;
    pd.kind = p->kind;
    // The primary stage as a persistent launch that refills its lanes from a pool of tiles (instead of one tile per wave, dealt by the
    // hardware): pays where the rays of a tile part ways early -- two-level scenes, whose primary waves run at 0.48 of their lanes
    // (4096 instances at 4K: 1.39 -> 1.31 ms) -- and costs 13 % where they stay together (the single-level bench scene: 0.69 of the lanes
    // as it is).  The option primary_persistent = 0 / 1 overrides.  profiles/r04/c4_variants.txt, primary_persistent.txt
    {   // (single-level scenes: one tile per wave, unless this scene's retry lists said otherwise -- rt_pipeline::PrimaryMode)
        rt_pipeline::PrimaryMode &pm = p->primary_mode;
        if (pm.gen != p->scene->generation) { pm.gen = p->scene->generation; pm.samples = 0; pm.persistent = false; pm.in_flight = false; }
        if (pm.in_flight && hipEventQuery(pm.landed) == hipSuccess) {
            pm.in_flight = false;
            pm.samples++;
            if ((double)*pm.h_count > 0.002 * (double)pm.asked_slots) pm.persistent = true;       // more than 0.2 % of the primary rays
        }
        pd.primary_persistent = (ctx->opt_primary_persistent < 0 ? (p->scene->two_level || pm.persistent) : ctx->opt_primary_persistent != 0) ? 1u : 0u;
    }
    pd.retry = nullptr;
    pd.retry_cap = 0;
    if (pd.sc.deep_stack && !pd.primary_persistent) {      // the primary launch keeps no rows beyond LDS: the list of the rays that would need one
        const uint32_t want = ctx->opt_primary_retry_cap ? ctx->opt_primary_retry_cap : (1u << 20);
        if (p->retry.bytes < (2 + (size_t)want) * 4) {
            RT_TRY(p->retry.reserve((2 + (size_t)want) * 4));
            HIP_TRY(hipMemsetAsync(p->retry.p, 0, 8, st));
        }
        pd.retry = p->retry.as<uint32_t>();
        pd.retry_cap = want;
    }
    pd.accum = p->accum;
    pd.aov_direct = p->accum;                       // realtime: output 0 = direct lighting, output 1 = indirect specular
    pd.aov_indirect = p->aov_own.as<float4>();
    pd.counters = p->counters.as<uint32_t>();
    for (int l = 0; l <= MAXD; l++) {          // (counted queues: launch_frame binds a level again once it has sized it)
        bind_level(p, pd, l);
        pd.lv[l].rstride = cap;
        pd.lv[l].hstride = l == 0 ? cap : 2u * cap;
    }
    static_assert(C_COUNT <= POOL_OFFSET_WORDS, "scalar counters overlap the chunk pools");
    pd.pools = pd.counters + POOL_OFFSET_WORDS;
    pd.totals = p->totals.as<unsigned long long>();
    RT_TRY(rt_frame_launch(p, pd, shadow_slots, counted, set_rows));
    HIP_TRY(hipGetLastError());
    if (pd.retry && !p->primary_mode.in_flight && p->primary_mode.samples < 8 && ctx->opt_primary_persistent < 0) {
        // this set's retry count, for the next sets' choice of primary launch (the set's compaction has cleared retry[0] and left the count in retry[1])
        rt_pipeline::PrimaryMode &pm = p->primary_mode;
        if (!pm.h_count) {
            if (hipHostMalloc((void **)&pm.h_count, 64, hipHostMallocDefault) != hipSuccess) pm.h_count = nullptr;
            else if (hipEventCreateWithFlags(&pm.landed, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(pm.h_count); pm.h_count = nullptr; pm.landed = nullptr; }
        }
        if (pm.h_count && hipMemcpyAsync(pm.h_count, pd.retry + 1, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(pm.landed, st) == hipSuccess) {
            pm.in_flight = true;
            pm.asked_slots = cap;
        }
    }
    p->last_pd = pd;
    p->last_scene_gen = p->scene->generation;
    p->last_tile[0] = x0; p->last_tile[1] = y0; p->last_tile[2] = x1; p->last_tile[3] = y1;
    p->last_pixels = pd.n_pixels * n_frames;
    p->rendered = true;
    return RT_OK;
}

// n frames through shared sets of launches; band_rows != 0: only the rank's interleaved bands of every frame
static int render_frames(rt_pipeline *p, uint32_t width, uint32_t height, const rt_per_frame_constants *constants, uint32_t n,
                         uint32_t band_rows, uint32_t band_rank, uint32_t band_world)
{
    uint32_t band_count = 0;
    if (band_rows) {
        RT_TRY(rt_tile_bands(height, band_rows, band_rank, band_world, nullptr, nullptr, 0, &band_count));
        if (band_count == 0) n = 0;             // more ranks than bands: nothing to render here (the constants still become current)
    }
    uint32_t batch_max = RT_MAX_BATCH;
    if (p->ctx->opt_batch_max >= 1 && p->ctx->opt_batch_max <= RT_MAX_BATCH) batch_max = p->ctx->opt_batch_max;
    // frames RayGen would leave at once (accumCount >= maxIterations, ProgressiveRaytracing.hlsl:14-16) are dropped here;
    // frames that disagree on what sizes the queues (the ambient-occlusion view) do not share a set of launches
    std::vector<rt_per_frame_constants> run;
    auto flush = [&]() -> int {
        // the queues grow with the batch: when the device cannot hold them, the same frames go through in smaller sets
        for (size_t at = 0; at < run.size();) {
            const size_t n_now = run.size() - at < batch_max ? run.size() - at : batch_max;
            const int rc = band_rows ? render_region(p, width, height, 0, 0, width, band_count * band_rows, band_rows, band_rank, band_world, run.data() + at, (uint32_t)n_now)
                                     : render_region(p, width, height, 0, 0, width, height, 0, 0, 1, run.data() + at, (uint32_t)n_now);
            if (rc == RT_ERR_OOM && n_now > 1) { batch_max = (uint32_t)(n_now / 2); continue; }
            if (rc != RT_OK) { run.clear(); return rc; }
            at += n_now;
        }
        run.clear();
        return RT_OK;
    };
    for (uint32_t i = 0; i < n; i++) {
        const rt_per_frame_constants &c = constants[i];
        if (c.cameraParams.accumCount >= c.options.maxIterations) continue;
        if (!run.empty() && (run.size() >= batch_max || (run[0].options.showAmbientOcclusionOnly != 0) != (c.options.showAmbientOcclusionOnly != 0))) RT_TRY(flush());
        run.push_back(c);
    }
    return flush();
}

}  // extern "C"

static int flush_pending_now(rt_pipeline *p)
{
    if (p->pending.empty()) return RT_OK;
    std::vector<rt_per_frame_constants> frames;
    frames.swap(p->pending);                    // (whatever happens, the frames are not rendered twice)
    std::vector<rt_pipeline *> &reg = p->ctx->deferred;
    for (size_t k = 0; k < reg.size(); k++) if (reg[k] == p) { reg.erase(reg.begin() + (long)k); break; }
    const rt_per_frame_constants keep = p->pfc;              // the constants of the last update(): a frame may have been updated and not rendered yet
    const int rc = render_frames(p, p->width, p->height, frames.data(), (uint32_t)frames.size(), 0, 0, 1);
    p->pfc = keep;
    return rc;
}

// every entry point of the pipeline that renders, reads or changes what recorded frames would see calls this first
int rt_pipeline_flush_pending(rt_pipeline *p)
{
    if (!p) return RT_OK;
    if (p->deferred_error != RT_OK) {           // a flush behind somebody else's call failed: this pipeline's caller hears of it here, once
        const int rc = p->deferred_error;
        rt_set_error("deferred frames were lost: %s", p->deferred_error_msg.c_str());
        p->deferred_error = RT_OK;
        p->deferred_error_msg.clear();
        return rc;
    }
    return flush_pending_now(p);
}

// Calls that are not about one pipeline (scene changes, rt_context_synchronize, the collectives) render what every deferred
// pipeline of the context still holds.  A failure there belongs to the pipeline, not to the caller's own work: it is parked on
// the pipeline (see deferred_error) and the call goes on.
int rt_context_flush_deferred(rt_context *ctx)
{
    if (!ctx) return RT_OK;
    while (!ctx->deferred.empty()) {
        rt_pipeline *p = ctx->deferred.back();
        const int rc = flush_pending_now(p);      // (a flush takes the pipeline off the list)
        if (rc != RT_OK && p->deferred_error == RT_OK) { p->deferred_error = rc; p->deferred_error_msg = rt_last_error(); }
    }
    return RT_OK;
}

extern "C" {

int rt_pipeline_flush(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    return rt_pipeline_flush_pending(p);
}

int rt_pipeline_set_deferred(rt_pipeline *p, uint32_t max_frames)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(max_frames <= RT_MAX_BATCH, "set_deferred: at most 32 frames share a set of launches");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE || max_frames <= 1, "set_deferred: only the progressive pipeline accumulates frames");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->deferred_max = max_frames;
    return RT_OK;
}

int rt_pipeline_get_deferred(const rt_pipeline *p, uint32_t *max_frames, uint32_t *pending)
{
    RT_REQUIRE(p, "null pipeline");
    if (max_frames) *max_frames = p->deferred_max;
    if (pending) *pending = (uint32_t)p->pending.size();
    return RT_OK;
}

int rt_pipeline_render_tile(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    RT_TRY(rt_pipeline_flush_pending(p));
    return render_region(p, width, height, x0, y0, x1, y1, 0, 0, 1);
}

int rt_pipeline_render_bands(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world)
{
    RT_REQUIRE(world > 0 && rank < world, "rank outside [0, world)");
    RT_REQUIRE(band_rows > 0 && band_rows % 8 == 0, "band_rows must be a positive multiple of 8 (pixel slots are 8x8 tiles)");
    RT_TRY(rt_pipeline_flush_pending(p));
    uint32_t n = 0;
    RT_TRY(rt_tile_bands(height, band_rows, rank, world, nullptr, nullptr, 0, &n));
    if (n == 0) return RT_OK;                   // more ranks than bands: nothing to render here
    return render_region(p, width, height, 0, 0, width, n * band_rows, band_rows, rank, world);
}

int rt_pipeline_render(rt_pipeline *p, uint32_t width, uint32_t height)
{
    RT_REQUIRE(p, "null pipeline");
    if (p->deferred_max > 1 && p->kind == RT_PIPELINE_PROGRESSIVE) {
        // the checks render_region would make now, so that a bad call fails where it is made and not at some later flush
        if (!p->scene || !p->scene->built) { rt_set_error("render: acceleration structures not built"); return RT_ERR_STATE; }
        if (!p->accum) { rt_set_error("render: no output resource"); return RT_ERR_STATE; }
        if (!p->have_pfc) { rt_set_error("render: update() has not been called"); return RT_ERR_STATE; }
        if (p->mats.empty()) { rt_set_error("render: no material"); return RT_ERR_STATE; }
        RT_REQUIRE(width == p->width && height == p->height, "width/height differ from the output resource");
        if (p->pending.empty()) p->ctx->deferred.push_back(p);
        p->pending.push_back(p->pfc);
        p->rendered = false;                    // (nothing of the LAST frame is on the device yet: count_work / stats flush first)
        if (p->pending.size() >= p->deferred_max) return rt_pipeline_flush_pending(p);
        return RT_OK;
    }
    return rt_pipeline_render_tile(p, width, height, 0, 0, width, height);
}

int rt_pipeline_render_batch(rt_pipeline *p, uint32_t width, uint32_t height, const rt_per_frame_constants *constants, uint32_t n)
{
    RT_REQUIRE(p && (constants || n == 0), "null argument");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "render_batch: only the progressive pipeline accumulates frames");
    RT_TRY(rt_pipeline_flush_pending(p));
    RT_TRY(render_frames(p, width, height, constants, n, 0, 0, 1));
    if (n) { p->pfc = constants[n - 1]; p->have_pfc = true; }     // as after n x (update, render)
    return RT_OK;
}

int rt_pipeline_render_bands_batch(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world,
                                   const rt_per_frame_constants *constants, uint32_t n)
{
    RT_REQUIRE(p && (constants || n == 0), "null argument");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "render_bands_batch: only the progressive pipeline accumulates frames");
    RT_REQUIRE(world > 0 && rank < world, "rank outside [0, world)");
    RT_REQUIRE(band_rows > 0 && band_rows % 8 == 0, "band_rows must be a positive multiple of 8 (pixel slots are 8x8 tiles)");
    RT_TRY(rt_pipeline_flush_pending(p));
    RT_TRY(render_frames(p, width, height, constants, n, band_rows, rank, world));
    if (n) { p->pfc = constants[n - 1]; p->have_pfc = true; }
    return RT_OK;
}

int rt_pipeline_set_queue_budget(rt_pipeline *p, size_t bytes)
{
    RT_REQUIRE(p, "null pipeline");
    p->queue_budget = bytes;
    return RT_OK;
}

int rt_pipeline_get_queue_memory(rt_pipeline *p, size_t *bytes_reserved, uint32_t *sized_by_count)
{
    RT_REQUIRE(p, "null pipeline");
    size_t total = p->counters.bytes + p->batch_consts.bytes + p->sh_hits.bytes + p->sh_O.bytes + p->sh_D.bytes + p->sh_vis.bytes;
    for (const rt_pipeline::LevelBuf &l : p->lv) {
        const DevBuf *lb[] = {&l.O, &l.D, &l.hit, &l.inst, &l.slot_j, &l.jlist, &l.pix, &l.color};
        for (const DevBuf *b : lb) total += b->bytes;
    }
    if (bytes_reserved) *bytes_reserved = total;
    if (sized_by_count) *sized_by_count = p->counted_queues ? 1u : 0u;
    return RT_OK;
}

int rt_pipeline_reserve_batch(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t frames)
{
    RT_REQUIRE(p && width > 0 && height > 0 && frames >= 1 && frames <= RT_MAX_BATCH, "reserve_batch: bad argument");
    HIP_TRY(hipSetDevice(p->ctx->device));
    const uint32_t fcap = ((width + 7u) / 8u) * ((height + 7u) / 8u) * 64u;
    RT_REQUIRE((uint64_t)fcap * frames < 0x40000000ull, "batch: more than 2^30 pixel slots in one set of launches");
    const bool ao_view = p->have_pfc && p->pfc.options.showAmbientOcclusionOnly != 0;
    // (a set whose worst case is over the budget sizes its levels by count as it goes: only the pixel slots are known now)
    const uint32_t levels_now = frame_levels(p);
    const size_t cap = (size_t)fcap * frames;
    // Any buffer this call grows is a buffer the last frame's replay calls (count_work, count_walk, read_primary_hits, ...) still hold
    // pointers to in last_pd: a reservation therefore ends what was "rendered" (the calls then refuse instead of
    // launching on freed memory).  Frames a deferred pipeline still holds are rendered first: they own the queues as they are.
    RT_TRY(rt_pipeline_flush_pending(p));
    p->rendered = false;
    if (p->counters.bytes < POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4) {
        RT_TRY(p->counters.reserve(POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4));
        HIP_TRY(hipMemsetAsync(p->counters.p, 0, p->counters.bytes, p->ctx->stream));
    }
    if (worst_case_queue_bytes(cap, levels_now, p->max_shadow, ao_view ? 4u : 2u, !ao_view) > queue_budget(p)) RT_TRY(reserve_level_rays(p, 0, cap, levels_now > 1));
    else RT_TRY(reserve_worst_case(p, cap, levels_now, p->max_shadow, ao_view ? 4u : 2u, !ao_view));
    if (frames > 1) RT_TRY(p->batch_consts.reserve((sizeof(rt_per_frame_constants) + sizeof(LightRays)) * RT_MAX_BATCH));
    if (p->scene && p->scene->built) {                 // the traversal kernels' global stack rows and the shadow cache's table as well
        SceneDev sc;
        RT_TRY(scene_for_set(p, frames, cap, &sc));
        LightRays only_allocate = no_light_rays();
        only_allocate.on = 0xffffffffu;
        RT_TRY(prepare_shadow_cache(p, p->pfc, only_allocate));
        p->shadow_cache_dev = ShadowCacheDev{};
    }
    return RT_OK;
}

int rt_pipeline_get_free_sphere(rt_pipeline *p, float *radius)
{
    RT_REQUIRE(p && radius, "get_free_sphere: null argument");
    *radius = 0.0f;
    rt_pipeline::FreeSphere &f = p->free_sphere;
    if (!p->scene || !p->have_pfc || !f.landed) return RT_OK;
    HIP_TRY(hipSetDevice(p->ctx->device));
    free_sphere_poll(f);
    const float *lp = &p->pfc.pointLight.worldPos.x;
    if (f.known_gen == p->scene->generation && memcmp(f.known_lp, lp, sizeof f.known_lp) == 0) *radius = f.known_radius;
    return RT_OK;
}

int rt_pipeline_count_work(rt_pipeline *p, rt_stage_work *out)
{
    RT_REQUIRE(p && out, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->rendered || !p->scene->built || p->scene->generation != p->last_scene_gen) { rt_set_error("count_work: nothing rendered since the last change of scene, materials or output"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    RT_TRY(p->work.reserve(RT_STAGE_COUNT * RT_WALK_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, RT_STAGE_COUNT * 3 * sizeof(unsigned long long), st));
    unsigned long long *w = p->work.as<unsigned long long>();
    RT_TRY(rt_frame_count_work(p, w));
    unsigned long long h[RT_STAGE_COUNT * 3];
    HIP_TRY(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < RT_STAGE_COUNT; k++) { out[k].rays = h[3 * k]; out[k].nodes = h[3 * k + 1]; out[k].tris = h[3 * k + 2]; }
    return RT_OK;
}

int rt_pipeline_count_walk(rt_pipeline *p, rt_stage_walk *out)
{
    RT_REQUIRE(p && out, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->rendered || !p->scene->built || p->scene->generation != p->last_scene_gen) { rt_set_error("count_walk: nothing rendered since the last change of scene, materials or output"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    const size_t bytes = RT_STAGE_COUNT * RT_WALK_WORDS * sizeof(unsigned long long);
    RT_TRY(p->work.reserve(bytes));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, bytes, st));
    unsigned long long *w = p->work.as<unsigned long long>();
    RT_TRY(rt_frame_count_walk(p, w));
    unsigned long long h[RT_STAGE_COUNT * RT_WALK_WORDS];
    HIP_TRY(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < RT_STAGE_COUNT; k++) {
        const unsigned long long *hk = h + RT_WALK_WORDS * k;
        out[k].rays = hk[0]; out[k].nodes_global = hk[1]; out[k].nodes_lds = hk[2];
        out[k].tris = hk[3]; out[k].instance_entries = hk[4]; out[k].lines = hk[5];
        out[k].longest_walk = hk[6] >> 32;
        out[k].longest_walk_ray = (uint32_t)hk[6];
        out[k].wave_node_steps = hk[7]; out[k].wave_leaf_phases = hk[8]; out[k].wave_tri_steps = hk[9]; out[k].node_lines = hk[10];
    }
    return RT_OK;
}

// debugging aid for rt_stage_walk.longest_walk_ray: ray `index` of the level-1 radiance queue (diffuse batch, then specular batch)
int rt_debug_read_secondary_ray(rt_pipeline *p, uint32_t index, float origin_tmin[4], float dir_tmax[4])
{
    RT_REQUIRE(p && origin_tmin && dir_tmax, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->rendered) { rt_set_error("nothing rendered yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    uint32_t n = 0;
    HIP_TRY(hipMemcpy(&n, p->last_pd.counters + C_NHIT, 4, hipMemcpyDeviceToHost));
    RT_REQUIRE(n > 0 && index < 2 * n, "ray index out of range");
    const size_t slot = (size_t)(index / n) * p->last_pd.lv[1].rstride + index % n;
    HIP_TRY(hipMemcpy(origin_tmin, p->last_pd.lv[1].O + slot, 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dir_tmax, p->last_pd.lv[1].D + slot, 16, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_pipeline_read_primary_hits(rt_pipeline *p, float *t, uint32_t *prim, uint32_t *inst)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->rendered) { rt_set_error("nothing rendered yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    const PipeDev &pd = p->last_pd;
    const size_t cap = pd.cap;
    std::vector<float4> h(cap);
    std::vector<uint32_t> hi(cap);
    HIP_TRY(hipMemcpy(h.data(), p->lv[0].hit.p, cap * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hi.data(), p->lv[0].inst.p, cap * 4, hipMemcpyDeviceToHost));
    for (size_t q = 0; q < cap; q++) {          // slots are 8x8-tiled: scatter back to scanline order
        const uint32_t tl = (uint32_t)(q >> 6), w = (uint32_t)(q & 63u);
        const uint32_t lx = (tl % pd.tiles_x) * 8u + (w & 7u), ly = (tl / pd.tiles_x) * 8u + (w >> 3);
        if (lx >= pd.tw || ly >= pd.th) continue;
        const size_t i = (size_t)ly * pd.tw + lx;
        if (t) t[i] = h[q].x;
        if (prim) memcpy(&prim[i], &h[q].w, 4);
        if (inst) inst[i] = hi[q];
    }
    return RT_OK;
}

int rt_debug_sample_cube(rt_context *ctx, const float *faces, uint32_t size, uint32_t filter, const float *dirs, float *out, size_t n)
{
    RT_REQUIRE(ctx && faces && dirs && out && size > 0, "bad argument");
    RT_REQUIRE(filter == RT_CUBE_SEAMLESS || filter == RT_CUBE_FACE_CLAMP, "unknown cube-map filter");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf *sb = ctx->scratch;
    const size_t fb = (size_t)6 * size * size * 16;
    RT_TRY(sb[0].reserve(fb)); RT_TRY(sb[1].reserve(n * 12)); RT_TRY(sb[2].reserve(n * 12));
    HIP_TRY(hipMemcpyAsync(sb[0].p, faces, fb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(sb[1].p, dirs, n * 12, hipMemcpyHostToDevice, ctx->stream));
    PipeDev pd;
    memset(&pd, 0, sizeof pd);
    pd.env = sb[0].as<float4>();
    pd.env_size = size;
    pd.env_filter = filter;
    RT_TRY(rt_frame_debug_cube(ctx->stream, pd, sb[1].as<float>(), sb[2].as<float>(), n));
    HIP_TRY(hipMemcpyAsync(out, sb[2].p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
