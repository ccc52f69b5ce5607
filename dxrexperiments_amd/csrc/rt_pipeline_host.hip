// rt_pipeline_host.hip -- the part of the pipeline object that launches no traversal or shading kernel: creation and the
// setters of the reference's interface (include/RaytracingPipeline.h:14-38), outputs and their read-back (RGBA32F and the
// RGBA16F view of the reference's storage format), accumulation checkpoints, stage timing and ray statistics.
// The frame itself (kernels, launches, render calls, work counting): rt_pipeline.hip.
#include <hip/hip_fp16.h>

#include <new>

#include "rt_pipeline_dev.h"

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size);

namespace {

__global__ void k_f32_to_f16(const float4 *__restrict__ in, ushort4 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    ushort4 o;
    o.x = __half_as_ushort(__float2half_rn(v.x));
    o.y = __half_as_ushort(__float2half_rn(v.y));
    o.z = __half_as_ushort(__float2half_rn(v.z));
    o.w = __half_as_ushort(__float2half_rn(v.w));
    out[i] = o;
}

}  // namespace

extern "C" {

int rt_pipeline_create(rt_context *ctx, uint32_t kind, rt_pipeline **out)
{
    RT_REQUIRE(ctx && out, "null argument");
    RT_REQUIRE(kind == RT_PIPELINE_PROGRESSIVE || kind == RT_PIPELINE_REALTIME, "unknown pipeline kind");
    rt_pipeline *p = new (std::nothrow) rt_pipeline();
    if (!p) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    p->ctx = ctx;
    p->kind = kind;
    rt_context_retain(ctx);
    memset(&p->pfc, 0, sizeof p->pfc);
    memset(&p->stats, 0, sizeof p->stats);
    *out = p;
    return RT_OK;
}

int rt_pipeline_destroy(rt_pipeline *p)
{
    if (!p) return RT_OK;
    (void)hipSetDevice(p->ctx->device);
    {   // Frames accepted and not rendered yet: an output in caller memory (rt_pipeline_bind_output) outlives the pipeline, so
        // they are rendered into it first; with an output of its own nothing could read them any more and they die here.
        if (!p->pending.empty() && p->accum && p->accum != p->accum_own.as<float4>()) (void)rt_pipeline_flush_pending(p);
        std::vector<rt_pipeline *> &reg = p->ctx->deferred;
        for (size_t k = 0; k < reg.size(); k++) if (reg[k] == p) { reg.erase(reg.begin() + (long)k); break; }
        p->pending.clear();
    }
    (void)hipStreamSynchronize(p->ctx->stream);
    DevBuf *all[] = {&p->d_mats, &p->d_env, &p->accum_own, &p->aov_own, &p->counters, &p->half_out, &p->totals, &p->work, &p->batch_consts, &p->shadow_cache,
                     &p->sh_hits, &p->sh_O, &p->sh_D, &p->sh_vis};
    for (DevBuf *b : all) b->release();
    for (rt_pipeline::LevelBuf &l : p->lv) {
        DevBuf *lb[] = {&l.O, &l.D, &l.hit, &l.inst, &l.slot_j, &l.jlist, &l.pix, &l.color};
        for (DevBuf *b : lb) b->release();
    }
    for (hipEvent_t e : p->ring) if (e) (void)hipEventDestroy(e);
    if (p->free_sphere.landed) (void)hipEventDestroy(p->free_sphere.landed);
    if (p->primary_mode.landed) (void)hipEventDestroy(p->primary_mode.landed);
    if (p->primary_mode.h_count) (void)hipHostFree(p->primary_mode.h_count);
    p->retry.release();
    if (p->free_sphere.h_min) (void)hipHostFree(p->free_sphere.h_min);
    p->free_sphere.d_min.release();
    if (p->scene) rt_scene_destroy(p->scene);
    rt_context *ctx = p->ctx;
    delete p;
    rt_context_release(ctx);
    return RT_OK;
}

const char *rt_pipeline_get_name(const rt_pipeline *p)
{
    // include/ProgressiveRaytracingPipeline.h:40, include/RealtimeRaytracingPipeline.h:40
    return p && p->kind == RT_PIPELINE_REALTIME ? "Realtime Ray Tracing Pipeline" : "Progressive Ray Tracing Pipeline";
}

int rt_pipeline_set_scene(rt_pipeline *p, rt_scene *s)
{
    RT_REQUIRE(p && s, "null argument");
    RT_REQUIRE(s->ctx == p->ctx, "scene belongs to a different context");
    RT_TRY(rt_pipeline_flush_pending(p));
    rt_scene_retain(s);
    if (p->scene) rt_scene_destroy(p->scene);
    p->scene = s;
    p->rendered = false;        // last_pd holds device pointers of the previous scene
    p->shadow_cache_gen = 0xffffffffu;      // ... and the shadow cache triangle indices of its arrays (another scene may carry the same generation number)
    p->free_sphere.known_gen = p->free_sphere.asked_gen = 0xffffffffu;      // ... and the free sphere its geometry (a pass in flight lands unused)
    return RT_OK;
}

int rt_pipeline_add_material(rt_pipeline *p, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->mats.push_back(*m);
    p->mats_dirty = true;
    p->rendered = false;        // d_mats may be reallocated by the next render
    return RT_OK;
}

int rt_pipeline_set_material(rt_pipeline *p, uint32_t index, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    RT_REQUIRE(index < p->mats.size(), "material index out of range");
    // (the reference rewrites every hit record every frame, libs/DXRFramework/RtBindings.cpp:100-129, and so does the C++ mirror:
    // a record that says what it said before changes nothing -- no upload, and no flush of a deferred set)
    if (memcmp(&p->mats[index], m, sizeof *m) == 0) return RT_OK;
    RT_TRY(rt_pipeline_flush_pending(p));
    p->mats[index] = *m;
    p->mats_dirty = true;
    p->rendered = false;
    return RT_OK;
}

int rt_pipeline_set_environment_cube(rt_pipeline *p, const float *faces, uint32_t size)
{
    RT_REQUIRE(p && faces && size > 0, "bad argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    const size_t bytes = (size_t)6 * size * size * 16;
    RT_TRY(p->d_env.reserve(bytes));
    HIP_TRY(hipMemcpyAsync(p->d_env.p, faces, bytes, hipMemcpyHostToDevice, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->env_size = size;
    return RT_OK;
}

int rt_pipeline_set_environment_constant(rt_pipeline *p, const float rgb[3])
{
    RT_REQUIRE(p && rgb, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->env_size = 0;
    for (int k = 0; k < 3; k++) p->env_const[k] = rgb[k];
    return RT_OK;
}

int rt_pipeline_set_environment_filter(rt_pipeline *p, uint32_t filter)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(filter == RT_CUBE_SEAMLESS || filter == RT_CUBE_FACE_CLAMP, "unknown cube-map filter");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->env_filter = filter;
    return RT_OK;
}

int rt_pipeline_load_environment_dds(rt_pipeline *p, const char *path)
{
    RT_REQUIRE(p && path, "null argument");
    std::vector<float> faces;
    uint32_t size = 0;
    RT_TRY(rt_dds_load_cube(path, faces, size));
    return rt_pipeline_set_environment_cube(p, faces.data(), size);
}

int rt_pipeline_create_output(rt_pipeline *p, uint32_t format, uint32_t width, uint32_t height)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(width > 0 && height > 0, "empty output");
    RT_REQUIRE(format == RT_FORMAT_R32G32B32A32_FLOAT || format == RT_FORMAT_R16G16B16A16_FLOAT, "unsupported output format");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    RT_TRY(p->accum_own.reserve((size_t)width * height * 16));
    if (p->kind == RT_PIPELINE_REALTIME) RT_TRY(p->aov_own.reserve((size_t)width * height * 16));     // kNumOutputResources = 2
    p->accum = p->accum_own.as<float4>();
    p->width = width; p->height = height; p->format = format;
    p->rendered = false;
    return rt_pipeline_clear_output(p);
}

int rt_pipeline_bind_output(rt_pipeline *p, void *device_rgba32f, uint32_t width, uint32_t height)
{
    RT_REQUIRE(p && device_rgba32f, "null argument");
    RT_REQUIRE(width > 0 && height > 0, "empty output");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "bind_output: only the progressive pipeline renders into caller memory");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->accum = (float4 *)device_rgba32f;
    p->width = width; p->height = height; p->format = RT_FORMAT_R32G32B32A32_FLOAT;
    p->rendered = false;
    return RT_OK;
}

int rt_pipeline_build_acceleration_structures(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->scene) { rt_set_error("buildAccelerationStructures: no scene set"); return RT_ERR_STATE; }
    if (p->scene->built) return RT_OK;       // built once, shared between pipelines
    p->rendered = false;                     // a rebuild reallocates what last_pd points at
    return rt_scene_build(p->scene, 2);
}

int rt_pipeline_set_depth_limits(rt_pipeline *p, uint32_t max_radiance_depth, uint32_t max_shadow_depth)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (max_radiance_depth > (uint32_t)MAXD) {
        rt_set_error("max radiance depth %u: the wavefront DAG holds at most %d radiance levels", max_radiance_depth, MAXD);
        return RT_ERR_UNSUPPORTED;
    }
    p->max_rad = max_radiance_depth;
    p->max_shadow = max_shadow_depth;
    return RT_OK;
}

int rt_pipeline_set_shadow_cache(rt_pipeline *p, int cells_per_side)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(cells_per_side >= -1 && cells_per_side <= 8192, "shadow cache: cells per side in [16, 8192], 0 = off, -1 = automatic");
    p->shadow_cache_res = cells_per_side;
    p->shadow_cache_gen = 0xffffffffu;          // (a table of another size starts empty)
    return RT_OK;
}

int rt_pipeline_get_shadow_cache(const rt_pipeline *p, int *cells_per_side)
{
    RT_REQUIRE(p && cells_per_side, "null argument");
    *cells_per_side = p->shadow_cache_dev.table ? (int)p->shadow_cache_dev.res : 0;       // what the last frame ran with
    return RT_OK;
}

int rt_pipeline_set_skip_unlit_shadow_rays(rt_pipeline *p, int on)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->skip_unlit = on ? 1u : 0u;
    return RT_OK;
}

int rt_pipeline_set_accumulation_mode(rt_pipeline *p, uint32_t mode)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(mode == RT_ACCUM_RUNNING_MEAN || mode == RT_ACCUM_SUM, "unknown accumulation mode");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->accum_mode = mode;
    return RT_OK;
}

int rt_pipeline_set_accumulation_storage(rt_pipeline *p, uint32_t format, uint32_t rounding)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(format == RT_FORMAT_R32G32B32A32_FLOAT || format == RT_FORMAT_R16G16B16A16_FLOAT, "unsupported accumulation storage format");
    RT_REQUIRE(rounding == RT_ROUND_NEAREST_EVEN || rounding == RT_ROUND_TOWARD_ZERO, "unknown rounding");
    RT_TRY(rt_pipeline_flush_pending(p));
    p->accum_f16 = format == RT_FORMAT_R16G16B16A16_FLOAT ? (rounding == RT_ROUND_TOWARD_ZERO ? 2u : 1u) : 0u;
    return RT_OK;
}

int rt_pipeline_clear_output(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->accum) { rt_set_error("no output resource"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipMemsetAsync(p->accum, 0, (size_t)p->width * p->height * 16, p->ctx->stream));
    if (p->aov_own.p) HIP_TRY(hipMemsetAsync(p->aov_own.p, 0, (size_t)p->width * p->height * 16, p->ctx->stream));
    return RT_OK;
}

int rt_pipeline_update(rt_pipeline *p, const rt_per_frame_constants *constants)
{
    RT_REQUIRE(p && constants, "null argument");
    p->pfc = *constants;
    p->have_pfc = true;
    return RT_OK;
}

int rt_pipeline_get_num_outputs(const rt_pipeline *p, int *n)
{
    RT_REQUIRE(p && n, "null argument");
    *n = p->kind == RT_PIPELINE_REALTIME ? 2 : 1;
    return RT_OK;
}

int rt_pipeline_get_output_device_ptr(rt_pipeline *p, uint32_t id, void **ptr)
{
    RT_REQUIRE(p && ptr, "null argument");
    RT_REQUIRE(id < (p->kind == RT_PIPELINE_REALTIME ? 2u : 1u), "output index out of range");
    RT_TRY(rt_pipeline_flush_pending(p));
    *ptr = id == 0 ? (void *)p->accum : p->aov_own.p;
    return RT_OK;
}

int rt_pipeline_read_output_n(rt_pipeline *p, uint32_t id, void *host, size_t bytes)
{
    RT_REQUIRE(p && host, "null argument");
    RT_REQUIRE(id < (p->kind == RT_PIPELINE_REALTIME ? 2u : 1u), "output index out of range");
    RT_TRY(rt_pipeline_flush_pending(p));
    const float4 *src = id == 0 ? p->accum : p->aov_own.as<float4>();
    if (!src) { rt_set_error("no output resource"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    const size_t npix = (size_t)p->width * p->height;
    hipStream_t st = p->ctx->stream;
    if (p->format == RT_FORMAT_R16G16B16A16_FLOAT) {
        RT_REQUIRE(bytes == npix * 8, "host buffer must be width*height*8 bytes for RGBA16F");
        RT_TRY(p->half_out.reserve(npix * 8));
        k_f32_to_f16<<<blocks(npix), PBLOCK, 0, st>>>(src, p->half_out.as<ushort4>(), npix);
        HIP_TRY(hipMemcpyAsync(host, p->half_out.p, bytes, hipMemcpyDeviceToHost, st));
    } else {
        RT_REQUIRE(bytes == npix * 16, "host buffer must be width*height*16 bytes for RGBA32F");
        HIP_TRY(hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_pipeline_read_output(rt_pipeline *p, void *host, size_t bytes) { return rt_pipeline_read_output_n(p, 0, host, bytes); }

int rt_pipeline_write_output(rt_pipeline *p, const void *host_rgba32f, size_t bytes)
{
    RT_REQUIRE(p && host_rgba32f, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->accum) { rt_set_error("no output resource"); return RT_ERR_STATE; }
    RT_REQUIRE(bytes == (size_t)p->width * p->height * 16, "host buffer must be width*height*16 bytes (RGBA32F)");
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipMemcpyAsync(p->accum, host_rgba32f, bytes, hipMemcpyHostToDevice, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    return RT_OK;
}

// checkpoint file: "DXRACCUM1\n", u32 width, u32 height, u64 state bytes, host state, width*height float4
static const char kCheckpointMagic[10] = {'D', 'X', 'R', 'A', 'C', 'C', 'U', 'M', '1', '\n'};

int rt_pipeline_save_checkpoint(rt_pipeline *p, const rt_progressive_host *h, const char *path)
{
    RT_REQUIRE(p && path, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->accum) { rt_set_error("no output resource"); return RT_ERR_STATE; }
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "checkpoint: only the progressive pipeline accumulates");
    HIP_TRY(hipSetDevice(p->ctx->device));
    const size_t bytes = (size_t)p->width * p->height * 16;
    std::vector<char> img(bytes), state;
    HIP_TRY(hipMemcpyAsync(img.data(), p->accum, bytes, hipMemcpyDeviceToHost, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    size_t sb = 0;
    if (h) {
        RT_TRY(rt_progressive_host_save_state(h, nullptr, 0, &sb));
        state.resize(sb);
        RT_TRY(rt_progressive_host_save_state(h, state.data(), sb, &sb));
    }
    FILE *f = fopen(path, "wb");
    if (!f) { rt_set_error("checkpoint: cannot create %s", path); return RT_ERR_IO; }
    const uint32_t wh[2] = {p->width, p->height};
    const uint64_t sb64 = sb;
    bool ok = fwrite(kCheckpointMagic, 1, sizeof kCheckpointMagic, f) == sizeof kCheckpointMagic && fwrite(wh, 4, 2, f) == 2 &&
              fwrite(&sb64, 8, 1, f) == 1 && (sb == 0 || fwrite(state.data(), 1, sb, f) == sb) && fwrite(img.data(), 1, bytes, f) == bytes;
    ok = (fclose(f) == 0) && ok;
    if (!ok) { rt_set_error("checkpoint: short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

int rt_pipeline_load_checkpoint(rt_pipeline *p, rt_progressive_host *h, const char *path)
{
    RT_REQUIRE(p && path, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->accum) { rt_set_error("no output resource"); return RT_ERR_STATE; }
    FILE *f = fopen(path, "rb");
    if (!f) { rt_set_error("checkpoint: cannot open %s", path); return RT_ERR_IO; }
    char magic[sizeof kCheckpointMagic];
    uint32_t wh[2] = {0, 0};
    uint64_t sb = 0;
    int rc = RT_OK;
    std::vector<char> state, img;
    do {
        if (fread(magic, 1, sizeof magic, f) != sizeof magic || memcmp(magic, kCheckpointMagic, sizeof magic) != 0 || fread(wh, 4, 2, f) != 2 ||
            fread(&sb, 8, 1, f) != 1 || sb > (1u << 20)) { rt_set_error("checkpoint: %s is not an accumulation checkpoint", path); rc = RT_ERR_IO; break; }
        if (wh[0] != p->width || wh[1] != p->height) {
            rt_set_error("checkpoint: %s holds a %ux%u image, the output is %ux%u", path, wh[0], wh[1], p->width, p->height);
            rc = RT_ERR_INVALID_ARG;
            break;
        }
        state.resize((size_t)sb);
        img.resize((size_t)wh[0] * wh[1] * 16);
        if ((sb && fread(state.data(), 1, (size_t)sb, f) != sb) || fread(img.data(), 1, img.size(), f) != img.size()) {
            rt_set_error("checkpoint: %s is truncated", path);
            rc = RT_ERR_IO;
        }
    } while (0);
    fclose(f);
    if (rc != RT_OK) return rc;
    if (h && sb) RT_TRY(rt_progressive_host_load_state(h, state.data(), (size_t)sb));
    return rt_pipeline_write_output(p, img.data(), img.size());
}

int rt_pipeline_enable_timing(rt_pipeline *p, int frames)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(frames >= 0 && frames <= 4096, "timing ring holds 0..4096 frames");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    for (hipEvent_t e : p->ring) if (e) (void)hipEventDestroy(e);
    p->ring.assign((size_t)frames * EV_COUNT, nullptr);
    p->ring_levels.assign((size_t)frames, 0);
    p->ring_nframes.assign((size_t)frames, 1);
    for (hipEvent_t &e : p->ring) HIP_TRY(hipEventCreate(&e));
    p->ring_frames = frames;
    p->ring_pos = 0;
    return RT_OK;
}

// ms: primary | shade 0 | secondary traces (all levels) | 0 | secondary shades (all levels) | shadow | resolve | total
static int stage_times(rt_pipeline *p, uint64_t frame, float ms[8])
{
    const size_t slot = (size_t)(frame % (uint64_t)p->ring_frames);
    hipEvent_t *ev = &p->ring[slot * EV_COUNT];
    const int levels = p->ring_levels[slot];
    for (int k = 0; k < 8; k++) ms[k] = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms[1], ev[1], ev[2]));
    int last = 2;
    for (int l = 1; l <= levels; l++) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, ev[last], ev[3 + 2 * (l - 1)]));
        ms[2] += t;
        HIP_TRY(hipEventElapsedTime(&t, ev[3 + 2 * (l - 1)], ev[4 + 2 * (l - 1)]));
        ms[4] += t;
        last = 4 + 2 * (l - 1);
    }
    HIP_TRY(hipEventElapsedTime(&ms[5], ev[last], ev[EV_SHADOW]));
    HIP_TRY(hipEventElapsedTime(&ms[6], ev[EV_SHADOW], ev[EV_RESOLVE]));
    HIP_TRY(hipEventElapsedTime(&ms[7], ev[0], ev[EV_RESOLVE]));
    return RT_OK;
}

static void add_times(rt_stats *out, const float ms[8])
{
    out->ms_primary += ms[0]; out->ms_shade0 += ms[1]; out->ms_trace_secondary += ms[2]; out->ms_trace_shadow0 += ms[3];
    out->ms_shade1 += ms[4]; out->ms_trace_shadow1 += ms[5]; out->ms_resolve += ms[6]; out->ms_total += ms[7];
}

int rt_pipeline_get_stats(rt_pipeline *p, rt_stats *out)
{
    RT_REQUIRE(p && out, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    memset(out, 0, sizeof *out);
    if (!p->rendered) return RT_OK;
    uint32_t c[C_COUNT];
    HIP_TRY(hipMemcpy(c, p->counters.p, sizeof c, hipMemcpyDeviceToHost));
    out->rays_primary = p->last_pixels;
    out->primary_hits = c[C_NHIT];
    out->secondary_hits = 0;
    for (int l = 1; l <= MAXD; l++) out->secondary_hits += c[C_NHIT + l];
    out->rays_secondary = c[C_SECONDARY];
    out->rays_shadow = (uint64_t)c[C_SHADOW] + c[C_SHADOW_SKIPPED];
    out->rays_shadow_skipped = c[C_SHADOW_SKIPPED];
    out->frames = p->last_pd.n_frames;
    if (p->ring_frames > 0 && p->ring_pos > 0) {
        float ms[8];
        RT_TRY(stage_times(p, p->ring_pos - 1, ms));
        add_times(out, ms);
    }
    p->stats = *out;
    return RT_OK;
}

int rt_pipeline_get_totals(rt_pipeline *p, rt_stats *out)
{
    RT_REQUIRE(p && out, "null argument");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    memset(out, 0, sizeof *out);
    if (!p->totals.p) return RT_OK;
    unsigned long long t[8];
    HIP_TRY(hipMemcpy(t, p->totals.p, sizeof t, hipMemcpyDeviceToHost));
    out->rays_primary = t[0]; out->rays_secondary = t[1]; out->rays_shadow = t[2]; out->rays_shadow_skipped = t[6];
    out->primary_hits = t[3]; out->secondary_hits = t[4];
    out->frames = t[5];
    if (p->ring_frames > 0) {
        const uint64_t have = p->ring_pos < (uint64_t)p->ring_frames ? p->ring_pos : (uint64_t)p->ring_frames;
        uint64_t covered = 0;
        for (uint64_t f = p->ring_pos - have; f < p->ring_pos; f++) {
            float ms[8];
            RT_TRY(stage_times(p, f, ms));
            add_times(out, ms);
            covered += p->ring_nframes[(size_t)(f % (uint64_t)p->ring_frames)];
        }
        if (covered < out->frames) out->frames = covered;      // times cover only the remembered frames
    }
    return RT_OK;
}

int rt_pipeline_reset_totals(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    RT_TRY(rt_pipeline_flush_pending(p));
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (p->totals.p) HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->ring_pos = 0;
    return RT_OK;
}

}  // extern "C"
