// rt_api.hip -- C-ABI entry points for context, model, scene and batch TraceRay
// (include/dxr_amd.h).  Stand-ins for libs/DXRFramework/Rt{Context,Model,Scene}.
#include <new>

#include "rt_trace_device.h"

static thread_local std::string g_last_error;

void rt_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

size_t &rt_alloc_limit_ref()
{
    static size_t limit = ~(size_t)0;
    return limit;
}

extern "C" int rt_debug_set_alloc_limit(size_t bytes)
{
    rt_alloc_limit_ref() = bytes ? bytes : ~(size_t)0;
    return RT_OK;
}

namespace {

using namespace rtd;

// ---- device math probes ------------------------------------------------------

__global__ void k_debug_math(int fn, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c, r = 0.0f;
    switch (fn) {
    case 0: sincos_det(x[i], s, c); r = s; break;
    case 1: sincos_det(x[i], s, c); r = c; break;
    case 2: r = exp_det(x[i]); break;
    case 3: r = log_det(x[i]); break;
    case 4: r = pow_det(x[i], y[i]); break;
    case 5: r = __builtin_sqrtf(x[i]); break;
    case 6: r = x[i] / y[i]; break;
    case 7: r = fmin2(x[i], y[i]); break;
    case 8: r = fmax2(x[i], y[i]); break;
    default: break;
    }
    out[i] = r;
}

__global__ void k_debug_sample(int kind, const uint32_t *__restrict__ seeds, const float *__restrict__ vin, float exponent,
                               float *__restrict__ vout, float *__restrict__ pdf_brdf, uint32_t *__restrict__ seeds_out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s = seeds[i];
    const f3 in = mk3(vin[3 * i], vin[3 * i + 1], vin[3 * i + 2]);
    f3 o = mk3(0.0f, 0.0f, 0.0f);
    float pdf = 0.0f, brdf = 0.0f;
    switch (kind) {
    case 0: o = cos_hemisphere(s, in); break;
    case 1: o = uniform_hemisphere(s, in); break;
    case 2: o = phong_lobe(s, in, exponent, pdf, brdf); break;
    case 3: o = perpendicular(in); break;
    default: break;
    }
    vout[3 * i] = o.x; vout[3 * i + 1] = o.y; vout[3 * i + 2] = o.z;
    if (pdf_brdf) { pdf_brdf[2 * i] = pdf; pdf_brdf[2 * i + 1] = brdf; }
    if (seeds_out) seeds_out[i] = s;
}

int upload(rt_context *ctx, DevBuf &b, const void *host, size_t bytes)
{
    RT_TRY(b.reserve(bytes));
    if (bytes) HIP_TRY(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

int use_device(const rt_context *ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_version(void) { return "dxrexperiments_amd 0.1 (gfx950)"; }

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        rt_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return n;
}

// ---- experiment / test knobs ------------------------------------------------------

extern "C" int rt_debug_set_option(rt_context *c, const char *name, const char *value)
{
    RT_REQUIRE(c && name && value, "null argument");
    const std::string n(name);
    // every option but fast_bvh takes a number: the whole of `value` must be one ('true', 'on', '4x' used to read as 0 and quietly switch things off)
    char *end_i = nullptr, *end_f = nullptr;
    const long iv = strtol(value, &end_i, 10);
    const double fv = strtod(value, &end_f);
    auto bad = [&]() { rt_set_error("rt_debug_set_option: value '%s' out of range for '%s'", value, name); return RT_ERR_INVALID_ARG; };
    const bool is_int = *value != '\0' && end_i && *end_i == '\0', is_num = *value != '\0' && end_f && *end_f == '\0';
    const bool wants_float = n == "sah_node" || n == "sah_prim" || n == "dist_check_seconds";
    if (n != "fast_bvh" && !(wants_float ? is_num : is_int)) { rt_set_error("rt_debug_set_option: '%s' is not a number (option '%s')", value, name); return RT_ERR_INVALID_ARG; }
    if (n == "lds_top") c->lds_top = iv != 0;
    else if (n == "lds_stack_rows") { if (iv != 0 && iv != RT_LDS_STACK_ROWS && iv != RT_LDS_STACK_ROWS_TEST) return bad(); c->lds_stack_rows = (uint32_t)iv; }
    else if (n == "persistent_blocks_per_cu") { if (iv < 0 || iv > 16) return bad(); c->blocks_per_cu_override = (uint32_t)iv; }
    else if (n == "fast_bvh") { if (strcmp(value, "lbvh") != 0 && strcmp(value, "ploc") != 0) return bad(); c->use_ploc = strcmp(value, "ploc") == 0; }
    else if (n == "build_batch") { if (iv < 0 || iv > 64) return bad(); c->build_batch = (uint32_t)iv; }
    else if (n == "leaf_max") { if (iv < 1 || iv > 8) return bad(); c->leaf_max = (uint32_t)iv; }
    else if (n == "wide_sah") c->wide_sah = iv != 0;
    else if (n == "sah_node") { if (!(fv > 0.0)) return bad(); c->sah_node = (float)fv; }
    else if (n == "sah_prim") { if (!(fv > 0.0)) return bad(); c->sah_prim = (float)fv; }
    else if (n == "verbose") c->verbose = iv != 0;
    else if (n == "shadow_cache_res") { if (iv < -1 || iv > 8192) return bad(); c->opt_shadow_cache_res = (int)iv; }
    else if (n == "shadow_cache_pixels") { if (iv < -1 || iv > 1) return bad(); c->opt_shadow_cache_pixels = (int)iv; }
    else if (n == "primary_persistent") { if (iv < -1 || iv > 1) return bad(); c->opt_primary_persistent = (int)iv; }
    else if (n == "seven_waves_always") c->opt_seven_waves_always = iv != 0;
    else if (n == "free_radius") c->opt_free_radius = iv != 0;
    else if (n == "split_refs") c->opt_split_refs = iv != 0;
    else if (n == "fail_ploc_rounds") c->opt_fail_ploc_rounds = iv != 0;
    else if (n == "repack") c->opt_repack = iv != 0;
    else if (n == "primary_retry_cap") { if (iv < 0 || iv > (1 << 24)) return bad(); c->opt_primary_retry_cap = (uint32_t)iv; }
    else if (n == "batch_max") { if (iv < 0 || iv > 32) return bad(); c->opt_batch_max = (uint32_t)iv; }
    else if (n == "queue_budget_mb") { if (iv < 0) return bad(); c->opt_queue_budget_mb = (size_t)iv; }
    else if (n == "dist_check_seconds") { if (!(fv >= 0.0)) return bad(); c->opt_dist_check_seconds = fv; }
    else { rt_set_error("rt_debug_set_option: unknown option '%s'", name); return RT_ERR_INVALID_ARG; }
    return RT_OK;
}

// the re-packed engine's tallies since they were last read (rt_trace_repack.h: node steps, lanes in them, leaf passes, lanes in them, rays through
// the leaf queue, through the node queue, refills, watchdog aborts); synchronises
extern "C" int rt_debug_repack_stats(rt_context *c, unsigned long long out[8])
{
    RT_REQUIRE(c && out, "null argument");
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (!c->rp_records.p || !c->rp_grid) return RT_OK;
    char *at = (char *)c->rp_records.p + (size_t)c->rp_grid * 256 * 64;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(out, at, 64, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(at, 0, 64, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ---- context -------------------------------------------------------------------

static int context_create(int device, void *stream, bool own, rt_context **out)
{
    RT_REQUIRE(out, "null argument");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        rt_set_error("rt_context_create: device %d not present (%d HIP devices visible)", device, n);
        return RT_ERR_HIP;
    }
    HIP_TRY(hipSetDevice(device));
    rt_context *c = new (std::nothrow) rt_context();
    if (!c) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    c->device = device;
    c->own_stream = own;
    if (own) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { rt_set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete c; return RT_ERR_HIP; }
    } else c->stream = (hipStream_t)stream;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        rt_set_error("hipEventCreate failed");
        delete c;
        return RT_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->cu_count = (uint32_t)prop.multiProcessorCount;
    if (hipHostMalloc((void **)&c->pinned, RT_PINNED_WORDS * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) c->pinned = nullptr;
    // the one environment variable the library reads for its behaviour: RT_DEBUG_OPTIONS="name=value,name=value" 
    if (const char *env = getenv("RT_DEBUG_OPTIONS")) {
        std::string all(env);
        size_t at = 0;
        while (at < all.size()) {
            size_t end = all.find(',', at);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(at, end - at);
            const size_t eq = item.find('=');
            if (item.empty()) { at = end + 1; continue; }
            if (eq == std::string::npos) {
                rt_set_error("RT_DEBUG_OPTIONS: '%s' is not name=value", item.c_str());
                rt_context_release(c);
                return RT_ERR_INVALID_ARG;
            }
            if (rt_debug_set_option(c, item.substr(0, eq).c_str(), item.substr(eq + 1).c_str()) != RT_OK) {
                const std::string why = rt_last_error();
                rt_set_error("RT_DEBUG_OPTIONS: %s", why.c_str());
                rt_context_release(c);
                return RT_ERR_INVALID_ARG;
            }
            at = end + 1;
        }
    }
    // the RT_* variables the library read one by one until round 4 are no longer looked at: say so once instead of quietly running the defaults
    {
        static const char *const legacy[] = {"RT_BATCH_MAX", "RT_BUILD_BATCH", "RT_DIST_CHECK_SECONDS", "RT_FAST_BVH", "RT_FREE_RADIUS", "RT_LDS_STACK_ROWS", "RT_LDS_TOP",
                                             "RT_LEAF_MAX", "RT_PERSISTENT_BLOCKS_PER_CU", "RT_PRIMARY_PERSISTENT", "RT_QUEUE_BUDGET_MB", "RT_SAH_NODE", "RT_SAH_PRIM",
                                             "RT_SEVEN_WAVES_ALWAYS", "RT_SHADOW_CACHE_PIXELS", "RT_SHADOW_CACHE_RES", "RT_VERBOSE", "RT_WIDE_SAH"};
        static bool warned = false;
        for (const char *v : legacy)
            if (!warned && getenv(v)) {
                fprintf(stderr, "[dxr_amd] %s is set but no longer read: use RT_DEBUG_OPTIONS=\"name=value,...\" or rt_debug_set_option (INTEGRATION.md)\n", v);
                warned = true;
            }
    }
    *out = c;
    return RT_OK;
}

int rt_context_create(int device, rt_context **out) { return context_create(device, nullptr, true, out); }

int rt_context_create_on_stream(int device, void *hip_stream, rt_context **out)
{
    return context_create(device, hip_stream, false, out);
}

int rt_context_destroy(rt_context *ctx)
{
    rt_context_release(ctx);
    return RT_OK;
}

}  // extern "C"

void rt_context_retain(rt_context *ctx) { if (ctx) ctx->refs++; }

void rt_scene_retain(rt_scene *s) { if (s) s->refs++; }

// The context lives until its handle AND every object created on it are gone.
void rt_context_release(rt_context *ctx)
{
    if (!ctx || --ctx->refs > 0) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf &b : ctx->scratch) b.release();
    ctx->pool.release();
    ctx->deep_stack.release();
    ctx->rp_records.release();
    ctx->build_arena.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" {

int rt_context_synchronize(rt_context *ctx)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(use_device(ctx));
    RT_TRY(rt_context_flush_deferred(ctx));          // frames a deferred pipeline still holds are part of "everything submitted"
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_context_get_stack_memory(rt_context *ctx, size_t *bytes)
{
    RT_REQUIRE(ctx && bytes, "null argument");
    *bytes = ctx->deep_stack.bytes;
    return RT_OK;
}

int rt_context_get_stream(rt_context *ctx, void **hip_stream_out)
{
    RT_REQUIRE(ctx && hip_stream_out, "null argument");
    // (a caller that asks for the stream is about to order its own work behind "everything submitted": the frames a deferred
    // pipeline still holds go onto the stream first -- ADVICE r4)
    *hip_stream_out = (void *)ctx->stream;
    if (!ctx->deferred.empty()) { RT_TRY(use_device(ctx)); RT_TRY(rt_context_flush_deferred(ctx)); }
    return RT_OK;
}

int rt_context_get_device(rt_context *ctx, int *device_out)
{
    RT_REQUIRE(ctx && device_out, "null argument");
    *device_out = ctx->device;
    return RT_OK;
}

int rt_device_alloc(rt_context *ctx, size_t bytes, void **device_ptr)
{
    RT_REQUIRE(ctx && device_ptr, "null argument");
    RT_TRY(use_device(ctx));
    hipError_t e = hipMalloc(device_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) {
        rt_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP;
    }
    return RT_OK;
}

int rt_device_free(rt_context *ctx, void *device_ptr)
{
    RT_REQUIRE(ctx, "null context");
    RT_TRY(use_device(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (device_ptr) HIP_TRY(hipFree(device_ptr));
    return RT_OK;
}

int rt_device_upload(rt_context *ctx, void *device_dst, const void *host_src, size_t bytes)
{
    RT_REQUIRE(ctx && (bytes == 0 || (device_dst && host_src)), "null argument");
    RT_TRY(use_device(ctx));
    if (bytes) HIP_TRY(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_device_download(rt_context *ctx, void *host_dst, const void *device_src, size_t bytes)
{
    RT_REQUIRE(ctx && (bytes == 0 || (host_dst && device_src)), "null argument");
    RT_TRY(use_device(ctx));
    RT_TRY(rt_context_flush_deferred(ctx));          // (device_src may be an output a deferred pipeline still owes frames to)
    if (bytes) HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- model -----------------------------------------------------------------------

static int model_finish(rt_context *ctx, rt_model *m, rt_model **out)
{
    m->ctx = ctx;
    rt_context_retain(ctx);
    m->n_verts = (uint32_t)m->h_verts.size();
    m->n_tris = (uint32_t)(m->h_idx.size() / 3);
    for (uint32_t i : m->h_idx)
        if (i >= m->n_verts) {
            rt_set_error("index %u out of range (%u vertices)", i, m->n_verts);
            rt_context_release(ctx);
            delete m;
            return RT_ERR_INVALID_ARG;
        }
    int rc = use_device(ctx);
    if (rc == RT_OK) rc = upload(ctx, m->d_verts, m->h_verts.data(), sizeof(rt_vertex) * m->h_verts.size());
    if (rc == RT_OK) rc = upload(ctx, m->d_idx, m->h_idx.data(), sizeof(uint32_t) * m->h_idx.size());
    if (rc == RT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) { rt_set_error("geometry upload failed"); rc = RT_ERR_HIP; }
    if (rc != RT_OK) {
        m->d_verts.release(); m->d_idx.release();
        rt_context_release(ctx);
        delete m;
        return rc;
    }
    *out = m;
    return RT_OK;
}

int rt_model_create_from_arrays(rt_context *ctx, const rt_vertex *verts, uint32_t n_verts, const uint32_t *indices,
                                uint32_t n_tris, rt_model **out)
{
    RT_REQUIRE(ctx && verts && indices && out, "null argument");
    RT_REQUIRE(n_verts > 0 && n_tris > 0, "a model needs at least one triangle");
    rt_model *m = new (std::nothrow) rt_model();
    if (!m) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    m->h_verts.assign(verts, verts + n_verts);
    m->h_idx.assign(indices, indices + 3 * (size_t)n_tris);
    return model_finish(ctx, m, out);
}

int rt_model_create_from_obj(rt_context *ctx, const char *path, rt_model **out)
{
    RT_REQUIRE(ctx && path && out, "null argument");
    rt_model *m = new (std::nothrow) rt_model();
    if (!m) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    int rc = rt_obj_parse(path, m->h_verts, m->h_idx);
    if (rc != RT_OK) { delete m; return rc; }
    return model_finish(ctx, m, out);
}

int rt_model_create_from_file(rt_context *ctx, const char *path, rt_model **out)
{
    RT_REQUIRE(ctx && path && out, "null argument");
    const size_t n = strlen(path);
    const bool fbx = n >= 4 && path[n - 4] == '.' && (path[n - 3] | 0x20) == 'f' && (path[n - 2] | 0x20) == 'b' && (path[n - 1] | 0x20) == 'x';
    if (!fbx) return rt_model_create_from_obj(ctx, path, out);
    rt_model *m = new (std::nothrow) rt_model();
    if (!m) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    int rc = rt_fbx_parse(path, m->h_verts, m->h_idx);
    if (rc != RT_OK) { delete m; return rc; }
    return model_finish(ctx, m, out);
}

int rt_model_get_counts(const rt_model *m, uint32_t *n_verts, uint32_t *n_tris)
{
    RT_REQUIRE(m, "null model");
    if (n_verts) *n_verts = m->n_verts;
    if (n_tris) *n_tris = m->n_tris;
    return RT_OK;
}

int rt_model_read_geometry(const rt_model *m, rt_vertex *verts, uint32_t *indices)
{
    RT_REQUIRE(m, "null model");
    if (verts) memcpy(verts, m->h_verts.data(), sizeof(rt_vertex) * m->h_verts.size());
    if (indices) memcpy(indices, m->h_idx.data(), sizeof(uint32_t) * m->h_idx.size());
    return RT_OK;
}

int rt_model_retain(rt_model *m)
{
    RT_REQUIRE(m, "null model");
    m->refs++;
    return RT_OK;
}

int rt_model_destroy(rt_model *m)
{
    if (!m) return RT_OK;
    if (--m->refs > 0) return RT_OK;
    (void)hipSetDevice(m->ctx->device);
    m->d_verts.release(); m->d_idx.release(); m->tris.release(); m->normals.release(); m->blas.release(); m->rec_boxes.release(); m->ref_off.release(); m->ref_boxes.release();
    rt_context *ctx = m->ctx;
    delete m;
    rt_context_release(ctx);
    return RT_OK;
}

// ---- scene -----------------------------------------------------------------------

int rt_scene_create(rt_context *ctx, rt_scene **out)
{
    RT_REQUIRE(ctx && out, "null argument");
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    s->ctx = ctx;
    rt_context_retain(ctx);
    *out = s;
    return RT_OK;
}

int rt_scene_add_model(rt_scene *s, rt_model *m, const float transform3x4[12])
{
    RT_REQUIRE(s && m && transform3x4, "null argument");
    RT_REQUIRE(m->ctx == s->ctx, "model and scene belong to different contexts");
    RT_TRY(rt_context_flush_deferred(s->ctx));       // frames accepted before this change see the scene as it was
    SceneInstance in;
    in.model = m;
    memcpy(in.xform, transform3x4, sizeof in.xform);
    m->refs++;
    s->inst.push_back(in);
    s->built = false;
    s->generation++;
    return RT_OK;
}

int rt_scene_get_num_instances(const rt_scene *s, uint32_t *n)
{
    RT_REQUIRE(s && n, "null argument");
    *n = (uint32_t)s->inst.size();
    return RT_OK;
}

int rt_scene_build(rt_scene *s, uint32_t hit_group_count)
{
    (void)hit_group_count;   // hit-group stride of the reference's shader table; material = f(instance) here
    RT_REQUIRE(s, "null scene");
    RT_REQUIRE(!s->inst.empty(), "scene has no instances");
    rt_context *ctx = s->ctx;
    RT_TRY(use_device(ctx));
    RT_TRY(rt_context_flush_deferred(ctx));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); rt_set_error("rt_scene_build: hipEventCreate failed"); return RT_ERR_HIP; }
    if (hipEventRecord(e0, ctx->stream) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); rt_set_error("rt_scene_build: hipEventRecord failed"); return RT_ERR_HIP; }
    s->generation++;          // device arrays are about to be reallocated: pipelines drop what they cached
    int rc = RT_OK;
    for (SceneInstance &in : s->inst)
        if ((rc = rt_build_blas(ctx, in.model)) != RT_OK) break;
    if (rc == RT_OK) rc = rt_build_tlas(ctx, s);
    if (rc == RT_OK) {
        if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&s->build_ms, e0, e1) != hipSuccess) {
            rt_set_error("rt_scene_build: timing events failed: %s", hipGetErrorString(hipGetLastError()));
            rc = RT_ERR_HIP;
        } else s->built = true;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int rt_scene_destroy(rt_scene *s)
{
    if (!s) return RT_OK;
    if (--s->refs > 0) return RT_OK;
    (void)hipSetDevice(s->ctx->device);
    for (SceneInstance &in : s->inst) rt_model_destroy(in.model);
    s->d_inst.release();
    s->tlas.release();
    rt_context *ctx = s->ctx;
    delete s;
    rt_context_release(ctx);
    return RT_OK;
}

static const BvhDev *pick_bvh(const rt_scene *s, int which)
{
    if (!s || !s->built) return nullptr;
    if (which < 0) return &s->tlas;
    if ((size_t)which >= s->inst.size()) return nullptr;
    return &s->inst[which].model->blas;
}

int rt_scene_bvh_info(const rt_scene *s, int which, uint32_t *n_prims, uint32_t *n_nodes, uint32_t *max_depth)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b) { rt_set_error("rt_scene_bvh_info: scene not built or index out of range"); return RT_ERR_STATE; }
    if (n_prims) *n_prims = b->n;
    if (n_nodes) *n_nodes = 2 * b->n - 1;
    if (max_depth) *max_depth = b->max_depth;
    return RT_OK;
}

int rt_scene_bvh_read(const rt_scene *s, int which, rt_bvh_node *nodes, uint64_t *sorted_keys, uint32_t *parents)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b) { rt_set_error("rt_scene_bvh_read: scene not built or index out of range"); return RT_ERR_STATE; }
    RT_TRY(use_device(s->ctx));
    const size_t nn = 2 * (size_t)b->n - 1;
    if (nodes) HIP_TRY(hipMemcpy(nodes, b->nodes.p, sizeof(rt_bvh_node) * nn, hipMemcpyDeviceToHost));
    if (sorted_keys) HIP_TRY(hipMemcpy(sorted_keys, b->keys.p, sizeof(uint64_t) * b->n, hipMemcpyDeviceToHost));
    if (parents) HIP_TRY(hipMemcpy(parents, b->parents.p, sizeof(uint32_t) * nn, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_wide_info(const rt_scene *s, int which, uint32_t *n_nodes, int32_t *root_code, uint32_t *n_records)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b) { rt_set_error("rt_scene_wide_info: scene not built or index out of range"); return RT_ERR_STATE; }
    if (n_nodes) *n_nodes = b->wide_n;
    if (root_code) *root_code = b->root_code;
    if (n_records) *n_records = which < 0 ? 0u : s->inst[which].model->n_recs;
    return RT_OK;
}

int rt_wide_layout_info(uint32_t *width, uint32_t *node_bytes)
{
    if (width) *width = RT_WIDE;
    if (node_bytes) *node_bytes = (uint32_t)sizeof(WNode);
    return RT_OK;
}

int rt_scene_wide_read(const rt_scene *s, int which, void *nodes, void *records)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b) { rt_set_error("rt_scene_wide_read: scene not built or index out of range"); return RT_ERR_STATE; }
    RT_TRY(use_device(s->ctx));
    if (nodes && b->wide_n) HIP_TRY(hipMemcpy(nodes, b->wide.p, sizeof(WNode) * (size_t)b->wide_n, hipMemcpyDeviceToHost));
    if (records && which >= 0) HIP_TRY(hipMemcpy(records, s->inst[which].model->tris.p, sizeof(TriRec) * (size_t)s->inst[which].model->n_recs, hipMemcpyDeviceToHost));
    return RT_OK;
}

/* the validation boxes of an instance's model (rt_refs.h): *n_refs = 0 when none of its triangles is split */
int rt_scene_refs_info(const rt_scene *s, int which, uint32_t *n_tris, uint32_t *n_refs)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b || which < 0) { rt_set_error("rt_scene_refs_info: scene not built or index out of range"); return RT_ERR_STATE; }
    const rt_model *m = s->inst[which].model;
    if (n_tris) *n_tris = m->n_tris;
    if (n_refs) *n_refs = m->ref_off.p ? (uint32_t)(m->ref_boxes.bytes / 24) : 0u;
    return RT_OK;
}

int rt_scene_refs_read(const rt_scene *s, int which, uint32_t *ref_off, float *ref_boxes, float *record_boxes)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b || which < 0) { rt_set_error("rt_scene_refs_read: scene not built or index out of range"); return RT_ERR_STATE; }
    const rt_model *m = s->inst[which].model;
    if (!m->ref_off.p) { rt_set_error("rt_scene_refs_read: no triangle of this model is split"); return RT_ERR_STATE; }
    RT_TRY(use_device(s->ctx));
    uint32_t total = 0;
    HIP_TRY(hipMemcpy(&total, m->ref_off.as<uint32_t>() + m->n_tris, 4, hipMemcpyDeviceToHost));
    if (ref_off) HIP_TRY(hipMemcpy(ref_off, m->ref_off.p, 4 * ((size_t)m->n_tris + 1), hipMemcpyDeviceToHost));
    if (ref_boxes) HIP_TRY(hipMemcpy(ref_boxes, m->ref_boxes.p, 24 * (size_t)total, hipMemcpyDeviceToHost));
    if (record_boxes && m->rec_boxes.p) HIP_TRY(hipMemcpy(record_boxes, m->rec_boxes.p, 24 * (size_t)m->n_recs, hipMemcpyDeviceToHost));
    return RT_OK;
}

/* layout experiments (tools/layout_estimate.py): replaces the node array by a renumbering of itself */
int rt_debug_wide_write(rt_scene *s, int which, const void *nodes, uint32_t n_nodes)
{
    const BvhDev *b = pick_bvh(s, which);
    if (!b || !nodes || n_nodes != b->wide_n) { rt_set_error("rt_debug_wide_write: scene not built, or a different node count"); return RT_ERR_STATE; }
    RT_TRY(use_device(s->ctx));
    HIP_TRY(hipMemcpy(b->wide.p, nodes, sizeof(WNode) * (size_t)n_nodes, hipMemcpyHostToDevice));
    return RT_OK;                                      // (same device pointers, same node count: nothing cached goes stale)
}

int rt_scene_instance_info(const rt_scene *s, uint32_t instance, float world_box[6], float world_to_object[12])
{
    RT_REQUIRE(s, "null scene");
    if (!s->built || instance >= s->h_inst.size()) { rt_set_error("scene not built or instance out of range"); return RT_ERR_STATE; }
    const InstanceRec &r = s->h_inst[instance];
    if (world_box) for (int c = 0; c < 3; c++) { world_box[c] = r.wlo[c]; world_box[3 + c] = r.whi[c]; }
    if (world_to_object) memcpy(world_to_object, r.inv, sizeof r.inv);
    return RT_OK;
}

int rt_scene_build_ms(const rt_scene *s, float *ms)
{
    RT_REQUIRE(s && ms, "null argument");
    *ms = s->build_ms;
    return RT_OK;
}

// ---- batch TraceRay ---------------------------------------------------------------

int rt_trace_batch(rt_context *ctx, const rt_scene *s, const float *origin_tmin, const float *dir_tmax, size_t n,
                   uint32_t ray_flags, uint32_t kernel, uint32_t mem, float *t, float *u, float *v, uint32_t *prim,
                   uint32_t *inst, uint32_t *cnt_nodes, uint32_t *cnt_tris)
{
    RT_REQUIRE(ctx && s, "null argument");
    if (!s->built) { rt_set_error("rt_trace_batch: scene not built"); return RT_ERR_STATE; }
    RT_REQUIRE(kernel == RT_TRACE_FAST || kernel == RT_TRACE_CANONICAL, "unknown kernel selector");
    if (n == 0) return RT_OK;
    RT_REQUIRE(origin_tmin && dir_tmax, "null ray arrays");
    RT_TRY(use_device(ctx));
    if (mem == RT_MEM_DEVICE) {
        TraceOut out = {t, u, v, prim, inst, cnt_nodes, cnt_tris};
        return rt_launch_trace(ctx, s, (const float4 *)origin_tmin, (const float4 *)dir_tmax, n, ray_flags, kernel, out);
    }
    RT_REQUIRE(mem == RT_MEM_HOST, "unknown memory selector");
    DevBuf *sb = ctx->scratch;
    RT_TRY(upload(ctx, sb[0], origin_tmin, n * 16));
    RT_TRY(upload(ctx, sb[1], dir_tmax, n * 16));
    void *host_out[7] = {t, u, v, prim, inst, cnt_nodes, cnt_tris};
    RT_TRY(sb[2].reserve(n * 4 * 7));
    char *base = sb[2].as<char>();
    void *dev_out[7];
    for (int k = 0; k < 7; k++) dev_out[k] = host_out[k] ? base + (size_t)k * n * 4 : nullptr;
    TraceOut out = {(float *)dev_out[0], (float *)dev_out[1], (float *)dev_out[2], (uint32_t *)dev_out[3],
                    (uint32_t *)dev_out[4], (uint32_t *)dev_out[5], (uint32_t *)dev_out[6]};
    if (kernel != RT_TRACE_CANONICAL) { out.cnt_nodes = nullptr; out.cnt_tris = nullptr; }
    RT_TRY(rt_launch_trace(ctx, s, sb[0].as<float4>(), sb[1].as<float4>(), n, ray_flags, kernel, out));
    for (int k = 0; k < 7; k++)
        if (host_out[k] && dev_out[k] && !((k >= 5) && kernel != RT_TRACE_CANONICAL))
            HIP_TRY(hipMemcpyAsync(host_out[k], dev_out[k], n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_last_ms(rt_context *ctx, float *ms)
{
    RT_REQUIRE(ctx && ms, "null argument");
    RT_TRY(use_device(ctx));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return RT_OK;
}

// ---- device math probes -------------------------------------------------------------

int rt_debug_math(rt_context *ctx, int fn, const float *x, const float *y, float *out, size_t n)
{
    RT_REQUIRE(ctx && x && out, "null argument");
    if (n == 0) return RT_OK;
    RT_TRY(use_device(ctx));
    DevBuf *sb = ctx->scratch;
    RT_TRY(upload(ctx, sb[0], x, n * 4));
    RT_TRY(upload(ctx, sb[1], y ? y : x, n * 4));
    RT_TRY(sb[2].reserve(n * 4));
    k_debug_math<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(fn, sb[0].as<float>(), sb[1].as<float>(), sb[2].as<float>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, sb[2].p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_debug_sample(rt_context *ctx, int kind, const uint32_t *seeds, const float *vec3_in, float exponent, float *vec3_out,
                    float *pdf_brdf, uint32_t *seeds_out, size_t n)
{
    RT_REQUIRE(ctx && seeds && vec3_in && vec3_out, "null argument");
    if (n == 0) return RT_OK;
    RT_TRY(use_device(ctx));
    DevBuf *sb = ctx->scratch;
    RT_TRY(upload(ctx, sb[0], seeds, n * 4));
    RT_TRY(upload(ctx, sb[1], vec3_in, n * 12));
    RT_TRY(sb[2].reserve(n * 12));
    RT_TRY(sb[3].reserve(n * 8));
    RT_TRY(sb[4].reserve(n * 4));
    k_debug_sample<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(kind, sb[0].as<uint32_t>(), sb[1].as<float>(), exponent,
                                                                         sb[2].as<float>(), sb[3].as<float>(), sb[4].as<uint32_t>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(vec3_out, sb[2].p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (pdf_brdf) HIP_TRY(hipMemcpyAsync(pdf_brdf, sb[3].p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (seeds_out) HIP_TRY(hipMemcpyAsync(seeds_out, sb[4].p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
