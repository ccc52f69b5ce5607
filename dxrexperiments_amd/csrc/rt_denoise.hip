// rt_denoise.hip -- DenoiseCompositor on gfx950.
//
// Reference: src/DenoiseCompositor.cpp:109-148 dispatches two compute passes
// (assets/shaders/DenoiseCompositorH/V.hlsl -> DenoiseCommon.hlsli:46-77 ->
// BilateralFilter.hlsli:75-118): a separable joint-bilateral filter of the indirect-specular
// AOV, guided by the direct-lighting AOV, horizontally then vertically, and in the second
// pass the composite (+ direct), exposure, Reinhard tone map and optional gamma.
//
// A stencil of 2 x 16 B in, 16 B out per pixel and pass, but bound by VALU issue, not by HBM: a tap costs ~16 vector
// instructions (no FMA: the HLSL's operation order is kept), 25 taps per pixel and pass.  The taps come from an LDS
// tile so each texel is fetched from memory once per block; the 41 tap weights are computed ONCE per dispatch on the
// host with the reference's expression (BilateralFilter.hlsli:82-90 computes them once per thread group) and reach the
// kernels as kernel arguments, i.e. scalar registers: round 1 recomputed a correctly rounded division per tap and pixel.
//   pass H  block = 256 x 1 pixels,  tile = (256 + 2*20) texels of both images
//   pass V  block = 8 x 64 pixels (256 threads, 2 rows each), tile = 8 x (64 + 2*20) texels
// The reference's own LDS cache (64 + 2*20 texels, BilateralFilter.hlsli:40-73) has an
// index-clamp race on its left halo; it is meant to equal "the texture with a zero border",
// which is what the tiles hold (out-of-image loads read 0, like D3D).
// Arithmetic order follows the HLSL exactly (no FMA), so results match the oracle bit for bit.
#include <hip/hip_fp16.h>

#include <new>

#include "rt_device_math.h"
#include "rt_internal.h"

using namespace rtd;

namespace {

constexpr int MAX_EXTENT = 20;      // BilateralFilter.hlsli:41

struct DenoiseArgs {
    const float4 *direct;           // gDirectLighting (joint image)
    const float4 *input;            // gInput
    float4 *output;
    int width, height;
    rt_denoiser_params prm;
    float w[2 * MAX_EXTENT + 1];    // sPrecalculatedGaussianWeights, BilateralFilter.hlsli:40,82-90
};

RT_DEV float4 load_texel(const float4 *img, int x, int y, int w, int h)
{
    if (x < 0 || y < 0 || x >= w || y >= h) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return img[(size_t)y * w + x];
}

// weights table of BilateralFilter.hlsli:82-90 (host: IEEE single arithmetic, the same bits as the device's correctly
// rounded division)
static float tap_weight(int i, float kernelRadius)
{
    const int a = i < 0 ? -i : i;
    int idx = (int)((float)(a * 5) / (0.001f + __builtin_fabsf(kernelRadius * 0.8f)));
    idx = idx < 0 ? 0 : (idx > 6 ? 6 : idx);
    return idx < 2 ? 1.0f : (idx < 3 ? 0.9f : (idx < 4 ? 0.75f : (idx < 5 ? 0.6f : (idx < 6 ? 0.5f : 0.0f))));
}

// one bilateral tap accumulated exactly like BilateralFilter.hlsli:103-114
RT_DEV void accumulate_tap(float4 s, float4 sj, float4 cj, float gw, float4 &color, float &weight)
{
    float dist = __builtin_fabsf(sj.x - cj.x);
    dist += __builtin_fabsf(sj.y - cj.y);
    dist += __builtin_fabsf(sj.z - cj.z);
    dist *= 10.0f;
    const float cw = 1.0f - fmin2(fmax2(dist, 0.0f), 1.0f);
    const float bw = gw * cw;
    color.x += s.x * bw; color.y += s.y * bw; color.z += s.z * bw; color.w += s.w * bw;
    weight += bw;
}

// A tile texel as one 16-B LDS read: without this the compiler fetches the three used components with ds_read_b96,
// which moves 96 B per clock per CU against ds_read_b128's 256 (MI355X_MICROARCH.md, LDS table), and the two reads per
// tap are what bounds these kernels.
RT_DEV float4 lds_texel(const float4 *p)
{
    float4 t = *p;
    asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
    return t;
}

// the 2K+1 taps of one pixel, in the reference's order (BilateralFilter.hlsli:101-114); `stride` texels between taps.
// Five taps per trip: their weights arrive as one batch of scalar loads and their LDS reads are in flight together.
RT_DEV void filter_taps(const DenoiseArgs &a, const float4 *s_in, const float4 *s_jn, int c, int stride, float4 cj, float4 &color, float &weight)
{
    const int K = a.prm.maxKernelSize;
    int i = -K;
    for (; i + 4 <= K; i += 5) {
        const float w0 = a.w[i + MAX_EXTENT], w1 = a.w[i + MAX_EXTENT + 1], w2 = a.w[i + MAX_EXTENT + 2], w3 = a.w[i + MAX_EXTENT + 3], w4 = a.w[i + MAX_EXTENT + 4];
        const int t = c + i * stride;
        const float4 i0 = lds_texel(s_in + t), j0 = lds_texel(s_jn + t), i1 = lds_texel(s_in + t + stride), j1 = lds_texel(s_jn + t + stride);
        const float4 i2 = lds_texel(s_in + t + 2 * stride), j2 = lds_texel(s_jn + t + 2 * stride), i3 = lds_texel(s_in + t + 3 * stride);
        const float4 j3 = lds_texel(s_jn + t + 3 * stride), i4 = lds_texel(s_in + t + 4 * stride), j4 = lds_texel(s_jn + t + 4 * stride);
        accumulate_tap(i0, j0, cj, w0, color, weight);
        accumulate_tap(i1, j1, cj, w1, color, weight);
        accumulate_tap(i2, j2, cj, w2, color, weight);
        accumulate_tap(i3, j3, cj, w3, color, weight);
        accumulate_tap(i4, j4, cj, w4, color, weight);
    }
    for (; i <= K; ++i) accumulate_tap(lds_texel(s_in + c + i * stride), lds_texel(s_jn + c + i * stride), cj, a.w[i + MAX_EXTENT], color, weight);
}

// composite + exposure + tone map + gamma of pass 1 (DenoiseCommon.hlsli:56-74)
RT_DEV float4 finish_pass1(const rt_denoiser_params &P, float4 c, float4 d)
{
    if (P.debugVisualize == 0) { c.x += d.x; c.y += d.y; c.z += d.z; }
    else if (P.debugVisualize == 3) { c.x = d.x; c.y = d.y; c.z = d.z; }
    c.x *= P.exposure; c.y *= P.exposure; c.z *= P.exposure;
    if (P.tonemap) {
        float lum = c.x * 0.299f;
        lum += c.y * 0.587f;
        lum += c.z * 0.114f;
        const float k = (lum / (lum + 1.0f)) / lum;
        c.x = fmax2(c.x * k, 0.0f); c.y = fmax2(c.y * k, 0.0f); c.z = fmax2(c.z * k, 0.0f);
    }
    if (P.gammaCorrect) {
        const float e = 1.0f / P.gamma;
        c.x = saturate(pow_det(c.x, e)); c.y = saturate(pow_det(c.y, e)); c.z = saturate(pow_det(c.z, e));
    }
    return c;
}

// ---- pass H: 256 pixels of one row per block --------------------------------------------------
constexpr int HB = 256;
__global__ void __launch_bounds__(HB) k_denoise_h(DenoiseArgs a)
{
    __shared__ float4 s_in[HB + 2 * MAX_EXTENT];
    __shared__ float4 s_jn[HB + 2 * MAX_EXTENT];
    const int y = blockIdx.y;
    const int x0 = blockIdx.x * HB;
    for (int t = threadIdx.x; t < HB + 2 * MAX_EXTENT; t += HB) {
        s_in[t] = load_texel(a.input, x0 - MAX_EXTENT + t, y, a.width, a.height);
        s_jn[t] = load_texel(a.direct, x0 - MAX_EXTENT + t, y, a.width, a.height);
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= a.width) return;
    const int c = threadIdx.x + MAX_EXTENT;
    float4 color;
    if (a.prm.debugVisualize == 2) color = s_in[c];
    else {
        const float4 cj = s_jn[c];
        color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float weight = 0.0f;
        filter_taps(a, s_in, s_jn, c, 1, cj, color, weight);
        color = make_float4(color.x / weight, color.y / weight, color.z / weight, color.w / weight);
    }
    a.output[(size_t)y * a.width + x] = make_float4(color.x, color.y, color.z, 1.0f);
}

// ---- pass V: 8 x 64 pixel tile per block, each thread 2 rows ------------------------------------------
// A narrow, tall tile: a tile row is 8 texels = one 128-B line of each image, the halo costs 104 / 64 of the rows
// (round 1's 32 x 32 tile: 72 / 32), the tile takes 26 KiB of LDS instead of 72 (6 blocks per CU instead of 2, and this
// kernel lives on occupancy: every tap waits for two LDS reads), and with the [row][8] layout the 64 lanes of a wave --
// 8 rows x 8 columns -- read 64 consecutive texels, which is conflict free.
constexpr int VW = 8, VH = 64;
__global__ void __launch_bounds__(256) k_denoise_v(DenoiseArgs a)
{
    __shared__ float4 s_in[(VH + 2 * MAX_EXTENT) * VW];
    __shared__ float4 s_jn[(VH + 2 * MAX_EXTENT) * VW];
    const int x0 = blockIdx.x * VW, y0 = blockIdx.y * VH;
    for (int t = threadIdx.x; t < (VH + 2 * MAX_EXTENT) * VW; t += 256) {
        const int ty = t / VW, tx = t % VW;
        s_in[t] = load_texel(a.input, x0 + tx, y0 - MAX_EXTENT + ty, a.width, a.height);
        s_jn[t] = load_texel(a.direct, x0 + tx, y0 - MAX_EXTENT + ty, a.width, a.height);
    }
    __syncthreads();
    const int tx = threadIdx.x % VW;
    const int x = x0 + tx;
    if (x >= a.width) return;
    for (int r = threadIdx.x / VW; r < VH; r += 256 / VW) {
        const int y = y0 + r;
        if (y >= a.height) break;
        const int c = (r + MAX_EXTENT) * VW + tx;
        float4 color;
        if (a.prm.debugVisualize == 2) color = s_in[c];
        else {
            const float4 cj = s_jn[c];
            color = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float weight = 0.0f;
            filter_taps(a, s_in, s_jn, c, VW, cj, color, weight);
            color = make_float4(color.x / weight, color.y / weight, color.z / weight, color.w / weight);
        }
        color = finish_pass1(a.prm, color, s_jn[c]);
        a.output[(size_t)y * a.width + x] = make_float4(color.x, color.y, color.z, 1.0f);
    }
}

__global__ void k_denoise_f16(const float4 *__restrict__ in, ushort4 *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = in[i];
    ushort4 o;
    o.x = __half_as_ushort(__float2half_rn(v.x)); o.y = __half_as_ushort(__float2half_rn(v.y));
    o.z = __half_as_ushort(__float2half_rn(v.z)); o.w = __half_as_ushort(__float2half_rn(v.w));
    out[i] = o;
}

}  // namespace

struct rt_denoiser {
    rt_context *ctx = nullptr;
    rt_denoiser_params prm;
    uint32_t width = 0, height = 0, format = RT_FORMAT_R32G32B32A32_FLOAT;
    DevBuf out[2], half_out;        // out[0] = pass H, out[1] = pass V (the result, DenoiseCompositor.h:23)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool dispatched = false;
};

extern "C" {

int rt_denoiser_create(rt_context *ctx, rt_denoiser **out)
{
    RT_REQUIRE(ctx && out, "null argument");
    rt_denoiser *d = new (std::nothrow) rt_denoiser();
    if (!d) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    d->ctx = ctx;
    rt_context_retain(ctx);
    d->prm.exposure = 1.0f;          // src/DenoiseCompositor.cpp:44-49
    d->prm.gamma = 2.2f;
    d->prm.tonemap = 1;
    d->prm.gammaCorrect = 0;
    d->prm.maxKernelSize = 12;
    d->prm.debugVisualize = 0;
    *out = d;
    return RT_OK;
}

int rt_denoiser_destroy(rt_denoiser *d)
{
    if (!d) return RT_OK;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    d->out[0].release(); d->out[1].release(); d->half_out.release();
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    rt_context *ctx = d->ctx;
    delete d;
    rt_context_release(ctx);
    return RT_OK;
}

int rt_denoiser_get_params(rt_denoiser *d, rt_denoiser_params **params)
{
    RT_REQUIRE(d && params, "null argument");
    *params = &d->prm;
    return RT_OK;
}

int rt_denoiser_create_output(rt_denoiser *d, uint32_t format, uint32_t width, uint32_t height)
{
    RT_REQUIRE(d, "null denoiser");
    RT_REQUIRE(width > 0 && height > 0, "empty output");
    RT_REQUIRE(format == RT_FORMAT_R32G32B32A32_FLOAT || format == RT_FORMAT_R16G16B16A16_FLOAT, "unsupported output format");
    HIP_TRY(hipSetDevice(d->ctx->device));
    for (DevBuf &b : d->out) RT_TRY(b.reserve((size_t)width * height * 16));
    d->width = width; d->height = height; d->format = format;
    return RT_OK;
}

int rt_denoiser_dispatch(rt_denoiser *d, const void *direct_lighting, const void *indirect_specular, uint32_t width, uint32_t height)
{
    RT_REQUIRE(d && direct_lighting && indirect_specular, "null argument");
    if (!d->out[0].p) { rt_set_error("dispatch: createOutputResource has not been called"); return RT_ERR_STATE; }
    RT_REQUIRE(width == d->width && height == d->height, "width/height differ from the output resource");
    if (d->prm.maxKernelSize < 0 || d->prm.maxKernelSize > MAX_EXTENT) {
        rt_set_error("maxKernelSize %d outside 0..%d (the filter's tile halo)", d->prm.maxKernelSize, MAX_EXTENT);
        return RT_ERR_INVALID_ARG;
    }
    HIP_TRY(hipSetDevice(d->ctx->device));
    hipStream_t st = d->ctx->stream;
    if (!d->ev0) { HIP_TRY(hipEventCreate(&d->ev0)); HIP_TRY(hipEventCreate(&d->ev1)); }
    DenoiseArgs a;
    a.direct = (const float4 *)direct_lighting;
    a.width = (int)width; a.height = (int)height;
    a.prm = d->prm;
    for (int i = -MAX_EXTENT; i <= MAX_EXTENT; i++) a.w[i + MAX_EXTENT] = tap_weight(i, (float)d->prm.maxKernelSize);
    HIP_TRY(hipEventRecord(d->ev0, st));
    a.input = (const float4 *)indirect_specular;                 // pass 0: Dispatch(ceil(w/64), h, 1) in the reference
    a.output = d->out[0].as<float4>();
    k_denoise_h<<<dim3((width + HB - 1) / HB, height), HB, 0, st>>>(a);
    a.input = d->out[0].as<float4>();                            // pass 1 reads pass 0's output
    a.output = d->out[1].as<float4>();
    k_denoise_v<<<dim3((width + VW - 1) / VW, (height + VH - 1) / VH), 256, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d->ev1, st));
    d->dispatched = true;
    return RT_OK;
}

int rt_denoiser_get_output_device_ptr(rt_denoiser *d, void **ptr)
{
    RT_REQUIRE(d && ptr, "null argument");
    *ptr = d->out[1].p;
    return RT_OK;
}

static int denoiser_read(rt_denoiser *d, int which, void *host, size_t bytes)
{
    RT_REQUIRE(d && host, "null argument");
    if (!d->dispatched) { rt_set_error("nothing dispatched yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(d->ctx->device));
    hipStream_t st = d->ctx->stream;
    const size_t npix = (size_t)d->width * d->height;
    if (d->format == RT_FORMAT_R16G16B16A16_FLOAT) {
        RT_REQUIRE(bytes == npix * 8, "host buffer must be width*height*8 bytes for RGBA16F");
        RT_TRY(d->half_out.reserve(npix * 8));
        k_denoise_f16<<<(unsigned)((npix + 255) / 256), 256, 0, st>>>(d->out[which].as<float4>(), d->half_out.as<ushort4>(), npix);
        HIP_TRY(hipMemcpyAsync(host, d->half_out.p, bytes, hipMemcpyDeviceToHost, st));
    } else {
        RT_REQUIRE(bytes == npix * 16, "host buffer must be width*height*16 bytes for RGBA32F");
        HIP_TRY(hipMemcpyAsync(host, d->out[which].p, bytes, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_denoiser_read_output(rt_denoiser *d, void *host, size_t bytes) { return denoiser_read(d, 1, host, bytes); }
int rt_denoiser_read_intermediate(rt_denoiser *d, void *host, size_t bytes) { return denoiser_read(d, 0, host, bytes); }

int rt_denoiser_last_ms(rt_denoiser *d, float *ms)
{
    RT_REQUIRE(d && ms, "null argument");
    if (!d->dispatched) { rt_set_error("nothing dispatched yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(d->ctx->device));
    HIP_TRY(hipEventSynchronize(d->ev1));
    HIP_TRY(hipEventElapsedTime(ms, d->ev0, d->ev1));
    return RT_OK;
}

}  // extern "C"
