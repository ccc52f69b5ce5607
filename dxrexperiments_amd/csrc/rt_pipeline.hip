// rt_pipeline.hip -- the ProgressiveRaytracingPipeline on gfx950.
//
// The reference renders a frame with ONE DispatchRays whose raygen shader recurses
// through TraceRay (src/ProgressiveRaytracingPipeline.cpp:215-247 ->
// assets/shaders/ProgressiveRaytracing.hlsl).  Here the same per-pixel recursion
// (depth <= 1 radiance, <= 2 shadow; RaytracingCommon.hlsli:11-12) is unrolled into
// a fixed wavefront DAG of ray queues in HBM, one kernel per stage:
//
//   primary      raygen + closest-hit traversal (cull back faces)  -> hit0, compaction of hit pixels
//   shade0/emit  PrimaryClosestHit -> shade(): emits 2 shadow rays + diffuse + specular secondary rays
//   trace        any-hit over the shadow queue; closest-hit over the secondary queue (+ compaction)
//   shade1/emit  closest-hit shading of secondary hits: emits their 2 shadow rays each
//   trace        any-hit over the second shadow queue
//   resolve      re-runs shade() with every TraceRay replaced by its stored result, then
//                gOutput = (n*prev + cur)/(n+1)                      (ProgressiveRaytracing.hlsl:36-38)
//
// shade() is ONE template used in both the emit and the resolve stage, so the
// arithmetic (and the RNG draw order) of both passes is identical by construction.
// Queues are SoA float4 arrays (origin|tmin, direction|tmax) so a wave reads 1 KiB
// per instruction; live rays are compacted with __ballot + popcount prefix sums and
// one atomic per wave; diffuse and specular secondaries sit in separate batches so
// waves stay as coherent as the sampling allows.
#include <hip/hip_fp16.h>

#include <new>

#include "rt_trace_wave.h"

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size);

using namespace rtd;

namespace {

constexpr int PBLOCK = 256;

#define RAY_MAX_T 1.0e+38f      // RaytracingCommon.hlsli:8
#define RAY_EPSILON 0.0001f     // RaytracingCommon.hlsli:9
#define HLSL_PI 3.1415927f      // RaytracingUtils.hlsli:22

#define HIT_MISS -1.0f
#define HIT_UNTRACED -2.0f

enum { C_N0 = 0, C_N1 = 1, C_SECONDARY = 2, C_SHADOW = 3, C_POOL_PRIMARY = 4, C_POOL_SECONDARY = 5, C_POOL_SHADOW0 = 6,
       C_POOL_SHADOW1 = 7, C_COUNT = 8 };

struct PipeDev {
    SceneDev sc;
    rt_per_frame_constants pfc;
    const rt_material_params *mats;
    uint32_t nmats;
    const float4 *env;
    uint32_t env_size;
    float env_const[3];
    uint32_t width, height;
    uint32_t x0, y0, tw, th, cap;       // tile rectangle; cap = tiles_x * tiles_y * 64 pixel slots
    uint32_t tiles_x;
    uint32_t max_rad, max_shadow;
    uint32_t accum_mode;
    uint32_t kind;                      // RT_PIPELINE_PROGRESSIVE / RT_PIPELINE_REALTIME
    float4 *accum;
    float4 *aov_direct, *aov_indirect;  // realtime pipeline outputs (RealtimeRaytracing.hlsl:3-4)
    float4 *hit0; uint32_t *inst0;
    uint32_t *pix_k, *klist;
    uint32_t *counters;
    float4 *secO, *secD, *hit1; uint32_t *inst1;
    uint32_t *slot_j, *jlist;
    float4 *sh0O, *sh0D; uint32_t *vis0;
    float4 *sh1O, *sh1D; uint32_t *vis1;
};

// ---- environment: TextureCube.SampleLevel(linear, dir, 0), RaytracingCommon.hlsli:149-159
RT_DEV f3 sample_cube(const PipeDev &pd, f3 d)
{
    if (pd.env_size == 0) return mk3(pd.env_const[0], pd.env_const[1], pd.env_const[2]);
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    int face; float ma, sc, tc;
    if (ax >= ay && ax >= az) { face = d.x > 0.0f ? 0 : 1; ma = ax; sc = d.x > 0.0f ? -d.z : d.z; tc = -d.y; }
    else if (ay >= az) { face = d.y > 0.0f ? 2 : 3; ma = ay; sc = d.x; tc = d.y > 0.0f ? d.z : -d.z; }
    else { face = d.z > 0.0f ? 4 : 5; ma = az; sc = d.z > 0.0f ? d.x : -d.x; tc = -d.y; }
    if (!(ma > 0.0f) || !(ma < __uint_as_float(0x7f800000u))) return mk3(0.0f, 0.0f, 0.0f);
    const float u = (sc / ma + 1.0f) * 0.5f;
    const float v = (tc / ma + 1.0f) * 0.5f;
    const float n = (float)pd.env_size;
    const float fx = u * n - 0.5f, fy = v * n - 0.5f;
    const float x0f = __builtin_floorf(fx), y0f = __builtin_floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int m = (int)pd.env_size - 1;
    int x0 = (int)x0f, y0 = (int)y0f;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), m); x1 = min(max(x1, 0), m);
    y0 = min(max(y0, 0), m); y1 = min(max(y1, 0), m);
    const float4 *f = pd.env + (size_t)face * pd.env_size * pd.env_size;
    const float4 c00 = f[(size_t)y0 * pd.env_size + x0], c10 = f[(size_t)y0 * pd.env_size + x1];
    const float4 c01 = f[(size_t)y1 * pd.env_size + x0], c11 = f[(size_t)y1 * pd.env_size + x1];
    const float tx = c00.x + (c10.x - c00.x) * wx, bx = c01.x + (c11.x - c01.x) * wx;
    const float ty = c00.y + (c10.y - c00.y) * wx, by = c01.y + (c11.y - c01.y) * wx;
    const float tz = c00.z + (c10.z - c00.z) * wx, bz = c01.z + (c11.z - c01.z) * wx;
    return mk3(tx + (bx - tx) * wy, ty + (by - ty) * wy, tz + (bz - tz) * wy);
}

RT_DEV f3 sample_environment(const PipeDev &pd, f3 dir)
{
    return sample_cube(pd, dir) * pd.pfc.options.environmentStrength;
}

// Pixel slot q -> pixel.  Slots are laid out as 8x8 pixel tiles (64 consecutive slots = one
// wave = one 8x8 screen tile), so a wave's primary rays -- and, through the order-preserving
// compaction, the secondary and shadow rays spawned from them -- share BVH nodes.  Slots of
// partial tiles that fall outside the rectangle are invalid.
RT_DEV bool pix_xy(const PipeDev &pd, uint32_t q, uint32_t &px, uint32_t &py)
{
    const uint32_t t = q >> 6, w = q & 63u;
    const uint32_t lx = (t % pd.tiles_x) * 8u + (w & 7u), ly = (t / pd.tiles_x) * 8u + (w >> 3);
    px = pd.x0 + lx;
    py = pd.y0 + ly;
    return lx < pd.tw && ly < pd.th;
}

// ---- RayGen (ProgressiveRaytracing.hlsl:18-32)
RT_DEV RayD primary_ray(const PipeDev &pd, uint32_t px, uint32_t py)
{
    const rt_camera_params &cp = pd.pfc.cameraParams;
    const float dx = ((float)px + 0.5f) / (float)pd.width * 2.0f - 1.0f;
    const float dy = ((float)py + 0.5f) / (float)pd.height * 2.0f - 1.0f;
    const float js = pd.kind == RT_PIPELINE_REALTIME ? 10.0f : 30.0f;   // ProgressiveRaytracing.hlsl:26 / RealtimeRaytracing.hlsl:33
    const float jx = cp.jitters.x * js, jy = cp.jitters.y * js;
    RayD r;
    r.o = mk3(cp.worldEyePos.x + jx, cp.worldEyePos.y + jy, cp.worldEyePos.z + 0.0f);
    f3 dir = mk3(cp.U.x, cp.U.y, cp.U.z) * dx;
    dir = dir + mk3(cp.V.x, cp.V.y, cp.V.z) * (-dy);
    dir = dir + mk3(cp.W.x, cp.W.y, cp.W.z);
    r.d = normalize(dir);
    r.tmin = 0.0f;
    r.tmax = RAY_MAX_T;
    return r;
}

// ---- interpolateVertexAttributes (RaytracingCommon.hlsli:53-82), normal only
RT_DEV f3 hit_normal(const InstanceRec &in, uint32_t prim, float bu, float bv)
{
    const float b0 = 1.0f - bu - bv;
    const rt_float3 n0 = in.verts[in.indices[3 * prim + 0]].normal;
    const rt_float3 n1 = in.verts[in.indices[3 * prim + 1]].normal;
    const rt_float3 n2 = in.verts[in.indices[3 * prim + 2]].normal;
    f3 n = mk3(n0.x, n0.y, n0.z) * b0;
    n = n + mk3(n1.x, n1.y, n1.z) * bu;
    n = n + mk3(n2.x, n2.y, n2.z) * bv;
    return n;
}

// ---- lights (RaytracingCommon.hlsli:126-147), AO (:98-124)
template <class IO>
RT_DEV f3 directional_light(const PipeDev &pd, IO &io, f3 P, f3 N, uint32_t depth)
{
    const rt_directional_light_params &dl = pd.pfc.directionalLight;
    const f3 L = normalize(mk3(-dl.forwardDir.x, -dl.forwardDir.y, -dl.forwardDir.z));
    const float NoL = saturate(dot(N, L));
    const float vis = io.shadow(0, P, L, RAY_EPSILON, RAY_MAX_T, depth);
    return mk3(dl.color.x, dl.color.y, dl.color.z) * dl.color.w * NoL * vis;
}

template <class IO>
RT_DEV f3 point_light(const PipeDev &pd, IO &io, f3 P, f3 N, uint32_t depth)
{
    const rt_point_light_params &pl = pd.pfc.pointLight;
    const f3 path = mk3(pl.worldPos.x, pl.worldPos.y, pl.worldPos.z) - P;
    const float dist = length(path);
    const f3 L = normalize(path);
    const float NoL = saturate(dot(N, L));
    const float vis = io.shadow(1, P, L, RAY_EPSILON, dist - RAY_EPSILON, depth);
    const float falloff = 1.0f / (2.0f * HLSL_PI * dist * dist);
    return mk3(pl.color.x, pl.color.y, pl.color.z) * pl.color.w * NoL * vis * falloff;
}

template <class IO>
RT_DEV f3 ambient_occlusion(const PipeDev &pd, IO &io, f3 P, f3 N, uint32_t pix)
{
    float visibility = 0.0f;
    uint32_t seed = init_rand(pix, pd.pfc.cameraParams.frameCount);
    for (int i = 0; i < 4; ++i) {
        f3 dir; float NoL, pdf;
        if (pd.pfc.options.cosineHemisphereSampling) {
            dir = cos_hemisphere(seed, N);
            NoL = saturate(dot(N, dir));
            pdf = NoL / HLSL_PI;
        } else {
            dir = uniform_hemisphere(seed, N);
            NoL = saturate(dot(N, dir));
            pdf = 1.0f / (2.0f * HLSL_PI);
        }
        visibility += io.shadow(i, P, dir, RAY_EPSILON, 10.0f, 1u) * NoL / pdf;
    }
    const float r = visibility / 4.0f;
    return mk3(r, r, r);
}

// ---- shade (ProgressiveRaytracing.hlsl:80-148) + evaluateIndirectDiffuse (:57-78)
template <class IO>
RT_DEV f3 shade(const PipeDev &pd, IO &io, const rt_material_params &mp, f3 P, f3 N, f3 D, uint32_t depth, uint32_t pix)
{
    const rt_debug_options &opt = pd.pfc.options;
    if (opt.showAmbientOcclusionOnly) return ambient_occlusion(pd, io, P, N, pix);

    uint32_t seed = init_rand(pix, pd.pfc.cameraParams.frameCount);

    f3 direct = mk3(0.0f, 0.0f, 0.0f);
    if (opt.debug == 2) {
        if (next_rand(seed) < 0.5f) direct = direct + directional_light(pd, io, P, N, depth) * 2.0f;
        else direct = direct + point_light(pd, io, P, N, depth) * 2.0f;
    } else {
        direct = direct + directional_light(pd, io, P, N, depth);
        direct = direct + point_light(pd, io, P, N, depth);
    }

    f3 indirect = mk3(0.0f, 0.0f, 0.0f);
    if (depth < 1 && !opt.noIndirectDiffuse) {
        f3 color = mk3(0.0f, 0.0f, 0.0f);
        if (opt.cosineHemisphereSampling) {
            const f3 dir = cos_hemisphere(seed, N);
            color = color + io.secondary(0, P, dir, RAY_EPSILON, depth) * HLSL_PI;
        } else {
            const f3 dir = uniform_hemisphere(seed, N);
            const float NoL = saturate(dot(N, dir));
            const float pdf = 1.0f / (2.0f * HLSL_PI);
            color = color + io.secondary(0, P, dir, RAY_EPSILON, depth) * NoL / pdf;
        }
        indirect = indirect + color / 1.0f;
    }

    const f3 diffuse = (direct + indirect) / HLSL_PI;

    f3 fresnel = mk3(0.0f, 0.0f, 0.0f);
    f3 specular = mk3(0.0f, 0.0f, 0.0f);
    if (mp.type == 1u || mp.type == 2u) {
        if (mp.reflectivity > 0.001f) {
            const float exponent = exp_det((1.0f - mp.roughness) * 12.0f);
            float pdf, brdf;
            const f3 mirror = reflect(D, N);
            const f3 dir = phong_lobe(seed, mirror, exponent, pdf, brdf);
            const f3 refl = io.secondary(1, P, dir, RAY_EPSILON, depth);
            specular = specular + refl * brdf / pdf;
            fresnel = fresnel_schlick(D, N, mk3(mp.specular.x, mp.specular.y, mp.specular.z));
        }
    }

    const f3 albedo = mk3(mp.albedo.x, mp.albedo.y, mp.albedo.z);
    if (depth == 0) {
        if (opt.showIndirectDiffuseOnly) return albedo * indirect / HLSL_PI;
        else if (opt.showIndirectSpecularOnly) return specular * mp.reflectivity * fresnel;
        else if (opt.showFresnelTerm) return fresnel;
        else if (opt.showGBufferAlbedoOnly) return albedo;
        else if (opt.showDirectLightingOnly) return albedo * direct / HLSL_PI;
    }
    f3 r = mk3(mp.emissive.x, mp.emissive.y, mp.emissive.z) * mp.emissive.w;
    r = r + albedo * diffuse;
    r = r + specular * mp.reflectivity * fresnel;
    return r;
}

// shadeAOV of the realtime pipeline (RealtimeRaytracing.hlsl:65-103): direct light + one Phong-lobe
// bounce, split into the two AOVs the denoiser consumes (written at depth 0 only)
template <class IO>
RT_DEV f3 shade_aov(const PipeDev &pd, IO &io, const rt_material_params &mp, f3 P, f3 N, f3 D, uint32_t depth, uint32_t pix,
                    f3 &aov_direct, f3 &aov_indirect)
{
    uint32_t seed = init_rand(pix, pd.pfc.cameraParams.frameCount);
    f3 direct = mk3(0.0f, 0.0f, 0.0f);
    direct = direct + directional_light(pd, io, P, N, depth);
    direct = direct + point_light(pd, io, P, N, depth);
    f3 fresnel = mk3(0.0f, 0.0f, 0.0f);
    f3 specular = mk3(0.0f, 0.0f, 0.0f);
    if (mp.type == 1u || mp.type == 2u) {
        if (mp.reflectivity > 0.001f) {
            const float exponent = exp_det((1.0f - mp.roughness) * 12.0f);
            float pdf, brdf;
            const f3 mirror = reflect(D, N);
            const f3 dir = phong_lobe(seed, mirror, exponent, pdf, brdf);
            const f3 refl = io.secondary(1, P, dir, RAY_EPSILON, depth);
            specular = specular + refl * brdf / pdf;
            fresnel = fresnel_schlick(D, N, mk3(mp.specular.x, mp.specular.y, mp.specular.z));
        }
    }
    const f3 albedo = mk3(mp.albedo.x, mp.albedo.y, mp.albedo.z);
    const f3 dl = albedo * direct / HLSL_PI;
    const f3 is = specular * mp.reflectivity * fresnel;
    if (depth == 0) { aov_direct = dl; aov_indirect = is; }
    return dl + is;
}

struct Shaded { f3 color, aov_direct, aov_indirect; };

// PrimaryClosestHit (ProgressiveRaytracing.hlsl:150-158, RealtimeRaytracing.hlsl:105-117) for a stored hit
template <class IO>
RT_DEV Shaded closest_hit_aov(const PipeDev &pd, IO &io, const RayD &r, float t, float u, float v, uint32_t prim, uint32_t inst,
                              uint32_t depth, uint32_t pix)
{
    const InstanceRec &in = pd.sc.inst[inst];
    const f3 N = normalize(hit_normal(in, prim, u, v));
    const f3 P = r.o + r.d * t;
    const rt_material_params mp = pd.mats[min(inst, pd.nmats - 1u)];
    Shaded s;
    s.aov_direct = mk3(0.0f, 0.0f, 0.0f);
    s.aov_indirect = mk3(0.0f, 0.0f, 0.0f);
    if (pd.kind == RT_PIPELINE_REALTIME) s.color = shade_aov(pd, io, mp, P, N, r.d, depth, pix, s.aov_direct, s.aov_indirect);
    else s.color = shade(pd, io, mp, P, N, r.d, depth, pix);
    return s;
}

template <class IO>
RT_DEV f3 closest_hit(const PipeDev &pd, IO &io, const RayD &r, float t, float u, float v, uint32_t prim, uint32_t inst,
                      uint32_t depth, uint32_t pix)
{
    return closest_hit_aov(pd, io, r, t, u, v, prim, inst, depth, pix).color;
}

RT_DEV void store_ray(float4 *O, float4 *D, size_t slot, f3 o, float tmin, f3 d, float tmax)
{
    O[slot] = make_float4(o.x, o.y, o.z, tmin);
    D[slot] = make_float4(d.x, d.y, d.z, tmax);
}
RT_DEV void store_invalid(float4 *O, float4 *D, size_t slot)
{
    O[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    D[slot] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);      // tmax < tmin: never traced
}
RT_DEV RayD load_ray(const float4 *O, const float4 *D, size_t slot)
{
    const float4 a = O[slot], b = D[slot];
    RayD r;
    r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
    r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
    return r;
}

// ---- the "TraceRay" providers of the two passes ---------------------------------

// emit pass, depth 0: shadow slot s -> sh0[s*cap + k], secondary w -> sec[w*cap + k]
struct EmitIO0 {
    const PipeDev &pd;
    uint32_t k;
    uint32_t shadow_mask, sec_mask;
    RT_DEV EmitIO0(const PipeDev &p, uint32_t kk) : pd(p), k(kk), shadow_mask(0), sec_mask(0) {}
    RT_DEV float shadow(int s, f3 o, f3 d, float tmin, float tmax, uint32_t depth)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        store_ray(pd.sh0O, pd.sh0D, (size_t)s * pd.cap + k, o, tmin, d, tmax);
        shadow_mask |= 1u << s;
        return 1.0f;
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if (depth >= pd.max_rad) return mk3(0.0f, 0.0f, 0.0f);
        store_ray(pd.secO, pd.secD, (size_t)w * pd.cap + k, o, tmin, d, RAY_MAX_T);
        sec_mask |= 1u << w;
        return mk3(0.0f, 0.0f, 0.0f);
    }
};

// emit pass, depth 1: shadow slot s -> sh1[s*2cap + j]; secondaries are never traced at depth 1
struct EmitIO1 {
    const PipeDev &pd;
    uint32_t j;
    uint32_t shadow_mask;
    RT_DEV EmitIO1(const PipeDev &p, uint32_t jj) : pd(p), j(jj), shadow_mask(0) {}
    RT_DEV float shadow(int s, f3 o, f3 d, float tmin, float tmax, uint32_t depth)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        store_ray(pd.sh1O, pd.sh1D, (size_t)s * 2u * pd.cap + j, o, tmin, d, tmax);
        shadow_mask |= 1u << s;
        return 1.0f;
    }
    RT_DEV f3 secondary(int, f3, f3, float, uint32_t) { return mk3(0.0f, 0.0f, 0.0f); }
};

struct ResolveIO1 {
    const PipeDev &pd;
    uint32_t j;
    RT_DEV ResolveIO1(const PipeDev &p, uint32_t jj) : pd(p), j(jj) {}
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        return pd.vis1[(size_t)s * 2u * pd.cap + j] ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int, f3, f3, float, uint32_t) { return mk3(0.0f, 0.0f, 0.0f); }
};

struct ResolveIO0 {
    const PipeDev &pd;
    uint32_t k, pix;
    RT_DEV ResolveIO0(const PipeDev &p, uint32_t kk, uint32_t px) : pd(p), k(kk), pix(px) {}
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        return pd.vis0[(size_t)s * pd.cap + k] ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if (depth >= pd.max_rad) return mk3(0.0f, 0.0f, 0.0f);
        const size_t slot = (size_t)w * pd.cap + k;
        const float4 h = pd.hit1[slot];
        if (h.x == HIT_MISS) return sample_environment(pd, d);             // PrimaryMiss, :160-164
        RayD r;
        r.o = o; r.tmin = tmin; r.d = d; r.tmax = RAY_MAX_T;
        ResolveIO1 io(pd, pd.slot_j[slot]);
        return closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.inst1[slot], depth + 1u, pix);
    }
};

// ---- wave compaction ---------------------------------------------------------------
// Order-preserving compaction inside a block of CBLOCK threads: ballot + popcount give the rank inside
// the wave, wave totals meet in LDS, and ONE atomic per block reserves the output range (same-address
// returning atomics cost ~11 ns each on this chip, so one per wave would dominate these tiny kernels).
constexpr int CBLOCK = 1024;
RT_DEV uint32_t block_compact(bool keep, uint32_t *counter)
{
    __shared__ uint32_t wave_total[CBLOCK / 64];
    __shared__ uint32_t block_base;
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < CBLOCK / 64; w++) { const uint32_t c = wave_total[w]; wave_total[w] = sum; sum += c; }
        block_base = sum ? atomicAdd(counter, sum) : 0u;
    }
    __syncthreads();
    return block_base + wave_total[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// every lane of the wave must call this (no early exits before it)
RT_DEV void wave_add(uint32_t v, uint32_t *counter)
{
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(counter, v);
}

// ---- kernels -------------------------------------------------------------------------

// primary stage: raygen is the ray source, hit0/inst0 the sink (pixel-indexed)
struct PrimarySrc {
    const PipeDev &pd;
    RT_DEV uint32_t count() const { return pd.cap; }
    RT_DEV uint32_t flags() const { return RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES; }      // ProgressiveRaytracing.hlsl:34
    RT_DEV bool load(uint32_t q, RayD &r) const
    {
        uint32_t px, py;
        const bool valid = pix_xy(pd, q, px, py);
        r = primary_ray(pd, px, py);
        return valid;
    }
};
struct PrimarySink {
    const PipeDev &pd;
    RT_DEV void store(uint32_t q, const HitD &h, bool) const
    {
        const bool hit = h.inst != RT_NO_HIT;
        pd.hit0[q] = make_float4(hit ? h.t : HIT_MISS, h.u, h.v, __uint_as_float(h.prim));
        pd.inst0[q] = h.inst;
    }
};

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_primary(PipeDev pd)
{
    __shared__ int smem[STACK * PBLOCK];
    PrimarySrc src = {pd};
    PrimarySink sink = {pd};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, 64u>(pd.sc, src, sink, &pd.counters[C_POOL_PRIMARY], smem, nullptr);   // one 8x8 tile per wave
}

// compaction of the pixels whose primary ray hit: ballot + popcount prefix sum, one atomic per wave
__global__ void __launch_bounds__(CBLOCK) k_compact_primary(PipeDev pd)
{
    const uint32_t q = blockIdx.x * CBLOCK + threadIdx.x;
    const bool hit = q < pd.cap && pd.hit0[q].x != HIT_MISS;
    const uint32_t k = block_compact(hit, &pd.counters[C_N0]);
    if (q < pd.cap) pd.pix_k[q] = hit ? k : RT_NO_HIT;
    if (hit) pd.klist[k] = q;
}

__global__ void __launch_bounds__(PBLOCK) k_shade0_emit(PipeDev pd, uint32_t shadow_slots)
{
    const uint32_t k = blockIdx.x * PBLOCK + threadIdx.x;
    uint32_t n_shadow = 0, n_sec = 0;
    if (k < pd.counters[C_N0]) {
        const uint32_t q = pd.klist[k];
        uint32_t px, py;
        (void)pix_xy(pd, q, px, py);
        const RayD r = primary_ray(pd, px, py);
        const float4 h = pd.hit0[q];
        EmitIO0 io(pd, k);
        (void)closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.inst0[q], 0u, px + py * pd.width);
        for (uint32_t s = 0; s < shadow_slots; s++)
            if (!(io.shadow_mask & (1u << s))) store_invalid(pd.sh0O, pd.sh0D, (size_t)s * pd.cap + k);
        for (uint32_t w = 0; w < 2; w++)
            if (!(io.sec_mask & (1u << w))) store_invalid(pd.secO, pd.secD, (size_t)w * pd.cap + k);
        n_shadow = (uint32_t)__popc(io.shadow_mask);
        n_sec = (uint32_t)__popc(io.sec_mask);
    }
    (void)n_shadow; (void)n_sec;       // rays are counted by the traversal waves (one atomic per persistent wave)
}

// a ray queue of `batches` batches of *count rays; batch b lives at [b*stride, b*stride + *count)
struct QueueSrc {
    const float4 *O, *D;
    const uint32_t *count_ptr;
    uint32_t stride, batches, fl;
    RT_DEV uint32_t n() const { return *count_ptr; }
    RT_DEV uint32_t count() const { return n() * batches; }
    RT_DEV uint32_t flags() const { return fl; }
    RT_DEV size_t slot(uint32_t i) const { const uint32_t c = n(); return (size_t)(i / c) * stride + i % c; }
    RT_DEV bool load(uint32_t i, RayD &r) const
    {
        const size_t sl = slot(i);
        const v4f a = ldg16(O, sl * 16), b = ldg16(D, sl * 16);
        r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
        r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
        return r.tmax > r.tmin;
    }
};

struct ShadowSink {      // ShadowMiss sets visibility 1 (ProgressiveRaytracing.hlsl:178-182)
    QueueSrc q;
    uint32_t *vis;
    RT_DEV void store(uint32_t i, const HitD &h, bool) const { vis[q.slot(i)] = h.inst == RT_NO_HIT ? 1u : 0u; }
};

struct SecondarySink {
    QueueSrc q;
    float4 *hit1;
    uint32_t *inst1;
    RT_DEV void store(uint32_t i, const HitD &h, bool traced) const
    {
        const size_t sl = q.slot(i);
        const bool hit = h.inst != RT_NO_HIT;
        hit1[sl] = make_float4(hit ? h.t : (traced ? HIT_MISS : HIT_UNTRACED), h.u, h.v, __uint_as_float(h.prim));
        inst1[sl] = h.inst;
    }
};

// every shadow ray of the frame in ONE launch: queue 0 (rays from first hits) then queue 1 (from second hits)
struct ShadowSrc2 {
    QueueSrc a, b;
    RT_DEV uint32_t count() const { return a.count() + b.count(); }
    RT_DEV uint32_t flags() const { return a.fl; }
    RT_DEV bool load(uint32_t i, RayD &r) const { const uint32_t ca = a.count(); return i < ca ? a.load(i, r) : b.load(i - ca, r); }
};
struct ShadowSink2 {
    ShadowSrc2 q;
    uint32_t *vis_a, *vis_b;
    RT_DEV void store(uint32_t i, const HitD &h, bool) const
    {
        const uint32_t ca = q.a.count();
        const uint32_t v = h.inst == RT_NO_HIT ? 1u : 0u;
        if (i < ca) vis_a[q.a.slot(i)] = v;
        else vis_b[q.b.slot(i - ca)] = v;
    }
};

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_trace_shadow(SceneDev sc, ShadowSrc2 src, uint32_t *vis_a, uint32_t *vis_b, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[STACK * PBLOCK];
    ShadowSink2 sink = {src, vis_a, vis_b};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK>(sc, src, sink, pool, smem, stat);
}

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_trace_secondary(SceneDev sc, QueueSrc src, float4 *hit1, uint32_t *inst1, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[STACK * PBLOCK];
    SecondarySink sink = {src, hit1, inst1};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK>(sc, src, sink, pool, smem, stat);
}

// compaction of the secondary rays that hit (they get shaded and emit shadow rays)
__global__ void __launch_bounds__(CBLOCK) k_compact_secondary(PipeDev pd)
{
    const uint32_t n = pd.counters[C_N0];
    const uint32_t idx = blockIdx.x * CBLOCK + threadIdx.x;
    const bool in_range = idx < 2u * n;
    const size_t slot = in_range ? (size_t)(idx / n) * pd.cap + idx % n : 0;
    const bool hit = in_range && pd.hit1[slot].x >= 0.0f;
    const uint32_t j = block_compact(hit, &pd.counters[C_N1]);
    if (in_range) pd.slot_j[slot] = hit ? j : RT_NO_HIT;
    if (hit) pd.jlist[j] = (uint32_t)slot;
}

__global__ void __launch_bounds__(PBLOCK) k_shade1_emit(PipeDev pd)
{
    const uint32_t j = blockIdx.x * PBLOCK + threadIdx.x;
    uint32_t n_shadow = 0;
    if (j < pd.counters[C_N1]) {
        const uint32_t slot = pd.jlist[j];
        const uint32_t k = slot % pd.cap;
        const uint32_t q = pd.klist[k];
        uint32_t px, py;
        (void)pix_xy(pd, q, px, py);
        const RayD r = load_ray(pd.secO, pd.secD, slot);
        const float4 h = pd.hit1[slot];
        EmitIO1 io(pd, j);
        (void)closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.inst1[slot], 1u, px + py * pd.width);
        for (uint32_t s = 0; s < 2; s++)
            if (!(io.shadow_mask & (1u << s))) store_invalid(pd.sh1O, pd.sh1D, (size_t)s * 2u * pd.cap + j);
        n_shadow = (uint32_t)__popc(io.shadow_mask);
    }
    (void)n_shadow;
}

__global__ void __launch_bounds__(PBLOCK) k_resolve(PipeDev pd)
{
    const uint32_t q = blockIdx.x * PBLOCK + threadIdx.x;
    if (q >= pd.cap) return;
    uint32_t px, py;
    if (!pix_xy(pd, q, px, py)) return;
    const RayD r = primary_ray(pd, px, py);
    const float4 h = pd.hit0[q];
    Shaded sh;
    if (h.x == HIT_MISS) {
        sh.color = sample_environment(pd, r.d);             // PrimaryMiss
        sh.aov_direct = sh.color;                           // RealtimeRaytracing.hlsl:119-126
        sh.aov_indirect = mk3(0.0f, 0.0f, 0.0f);
    } else {
        ResolveIO0 io(pd, pd.pix_k[q], px + py * pd.width);
        sh = closest_hit_aov(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.inst0[q], 0u, px + py * pd.width);
    }
    const size_t pixel = (size_t)py * pd.width + px;
    if (pd.kind == RT_PIPELINE_REALTIME) {                  // RealtimeRaytracing.hlsl:44-45: two AOVs, no accumulation
        pd.aov_direct[pixel] = make_float4(fmax2(sh.aov_direct.x, 0.0f), fmax2(sh.aov_direct.y, 0.0f), fmax2(sh.aov_direct.z, 0.0f), 1.0f);
        pd.aov_indirect[pixel] = make_float4(fmax2(sh.aov_indirect.x, 0.0f), fmax2(sh.aov_indirect.y, 0.0f), fmax2(sh.aov_indirect.z, 0.0f), 1.0f);
        return;
    }
    const f3 c = sh.color;
    const float4 cur = make_float4(fmax2(c.x, 0.0f), fmax2(c.y, 0.0f), fmax2(c.z, 0.0f), 1.0f);
    float4 *dst = pd.accum + pixel;
    const float4 prev = *dst;
    float4 o;
    if (pd.accum_mode == RT_ACCUM_SUM) {
        o = make_float4(prev.x + cur.x, prev.y + cur.y, prev.z + cur.z, prev.w + cur.w);
    } else {
        const float n = (float)pd.pfc.cameraParams.accumCount;
        const float n1 = (float)(pd.pfc.cameraParams.accumCount + 1u);
        o = make_float4((n * prev.x + cur.x) / n1, (n * prev.y + cur.y) / n1, (n * prev.z + cur.z) / n1, (n * prev.w + cur.w) / n1);
    }
    *dst = o;
}

__global__ void k_add_totals(const uint32_t *__restrict__ counters, unsigned long long *__restrict__ totals, uint32_t cap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    totals[0] += cap;
    totals[1] += counters[C_SECONDARY];
    totals[2] += counters[C_SHADOW];
    totals[3] += counters[C_N0];
    totals[4] += counters[C_N1];
    totals[5] += 1;
}

RT_DEV void wave_add64(unsigned long long v, unsigned long long *counter)
{
    for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(counter, v);
}

// canonical-order re-trace of a queue: sums rays / nodes / triangles into out[0..2]
__global__ void __launch_bounds__(PBLOCK)
k_count_queue(SceneDev sc, const float4 *__restrict__ O, const float4 *__restrict__ D, const uint32_t *__restrict__ count,
              uint32_t stride, uint32_t batches, uint32_t flags, unsigned long long *__restrict__ out)
{
    const uint32_t n = *count;
    const uint32_t idx = blockIdx.x * PBLOCK + threadIdx.x;
    const uint32_t per = (n + PBLOCK - 1) / PBLOCK * PBLOCK;
    const uint32_t b = per ? idx / per : batches, k = per ? idx % per : 0;
    unsigned long long rays = 0, nodes = 0, tris = 0;
    if (b < batches && k < n) {
        const RayD r = load_ray(O, D, (size_t)b * stride + k);
        if (r.tmax > r.tmin) {
            uint32_t cn, ct;
            (void)trace_canonical(sc, r, flags, cn, ct);
            rays = 1; nodes = cn; tris = ct;
        }
    }
    wave_add64(rays, &out[0]);
    wave_add64(nodes, &out[1]);
    wave_add64(tris, &out[2]);
}

__global__ void __launch_bounds__(PBLOCK) k_count_primary(PipeDev pd, unsigned long long *__restrict__ out)
{
    const uint32_t q = blockIdx.x * PBLOCK + threadIdx.x;
    unsigned long long rays = 0, nodes = 0, tris = 0;
    uint32_t px, py;
    if (q < pd.cap && pix_xy(pd, q, px, py)) {
        const RayD r = primary_ray(pd, px, py);
        uint32_t cn, ct;
        (void)trace_canonical(pd.sc, r, RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES, cn, ct);
        rays = 1; nodes = cn; tris = ct;
    }
    wave_add64(rays, &out[0]);
    wave_add64(nodes, &out[1]);
    wave_add64(tris, &out[2]);
}

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

__global__ void k_debug_cube(PipeDev pd, const float *__restrict__ dirs, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 c = sample_cube(pd, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + PBLOCK - 1) / PBLOCK); }

}  // namespace

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
    float env_const[3] = {0.5f, 0.5f, 0.5f};
    uint32_t width = 0, height = 0, format = RT_FORMAT_R32G32B32A32_FLOAT;
    DevBuf accum_own;
    float4 *accum = nullptr;
    rt_per_frame_constants pfc;
    bool have_pfc = false;
    uint32_t max_rad = 1, max_shadow = 2, accum_mode = RT_ACCUM_RUNNING_MEAN;
    // queues (sized for `cap` pixels)
    uint32_t cap = 0, sh0_batches = 0;
    DevBuf hit0, inst0, pix_k, klist, counters, secO, secD, hit1, inst1, slot_j, jlist, sh0O, sh0D, vis0, sh1O, sh1D, vis1;
    DevBuf half_out;
    std::vector<hipEvent_t> ring;      // 8 events per remembered frame
    int ring_frames = 0;               // 0 = timing off
    uint64_t ring_pos = 0;             // frames recorded since enable / reset
    DevBuf totals, work;
    PipeDev last_pd;
    uint32_t last_shadow_slots = 2;
    rt_stats stats;
    uint32_t last_tile[4] = {0, 0, 0, 0};
    bool rendered = false;
};

namespace {

int ensure_queues(rt_pipeline *p, uint32_t cap, uint32_t sh0_batches)
{
    if (cap <= p->cap && sh0_batches <= p->sh0_batches) return RT_OK;
    const size_t c = cap > p->cap ? cap : p->cap;
    const size_t sb = sh0_batches > p->sh0_batches ? sh0_batches : p->sh0_batches;
    RT_TRY(p->hit0.reserve(c * 16)); RT_TRY(p->inst0.reserve(c * 4));
    RT_TRY(p->pix_k.reserve(c * 4)); RT_TRY(p->klist.reserve(c * 4));
    RT_TRY(p->counters.reserve(C_COUNT * 4));
    RT_TRY(p->secO.reserve(2 * c * 16)); RT_TRY(p->secD.reserve(2 * c * 16));
    RT_TRY(p->hit1.reserve(2 * c * 16)); RT_TRY(p->inst1.reserve(2 * c * 4));
    RT_TRY(p->slot_j.reserve(2 * c * 4)); RT_TRY(p->jlist.reserve(2 * c * 4));
    RT_TRY(p->sh0O.reserve(sb * c * 16)); RT_TRY(p->sh0D.reserve(sb * c * 16)); RT_TRY(p->vis0.reserve(sb * c * 4));
    RT_TRY(p->sh1O.reserve(4 * c * 16)); RT_TRY(p->sh1D.reserve(4 * c * 16)); RT_TRY(p->vis1.reserve(4 * c * 4));
    p->cap = (uint32_t)c;
    p->sh0_batches = (uint32_t)sb;
    return RT_OK;
}

template <int STACK, bool TWO_LEVEL>
void launch_frame(rt_pipeline *p, const PipeDev &pd, uint32_t shadow_slots)
{
    hipStream_t st = p->ctx->stream;
    const bool T = p->ring_frames > 0;
    hipEvent_t *ev = T ? &p->ring[(size_t)(p->ring_pos % (uint64_t)p->ring_frames) * 8] : nullptr;
    const uint32_t cap = pd.cap;
    const rt_context *ctx = p->ctx;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    const QueueSrc sec = {pd.secO, pd.secD, &pd.counters[C_N0], cap, 2u, RT_RAY_FLAG_NONE};          // ProgressiveRaytracing.hlsl:53
    const QueueSrc sh0 = {pd.sh0O, pd.sh0D, &pd.counters[C_N0], cap, shadow_slots, any};              // RaytracingCommon.hlsli:94
    const QueueSrc sh1 = {pd.sh1O, pd.sh1D, &pd.counters[C_N1], 2u * cap, 2u, any};
    if (T) (void)hipEventRecord(ev[0], st);
    // primary rays are coherent: one 8x8 tile per wave, scheduled by the hardware dispatcher
    k_primary<STACK, TWO_LEVEL><<<blocks(cap), PBLOCK, 0, st>>>(pd);
    k_compact_primary<<<(cap + CBLOCK - 1) / CBLOCK, CBLOCK, 0, st>>>(pd);
    if (T) (void)hipEventRecord(ev[1], st);
    k_shade0_emit<<<blocks(cap), PBLOCK, 0, st>>>(pd, shadow_slots);
    if (T) (void)hipEventRecord(ev[2], st);
    k_trace_secondary<STACK, TWO_LEVEL><<<rt_persistent_grid(ctx, k_trace_secondary<STACK, TWO_LEVEL>, PBLOCK, (size_t)cap * 2), PBLOCK, 0, st>>>(pd.sc, sec, pd.hit1, pd.inst1, &pd.counters[C_POOL_SECONDARY],
                                                                                      &pd.counters[C_SECONDARY]);
    k_compact_secondary<<<(2 * cap + CBLOCK - 1) / CBLOCK, CBLOCK, 0, st>>>(pd);
    if (T) { (void)hipEventRecord(ev[3], st); (void)hipEventRecord(ev[4], st); }      // (shadow stage 0 is merged into the launch below)
    k_shade1_emit<<<blocks((size_t)cap * 2), PBLOCK, 0, st>>>(pd);
    if (T) (void)hipEventRecord(ev[5], st);
    const ShadowSrc2 shadows = {sh0, sh1};
    k_trace_shadow<STACK, TWO_LEVEL><<<rt_persistent_grid(ctx, k_trace_shadow<STACK, TWO_LEVEL>, PBLOCK, (size_t)cap * (shadow_slots + 4)), PBLOCK, 0, st>>>(
        pd.sc, shadows, pd.vis0, pd.vis1, &pd.counters[C_POOL_SHADOW0], &pd.counters[C_SHADOW]);
    if (T) (void)hipEventRecord(ev[6], st);
    k_resolve<<<blocks(cap), PBLOCK, 0, st>>>(pd);
    if (T) { (void)hipEventRecord(ev[7], st); p->ring_pos++; }
    k_add_totals<<<1, 64, 0, st>>>(pd.counters, p->totals.as<unsigned long long>(), pd.tw * pd.th);
}

template <int STACK>
void launch_frame_any(rt_pipeline *p, const PipeDev &pd, uint32_t shadow_slots)
{
    if (p->scene->two_level) launch_frame<STACK, true>(p, pd, shadow_slots);
    else launch_frame<STACK, false>(p, pd, shadow_slots);
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
    (void)hipStreamSynchronize(p->ctx->stream);
    DevBuf *all[] = {&p->d_mats, &p->d_env, &p->accum_own, &p->aov_own, &p->hit0, &p->inst0, &p->pix_k, &p->klist, &p->counters, &p->secO, &p->secD,
                     &p->hit1, &p->inst1, &p->slot_j, &p->jlist, &p->sh0O, &p->sh0D, &p->vis0, &p->sh1O, &p->sh1D, &p->vis1, &p->half_out,
                     &p->totals, &p->work};
    for (DevBuf *b : all) b->release();
    for (hipEvent_t e : p->ring) if (e) (void)hipEventDestroy(e);
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
    rt_scene_retain(s);
    if (p->scene) rt_scene_destroy(p->scene);
    p->scene = s;
    return RT_OK;
}

int rt_pipeline_add_material(rt_pipeline *p, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    p->mats.push_back(*m);
    p->mats_dirty = true;
    return RT_OK;
}

int rt_pipeline_set_material(rt_pipeline *p, uint32_t index, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    RT_REQUIRE(index < p->mats.size(), "material index out of range");
    p->mats[index] = *m;
    p->mats_dirty = true;
    return RT_OK;
}

int rt_pipeline_set_environment_cube(rt_pipeline *p, const float *faces, uint32_t size)
{
    RT_REQUIRE(p && faces && size > 0, "bad argument");
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
    p->env_size = 0;
    for (int k = 0; k < 3; k++) p->env_const[k] = rgb[k];
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
    HIP_TRY(hipSetDevice(p->ctx->device));
    RT_TRY(p->accum_own.reserve((size_t)width * height * 16));
    if (p->kind == RT_PIPELINE_REALTIME) RT_TRY(p->aov_own.reserve((size_t)width * height * 16));     // kNumOutputResources = 2
    p->accum = p->accum_own.as<float4>();
    p->width = width; p->height = height; p->format = format;
    return rt_pipeline_clear_output(p);
}

int rt_pipeline_bind_output(rt_pipeline *p, void *device_rgba32f, uint32_t width, uint32_t height)
{
    RT_REQUIRE(p && device_rgba32f, "null argument");
    RT_REQUIRE(width > 0 && height > 0, "empty output");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "bind_output: only the progressive pipeline renders into caller memory");
    p->accum = (float4 *)device_rgba32f;
    p->width = width; p->height = height; p->format = RT_FORMAT_R32G32B32A32_FLOAT;
    return RT_OK;
}

int rt_pipeline_build_acceleration_structures(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    if (!p->scene) { rt_set_error("buildAccelerationStructures: no scene set"); return RT_ERR_STATE; }
    if (p->scene->built) return RT_OK;       // built once, shared between pipelines
    return rt_scene_build(p->scene, 2);
}

int rt_pipeline_set_depth_limits(rt_pipeline *p, uint32_t max_radiance_depth, uint32_t max_shadow_depth)
{
    RT_REQUIRE(p, "null pipeline");
    if (max_radiance_depth > 1) {
        rt_set_error("max radiance depth %u: the wavefront DAG is unrolled for depth <= 1", max_radiance_depth);
        return RT_ERR_UNSUPPORTED;
    }
    p->max_rad = max_radiance_depth;
    p->max_shadow = max_shadow_depth;
    return RT_OK;
}

int rt_pipeline_set_accumulation_mode(rt_pipeline *p, uint32_t mode)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(mode == RT_ACCUM_RUNNING_MEAN || mode == RT_ACCUM_SUM, "unknown accumulation mode");
    p->accum_mode = mode;
    return RT_OK;
}

int rt_pipeline_clear_output(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
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

int rt_pipeline_render_tile(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    RT_REQUIRE(p, "null pipeline");
    if (!p->scene || !p->scene->built) { rt_set_error("render: acceleration structures not built"); return RT_ERR_STATE; }
    if (!p->accum) { rt_set_error("render: no output resource"); return RT_ERR_STATE; }
    if (!p->have_pfc) { rt_set_error("render: update() has not been called"); return RT_ERR_STATE; }
    if (p->mats.empty()) { rt_set_error("render: no material"); return RT_ERR_STATE; }
    RT_REQUIRE(width == p->width && height == p->height, "width/height differ from the output resource");
    if (x1 > width) x1 = width;
    if (y1 > height) y1 = height;
    RT_REQUIRE(x0 < x1 && y0 < y1, "empty tile");
    rt_context *ctx = p->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    p->rendered = false;
    // RayGen early-out (ProgressiveRaytracing.hlsl:14-16): nothing is traced or written
    if (p->kind == RT_PIPELINE_PROGRESSIVE && p->pfc.cameraParams.accumCount >= p->pfc.options.maxIterations) {
        memset(&p->stats, 0, sizeof p->stats);
        return RT_OK;
    }
    if (p->mats_dirty) {
        RT_TRY(p->d_mats.reserve(sizeof(rt_material_params) * p->mats.size()));
        HIP_TRY(hipMemcpyAsync(p->d_mats.p, p->mats.data(), sizeof(rt_material_params) * p->mats.size(), hipMemcpyHostToDevice, st));
        p->mats_dirty = false;
    }
    const uint32_t tw = x1 - x0, th = y1 - y0;
    const uint32_t tiles_x = (tw + 7u) / 8u, cap = tiles_x * ((th + 7u) / 8u) * 64u;
    const uint32_t shadow_slots = (p->kind == RT_PIPELINE_PROGRESSIVE && p->pfc.options.showAmbientOcclusionOnly) ? 4u : 2u;
    RT_TRY(ensure_queues(p, cap, shadow_slots));
    if (!p->totals.p) {
        RT_TRY(p->totals.reserve(8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), st));
    }
    PipeDev pd;
    pd.sc = p->scene->dev();
    pd.pfc = p->pfc;
    pd.mats = p->d_mats.as<rt_material_params>();
    pd.nmats = (uint32_t)p->mats.size();
    pd.env = p->d_env.as<float4>();
    pd.env_size = p->env_size;
    for (int k = 0; k < 3; k++) pd.env_const[k] = p->env_const[k];
    pd.width = width; pd.height = height;
    pd.x0 = x0; pd.y0 = y0; pd.tw = tw; pd.th = th; pd.cap = cap; pd.tiles_x = tiles_x;
    pd.max_rad = p->max_rad; pd.max_shadow = p->max_shadow;
    pd.accum_mode = p->accum_mode;
    pd.kind = p->kind;
    pd.accum = p->accum;
    pd.aov_direct = p->accum;                       // realtime: output 0 = direct lighting, output 1 = indirect specular
    pd.aov_indirect = p->aov_own.as<float4>();
    pd.hit0 = p->hit0.as<float4>(); pd.inst0 = p->inst0.as<uint32_t>();
    pd.pix_k = p->pix_k.as<uint32_t>(); pd.klist = p->klist.as<uint32_t>();
    pd.counters = p->counters.as<uint32_t>();
    pd.secO = p->secO.as<float4>(); pd.secD = p->secD.as<float4>();
    pd.hit1 = p->hit1.as<float4>(); pd.inst1 = p->inst1.as<uint32_t>();
    pd.slot_j = p->slot_j.as<uint32_t>(); pd.jlist = p->jlist.as<uint32_t>();
    pd.sh0O = p->sh0O.as<float4>(); pd.sh0D = p->sh0D.as<float4>(); pd.vis0 = p->vis0.as<uint32_t>();
    pd.sh1O = p->sh1O.as<float4>(); pd.sh1D = p->sh1D.as<float4>(); pd.vis1 = p->vis1.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(pd.counters, 0, C_COUNT * 4, st));
    const uint32_t need = p->scene->stack_need;
    // LDS stack rows -> resident 256-thread blocks per CU: 24 -> 6, 31 -> 5, 39 -> 4, 52 -> 3, 78 -> 2, 160 -> 1
    if (need < 24) launch_frame_any<24>(p, pd, shadow_slots);
    else if (need < 31) launch_frame_any<31>(p, pd, shadow_slots);
    else if (need < 39) launch_frame_any<39>(p, pd, shadow_slots);
    else if (need < 52) launch_frame_any<52>(p, pd, shadow_slots);
    else if (need < 78) launch_frame_any<78>(p, pd, shadow_slots);
    else if (need < 160) launch_frame_any<160>(p, pd, shadow_slots);
    else { rt_set_error("traversal stack need %u exceeds 159 entries", need); return RT_ERR_UNSUPPORTED; }
    HIP_TRY(hipGetLastError());
    p->last_pd = pd;
    p->last_shadow_slots = shadow_slots;
    p->last_tile[0] = x0; p->last_tile[1] = y0; p->last_tile[2] = x1; p->last_tile[3] = y1;
    p->rendered = true;
    return RT_OK;
}

int rt_pipeline_render(rt_pipeline *p, uint32_t width, uint32_t height)
{
    return rt_pipeline_render_tile(p, width, height, 0, 0, width, height);
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
    *ptr = id == 0 ? (void *)p->accum : p->aov_own.p;
    return RT_OK;
}

int rt_pipeline_read_output_n(rt_pipeline *p, uint32_t id, void *host, size_t bytes)
{
    RT_REQUIRE(p && host, "null argument");
    RT_REQUIRE(id < (p->kind == RT_PIPELINE_REALTIME ? 2u : 1u), "output index out of range");
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

int rt_pipeline_enable_timing(rt_pipeline *p, int frames)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(frames >= 0 && frames <= 4096, "timing ring holds 0..4096 frames");
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    for (hipEvent_t e : p->ring) if (e) (void)hipEventDestroy(e);
    p->ring.assign((size_t)frames * 8, nullptr);
    for (hipEvent_t &e : p->ring) HIP_TRY(hipEventCreate(&e));
    p->ring_frames = frames;
    p->ring_pos = 0;
    return RT_OK;
}

static int stage_times(rt_pipeline *p, uint64_t frame, float ms[8])
{
    hipEvent_t *ev = &p->ring[(size_t)(frame % (uint64_t)p->ring_frames) * 8];
    for (int k = 0; k < 7; k++) HIP_TRY(hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]));
    HIP_TRY(hipEventElapsedTime(&ms[7], ev[0], ev[7]));
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
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    memset(out, 0, sizeof *out);
    if (!p->rendered) return RT_OK;
    uint32_t c[C_COUNT];
    HIP_TRY(hipMemcpy(c, p->counters.p, sizeof c, hipMemcpyDeviceToHost));
    out->rays_primary = (uint64_t)(p->last_tile[2] - p->last_tile[0]) * (p->last_tile[3] - p->last_tile[1]);
    out->primary_hits = c[C_N0];
    out->secondary_hits = c[C_N1];
    out->rays_secondary = c[C_SECONDARY];
    out->rays_shadow = c[C_SHADOW];
    out->frames = 1;
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
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    memset(out, 0, sizeof *out);
    if (!p->totals.p) return RT_OK;
    unsigned long long t[8];
    HIP_TRY(hipMemcpy(t, p->totals.p, sizeof t, hipMemcpyDeviceToHost));
    out->rays_primary = t[0]; out->rays_secondary = t[1]; out->rays_shadow = t[2];
    out->primary_hits = t[3]; out->secondary_hits = t[4];
    out->frames = t[5];
    if (p->ring_frames > 0) {
        const uint64_t have = p->ring_pos < (uint64_t)p->ring_frames ? p->ring_pos : (uint64_t)p->ring_frames;
        for (uint64_t f = p->ring_pos - have; f < p->ring_pos; f++) {
            float ms[8];
            RT_TRY(stage_times(p, f, ms));
            add_times(out, ms);
        }
        if (have < out->frames) out->frames = have;      // times cover only the remembered frames
    }
    return RT_OK;
}

int rt_pipeline_reset_totals(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (p->totals.p) HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->ring_pos = 0;
    return RT_OK;
}

int rt_pipeline_count_work(rt_pipeline *p, rt_stage_work *out)
{
    RT_REQUIRE(p && out, "null argument");
    if (!p->rendered) { rt_set_error("count_work: nothing rendered yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    RT_TRY(p->work.reserve(RT_STAGE_COUNT * 3 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, RT_STAGE_COUNT * 3 * sizeof(unsigned long long), st));
    unsigned long long *w = p->work.as<unsigned long long>();
    const PipeDev &pd = p->last_pd;
    const uint32_t cap = pd.cap, ss = p->last_shadow_slots;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    k_count_primary<<<blocks(cap), PBLOCK, 0, st>>>(pd, w + 3 * RT_STAGE_PRIMARY);
    k_count_queue<<<blocks(cap) * 2 + 2, PBLOCK, 0, st>>>(pd.sc, pd.secO, pd.secD, &pd.counters[C_N0], cap, 2, RT_RAY_FLAG_NONE,
                                                          w + 3 * RT_STAGE_SECONDARY);
    k_count_queue<<<(blocks(cap) + 1) * ss, PBLOCK, 0, st>>>(pd.sc, pd.sh0O, pd.sh0D, &pd.counters[C_N0], cap, ss, any,
                                                             w + 3 * RT_STAGE_SHADOW0);
    k_count_queue<<<(blocks((size_t)cap * 2) + 1) * 2, PBLOCK, 0, st>>>(pd.sc, pd.sh1O, pd.sh1D, &pd.counters[C_N1], 2 * cap, 2, any,
                                                                        w + 3 * RT_STAGE_SHADOW1);
    HIP_TRY(hipGetLastError());
    unsigned long long h[RT_STAGE_COUNT * 3];
    HIP_TRY(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < RT_STAGE_COUNT; k++) { out[k].rays = h[3 * k]; out[k].nodes = h[3 * k + 1]; out[k].tris = h[3 * k + 2]; }
    return RT_OK;
}

int rt_pipeline_read_primary_hits(rt_pipeline *p, float *t, uint32_t *prim, uint32_t *inst)
{
    RT_REQUIRE(p, "null pipeline");
    if (!p->rendered) { rt_set_error("nothing rendered yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    const PipeDev &pd = p->last_pd;
    const size_t cap = pd.cap;
    std::vector<float4> h(cap);
    std::vector<uint32_t> hi(cap);
    HIP_TRY(hipMemcpy(h.data(), p->hit0.p, cap * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hi.data(), p->inst0.p, cap * 4, hipMemcpyDeviceToHost));
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

int rt_debug_sample_cube(rt_context *ctx, const float *faces, uint32_t size, const float *dirs, float *out, size_t n)
{
    RT_REQUIRE(ctx && faces && dirs && out && size > 0, "bad argument");
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
    k_debug_cube<<<blocks(n), PBLOCK, 0, ctx->stream>>>(pd, sb[1].as<float>(), sb[2].as<float>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, sb[2].p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
