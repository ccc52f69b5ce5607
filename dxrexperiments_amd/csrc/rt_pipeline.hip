// rt_pipeline.hip -- the ProgressiveRaytracingPipeline on gfx950.
//
// The reference renders a frame with ONE DispatchRays whose raygen shader recurses
// through TraceRay (src/ProgressiveRaytracingPipeline.cpp:215-247 ->
// assets/shaders/ProgressiveRaytracing.hlsl).  Here the same per-pixel recursion
// (reference limits: depth <= 1 radiance, <= 2 shadow, RaytracingCommon.hlsli:11-12;
// this engine: radiance depth <= MAXD = 4) becomes a wavefront DAG of ray queues in
// HBM, one kernel per stage:
//
//   primary        raygen + closest-hit traversal (cull back faces) -> level-0 hits, compaction
//   shade 0 / emit PrimaryClosestHit -> shade(): emits 2 (AO: 4) shadow rays + the diffuse and
//                  the specular ray of level 1
//   for l = 1..max radiance depth:
//     trace l      closest-hit over the ray queue of level l, compaction of its hits
//     shade l/emit closest-hit shading of those hits: 2 shadow rays each + the specular ray of l+1
//   shadow         any-hit over the shadow queues of ALL levels in one persistent launch
//   resolve        re-runs shade() with every TraceRay replaced by its stored result, then
//                  gOutput = (n*prev + cur)/(n+1)                      (ProgressiveRaytracing.hlsl:36-38)
//
// shade() is ONE template used in both the emit and the resolve stage, so the
// arithmetic (and the RNG draw order) of both passes is identical by construction.
// Queues are SoA float4 arrays (origin|tmin, direction|tmax) so a wave reads 1 KiB
// per instruction; hits are compacted with __ballot + popcount prefix sums and one
// atomic per 1024-thread block; diffuse and specular secondaries sit in separate
// batches so waves stay as coherent as the sampling allows.
#include <hip/hip_fp16.h>

#include <new>

#include "rt_trace_wave.h"

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size);

using namespace rtd;

namespace {

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
//   slots of level 1: w*cap + k  (w = 0 diffuse / 1 specular batch, k = compact index of the primary hit)
//   slots of level L >= 2: j     (compact index of the level L-1 hit that spawned the ray)
//   shadow queue of level L: s*hcap(L) + idx, idx = compact hit index, hcap(0) = cap, hcap(L>=1) = 2 cap
struct LevelDev {
    float4 *O, *D;              // ray queue (unused at level 0)
    float4 *hit; uint32_t *inst;   // hit records, indexed by slot (level 0: by pixel slot q)
    uint32_t *slot_j, *jlist;   // slot -> compact hit index (RT_NO_HIT if none), compact index -> slot
    uint32_t *pix;              // slot -> pixel slot q (unused at level 0)
    float4 *shO, *shD; uint32_t *vis;
    float4 *color;              // deep paths (more than one radiance level): the shaded colour of every hit of this level, by slot
};

// the frame's two light rays as the shadow-queue loader rebuilds them (QueueSrc::load)
struct LightRays {
    uint32_t on;
    float dir_to_light[3];      // normalize(-directionalLight.forwardDir), computed once per frame on the host with the
                                //   device's expression (IEEE sqrt and division, left-to-right sums, no contraction)
    float point_pos[3];         // pointLight.worldPos
};

#define RT_MAX_BATCH 16u                // frames one set of launches renders (rt_pipeline_render_batch)

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
    uint32_t skip_unlit;                // do not traverse shadow rays of lights with N.L == 0 (their visibility is multiplied by 0)
    uint32_t shadow_compact;            // shadow queues hold ONE float4 per shaded hit (QueueSrc, "light rays")
    uint32_t kind;                      // RT_PIPELINE_PROGRESSIVE / RT_PIPELINE_REALTIME
    float4 *accum;
    float4 *aov_direct, *aov_indirect;  // realtime pipeline outputs (RealtimeRaytracing.hlsl:3-4)
    uint32_t *counters;
    unsigned long long *totals;         // running sums over frames (rt_pipeline_get_totals); updated by the frame's last kernel
    uint32_t *pools;            // chunk counters of the persistent launches: [1 + MAXD][RT_POOL_GROUPS], 128 B apart
    LevelDev lv[MAXD + 1];
};

constexpr size_t POOL_BYTES = (size_t)(1 + MAXD) * RT_POOL_GROUPS * RT_POOL_STRIDE * 4;      // shadow launch, levels 1..MAXD
constexpr size_t POOL_OFFSET_WORDS = 64;      // the pools start on a 256-B boundary after the scalar counters

RT_DEV uint32_t hcap(const PipeDev &pd, int L) { return L == 0 ? pd.cap : 2u * pd.cap; }

// ---- environment: TextureCube.SampleLevel(linear, dir, 0), RaytracingCommon.hlsli:149-159
// The sampler is MIN_MAG_LINEAR (ProgressiveRaytracingPipeline.cpp:48-55).  On D3D10+ hardware cube maps are
// always filtered seamlessly: a bilinear tap that falls off the selected face comes from the face across that
// edge.  RT_CUBE_SEAMLESS (default) models that with the cube's face-adjacency table; a tap off a CORNER has no
// texel (three faces meet there) and takes the mean of the footprint's other three, the D3D11 functional
// spec's suggestion.  RT_CUBE_FACE_CLAMP clamps taps to the selected face (round 1's behaviour).
//
// Face f = +X -X +Y -Y +Z -Z, edge e = x<0, x>=N, y<0, y>=N -> the face across the edge and where the texel at
// position k along the edge lands there: bit 0 set: x' is the fixed coordinate (else y'), bit 1: fixed = N-1
// (else 0), bit 2: the running coordinate is N-1-k (else k).  Derived from the D3D face parameterisation above.
__constant__ const unsigned char kCubeEdge[24] = {
    (4 << 3) | 3, (5 << 3) | 1, (2 << 3) | 7, (3 << 3) | 3,      // +X
    (5 << 3) | 3, (4 << 3) | 1, (2 << 3) | 1, (3 << 3) | 5,      // -X
    (1 << 3) | 0, (0 << 3) | 4, (5 << 3) | 4, (4 << 3) | 0,      // +Y
    (1 << 3) | 6, (0 << 3) | 2, (4 << 3) | 2, (5 << 3) | 6,      // -Y
    (1 << 3) | 3, (0 << 3) | 1, (2 << 3) | 2, (3 << 3) | 0,      // +Z
    (0 << 3) | 3, (1 << 3) | 1, (2 << 3) | 4, (3 << 3) | 6};     // -Z

// texel (x, y) of `face`, x and y in [-1, N]; false: the tap hangs over a cube corner
RT_DEV bool cube_tap(const PipeDev &pd, int face, int x, int y, float4 &out)
{
    const int m = (int)pd.env_size - 1;
    const bool ox = x < 0 || x > m, oy = y < 0 || y > m;
    if (pd.env_filter == RT_CUBE_FACE_CLAMP) {
        x = min(max(x, 0), m); y = min(max(y, 0), m);
    } else if (ox && oy) {
        return false;
    } else if (ox || oy) {
        const int e = ox ? (x < 0 ? 0 : 1) : (y < 0 ? 2 : 3);
        const int k = ox ? y : x;
        const unsigned code = kCubeEdge[face * 4 + e];
        const int fixed = (code & 2u) ? m : 0, run = (code & 4u) ? m - k : k;
        face = (int)(code >> 3);
        x = (code & 1u) ? fixed : run;
        y = (code & 1u) ? run : fixed;
    }
    out = pd.env[((size_t)face * pd.env_size + (size_t)y) * pd.env_size + (size_t)x];
    return true;
}

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
    const int x0 = (int)x0f, y0 = (int)y0f;
    float4 c[4];
    bool have[4];
    have[0] = cube_tap(pd, face, x0, y0, c[0]);
    have[1] = cube_tap(pd, face, x0 + 1, y0, c[1]);
    have[2] = cube_tap(pd, face, x0, y0 + 1, c[2]);
    have[3] = cube_tap(pd, face, x0 + 1, y0 + 1, c[3]);
    for (int k = 0; k < 4; k++) {
        if (have[k]) continue;                    // at most one tap of a footprint hangs over a corner
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (int j = 0; j < 4; j++)
            if (j != k) { sx = sx + c[j].x; sy = sy + c[j].y; sz = sz + c[j].z; }
        c[k] = make_float4(sx / 3.0f, sy / 3.0f, sz / 3.0f, 1.0f);
    }
    const float4 c00 = c[0], c10 = c[1], c01 = c[2], c11 = c[3];
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
    if (pd.band_rows) {         // rows of the rectangle = this rank's interleaved bands, packed (rt_pipeline_render_bands)
        py = ((ly / pd.band_rows) * pd.band_world + pd.band_rank) * pd.band_rows + ly % pd.band_rows;
        return lx < pd.tw && ly < pd.th && py < pd.height;
    }
    py = pd.y0 + ly;
    return lx < pd.tw && ly < pd.th;
}

// ---- RayGen (ProgressiveRaytracing.hlsl:18-32)
RT_DEV RayD primary_ray(const PipeDev &pd, const rt_camera_params &cp, uint32_t px, uint32_t py)
{
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
RT_DEV RayD primary_ray(const PipeDev &pd, uint32_t px, uint32_t py) { return primary_ray(pd, pd.pfc.cameraParams, px, py); }

// The frame a pixel slot belongs to, and the slot inside that frame (single frames: 0 and q itself)
RT_DEV uint32_t slot_frame(const PipeDev &pd, uint32_t q, uint32_t &q_in_frame)
{
    if (pd.n_frames <= 1u) { q_in_frame = q; return 0u; }
    const uint32_t f = q / pd.fcap;
    q_in_frame = q - f * pd.fcap;
    return f;
}

// ---- interpolateVertexAttributes (RaytracingCommon.hlsli:53-82), normal only
RT_DEV f3 hit_normal(const InstanceRec &in, uint32_t prim, float bu, float bv)
{
    const float b0 = 1.0f - bu - bv;
    // verts[indices[3 prim + k]].normal, gathered per primitive at build time (InstanceRec::normals): one 48-B record
    const TriRec nr = in.normals[prim];
    f3 n = mk3(nr.a.x, nr.a.y, nr.a.z) * b0;
    n = n + mk3(nr.a.w, nr.b.x, nr.b.y) * bu;
    n = n + mk3(nr.b.z, nr.b.w, nr.c.x) * bv;
    return n;
}

// ---- lights (RaytracingCommon.hlsli:126-147), AO (:98-124)
template <class IO>
RT_DEV f3 directional_light(const PipeDev &pd, IO &io, f3 P, f3 N, uint32_t depth)
{
    const rt_directional_light_params &dl = pd.pfc.directionalLight;
    const f3 L = normalize(mk3(-dl.forwardDir.x, -dl.forwardDir.y, -dl.forwardDir.z));
    const float NoL = saturate(dot(N, L));
    // the reference traces this ray even when NoL == 0 (RaytracingCommon.hlsli:132-133); its visibility is then multiplied
    // by 0, so the ray is emitted and counted but need not be traversed (io.shadow's last argument)
    const float vis = io.shadow(0, P, L, RAY_EPSILON, RAY_MAX_T, depth, NoL > 0.0f);
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
    const float vis = io.shadow(1, P, L, RAY_EPSILON, dist - RAY_EPSILON, depth, NoL > 0.0f);
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
        visibility += io.shadow(i, P, dir, RAY_EPSILON, 10.0f, 1u, true) * NoL / pdf;
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

// emit pass at radiance depth L: shadow ray s -> shadow queue of level L, secondary ray -> ray queue of level L+1
struct EmitIO {
    const PipeDev &pd;
    int L;
    uint32_t idx, q;            // compact hit index at level L, pixel slot
    uint32_t frame;             // frame of the batch the hit belongs to (single frames: 0)
    uint32_t shadow_mask, skip_mask, sec_mask;
    f3 shadow_origin;           // compact shadow queues: the hit point both light rays start from
    RT_DEV EmitIO(const PipeDev &p, int level, uint32_t i, uint32_t qq, uint32_t f) : pd(p), L(level), idx(i), q(qq), frame(f), shadow_mask(0), skip_mask(0), sec_mask(0)
    {
        shadow_origin = mk3(0.0f, 0.0f, 0.0f);
    }
    // matters = false: whatever this ray finds is multiplied by zero by the caller
    RT_DEV float shadow(int s, f3 o, f3 d, float tmin, float tmax, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        const bool skipped = !matters && pd.skip_unlit;     // "emitted but not worth traversing"; the trace kernel counts these
        shadow_mask |= 1u << s;
        if (skipped) skip_mask |= 1u << s;
        if (pd.shadow_compact) {                            // the ray is rebuilt from the hit point by QueueSrc::load
            shadow_origin = o;
            return 1.0f;
        }
        if (skipped) store_ray(pd.lv[L].shO, pd.lv[L].shD, (size_t)s * hcap(pd, L) + idx, mk3(0.0f, 0.0f, 0.0f), 0.0f, mk3(0.0f, 0.0f, 0.0f), RT_TMAX_SKIPPED);
        else store_ray(pd.lv[L].shO, pd.lv[L].shD, (size_t)s * hcap(pd, L) + idx, o, tmin, d, tmax);
        return 1.0f;
    }
    // the shadow slots of this hit that no ray went to are marked "not traced"
    RT_DEV void finish_shadows(uint32_t shadow_slots) const
    {
        if (pd.shadow_compact) {
            pd.lv[L].shO[idx] = make_float4(shadow_origin.x, shadow_origin.y, shadow_origin.z, __uint_as_float(shadow_mask | (skip_mask << 2) | (frame << 8)));
            return;
        }
        for (uint32_t s = 0; s < shadow_slots; s++)
            if (!(shadow_mask & (1u << s))) store_invalid(pd.lv[L].shO, pd.lv[L].shD, (size_t)s * hcap(pd, L) + idx);
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if (depth >= pd.max_rad || L >= MAXD) return mk3(0.0f, 0.0f, 0.0f);
        const size_t slot = L == 0 ? (size_t)w * pd.cap + idx : idx;
        store_ray(pd.lv[L + 1].O, pd.lv[L + 1].D, slot, o, tmin, d, RAY_MAX_T);
        pd.lv[L + 1].pix[slot] = q;
        sec_mask |= 1u << w;
        return mk3(0.0f, 0.0f, 0.0f);
    }
};

// resolve pass at radiance depth L: every TraceRay is replaced by its stored result; a secondary hit
// recurses into the next level (compile-time recursion, MAXD deep)
template <int L, int MAXL>
struct ResolveIO {
    const PipeDev &pd;
    uint32_t idx, pix;
    RT_DEV ResolveIO(const PipeDev &p, uint32_t i, uint32_t px) : pd(p), idx(i), pix(px) {}
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        if (!matters && pd.skip_unlit) return 1.0f;                       // never traced; the caller multiplies by zero
        return pd.lv[L].vis[(size_t)s * hcap(pd, L) + idx] ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if constexpr (L >= MAXL) {
            return mk3(0.0f, 0.0f, 0.0f);
        } else {
            if (depth >= pd.max_rad) return mk3(0.0f, 0.0f, 0.0f);
            const size_t slot = L == 0 ? (size_t)w * pd.cap + idx : idx;
            const float4 h = pd.lv[L + 1].hit[slot];
            if (h.x == HIT_MISS) return sample_environment(pd, d);             // PrimaryMiss, :160-164
            if (h.x == HIT_UNTRACED) return mk3(0.0f, 0.0f, 0.0f);
            RayD r;
            r.o = o; r.tmin = tmin; r.d = d; r.tmax = RAY_MAX_T;
            ResolveIO<L + 1, MAXL> io(pd, pd.lv[L + 1].slot_j[slot], pix);
            return closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[L + 1].inst[slot], depth + 1u, pix);
        }
    }
};

// The same pass for paths of more than one bounce, level by level from the deepest up (k_shade_level, then k_resolve_flat):
// the colour of a secondary hit is not recomputed by compile-time recursion -- five nested shade() bodies cost 142 VGPRs,
// three waves per SIMD -- but read from the colour buffer the pass of the level below has just written.  It is the value
// the recursion would have produced (same function, same inputs), so the image does not change by a bit.
struct LevelResolveIO {
    const PipeDev &pd;
    int L;
    uint32_t idx;
    RT_DEV LevelResolveIO(const PipeDev &p, int level, uint32_t i) : pd(p), L(level), idx(i) {}
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        if (!matters && pd.skip_unlit) return 1.0f;
        return pd.lv[L].vis[(size_t)s * hcap(pd, L) + idx] ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int w, f3, f3 d, float, uint32_t depth)
    {
        if (depth >= pd.max_rad || L >= MAXD) return mk3(0.0f, 0.0f, 0.0f);
        const size_t slot = L == 0 ? (size_t)w * pd.cap + idx : idx;
        const float4 h = pd.lv[L + 1].hit[slot];
        if (h.x == HIT_MISS) return sample_environment(pd, d);             // PrimaryMiss, :160-164
        if (h.x == HIT_UNTRACED) return mk3(0.0f, 0.0f, 0.0f);
        const float4 c = pd.lv[L + 1].color[slot];
        return mk3(c.x, c.y, c.z);
    }
};

// ---- compaction ----------------------------------------------------------------------
constexpr int CBLOCK = 1024;
// every lane of the wave must call this (no early exits before it)
RT_DEV void wave_add(uint32_t v, uint32_t *counter)
{
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(counter, v);
}

// ---- kernels -------------------------------------------------------------------------

// primary stage: raygen is the ray source, the level-0 hit records the sink (indexed by pixel slot)
// BATCH = false: one frame, the code of round 2; true: the slot's frame picks the camera (also right for one frame)
template <bool BATCH>
struct PrimarySrcT {
    const PipeDev &pd;
    RT_DEV uint32_t count() const { return pd.cap; }
    RT_DEV uint32_t flags() const { return RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES; }      // ProgressiveRaytracing.hlsl:34
    RT_DEV bool load(uint32_t q, RayD &r) const
    {
        uint32_t px, py;
        if (BATCH && pd.n_frames > 1u) {         // (a kernel argument: the branch is uniform)
            // 64 consecutive slots = one tile = one chunk of a wave: the frame is the same for every lane that loads here
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q / pd.fcap));
            const bool valid = pix_xy(pd, q - f * pd.fcap, px, py);
            r = primary_ray(pd, pd.pfcs[f].cameraParams, px, py);
            return valid;
        }
        const bool valid = pix_xy(pd, q, px, py);
        r = primary_ray(pd, px, py);
        return valid;
    }
};
typedef PrimarySrcT<true> PrimarySrc;        // (the counting kernels)
struct PrimarySink {
    const PipeDev &pd;
    RT_DEV void store(uint32_t q, const HitD &h, bool) const
    {
        const bool hit = h.inst != RT_NO_HIT;
        pd.lv[0].hit[q] = make_float4(hit ? h.t : HIT_MISS, h.u, h.v, __uint_as_float(h.prim));
        pd.lv[0].inst[q] = h.inst;
    }
};

template <int STACK, bool TWO_LEVEL, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) k_primary(PipeDev pd)
{
    __shared__ int smem[(STACK + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    // the frame's counters and chunk pools start at zero: its first kernel clears them (nothing here uses them, every later
    // kernel of the frame does) instead of a 20-KB fill launch of its own
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)(POOL_OFFSET_WORDS + POOL_BYTES / 4); i += PBLOCK) pd.counters[i] = 0u;
    PrimarySrcT<BATCH> src = {pd};
    PrimarySink sink = {pd};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, 64u>(pd.sc, src, sink, nullptr, smem, nullptr);   // one 8x8 tile per wave, dealt by the hardware dispatcher
}

// Compaction of the hits of level L (they get shaded).  Level 0 runs over the pixel slots, level 1 over its two batches,
// deeper levels over one batch.  A block owns CTILES consecutive tiles of CBLOCK slots: it first counts its hits (each
// thread remembers its CTILES flags in a bit mask), reserves its output range with ONE atomic, then writes tile after
// tile in slot order -- same-address returning atomics serialise at ~11 ns, and one per 1024 slots (2,000 - 4,000 per
// launch at 1080p) was most of this kernel's 36 us.
constexpr int CTILES = 8;            // (tiles per workgroup 4 / 8 / 16: 41 / 40 / 45 us for the frame's two compactions)
__global__ void __launch_bounds__(CBLOCK) k_compact_level(PipeDev pd, int L)
{
    __shared__ uint32_t wave_total[CBLOCK / 64];
    __shared__ uint32_t block_base;
    uint32_t total_items, n = 0;
    if (L == 0) total_items = pd.cap;
    else { n = pd.counters[C_NHIT + L - 1]; total_items = (L == 1 ? 2u : 1u) * n; }
    const uint32_t first = blockIdx.x * (uint32_t)(CTILES * CBLOCK);
    if (first >= total_items) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    auto slot_of = [&](uint32_t idx) -> size_t { return L == 1 ? (size_t)(idx / n) * pd.cap + idx % n : (size_t)idx; };
    uint32_t flags = 0, mine = 0;
#pragma unroll 4
    for (int t = 0; t < CTILES; t++) {
        const uint32_t idx = first + (uint32_t)t * CBLOCK + threadIdx.x;
        const bool hit = idx < total_items && pd.lv[L].hit[slot_of(idx)].x >= 0.0f;
        flags |= (hit ? 1u : 0u) << t;
        mine += hit ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
    if (lane == 0) wave_total[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < CBLOCK / 64; w++) sum += wave_total[w];
        block_base = sum ? atomicAdd(&pd.counters[C_NHIT + L], sum) : 0u;
    }
    __syncthreads();
    uint32_t running = block_base;
    for (int t = 0; t < CTILES; t++) {
        const uint32_t idx = first + (uint32_t)t * CBLOCK + threadIdx.x;
        if (first + (uint32_t)t * CBLOCK >= total_items) break;          // (block-uniform)
        const bool hit = (flags >> t) & 1u;
        const unsigned long long mask = __ballot(hit);
        __syncthreads();                                                 // wave_total is reused tile after tile
        if (lane == 0) wave_total[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, tile_total = 0;
        for (int w = 0; w < CBLOCK / 64; w++) { const uint32_t c = wave_total[w]; before += w < (int)wave ? c : 0u; tile_total += c; }
        const uint32_t j = running + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (idx < total_items) {
            const size_t slot = slot_of(idx);
            pd.lv[L].slot_j[slot] = hit ? j : RT_NO_HIT;
            if (hit) pd.lv[L].jlist[j] = (uint32_t)slot;
        }
        running += tile_total;
    }
}

// closest-hit shading of the compacted hits of level L in emit mode: writes their shadow rays and the rays
// of level L+1; slots a hit does not use are marked "not traced"
// BATCH: the launch covers several frames; a hit takes the constants of the frame its pixel slot lies in
// (LC >= 0: the level as a compile-time constant.  The batch kernels work on a per-thread copy of the arguments whose pfc
// they replace; with a run-time level the copy's lv[] would be indexed dynamically and live in scratch memory.)
template <bool PRIMARY, bool BATCH, int LC>
RT_DEV void shade_emit_body(const PipeDev &pd_arg, int level, uint32_t shadow_slots, uint32_t emit_next)
{
    PipeDev pd = pd_arg;
    const int L = PRIMARY ? 0 : (LC >= 0 ? LC : level);          // (depth 0 compiles to its own kernel: it alone samples indirect diffuse)
    __builtin_assume(PRIMARY || L >= 1);
    const uint32_t idx = blockIdx.x * PBLOCK + threadIdx.x;
    if (idx >= pd.counters[C_NHIT + L]) return;
    const uint32_t slot = pd.lv[L].jlist[idx];
    const uint32_t q = L == 0 ? slot : pd.lv[L].pix[slot];
    uint32_t px, py, ql = q, frame = 0;
    if (BATCH) { frame = slot_frame(pd, q, ql); pd.pfc = pd.pfcs[frame]; }
    (void)pix_xy(pd, ql, px, py);
    const RayD r = L == 0 ? primary_ray(pd, px, py) : load_ray(pd.lv[L].O, pd.lv[L].D, slot);
    const float4 h = pd.lv[L].hit[slot];
    EmitIO io(pd, L, idx, q, frame);
    (void)closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[L].inst[slot], (uint32_t)L, px + py * pd.width);
    io.finish_shadows(shadow_slots);
    if (emit_next) {
        if (L == 0) {
            for (uint32_t w = 0; w < 2; w++)
                if (!(io.sec_mask & (1u << w))) store_invalid(pd.lv[1].O, pd.lv[1].D, (size_t)w * pd.cap + idx);
        } else if (L < MAXD && !io.sec_mask) store_invalid(pd.lv[L < MAXD ? L + 1 : MAXD].O, pd.lv[L < MAXD ? L + 1 : MAXD].D, idx);
    }
}
template <bool PRIMARY, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) k_shade_emit(PipeDev pd, int level, uint32_t shadow_slots, uint32_t emit_next)
{
    if (PRIMARY || !BATCH) shade_emit_body<PRIMARY, BATCH, -1>(pd, level, shadow_slots, emit_next);
    else if (level == 1) shade_emit_body<PRIMARY, BATCH, 1>(pd, level, shadow_slots, emit_next);
    else if (level == 2) shade_emit_body<PRIMARY, BATCH, 2>(pd, level, shadow_slots, emit_next);
    else if (level == 3) shade_emit_body<PRIMARY, BATCH, 3>(pd, level, shadow_slots, emit_next);
    else shade_emit_body<PRIMARY, BATCH, MAXD>(pd, level, shadow_slots, emit_next);
}

// a ray queue of `batches` batches of *count rays; batch b lives at [b*stride, b*stride + *count).
// Shadow queues are normally COMPACT ("light rays"): both shadow rays of a shaded hit start at the hit point and go to the
// frame's two lights, so the emit pass stores one float4 per hit -- the point and, in the bits of w, which of the two rays
// exist (bits 0-1) and which need not be traversed (bits 2-3) -- and this loader rebuilds ray b of hit k with the very
// expressions of evaluateDirectionalLight / evaluatePointLight (RaytracingCommon.hlsli:126-147; directional_light /
// point_light above): 16 B written and read per hit instead of 128 B.  The four rays of the ambient-occlusion view have
// random directions and keep the explicit origin / direction form.
struct QueueSrc {
    const float4 *O, *D;
    const uint32_t *count_ptr;
    uint32_t stride, batches, fl;
    LightRays lights;
    RT_DEV uint32_t n() const { return *count_ptr; }
    RT_DEV uint32_t count() const { return n() * batches; }
    RT_DEV uint32_t flags() const { return fl; }
    RT_DEV size_t slot(uint32_t i) const { const uint32_t c = n(); return (size_t)(i / c) * stride + i % c; }
    RT_DEV bool load(uint32_t i, RayD &r) const { return load_lit(i, r, nullptr); }
    // per_frame: a batch of frames -- the lights of frame f (bits 8.. of the hit's word).  The single-frame kernels call this
    // with a literal nullptr: the branch folds away and their code is what it was before batches existed (with the branch
    // compiled in, the five inlined copies of this loader cost the any-hit kernel 30 VGPRs and 128 B of scratch).
    RT_DEV bool load_lit(uint32_t i, RayD &r, const LightRays *per_frame) const
    {
        if (lights.on) {
            const uint32_t c = n(), b = i / c;
            const v4f a = ldg16(O, (size_t)(i % c) * 16);
            const uint32_t bits = __float_as_uint(a.w);
            r.o = mk3(a.x, a.y, a.z);
            r.d = mk3(0.0f, 0.0f, 0.0f);
            r.tmin = 0.0f;
            r.tmax = -1.0f;                                  // no such ray: never traced
            if (!((bits >> b) & 1u)) return false;
            if ((bits >> (2u + b)) & 1u) { r.tmax = RT_TMAX_SKIPPED; return false; }
            r.tmin = RAY_EPSILON;
            if (per_frame) {                                 // (three floats by hand: a struct copy ends up in scratch memory)
                const float *fl = b == 0u ? per_frame[(bits >> 8) & 0xffu].dir_to_light : per_frame[(bits >> 8) & 0xffu].point_pos;
                const f3 l = mk3(fl[0], fl[1], fl[2]);
                if (b == 0u) { r.d = l; r.tmax = RAY_MAX_T; }
                else {
                    const f3 path = l - r.o;
                    const float dist = length(path);
                    r.d = normalize(path);
                    r.tmax = dist - RAY_EPSILON;
                }
                return r.tmax > r.tmin;
            }
            if (b == 0u) {
                r.d = mk3(lights.dir_to_light[0], lights.dir_to_light[1], lights.dir_to_light[2]);
                r.tmax = RAY_MAX_T;
            } else {
                const f3 path = mk3(lights.point_pos[0], lights.point_pos[1], lights.point_pos[2]) - r.o;
                const float dist = length(path);
                r.d = normalize(path);
                r.tmax = dist - RAY_EPSILON;
            }
            return r.tmax > r.tmin;
        }
        const size_t sl = slot(i);
        const v4f a = ldg16(O, sl * 16), b = ldg16(D, sl * 16);
        r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
        r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
        return r.tmax > r.tmin;
    }
};
static inline LightRays light_rays(uint32_t shadow_compact, const rt_per_frame_constants &pfc)
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
    return l;
}
static inline LightRays light_rays(const PipeDev &pd) { return light_rays(pd.shadow_compact, pd.pfc); }
static inline LightRays no_light_rays()
{
    LightRays l;
    memset(&l, 0, sizeof l);
    return l;
}

// a queue read with the batch's per-frame lights (nullptr: a single frame); the counting kernels are not register critical
struct LitQueueSrc {
    QueueSrc q;
    const LightRays *per_frame;
    RT_DEV uint32_t n() const { return q.n(); }
    RT_DEV uint32_t count() const { return q.count(); }
    RT_DEV uint32_t flags() const { return q.flags(); }
    RT_DEV bool load(uint32_t i, RayD &r) const { return q.load_lit(i, r, per_frame); }
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

// every shadow ray of the frame in ONE launch: the shadow queues of all shaded levels, back to back
// BATCH: the launch covers several frames, a ray takes the light rays of its hit's frame
struct ShadowQueues {
    QueueSrc q[MAXD + 1];
    uint32_t *vis[MAXD + 1];
    int nq;
    const LightRays *frame_lights;      // device array [n_frames] (batches)
};
template <bool BATCH>
struct ShadowSrcN : ShadowQueues {
    // (the loops are fully unrolled so that q[k] is always a compile-time member of the kernel argument)
    RT_DEV uint32_t count() const
    {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k <= MAXD; k++) if (k < nq) c += q[k].count();
        return c;
    }
    RT_DEV uint32_t flags() const { return q[0].fl; }
    RT_DEV bool load(uint32_t i, RayD &r) const
    {
#pragma unroll
        for (int k = 0; k <= MAXD; k++) {
            if (k < nq) {
                const uint32_t c = q[k].count();
                if (i < c) return q[k].load_lit(i, r, BATCH ? frame_lights : nullptr);
                i -= c;
            }
        }
        r.o = mk3(0.0f, 0.0f, 0.0f); r.d = r.o; r.tmin = 0.0f; r.tmax = -1.0f;
        return false;
    }
};
struct ShadowSinkN {      // ShadowMiss sets visibility 1 (ProgressiveRaytracing.hlsl:178-182)
    ShadowQueues s;
    RT_DEV void store(uint32_t i, const HitD &h, bool) const
    {
#pragma unroll
        for (int k = 0; k <= MAXD; k++) {
            if (k < s.nq) {
                const uint32_t c = s.q[k].count();
                if (i < c) { s.vis[k][s.q[k].slot(i)] = h.inst == RT_NO_HIT ? 1u : 0u; return; }
                i -= c;
            }
        }
    }
};

template <int STACK, bool TWO_LEVEL, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) k_trace_shadow(SceneDev sc, ShadowQueues queues, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[(STACK + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    ShadowSrcN<BATCH> src;
    static_cast<ShadowQueues &>(src) = queues;
    ShadowSinkN sink = {queues};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, RT_SHADOW_UNORDERED != 0>(sc, src, sink, pool, smem, stat);
}

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_trace_secondary(SceneDev sc, QueueSrc src, float4 *hit1, uint32_t *inst1, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[(STACK + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    SecondarySink sink = {src, hit1, inst1};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK>(sc, src, sink, pool, smem, stat);
}

// ---- walk counting (rt_pipeline_count_walk): the production walk over the last frame's queues with per-lane
// tallies of what it fetches; results are not stored (the frame already holds them)
struct NullSink { RT_DEV void store(uint32_t, const HitD &, bool) const {} };

template <bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_walk_primary(PipeDev pd, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    PrimarySrc src = {pd};
    NullSink sink;
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, 64u, false, true>(pd.sc, src, sink, nullptr, smem, nullptr, walk);
}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_walk_queue(SceneDev sc, QueueSrc src, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    NullSink sink;
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, false, true>(sc, src, sink, nullptr, smem, nullptr, walk);
}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_walk_shadow(SceneDev sc, LitQueueSrc src, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    NullSink sink;
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, RT_SHADOW_UNORDERED != 0, true>(sc, src, sink, nullptr, smem, nullptr, walk);
}

// the frame's ray / hit counts into the running totals (one thread, once per frame)
RT_DEV void add_totals(const uint32_t *__restrict__ counters, unsigned long long *__restrict__ totals, uint32_t pixels, uint32_t frames)
{
    totals[0] += (unsigned long long)pixels * frames;
    totals[1] += counters[C_SECONDARY];
    totals[2] += counters[C_SHADOW] + counters[C_SHADOW_SKIPPED];
    totals[6] += counters[C_SHADOW_SKIPPED];
    totals[3] += counters[C_NHIT + 0];
    for (int l = 1; l <= MAXD; l++) totals[4] += counters[C_NHIT + l];
    totals[5] += frames;
}

// what the frame's last kernel does with the colour of pixel slot q (RayGen's tail, ProgressiveRaytracing.hlsl:36-38 /
// RealtimeRaytracing.hlsl:44-45)
RT_DEV void write_pixel(const PipeDev &pd, uint32_t px, uint32_t py, const Shaded &sh)
{
    const size_t pixel = (size_t)py * pd.width + px;
    if (pd.kind == RT_PIPELINE_REALTIME) {                  // two AOVs, no accumulation
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

// FLAT = false: one bounce at most, the secondary hits are shaded inline (ResolveIO<0, 1>); FLAT = true: their colours
// come from k_shade_level (LevelResolveIO)
template <bool FLAT, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) k_resolve(PipeDev pd_arg)
{
    PipeDev pd = pd_arg;
    const uint32_t ql = blockIdx.x * PBLOCK + threadIdx.x;
    if (ql == 0) add_totals(pd.counters, pd.totals, pd.n_pixels, pd.n_frames);      // every counter of the frame is final when this kernel starts
    if (ql >= pd.fcap) return;
    uint32_t px, py;
    if (!pix_xy(pd, ql, px, py)) return;
    // a batch: the frames of a pixel one after the other, in frame order, so that the running mean is the one S single
    // frames would have left (each frame with its own accumCount)
    for (uint32_t f = 0; f < (BATCH ? pd.n_frames : 1u); f++) {
        if (BATCH) pd.pfc = pd.pfcs[f];
        const uint32_t q = f * pd.fcap + ql;
        const RayD r = primary_ray(pd, px, py);
        const float4 h = pd.lv[0].hit[q];
        Shaded sh;
        if (h.x == HIT_MISS) {
            sh.color = sample_environment(pd, r.d);             // PrimaryMiss
            sh.aov_direct = sh.color;                           // RealtimeRaytracing.hlsl:119-126
            sh.aov_indirect = mk3(0.0f, 0.0f, 0.0f);
        } else if (FLAT) {
            LevelResolveIO io(pd, 0, pd.lv[0].slot_j[q]);
            sh = closest_hit_aov(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[0].inst[q], 0u, px + py * pd.width);
        } else {
            ResolveIO<0, 1> io(pd, pd.lv[0].slot_j[q], px + py * pd.width);
            sh = closest_hit_aov(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[0].inst[q], 0u, px + py * pd.width);
        }
        write_pixel(pd, px, py, sh);
    }
}

// deep paths: the colour of every hit of level L >= 1 (its shadow rays are traced, the hits of level L + 1 already shaded)
template <bool BATCH, int LC>
RT_DEV void shade_level_body(const PipeDev &pd_arg, int level)
{
    PipeDev pd = pd_arg;
    const int L = LC >= 0 ? LC : level;
    const uint32_t idx = blockIdx.x * PBLOCK + threadIdx.x;
    if (idx >= pd.counters[C_NHIT + L]) return;
    const uint32_t slot = pd.lv[L].jlist[idx];
    const uint32_t q = pd.lv[L].pix[slot];
    uint32_t px, py, ql = q;
    if (BATCH) pd.pfc = pd.pfcs[slot_frame(pd, q, ql)];
    (void)pix_xy(pd, ql, px, py);
    const RayD r = load_ray(pd.lv[L].O, pd.lv[L].D, slot);
    const float4 h = pd.lv[L].hit[slot];
    LevelResolveIO io(pd, L, idx);
    const f3 c = closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[L].inst[slot], (uint32_t)L, px + py * pd.width);
    pd.lv[L].color[slot] = make_float4(c.x, c.y, c.z, 0.0f);
}
template <bool BATCH>
__global__ void __launch_bounds__(PBLOCK) k_shade_level(PipeDev pd, int L)
{
    if (!BATCH) shade_level_body<BATCH, -1>(pd, L);
    else if (L == 1) shade_level_body<BATCH, 1>(pd, L);
    else if (L == 2) shade_level_body<BATCH, 2>(pd, L);
    else if (L == 3) shade_level_body<BATCH, 3>(pd, L);
    else shade_level_body<BATCH, MAXD>(pd, L);
}

RT_DEV void wave_add64(unsigned long long v, unsigned long long *counter)
{
    for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((long long)v, o, 64);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(counter, v);
}

// canonical-order re-trace of a queue: sums rays / nodes / triangles into out[0..2]
__global__ void __launch_bounds__(PBLOCK)
k_count_queue(SceneDev sc, LitQueueSrc src, unsigned long long *__restrict__ out)
{
    const uint32_t n = src.n();
    const uint32_t idx = blockIdx.x * PBLOCK + threadIdx.x;
    const uint32_t per = (n + PBLOCK - 1) / PBLOCK * PBLOCK;
    const uint32_t b = per ? idx / per : src.q.batches, k = per ? idx % per : 0;
    unsigned long long rays = 0, nodes = 0, tris = 0;
    if (b < src.q.batches && k < n) {
        RayD r;
        if (src.load(b * n + k, r)) {
            uint32_t cn, ct;
            (void)trace_canonical(sc, r, src.q.fl, cn, ct);
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
    RayD r;
    const PrimarySrc src = {pd};
    if (q < pd.cap && src.load(q, r)) {
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
    uint32_t env_filter = RT_CUBE_SEAMLESS;
    float env_const[3] = {0.5f, 0.5f, 0.5f};
    uint32_t width = 0, height = 0, format = RT_FORMAT_R32G32B32A32_FLOAT;
    DevBuf accum_own;
    float4 *accum = nullptr;
    rt_per_frame_constants pfc;
    bool have_pfc = false;
    uint32_t max_rad = 1, max_shadow = 2, accum_mode = RT_ACCUM_RUNNING_MEAN;
    uint32_t skip_unlit = 0;           // off by default: every shadow ray the reference traces is traversed (rt_pipeline_set_skip_unlit_shadow_rays)
    // queues (sized for `cap` pixels)
    uint32_t cap = 0, sh0_batches = 0, levels = 0;
    struct LevelBuf { DevBuf O, D, hit, inst, slot_j, jlist, pix, shO, shD, vis, color; } lv[MAXD + 1];
    DevBuf counters;
    DevBuf half_out;
    std::vector<hipEvent_t> ring;      // EV_COUNT events per remembered frame
    std::vector<uint8_t> ring_levels;  // radiance levels each remembered frame ran
    std::vector<uint8_t> ring_nframes; // frames each remembered entry covers (a batch is one entry)
    DevBuf batch_consts;               // per-frame constants and light rays of a batch (rt_pipeline_render_batch)
    int ring_frames = 0;               // 0 = timing off
    uint64_t ring_pos = 0;             // frames recorded since enable / reset
    DevBuf totals, work;
    PipeDev last_pd;
    uint32_t last_shadow_slots = 2;
    rt_stats stats;
    uint32_t last_tile[4] = {0, 0, 0, 0};
    uint32_t last_pixels = 0;
    bool rendered = false;
    uint32_t last_scene_gen = 0;       // generation of the scene last_pd was filled from
};

namespace {

// events of one frame: start | primary | shade 0 | (trace l, shade l) for l = 1..MAXD | shadow | resolve
constexpr int EV_COUNT = 5 + 2 * MAXD;
constexpr int EV_SHADOW = 3 + 2 * MAXD, EV_RESOLVE = 4 + 2 * MAXD;

int ensure_queues(rt_pipeline *p, uint32_t cap, uint32_t sh0_batches, uint32_t levels)
{
    if (cap <= p->cap && sh0_batches <= p->sh0_batches && levels <= p->levels) return RT_OK;
    const size_t c = cap > p->cap ? cap : p->cap;
    const size_t sb = sh0_batches > p->sh0_batches ? sh0_batches : p->sh0_batches;
    const uint32_t nl = levels > p->levels ? levels : p->levels;
    RT_TRY(p->counters.reserve(POOL_OFFSET_WORDS * 4 + POOL_BYTES));
    for (uint32_t l = 0; l <= nl; l++) {
        rt_pipeline::LevelBuf &b = p->lv[l];
        const size_t slots = l == 0 ? c : 2 * c, shadow = l == 0 ? sb * c : 4 * c;
        if (l > 0) { RT_TRY(b.O.reserve(slots * 16)); RT_TRY(b.D.reserve(slots * 16)); RT_TRY(b.pix.reserve(slots * 4)); }
        RT_TRY(b.hit.reserve(slots * 16)); RT_TRY(b.inst.reserve(slots * 4));
        RT_TRY(b.slot_j.reserve(slots * 4)); RT_TRY(b.jlist.reserve(slots * 4));
        RT_TRY(b.shO.reserve(shadow * 16)); RT_TRY(b.shD.reserve(shadow * 16)); RT_TRY(b.vis.reserve(shadow * 4));
        if (l > 0 && nl > 1) RT_TRY(b.color.reserve(slots * 16));          // deep paths only (k_shade_level)
    }
    p->cap = (uint32_t)c;
    p->sh0_batches = (uint32_t)sb;
    p->levels = nl;
    return RT_OK;
}

// radiance levels a frame traces: level l exists when hits of depth l-1 may spawn rays
inline uint32_t frame_levels(const rt_pipeline *p) { return p->max_rad < (uint32_t)MAXD ? p->max_rad : (uint32_t)MAXD; }

// returns the first error of an event record (kernel launch errors surface in hipGetLastError at the call site)
template <int STACK, bool TWO_LEVEL>
hipError_t launch_frame(rt_pipeline *p, const PipeDev &pd, uint32_t shadow_slots)
{
    hipError_t first_error = hipSuccess;
    auto record = [&](hipEvent_t e, hipStream_t s) { const hipError_t rc = hipEventRecord(e, s); if (first_error == hipSuccess) first_error = rc; };
    hipStream_t st = p->ctx->stream;
    const bool T = p->ring_frames > 0;
    const size_t ring_slot = T ? (size_t)(p->ring_pos % (uint64_t)p->ring_frames) : 0;
    hipEvent_t *ev = T ? &p->ring[ring_slot * EV_COUNT] : nullptr;
    const uint32_t cap = pd.cap;
    const rt_context *ctx = p->ctx;
    const uint32_t levels = frame_levels(p);
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    if (T) record(ev[0], st);
    // primary rays are coherent: one 8x8 tile per wave, scheduled by the hardware dispatcher
    if (pd.n_frames > 1u) k_primary<STACK, TWO_LEVEL, true><<<blocks(cap), PBLOCK, 0, st>>>(pd);
    else k_primary<STACK, TWO_LEVEL, false><<<blocks(cap), PBLOCK, 0, st>>>(pd);
    k_compact_level<<<(cap + CTILES * CBLOCK - 1) / (CTILES * CBLOCK), CBLOCK, 0, st>>>(pd, 0);
    if (T) record(ev[1], st);
    const bool B = pd.n_frames > 1u;            // a batch of frames: the shading kernels pick the constants of every hit's frame
    if (B) k_shade_emit<true, true><<<blocks(cap), PBLOCK, 0, st>>>(pd, 0, shadow_slots, levels >= 1 ? 1u : 0u);
    else k_shade_emit<true, false><<<blocks(cap), PBLOCK, 0, st>>>(pd, 0, shadow_slots, levels >= 1 ? 1u : 0u);
    if (T) record(ev[2], st);
    ShadowQueues shadows;
    memset(&shadows, 0, sizeof shadows);
    const LightRays lr = light_rays(pd), none = no_light_rays();
    const LightRays *fl = B ? pd.frame_lights : nullptr;
    shadows.frame_lights = fl;
    shadows.q[0] = QueueSrc{pd.lv[0].shO, pd.lv[0].shD, &pd.counters[C_NHIT], cap, shadow_slots, any, lr};     // RaytracingCommon.hlsli:94
    shadows.vis[0] = pd.lv[0].vis;
    shadows.nq = 1;
    size_t shadow_max = (size_t)cap * shadow_slots;
    for (uint32_t l = 1; l <= levels; l++) {
        // level 1: the diffuse and the specular batch of the primary hits; deeper: one ray per hit of level l-1
        const QueueSrc rays = {pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], cap, l == 1 ? 2u : 1u, RT_RAY_FLAG_NONE, none};   // ProgressiveRaytracing.hlsl:53
        k_trace_secondary<STACK, TWO_LEVEL><<<rt_persistent_grid(ctx, k_trace_secondary<STACK, TWO_LEVEL>, PBLOCK, (size_t)cap * 2), PBLOCK, 0, st>>>(
            pd.sc, rays, pd.lv[l].hit, pd.lv[l].inst, pd.pools + (size_t)l * RT_POOL_GROUPS * RT_POOL_STRIDE, &pd.counters[C_SECONDARY]);
        k_compact_level<<<(2 * cap + CTILES * CBLOCK - 1) / (CTILES * CBLOCK), CBLOCK, 0, st>>>(pd, (int)l);
        if (T) record(ev[3 + 2 * (l - 1)], st);
        const bool casts_shadows = l < pd.max_shadow, spawns = l < levels;
        if (casts_shadows || spawns) {
            if (B) k_shade_emit<false, true><<<blocks((size_t)cap * 2), PBLOCK, 0, st>>>(pd, (int)l, 2u, spawns ? 1u : 0u);
            else k_shade_emit<false, false><<<blocks((size_t)cap * 2), PBLOCK, 0, st>>>(pd, (int)l, 2u, spawns ? 1u : 0u);
        }
        if (T) record(ev[4 + 2 * (l - 1)], st);
        if (casts_shadows) {
            shadows.q[shadows.nq] = QueueSrc{pd.lv[l].shO, pd.lv[l].shD, &pd.counters[C_NHIT + l], 2u * cap, 2u, any, lr};
            shadows.vis[shadows.nq] = pd.lv[l].vis;
            shadows.nq++;
            shadow_max += (size_t)cap * 4;
        }
    }
    if (B) k_trace_shadow<STACK, TWO_LEVEL, true><<<rt_persistent_grid(ctx, k_trace_shadow<STACK, TWO_LEVEL, true>, PBLOCK, shadow_max), PBLOCK, 0, st>>>(
        pd.sc, shadows, pd.pools, &pd.counters[C_SHADOW]);
    else k_trace_shadow<STACK, TWO_LEVEL, false><<<rt_persistent_grid(ctx, k_trace_shadow<STACK, TWO_LEVEL, false>, PBLOCK, shadow_max), PBLOCK, 0, st>>>(
        pd.sc, shadows, pd.pools, &pd.counters[C_SHADOW]);
    if (T) record(ev[EV_SHADOW], st);
    // (resolve: one thread per pixel slot of ONE frame; a batch's frames are accumulated in order inside the thread)
    if (levels <= 1) {                          // (level by level is slower here: 0.143 vs 0.118 ms at 1080p)
        if (B) k_resolve<false, true><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
        else k_resolve<false, false><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
    } else {
        for (uint32_t l = levels; l >= 1; l--) {
            if (B) k_shade_level<true><<<blocks((size_t)cap * 2), PBLOCK, 0, st>>>(pd, (int)l);
            else k_shade_level<false><<<blocks((size_t)cap * 2), PBLOCK, 0, st>>>(pd, (int)l);
        }
        if (B) k_resolve<true, true><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
        else k_resolve<true, false><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
    }
    if (T) { record(ev[EV_RESOLVE], st); p->ring_levels[ring_slot] = (uint8_t)levels; p->ring_nframes[ring_slot] = (uint8_t)pd.n_frames; p->ring_pos++; }
    return first_error;
}

template <int STACK>
hipError_t launch_frame_any(rt_pipeline *p, const PipeDev &pd, uint32_t shadow_slots)
{
    return p->scene->two_level ? launch_frame<STACK, true>(p, pd, shadow_slots) : launch_frame<STACK, false>(p, pd, shadow_slots);
}

template <bool TWO_LEVEL>
static int count_walk_launch(rt_pipeline *p, unsigned long long *w)
{
    hipStream_t st = p->ctx->stream;
    const rt_context *ctx = p->ctx;
    const PipeDev &pd = p->last_pd;
    const uint32_t cap = pd.cap, ss = p->last_shadow_slots;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    k_walk_primary<TWO_LEVEL><<<rt_persistent_grid(ctx, k_walk_primary<TWO_LEVEL>, PBLOCK, cap), PBLOCK, 0, st>>>(pd, w + 7 * RT_STAGE_PRIMARY);
    const unsigned gq = rt_persistent_grid(ctx, k_walk_queue<TWO_LEVEL>, PBLOCK, (size_t)cap * 2);
    const unsigned gs = rt_persistent_grid(ctx, k_walk_shadow<TWO_LEVEL>, PBLOCK, (size_t)cap * 2);
    const LightRays lr = light_rays(pd), none = no_light_rays();
    const LightRays *fl = pd.n_frames > 1u ? pd.frame_lights : nullptr;
    k_walk_shadow<TWO_LEVEL><<<gs, PBLOCK, 0, st>>>(pd.sc, LitQueueSrc{QueueSrc{pd.lv[0].shO, pd.lv[0].shD, &pd.counters[C_NHIT], cap, ss, any, lr}, fl}, w + 7 * RT_STAGE_SHADOW0);
    const uint32_t levels = pd.max_rad < (uint32_t)MAXD ? pd.max_rad : (uint32_t)MAXD;
    for (uint32_t l = 1; l <= levels; l++) {
        k_walk_queue<TWO_LEVEL><<<gq, PBLOCK, 0, st>>>(pd.sc, QueueSrc{pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], cap, l == 1 ? 2u : 1u, RT_RAY_FLAG_NONE, none},
                                                       w + 7 * RT_STAGE_SECONDARY);
        if (l < pd.max_shadow)
            k_walk_shadow<TWO_LEVEL><<<gs, PBLOCK, 0, st>>>(pd.sc, LitQueueSrc{QueueSrc{pd.lv[l].shO, pd.lv[l].shD, &pd.counters[C_NHIT + l], 2u * cap, 2u, any, lr}, fl}, w + 7 * RT_STAGE_SHADOW1);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
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
    DevBuf *all[] = {&p->d_mats, &p->d_env, &p->accum_own, &p->aov_own, &p->counters, &p->half_out, &p->totals, &p->work, &p->batch_consts};
    for (DevBuf *b : all) b->release();
    for (rt_pipeline::LevelBuf &l : p->lv) {
        DevBuf *lb[] = {&l.O, &l.D, &l.hit, &l.inst, &l.slot_j, &l.jlist, &l.pix, &l.shO, &l.shD, &l.vis, &l.color};
        for (DevBuf *b : lb) b->release();
    }
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
    p->rendered = false;        // last_pd holds device pointers of the previous scene
    return RT_OK;
}

int rt_pipeline_add_material(rt_pipeline *p, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    p->mats.push_back(*m);
    p->mats_dirty = true;
    p->rendered = false;        // d_mats may be reallocated by the next render
    return RT_OK;
}

int rt_pipeline_set_material(rt_pipeline *p, uint32_t index, const rt_material_params *m)
{
    RT_REQUIRE(p && m, "null argument");
    RT_REQUIRE(index < p->mats.size(), "material index out of range");
    p->mats[index] = *m;
    p->mats_dirty = true;
    p->rendered = false;
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

int rt_pipeline_set_environment_filter(rt_pipeline *p, uint32_t filter)
{
    RT_REQUIRE(p, "null pipeline");
    RT_REQUIRE(filter == RT_CUBE_SEAMLESS || filter == RT_CUBE_FACE_CLAMP, "unknown cube-map filter");
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
    p->accum = (float4 *)device_rgba32f;
    p->width = width; p->height = height; p->format = RT_FORMAT_R32G32B32A32_FLOAT;
    p->rendered = false;
    return RT_OK;
}

int rt_pipeline_build_acceleration_structures(rt_pipeline *p)
{
    RT_REQUIRE(p, "null pipeline");
    if (!p->scene) { rt_set_error("buildAccelerationStructures: no scene set"); return RT_ERR_STATE; }
    if (p->scene->built) return RT_OK;       // built once, shared between pipelines
    p->rendered = false;                     // a rebuild reallocates what last_pd points at
    return rt_scene_build(p->scene, 2);
}

int rt_pipeline_set_depth_limits(rt_pipeline *p, uint32_t max_radiance_depth, uint32_t max_shadow_depth)
{
    RT_REQUIRE(p, "null pipeline");
    if (max_radiance_depth > (uint32_t)MAXD) {
        rt_set_error("max radiance depth %u: the wavefront DAG holds at most %d radiance levels", max_radiance_depth, MAXD);
        return RT_ERR_UNSUPPORTED;
    }
    p->max_rad = max_radiance_depth;
    p->max_shadow = max_shadow_depth;
    return RT_OK;
}

int rt_pipeline_set_skip_unlit_shadow_rays(rt_pipeline *p, int on)
{
    RT_REQUIRE(p, "null pipeline");
    p->skip_unlit = on ? 1u : 0u;
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
    RT_TRY(ensure_queues(p, cap, shadow_slots, frame_levels(p)));
    if (!p->totals.p) {
        RT_TRY(p->totals.reserve(8 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), st));
    }
    PipeDev pd;
    // (threads of the largest launch: the primary stage runs one thread per pixel slot, the persistent stages fewer)
    RT_TRY(rt_scene_dev_for_launch(ctx, p->scene, rt_lds_stack_rows(ctx), cap > ctx->cu_count * 16u * PBLOCK ? cap : ctx->cu_count * 16u * PBLOCK, &pd.sc));
    pd.pfc = frames[0];
    pd.n_frames = n_frames; pd.fcap = fcap;
    pd.pfcs = nullptr; pd.frame_lights = nullptr;
    pd.shadow_compact = ao_view ? 0u : 1u;          // the AO view's four rays have random directions
    if (n_frames > 1) {
        // the batch's constants and light rays go to device memory: kernels index them by the frame of a slot
        const size_t cb = sizeof(rt_per_frame_constants) * RT_MAX_BATCH, lb = sizeof(LightRays) * RT_MAX_BATCH;
        RT_TRY(p->batch_consts.reserve(cb + lb));
        std::vector<unsigned char> stage(cb + lb, 0);
        for (uint32_t f = 0; f < n_frames; f++) {
            memcpy(&stage[sizeof(rt_per_frame_constants) * f], &frames[f], sizeof(rt_per_frame_constants));
            const LightRays lr = light_rays(pd.shadow_compact, frames[f]);
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
    pd.accum_mode = p->accum_mode;
    pd.skip_unlit = p->skip_unlit;
    pd.kind = p->kind;
    pd.accum = p->accum;
    pd.aov_direct = p->accum;                       // realtime: output 0 = direct lighting, output 1 = indirect specular
    pd.aov_indirect = p->aov_own.as<float4>();
    pd.counters = p->counters.as<uint32_t>();
    for (int l = 0; l <= MAXD; l++) {
        rt_pipeline::LevelBuf &b = p->lv[l];
        LevelDev &d = pd.lv[l];
        d.O = b.O.as<float4>(); d.D = b.D.as<float4>(); d.hit = b.hit.as<float4>(); d.inst = b.inst.as<uint32_t>();
        d.slot_j = b.slot_j.as<uint32_t>(); d.jlist = b.jlist.as<uint32_t>(); d.pix = b.pix.as<uint32_t>();
        d.shO = b.shO.as<float4>(); d.shD = b.shD.as<float4>(); d.vis = b.vis.as<uint32_t>(); d.color = b.color.as<float4>();
    }
    static_assert(C_COUNT <= POOL_OFFSET_WORDS, "scalar counters overlap the chunk pools");
    pd.pools = pd.counters + POOL_OFFSET_WORDS;
    pd.totals = p->totals.as<unsigned long long>();
    // 18 LDS stack rows + the 8-row top table = 26 KiB per 256-thread block = 6 resident blocks per CU, whatever
    // the depth of the tree; the rare deeper walk continues in global rows (rt_trace_wave.h)
    HIP_TRY(ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST ? launch_frame_any<RT_LDS_STACK_ROWS_TEST>(p, pd, shadow_slots)
                                                           : launch_frame_any<RT_LDS_STACK_ROWS>(p, pd, shadow_slots));
    HIP_TRY(hipGetLastError());
    p->last_pd = pd;
    p->last_shadow_slots = shadow_slots;
    p->last_scene_gen = p->scene->generation;
    p->last_tile[0] = x0; p->last_tile[1] = y0; p->last_tile[2] = x1; p->last_tile[3] = y1;
    p->last_pixels = pd.n_pixels * n_frames;
    p->rendered = true;
    return RT_OK;
}

int rt_pipeline_render_tile(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
    return render_region(p, width, height, x0, y0, x1, y1, 0, 0, 1);
}

int rt_pipeline_render_bands(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world)
{
    RT_REQUIRE(world > 0 && rank < world, "rank outside [0, world)");
    RT_REQUIRE(band_rows > 0 && band_rows % 8 == 0, "band_rows must be a positive multiple of 8 (pixel slots are 8x8 tiles)");
    uint32_t n = 0;
    RT_TRY(rt_tile_bands(height, band_rows, rank, world, nullptr, nullptr, 0, &n));
    if (n == 0) return RT_OK;                   // more ranks than bands: nothing to render here
    return render_region(p, width, height, 0, 0, width, n * band_rows, band_rows, rank, world);
}

int rt_pipeline_render(rt_pipeline *p, uint32_t width, uint32_t height)
{
    return rt_pipeline_render_tile(p, width, height, 0, 0, width, height);
}

int rt_pipeline_render_batch(rt_pipeline *p, uint32_t width, uint32_t height, const rt_per_frame_constants *constants, uint32_t n)
{
    RT_REQUIRE(p && (constants || n == 0), "null argument");
    RT_REQUIRE(p->kind == RT_PIPELINE_PROGRESSIVE, "render_batch: only the progressive pipeline accumulates frames");
    uint32_t batch_max = RT_MAX_BATCH;
    if (const char *e = getenv("RT_BATCH_MAX")) { const int v = atoi(e); if (v >= 1 && v <= (int)RT_MAX_BATCH) batch_max = (uint32_t)v; }
    // frames RayGen would leave at once (accumCount >= maxIterations, ProgressiveRaytracing.hlsl:14-16) are dropped here;
    // frames that disagree on what sizes the queues (the ambient-occlusion view) do not share a set of launches
    std::vector<rt_per_frame_constants> run;
    auto flush = [&]() -> int {
        if (run.empty()) return RT_OK;
        const int rc = render_region(p, width, height, 0, 0, width, height, 0, 0, 1, run.data(), (uint32_t)run.size());
        run.clear();
        return rc;
    };
    for (uint32_t i = 0; i < n; i++) {
        const rt_per_frame_constants &c = constants[i];
        if (c.cameraParams.accumCount >= c.options.maxIterations) continue;
        if (!run.empty() && (run.size() >= batch_max || (run[0].options.showAmbientOcclusionOnly != 0) != (c.options.showAmbientOcclusionOnly != 0))) RT_TRY(flush());
        run.push_back(c);
    }
    RT_TRY(flush());
    if (n) { p->pfc = constants[n - 1]; p->have_pfc = true; }     // as after n x (update, render)
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

int rt_pipeline_write_output(rt_pipeline *p, const void *host_rgba32f, size_t bytes)
{
    RT_REQUIRE(p && host_rgba32f, "null argument");
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
    HIP_TRY(hipSetDevice(p->ctx->device));
    if (p->totals.p) HIP_TRY(hipMemsetAsync(p->totals.p, 0, 8 * sizeof(unsigned long long), p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->ring_pos = 0;
    return RT_OK;
}

int rt_pipeline_count_work(rt_pipeline *p, rt_stage_work *out)
{
    RT_REQUIRE(p && out, "null argument");
    if (!p->rendered || !p->scene->built || p->scene->generation != p->last_scene_gen) { rt_set_error("count_work: nothing rendered since the last change of scene, materials or output"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    RT_TRY(p->work.reserve(RT_STAGE_COUNT * 7 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, RT_STAGE_COUNT * 3 * sizeof(unsigned long long), st));
    unsigned long long *w = p->work.as<unsigned long long>();
    const PipeDev &pd = p->last_pd;
    const uint32_t cap = pd.cap, ss = p->last_shadow_slots;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    k_count_primary<<<blocks(cap), PBLOCK, 0, st>>>(pd, w + 3 * RT_STAGE_PRIMARY);
    const LightRays lr = light_rays(pd), none = no_light_rays();
    const LightRays *fl = pd.n_frames > 1u ? pd.frame_lights : nullptr;
    k_count_queue<<<(blocks(cap) + 1) * ss, PBLOCK, 0, st>>>(pd.sc, LitQueueSrc{QueueSrc{pd.lv[0].shO, pd.lv[0].shD, &pd.counters[C_NHIT], cap, ss, any, lr}, fl},
                                                             w + 3 * RT_STAGE_SHADOW0);
    const uint32_t levels = pd.max_rad < (uint32_t)MAXD ? pd.max_rad : (uint32_t)MAXD;
    for (uint32_t l = 1; l <= levels; l++) {        // every secondary level adds into the same two rows
        const uint32_t batches = l == 1 ? 2u : 1u;
        k_count_queue<<<(blocks((size_t)cap * 2) + 1) * batches, PBLOCK, 0, st>>>(
            pd.sc, LitQueueSrc{QueueSrc{pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], cap, batches, RT_RAY_FLAG_NONE, none}, nullptr}, w + 3 * RT_STAGE_SECONDARY);
        if (l < pd.max_shadow)
            k_count_queue<<<(blocks((size_t)cap * 2) + 1) * 2, PBLOCK, 0, st>>>(
                pd.sc, LitQueueSrc{QueueSrc{pd.lv[l].shO, pd.lv[l].shD, &pd.counters[C_NHIT + l], 2 * cap, 2, any, lr}, fl}, w + 3 * RT_STAGE_SHADOW1);
    }
    HIP_TRY(hipGetLastError());
    unsigned long long h[RT_STAGE_COUNT * 3];
    HIP_TRY(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < RT_STAGE_COUNT; k++) { out[k].rays = h[3 * k]; out[k].nodes = h[3 * k + 1]; out[k].tris = h[3 * k + 2]; }
    return RT_OK;
}

int rt_pipeline_count_walk(rt_pipeline *p, rt_stage_walk *out)
{
    RT_REQUIRE(p && out, "null argument");
    if (!p->rendered || !p->scene->built || p->scene->generation != p->last_scene_gen) { rt_set_error("count_walk: nothing rendered since the last change of scene, materials or output"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    const size_t bytes = RT_STAGE_COUNT * 7 * sizeof(unsigned long long);
    RT_TRY(p->work.reserve(bytes));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, bytes, st));
    unsigned long long *w = p->work.as<unsigned long long>();
    if (p->scene->two_level) RT_TRY(count_walk_launch<true>(p, w));
    else RT_TRY(count_walk_launch<false>(p, w));
    unsigned long long h[RT_STAGE_COUNT * 7];
    HIP_TRY(hipMemcpyAsync(h, w, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < RT_STAGE_COUNT; k++) {
        out[k].rays = h[7 * k]; out[k].nodes_global = h[7 * k + 1]; out[k].nodes_lds = h[7 * k + 2];
        out[k].tris = h[7 * k + 3]; out[k].instance_entries = h[7 * k + 4]; out[k].lines = h[7 * k + 5];
        out[k].longest_walk = h[7 * k + 6] >> 32;
        out[k].longest_walk_ray = (uint32_t)h[7 * k + 6];
    }
    return RT_OK;
}

// debugging aid for rt_stage_walk.longest_walk_ray: ray `index` of the level-1 radiance queue (diffuse batch, then specular batch)
int rt_debug_read_secondary_ray(rt_pipeline *p, uint32_t index, float origin_tmin[4], float dir_tmax[4])
{
    RT_REQUIRE(p && origin_tmin && dir_tmax, "null argument");
    if (!p->rendered) { rt_set_error("nothing rendered yet"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    uint32_t n = 0;
    HIP_TRY(hipMemcpy(&n, p->last_pd.counters + C_NHIT, 4, hipMemcpyDeviceToHost));
    RT_REQUIRE(n > 0 && index < 2 * n, "ray index out of range");
    const size_t slot = (size_t)(index / n) * p->last_pd.cap + index % n;
    HIP_TRY(hipMemcpy(origin_tmin, p->last_pd.lv[1].O + slot, 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dir_tmax, p->last_pd.lv[1].D + slot, 16, hipMemcpyDeviceToHost));
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

#ifdef RT_TRACE_STATS
// instrumentation build only: wave / lane counters of the pipeline's traversal kernels since the last call
int rt_debug_trace_stats(unsigned long long out[8 + 64])
{
    unsigned long long zero[64];
    memset(zero, 0, sizeof zero);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_trace_stats), 8 * sizeof zero[0]));
    HIP_TRY(hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(rtd::g_trace_sp_hist), sizeof zero));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_trace_stats), zero, 8 * sizeof zero[0]));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_trace_sp_hist), zero, sizeof zero));
    return RT_OK;
}
#endif

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
    k_debug_cube<<<blocks(n), PBLOCK, 0, ctx->stream>>>(pd, sb[1].as<float>(), sb[2].as<float>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, sb[2].p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
