// rt_shade.h -- the shaders of the reference as device functions: environment sampling, RayGen's ray set-up, the lights,
// shade() / shadeAOV() (assets/shaders/ProgressiveRaytracing.hlsl, RealtimeRaytracing.hlsl, RaytracingCommon.hlsli) and the
// "TraceRay providers" of the emit and resolve passes.  shade() is ONE template instantiated with an emit and a resolve
// provider, so both passes execute the same arithmetic and RNG draw order by construction (rt_pipeline.hip).
#pragma once

#include "rt_pipeline_dev.h"

namespace rtp {

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

// The frame a pixel slot belongs to, and the slot inside that frame (single frames: 0 and q itself).  The slots of a set of n
// frames are TILE major: the 64 slots of tile t of frame f are chunk t * n + f, so that the same 8x8 pixels of consecutive frames --
// rays that differ by a sub-pixel jitter -- sit next to each other in the primary launch and, since every queue is compacted in slot
// order, in every queue after it (frame-major order put them a whole frame apart: the waves that walk them found none of the other's
// nodes in the caches).
RT_DEV uint32_t slot_frame(const PipeDev &pd, uint32_t q, uint32_t &q_in_frame)
{
    if (pd.n_frames <= 1u) { q_in_frame = q; return 0u; }
    const uint32_t c = q >> 6, tile = c / pd.n_frames, f = c - tile * pd.n_frames;
    q_in_frame = (tile << 6) | (q & 63u);
    return f;
}
RT_DEV uint32_t frame_slot(const PipeDev &pd, uint32_t frame, uint32_t q_in_frame)
{
    return pd.n_frames <= 1u ? q_in_frame : ((((q_in_frame >> 6) * pd.n_frames + frame) << 6) | (q_in_frame & 63u));
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
    uint32_t seed = io.pixel_seed(pix);      // initRand(pixel, frameCount): the shaders re-seed in every shade() call of the pixel
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

// The specular lobe of a hit at the depth limit (every secondary hit with the reference's limits: MAX_RADIANCE_RAY_DEPTH 1,
// RaytracingCommon.hlsli:11) multiplies its sample's weight by the radiance of a ray that is not traced: refl = (0, 0, 0), and
// `specular + refl * brdf / pdf` is (+0, +0, +0) -- unless brdf / pdf is not a number, which the shaders would then propagate.  When can it
// be?  pdf = (e + 1) / 2pi * pc, brdf = (e + 2) / 2pi * pc, pc = pow(ct, e), ct = pow(r0, 1 / (e + 1)), e = exp((1 - roughness) * 12)
// (RaytracingUtils.hlsli:95-123; pow = exp(y log x) with the kernels of rt_device_math.h, <= 1.5 ulp each).  For a roughness in
// [-0.15, 1.6], e lies in [7e-4, 1e6]; for a first random number r0 > 0 (it is a multiple of 2^-24) log r0 >= -16.7, so ct is in
// (5e-8, 1 + 2e-7], e log ct >= -16.7 e / (e + 1) - 1.6 (the 1.6: 2^20 times the kernels' absolute error on log ct), pc in [1e-8, 5],
// and pdf and brdf are positive and finite: 0 * brdf is +0, +0 / pdf is +0, +0 + +0 is +0.  The rest of shade() -- the Fresnel term,
// `specular * reflectivity * fresnel`, the sum -- runs as written on that +0.  Outside those ranges (a first draw of exactly 0: one in
// 2^24; a material from outside) the lobe is sampled as always.  What it saves: 26 % of the resolve pass's instructions.
template <class IO>
RT_DEV bool black_lobe(const IO &io, const rt_material_params &mp, uint32_t seed, uint32_t depth)
{
    if (!io.secondary_is_black(depth)) return false;
    if (!(mp.roughness >= -0.15f && mp.roughness <= 1.6f)) return false;
    return next_rand(seed) > 0.0f;              // (the lobe's first draw, on a copy of the sequence)
}

// ---- shade (ProgressiveRaytracing.hlsl:80-148) + evaluateIndirectDiffuse (:57-78)
template <class IO>
RT_DEV f3 shade(const PipeDev &pd, IO &io, const rt_material_params &mp, f3 P, f3 N, f3 D, uint32_t depth, uint32_t pix)
{
    const rt_debug_options &opt = pd.pfc.options;
    if (opt.showAmbientOcclusionOnly) return ambient_occlusion(pd, io, P, N, pix);

    uint32_t seed = io.pixel_seed(pix);      // initRand(pixel, frameCount): the shaders re-seed in every shade() call of the pixel

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
            if (black_lobe(io, mp, seed, depth)) {
                // (round 5) the lobe's ray is beyond the depth limit: io.secondary() returns black whatever its arguments, and
                // 0 * brdf / pdf is +0 -- see black_lobe() -- so the lobe (two pows, a sine and a cosine, a frame) need not be sampled
            } else {
                const float exponent = exp_det((1.0f - mp.roughness) * 12.0f);
                float pdf, brdf;
                const f3 mirror = reflect(D, N);
                const f3 dir = phong_lobe(seed, mirror, exponent, pdf, brdf);
                const f3 refl = io.secondary(1, P, dir, RAY_EPSILON, depth);
                specular = specular + refl * brdf / pdf;
            }
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
    uint32_t seed = io.pixel_seed(pix);      // initRand(pixel, frameCount): the shaders re-seed in every shade() call of the pixel
    f3 direct = mk3(0.0f, 0.0f, 0.0f);
    direct = direct + directional_light(pd, io, P, N, depth);
    direct = direct + point_light(pd, io, P, N, depth);
    f3 fresnel = mk3(0.0f, 0.0f, 0.0f);
    f3 specular = mk3(0.0f, 0.0f, 0.0f);
    if (mp.type == 1u || mp.type == 2u) {
        if (mp.reflectivity > 0.001f) {
            if (black_lobe(io, mp, seed, depth)) {
                // (round 5) the lobe's ray is beyond the depth limit: io.secondary() returns black whatever its arguments, and
                // 0 * brdf / pdf is +0 -- see black_lobe() -- so the lobe (two pows, a sine and a cosine, a frame) need not be sampled
            } else {
                const float exponent = exp_det((1.0f - mp.roughness) * 12.0f);
                float pdf, brdf;
                const f3 mirror = reflect(D, N);
                const f3 dir = phong_lobe(seed, mirror, exponent, pdf, brdf);
                const f3 refl = io.secondary(1, P, dir, RAY_EPSILON, depth);
                specular = specular + refl * brdf / pdf;
            }
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
    f3 shadow_origin;           // compact shadow queue: the hit point both light rays start from
    RT_DEV EmitIO(const PipeDev &p, int level, uint32_t i, uint32_t qq, uint32_t f) : pd(p), L(level), idx(i), q(qq), frame(f), shadow_mask(0), skip_mask(0), sec_mask(0)
    {
        shadow_origin = mk3(0.0f, 0.0f, 0.0f);
    }
    RT_DEV uint32_t pixel_seed(uint32_t pix) const { return init_rand(pix, pd.pfc.cameraParams.frameCount); }
    RT_DEV bool secondary_is_black(uint32_t depth) const { return depth >= pd.max_rad || L >= MAXD; }      // secondary() below returns at once
    // matters = false: whatever this ray finds is multiplied by zero by the caller
    RT_DEV float shadow(int s, f3 o, f3 d, float tmin, float tmax, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        const bool skipped = !matters && pd.skip_unlit;     // "emitted but not worth traversing"; the trace kernel counts these
        shadow_mask |= 1u << s;
        if (skipped) skip_mask |= 1u << s;
        if (pd.shadow_compact) {                            // the ray is rebuilt from the hit point by the loader (ShadowSrc::load)
            shadow_origin = o;
            return 1.0f;
        }
        if (skipped) store_ray(pd.sh_O, pd.sh_D, sh_ray(pd, L, idx, (uint32_t)s), mk3(0.0f, 0.0f, 0.0f), 0.0f, mk3(0.0f, 0.0f, 0.0f), RT_TMAX_SKIPPED);
        else store_ray(pd.sh_O, pd.sh_D, sh_ray(pd, L, idx, (uint32_t)s), o, tmin, d, tmax);
        return 1.0f;
    }
    // the shadow slots of this hit that no ray went to are marked "not traced"
    RT_DEV void finish_shadows(uint32_t shadow_slots) const
    {
        if ((uint32_t)L >= pd.sh_levels) return;            // a level that casts no shadow rays has no entries in the queue
        if (pd.shadow_compact) {
            pd.sh_hits[pd.sh_cbase[L] + idx] = make_float4(shadow_origin.x, shadow_origin.y, shadow_origin.z, __uint_as_float(shadow_mask | (skip_mask << 2) | (frame << 8)));
            return;
        }
        for (uint32_t s = 0; s < shadow_slots; s++)
            if (!(shadow_mask & (1u << s))) store_invalid(pd.sh_O, pd.sh_D, sh_ray(pd, L, idx, s));
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if (depth >= pd.max_rad || L >= MAXD) return mk3(0.0f, 0.0f, 0.0f);
        const size_t slot = L == 0 ? (size_t)w * pd.lv[1].rstride + idx : idx;
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
    uint32_t seed0;             // initRand(pixel, frameCount): every shade() of the pixel starts from it, so the 16 rounds run once per
                                //   pixel and not once per level of the inline recursion (-10 % of k_resolve's instructions)
    RT_DEV ResolveIO(const PipeDev &p, uint32_t i, uint32_t px) : pd(p), idx(i), pix(px), seed0(init_rand(px, p.pfc.cameraParams.frameCount)) {}
    RT_DEV ResolveIO(const PipeDev &p, uint32_t i, uint32_t px, uint32_t seed) : pd(p), idx(i), pix(px), seed0(seed) {}
    RT_DEV uint32_t pixel_seed(uint32_t) const { return seed0; }
    RT_DEV bool secondary_is_black(uint32_t depth) const { return L >= MAXL || depth >= pd.max_rad; }
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        if (!matters && pd.skip_unlit) return 1.0f;                       // never traced; the caller multiplies by zero
        return shadow_bit(pd, sh_ray(pd, L, idx, (uint32_t)s)) ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int w, f3 o, f3 d, float tmin, uint32_t depth)
    {
        if constexpr (L >= MAXL) {
            return mk3(0.0f, 0.0f, 0.0f);
        } else {
            if (depth >= pd.max_rad) return mk3(0.0f, 0.0f, 0.0f);
            const size_t slot = L == 0 ? (size_t)w * pd.lv[1].rstride + idx : idx;
            const float4 h = pd.lv[L + 1].hit[slot];
            if (h.x == HIT_MISS) return sample_environment(pd, d);             // PrimaryMiss, :160-164
            if (h.x == HIT_UNTRACED) return mk3(0.0f, 0.0f, 0.0f);
            RayD r;
            r.o = o; r.tmin = tmin; r.d = d; r.tmax = RAY_MAX_T;
            ResolveIO<L + 1, MAXL> io(pd, pd.lv[L + 1].slot_j[slot], pix, seed0);
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
    RT_DEV uint32_t pixel_seed(uint32_t pix) const { return init_rand(pix, pd.pfc.cameraParams.frameCount); }
    RT_DEV bool secondary_is_black(uint32_t depth) const { return depth >= pd.max_rad || L >= MAXD; }
    RT_DEV float shadow(int s, f3, f3, float, float, uint32_t depth, bool matters)
    {
        if (depth >= pd.max_shadow) return 1.0f;
        if (!matters && pd.skip_unlit) return 1.0f;
        return shadow_bit(pd, sh_ray(pd, L, idx, (uint32_t)s)) ? 1.0f : 0.0f;
    }
    RT_DEV f3 secondary(int w, f3, f3 d, float, uint32_t depth)
    {
        if (depth >= pd.max_rad || L >= MAXD) return mk3(0.0f, 0.0f, 0.0f);
        const size_t slot = L == 0 ? (size_t)w * pd.lv[1].rstride + idx : idx;
        const float4 h = pd.lv[L + 1].hit[slot];
        if (h.x == HIT_MISS) return sample_environment(pd, d);             // PrimaryMiss, :160-164
        if (h.x == HIT_UNTRACED) return mk3(0.0f, 0.0f, 0.0f);
        const float4 c = pd.lv[L + 1].color[slot];
        return mk3(c.x, c.y, c.z);
    }
};

}  // namespace rtp
