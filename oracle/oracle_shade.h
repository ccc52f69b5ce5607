/*
 * oracle_shade.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * 1:1 scalar restatement of the reference's HLSL entry points, recursion and
 * all, for one pixel at a time:
 *   RayGen / shootSecondaryRay / evaluateIndirectDiffuse / shade /
 *   PrimaryClosestHit / PrimaryMiss / Shadow*   assets/shaders/ProgressiveRaytracing.hlsl:11-182
 *   interpolateVertexAttributes / shootShadowRay / evaluateAO /
 *   evaluateDirectionalLight / evaluatePointLight / sampleEnvironment
 *                                               assets/shaders/RaytracingCommon.hlsli:53-159
 *   HitWorldPosition                            assets/shaders/RaytracingUtils.hlsli:209-212
 * Material = f(instance) as the reference's shader-table layout implies
 * (RtBindings.cpp:131-164, ProgressiveRaytracingPipeline.cpp:220-227).
 */
#ifndef ORACLE_SHADE_H
#define ORACLE_SHADE_H

#include "oracle_bvh.h"

namespace orc {

#define ORC_RAY_MAX_T 1.0e+38f
#define ORC_RAY_EPSILON 0.0001f
#define ORC_M_PI 3.1415927f

struct Env {
    const float *faces;     /* 6 * size * size * 4 floats, D3D face order +X -X +Y -Y +Z -Z, or NULL */
    int size;
    float constant[3];      /* used when faces == NULL */
    bool seamless;          /* cross-face bilinear taps (D3D10+ behaviour) */
};

struct RenderCtx {
    const Scene *scene;
    const rt_material_params *mats;
    uint32_t nmats;
    Env env;
    rt_per_frame_constants pfc;
    uint32_t width, height;
    uint32_t max_radiance_depth;   /* MAX_RADIANCE_RAY_DEPTH, RaytracingCommon.hlsli:11 */
    uint32_t max_shadow_depth;     /* MAX_SHADOW_RAY_DEPTH,   RaytracingCommon.hlsli:12 */
    bool use_brute;                /* trace through the brute-force loop instead of the BVH */
    /* tests/test_s2_truth.py: trace through the float64 geometric truth instead (truth64.h: no box of any kind); the hook is filled
     * by oracle.cpp, NULL otherwise */
    Hit (*truth)(const void *truth_scene, const Ray &r, uint32_t flags) = nullptr;
    const void *truth_scene = nullptr;
};

struct PixelStats {
    uint64_t rays_primary, rays_secondary, rays_shadow;
    uint64_t primary_hits, secondary_hits;
    uint64_t nodes, tris;          /* traversal counters summed over every ray */
    uint64_t shaded_hits;
};

struct PixelCtx {
    const RenderCtx *rc;
    uint32_t px, py;
    PixelStats *st;
};

static inline Hit trace(const PixelCtx &pc, const Ray &r, uint32_t flags)
{
    if (pc.rc->truth) return pc.rc->truth(pc.rc->truth_scene, r, flags);
    if (pc.rc->use_brute) return trace_brute(*pc.rc->scene, r, flags);
    Counters c;
    Hit h = trace_bvh(*pc.rc->scene, r, flags, c);
    pc.st->nodes += c.nodes;
    pc.st->tris += c.tris;
    return h;
}

/* TextureCube.SampleLevel(linear, dir, 0) (RaytracingCommon.hlsli:152; sampler MIN_MAG_LINEAR,
 * ProgressiveRaytracingPipeline.cpp:48-55): D3D major-axis face selection and bilinear filtering.  D3D10+
 * devices filter cube maps seamlessly: a tap that falls off the selected face is taken from the face across
 * that edge (seamless = true, the default); a tap that hangs over a cube CORNER has no texel and takes the
 * mean of the other three taps of the footprint (D3D11 functional spec's suggestion; hardware differs there,
 * parity unpinned).  seamless = false clamps the taps to the selected face.
 *
 * cube_edge[face][edge] with edge = x<0, x>=N, y<0, y>=N: { face across the edge, which coordinate is fixed
 * there (0 = x, 1 = y), whether it is fixed at N-1 (else 0), whether the position along the edge is mirrored }. */
struct CubeEdge { int face, fixed_is_y, at_max, mirrored; };
static const CubeEdge cube_edge[6][4] = {
    /* +X */ {{4, 0, 1, 0}, {5, 0, 0, 0}, {2, 0, 1, 1}, {3, 0, 1, 0}},
    /* -X */ {{5, 0, 1, 0}, {4, 0, 0, 0}, {2, 0, 0, 0}, {3, 0, 0, 1}},
    /* +Y */ {{1, 1, 0, 0}, {0, 1, 0, 1}, {5, 1, 0, 1}, {4, 1, 0, 0}},
    /* -Y */ {{1, 1, 1, 1}, {0, 1, 1, 0}, {4, 1, 1, 0}, {5, 1, 1, 1}},
    /* +Z */ {{1, 0, 1, 0}, {0, 0, 0, 0}, {2, 1, 1, 0}, {3, 1, 0, 0}},
    /* -Z */ {{0, 0, 1, 0}, {1, 0, 0, 0}, {2, 1, 0, 1}, {3, 1, 1, 1}},
};

/* texel (x, y) of a face, x and y in [-1, size]; NULL when the tap hangs over a corner of the cube */
static inline const float *cubeTexel(const Env &env, int face, int x, int y)
{
    const int m = env.size - 1;
    const bool offx = x < 0 || x > m, offy = y < 0 || y > m;
    if (!env.seamless) {
        x = x < 0 ? 0 : (x > m ? m : x);
        y = y < 0 ? 0 : (y > m ? m : y);
    } else if (offx && offy) {
        return NULL;
    } else if (offx || offy) {
        const CubeEdge &e = cube_edge[face][offx ? (x < 0 ? 0 : 1) : (y < 0 ? 2 : 3)];
        const int along = offx ? y : x;
        const int pos = e.mirrored ? m - along : along, edge = e.at_max ? m : 0;
        face = e.face;
        if (e.fixed_is_y) { x = pos; y = edge; } else { x = edge; y = pos; }
    }
    return env.faces + (((size_t)face * env.size + (size_t)y) * env.size + (size_t)x) * 4;
}

static inline V3 sampleCube(const Env &env, V3 d)
{
    if (!env.faces) return v3(env.constant[0], env.constant[1], env.constant[2]);
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face; float ma, sc, tc;
    if (ax >= ay && ax >= az) { face = d.x > 0 ? 0 : 1; ma = ax; sc = d.x > 0 ? -d.z : d.z; tc = -d.y; }
    else if (ay >= az)        { face = d.y > 0 ? 2 : 3; ma = ay; sc = d.x; tc = d.y > 0 ? d.z : -d.z; }
    else                      { face = d.z > 0 ? 4 : 5; ma = az; sc = d.z > 0 ? d.x : -d.x; tc = -d.y; }
    if (!(ma > 0.0f) || !(ma < u2f(0x7f800000u))) return v3(0, 0, 0);
    float u = (sc / ma + 1.0f) * 0.5f;
    float v = (tc / ma + 1.0f) * 0.5f;
    float n = (float)env.size;
    float fx = u * n - 0.5f;
    float fy = v * n - 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = (int)x0f, y0 = (int)y0f;
    const float *tap[4] = { cubeTexel(env, face, x0, y0), cubeTexel(env, face, x0 + 1, y0),
                            cubeTexel(env, face, x0, y0 + 1), cubeTexel(env, face, x0 + 1, y0 + 1) };
    float corner[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 4; k++) {
        if (tap[k]) continue;
        for (int j = 0; j < 4; j++)
            if (j != k) for (int ch = 0; ch < 3; ch++) corner[ch] = corner[ch] + tap[j][ch];
        for (int ch = 0; ch < 3; ch++) corner[ch] = corner[ch] / 3.0f;
        tap[k] = corner;
    }
    float out[3];
    for (int k = 0; k < 3; k++) {
        float top = tap[0][k] + (tap[1][k] - tap[0][k]) * wx;
        float bot = tap[2][k] + (tap[3][k] - tap[2][k]) * wx;
        out[k] = top + (bot - top) * wy;
    }
    return v3(out[0], out[1], out[2]);
}

/* RaytracingCommon.hlsli:149-159 */
static inline V3 sampleEnvironment(const PixelCtx &pc, V3 rayDir)
{
    V3 e = sampleCube(pc.rc->env, rayDir);
    return vscale(e, pc.rc->pfc.options.environmentStrength);
}

/* RaytracingCommon.hlsli:84-96 */
static inline float shootShadowRay(const PixelCtx &pc, V3 orig, V3 dir, float minT, float maxT, uint32_t currentDepth)
{
    if (currentDepth >= pc.rc->max_shadow_depth) return 1.0f;
    Ray r = { orig, minT, dir, maxT };
    pc.st->rays_shadow++;
    Hit h = trace(pc, r, RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER);
    return h.inst == RT_NO_HIT ? 1.0f : 0.0f;    /* ShadowMiss sets 1, ProgressiveRaytracing.hlsl:178-182 */
}

/* RaytracingCommon.hlsli:98-124 */
static inline V3 evaluateAO(const PixelCtx &pc, V3 position, V3 normal)
{
    float visibility = 0.0f;
    const int aoRayCount = 4;
    uint32_t seed = initRand(pc.px + pc.py * pc.rc->width, pc.rc->pfc.cameraParams.frameCount);
    for (int i = 0; i < aoRayCount; ++i) {
        V3 sampleDir; float NoL, pdf;
        if (pc.rc->pfc.options.cosineHemisphereSampling) {
            sampleDir = getCosHemisphereSample(&seed, normal);
            NoL = saturate(dot3(normal, sampleDir));
            pdf = NoL / ORC_M_PI;
        } else {
            sampleDir = getUniformHemisphereSample(&seed, normal);
            NoL = saturate(dot3(normal, sampleDir));
            pdf = 1.0f / (2.0f * ORC_M_PI);
        }
        float vis = shootShadowRay(pc, position, sampleDir, ORC_RAY_EPSILON, 10.0f, 1);
        float term = vis * NoL;
        term = term / pdf;
        visibility = visibility + term;
    }
    float r = visibility / (float)aoRayCount;
    return v3(r, r, r);
}

/* RaytracingCommon.hlsli:126-134 */
static inline V3 evaluateDirectionalLight(const PixelCtx &pc, V3 position, V3 normal, uint32_t currentDepth)
{
    const rt_directional_light_params &dl = pc.rc->pfc.directionalLight;
    V3 L = normalize3(v3(-dl.forwardDir.x, -dl.forwardDir.y, -dl.forwardDir.z));
    float NoL = saturate(dot3(normal, L));
    float visible = shootShadowRay(pc, position, L, ORC_RAY_EPSILON, ORC_RAY_MAX_T, currentDepth);
    V3 c = v3(dl.color.x, dl.color.y, dl.color.z);
    c = vscale(c, dl.color.w);
    c = vscale(c, NoL);
    c = vscale(c, visible);
    return c;
}

/* RaytracingCommon.hlsli:136-147 */
static inline V3 evaluatePointLight(const PixelCtx &pc, V3 position, V3 normal, uint32_t currentDepth)
{
    const rt_point_light_params &pl = pc.rc->pfc.pointLight;
    V3 lightPath = vsub(v3(pl.worldPos.x, pl.worldPos.y, pl.worldPos.z), position);
    float lightDistance = length3(lightPath);
    V3 L = normalize3(lightPath);
    float NoL = saturate(dot3(normal, L));
    float visible = shootShadowRay(pc, position, L, ORC_RAY_EPSILON, lightDistance - ORC_RAY_EPSILON, currentDepth);
    float den = 2.0f * ORC_M_PI;
    den = den * lightDistance;
    den = den * lightDistance;
    float falloff = 1.0f / den;
    V3 c = v3(pl.color.x, pl.color.y, pl.color.z);
    c = vscale(c, pl.color.w);
    c = vscale(c, NoL);
    c = vscale(c, visible);
    c = vscale(c, falloff);
    return c;
}

static V3 traceRadiance(const PixelCtx &pc, const Ray &r, uint32_t flags, uint32_t depth, float *distance);

/* ProgressiveRaytracing.hlsl:41-55 */
static inline V3 shootSecondaryRay(const PixelCtx &pc, V3 orig, V3 dir, float minT, uint32_t currentDepth)
{
    if (currentDepth >= pc.rc->max_radiance_depth) return v3(0, 0, 0);
    Ray r = { orig, minT, dir, ORC_RAY_MAX_T };
    pc.st->rays_secondary++;
    float dist;
    return traceRadiance(pc, r, RT_RAY_FLAG_NONE, currentDepth + 1, &dist);
}

/* ProgressiveRaytracing.hlsl:57-78 */
static inline V3 evaluateIndirectDiffuse(const PixelCtx &pc, V3 position, V3 normal, uint32_t *seed, uint32_t currentDepth)
{
    V3 color = v3(0, 0, 0);
    if (pc.rc->pfc.options.cosineHemisphereSampling) {
        V3 sampleDir = getCosHemisphereSample(seed, normal);
        V3 L = shootSecondaryRay(pc, position, sampleDir, ORC_RAY_EPSILON, currentDepth);
        color = vadd(color, vscale(L, ORC_M_PI));
    } else {
        V3 sampleDir = getUniformHemisphereSample(seed, normal);
        float NoL = saturate(dot3(normal, sampleDir));
        float pdf = 1.0f / (2.0f * ORC_M_PI);
        V3 L = shootSecondaryRay(pc, position, sampleDir, ORC_RAY_EPSILON, currentDepth);
        L = vscale(L, NoL);
        L = vdivs(L, pdf);
        color = vadd(color, L);
    }
    return vdivs(color, 1.0f);
}

/* ProgressiveRaytracing.hlsl:80-148 */
static inline V3 shade(const PixelCtx &pc, const rt_material_params &mp, V3 position, V3 normal, V3 rayDir, uint32_t currentDepth)
{
    const rt_debug_options &opt = pc.rc->pfc.options;
    if (opt.showAmbientOcclusionOnly) return evaluateAO(pc, position, normal);

    uint32_t randSeed = initRand(pc.px + pc.py * pc.rc->width, pc.rc->pfc.cameraParams.frameCount);

    V3 directContrib = v3(0, 0, 0);
    if (opt.debug == 2) {
        const float numLights = 2.0f;
        if (nextRand(&randSeed) < 0.5f)
            directContrib = vadd(directContrib, vscale(evaluateDirectionalLight(pc, position, normal, currentDepth), numLights));
        else
            directContrib = vadd(directContrib, vscale(evaluatePointLight(pc, position, normal, currentDepth), numLights));
    } else {
        directContrib = vadd(directContrib, evaluateDirectionalLight(pc, position, normal, currentDepth));
        directContrib = vadd(directContrib, evaluatePointLight(pc, position, normal, currentDepth));
    }

    V3 indirectContrib = v3(0, 0, 0);
    if (currentDepth < 1 && !opt.noIndirectDiffuse)
        indirectContrib = vadd(indirectContrib, evaluateIndirectDiffuse(pc, position, normal, &randSeed, currentDepth));

    V3 diffuseComponent = vdivs(vadd(directContrib, indirectContrib), ORC_M_PI);

    V3 fresnel = v3(0, 0, 0);
    V3 specularComponent = v3(0, 0, 0);
    if (mp.type == 1 || mp.type == 2) {
        if (mp.reflectivity > 0.001f) {
            float exponent = exp_((1.0f - mp.roughness) * 12.0f);
            float pdf, brdf;
            V3 mirrorDir = reflect3(rayDir, normal);
            V3 sampleDir = samplePhongLobe(&randSeed, mirrorDir, exponent, &pdf, &brdf);
            V3 reflectionColor = shootSecondaryRay(pc, position, sampleDir, ORC_RAY_EPSILON, currentDepth);
            V3 s = vscale(reflectionColor, brdf);
            s = vdivs(s, pdf);
            specularComponent = vadd(specularComponent, s);
            fresnel = FresnelReflectanceSchlick(rayDir, normal, v3(mp.specular.x, mp.specular.y, mp.specular.z));
        }
    }

    V3 albedo = v3(mp.albedo.x, mp.albedo.y, mp.albedo.z);
    if (currentDepth == 0) {
        if (opt.showIndirectDiffuseOnly)        return vdivs(vmul(albedo, indirectContrib), ORC_M_PI);
        else if (opt.showIndirectSpecularOnly)  return vmul(vscale(specularComponent, mp.reflectivity), fresnel);
        else if (opt.showFresnelTerm)           return fresnel;
        else if (opt.showGBufferAlbedoOnly)     return albedo;
        else if (opt.showDirectLightingOnly)    return vdivs(vmul(albedo, directContrib), ORC_M_PI);
    }

    V3 r = vscale(v3(mp.emissive.x, mp.emissive.y, mp.emissive.z), mp.emissive.w);
    r = vadd(r, vmul(albedo, diffuseComponent));
    r = vadd(r, vmul(vscale(specularComponent, mp.reflectivity), fresnel));
    return r;
}

/* RaytracingCommon.hlsli:53-82 : normal only (position is computed there but unused) */
static inline V3 interpolateNormal(const Model &m, uint32_t prim, float bu, float bv)
{
    float b0 = 1.0f - bu;
    b0 = b0 - bv;
    const rt_float3 &n0 = m.verts[m.idx[3 * prim + 0]].normal;
    const rt_float3 &n1 = m.verts[m.idx[3 * prim + 1]].normal;
    const rt_float3 &n2 = m.verts[m.idx[3 * prim + 2]].normal;
    V3 n = vscale(v3(n0.x, n0.y, n0.z), b0);
    n = vadd(n, vscale(v3(n1.x, n1.y, n1.z), bu));
    n = vadd(n, vscale(v3(n2.x, n2.y, n2.z), bv));
    return n;
}

/* TraceRay for ray type 0 + PrimaryClosestHit / PrimaryMiss
 * (ProgressiveRaytracing.hlsl:150-164).  depth = payload.depth. */
static V3 traceRadiance(const PixelCtx &pc, const Ray &r, uint32_t flags, uint32_t depth, float *distance)
{
    Hit h = trace(pc, r, flags);
    if (h.inst == RT_NO_HIT) {
        *distance = -1.0f;
        return sampleEnvironment(pc, r.d);
    }
    if (depth == 0) pc.st->primary_hits++; else pc.st->secondary_hits++;
    pc.st->shaded_hits++;
    const Scene &s = *pc.rc->scene;
    const Model &m = s.models[s.inst[h.inst].model];
    V3 n = normalize3(interpolateNormal(m, h.prim, h.u, h.v));
    V3 position = vadd(r.o, vscale(r.d, h.t));
    uint32_t mi = h.inst < pc.rc->nmats ? h.inst : pc.rc->nmats - 1;
    *distance = h.t;
    return shade(pc, pc.rc->mats[mi], position, n, r.d, depth);
}

/* ProgressiveRaytracing.hlsl:11-39 minus the accumulate; returns false on the
 * maxIterations early-out. */
static inline bool rayGen(const PixelCtx &pc, float out[4])
{
    const rt_camera_params &cp = pc.rc->pfc.cameraParams;
    if (cp.accumCount >= pc.rc->pfc.options.maxIterations) return false;
    float dimx = (float)pc.rc->width, dimy = (float)pc.rc->height;
    float dx = ((float)pc.px + 0.5f) / dimx;
    dx = dx * 2.0f;
    dx = dx - 1.0f;
    float dy = ((float)pc.py + 0.5f) / dimy;
    dy = dy * 2.0f;
    dy = dy - 1.0f;
    float jx = cp.jitters.x * 30.0f;
    float jy = cp.jitters.y * 30.0f;
    Ray r;
    r.o = v3(cp.worldEyePos.x + jx, cp.worldEyePos.y + jy, cp.worldEyePos.z + 0.0f);
    float ndy = -dy;
    V3 dir = vscale(v3(cp.U.x, cp.U.y, cp.U.z), dx);
    dir = vadd(dir, vscale(v3(cp.V.x, cp.V.y, cp.V.z), ndy));
    dir = vadd(dir, v3(cp.W.x, cp.W.y, cp.W.z));
    r.d = normalize3(dir);
    r.tmin = 0.0f;
    r.tmax = ORC_RAY_MAX_T;
    pc.st->rays_primary++;
    float dist;
    V3 c = traceRadiance(pc, r, RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES, 0, &dist);
    out[0] = fmax_(c.x, 0.0f);
    out[1] = fmax_(c.y, 0.0f);
    out[2] = fmax_(c.z, 0.0f);
    out[3] = 1.0f;
    return true;
}

/* ProgressiveRaytracing.hlsl:36-38 */
static inline void accumulate(float *px, const float cur[4], uint32_t accumCount, uint32_t mode)
{
    if ((mode & 0xFFu) == RT_ACCUM_SUM) {
        for (int k = 0; k < 4; k++) px[k] = px[k] + cur[k];
        return;
    }
    /* mode bits 8-9: the accumulation is STORED as RGBA16F, as the reference's output texture is (src/DXRExperimentsApp.cpp:28 ->
     * src/ProgressiveRaytracingPipeline.cpp:127-131; read-modify-write at ProgressiveRaytracing.hlsl:36-38): every frame's mean is rounded
     * to fp16 -- 1: to nearest even, 2: toward zero -- before the next frame reads it */
    uint32_t f16 = (mode >> 8) & 3u;
    float n = (float)accumCount;
    float n1 = (float)(accumCount + 1u);
    for (int k = 0; k < 4; k++) {
        float a = n * px[k];
        a = a + cur[k];
        px[k] = a / n1;
        if (f16) px[k] = round_to_half(px[k], f16 == 1u);
    }
}

}  // namespace orc


/* ---------------------------------------------------------------------------
 * "Next" rows N2 / N3: RealtimeRaytracingPipeline and DenoiseCompositor.
 * ------------------------------------------------------------------------- */
namespace orc {

/* assets/shaders/RealtimeRaytracing.hlsl:7-13 */
struct ShadingAOV { V3 albedo; float roughness; V3 directLighting; V3 indirectSpecular; };

static V3 traceRealtime(const PixelCtx &pc, const Ray &r, uint32_t flags, uint32_t depth, ShadingAOV *aov);

/* RealtimeRaytracing.hlsl:48-63 */
static inline V3 shootSecondaryRayRT(const PixelCtx &pc, V3 orig, V3 dir, float minT, uint32_t currentDepth)
{
    if (currentDepth >= pc.rc->max_radiance_depth) return v3(0, 0, 0);
    Ray r = { orig, minT, dir, ORC_RAY_MAX_T };
    pc.st->rays_secondary++;
    ShadingAOV unused;
    return traceRealtime(pc, r, RT_RAY_FLAG_NONE, currentDepth + 1, &unused);
}

/* RealtimeRaytracing.hlsl:65-103 */
static inline V3 shadeAOV(const PixelCtx &pc, const rt_material_params &mp, V3 position, V3 normal, V3 rayDir, uint32_t currentDepth, ShadingAOV *aov)
{
    uint32_t randSeed = initRand(pc.px + pc.py * pc.rc->width, pc.rc->pfc.cameraParams.frameCount);
    V3 directContrib = v3(0, 0, 0);
    directContrib = vadd(directContrib, evaluateDirectionalLight(pc, position, normal, currentDepth));
    directContrib = vadd(directContrib, evaluatePointLight(pc, position, normal, currentDepth));
    V3 fresnel = v3(0, 0, 0);
    V3 specularComponent = v3(0, 0, 0);
    if (mp.type == 1 || mp.type == 2) {
        if (mp.reflectivity > 0.001f) {
            float exponent = exp_((1.0f - mp.roughness) * 12.0f);
            float pdf, brdf;
            V3 mirrorDir = reflect3(rayDir, normal);
            V3 sampleDir = samplePhongLobe(&randSeed, mirrorDir, exponent, &pdf, &brdf);
            V3 reflectionColor = shootSecondaryRayRT(pc, position, sampleDir, ORC_RAY_EPSILON, currentDepth);
            V3 s = vscale(reflectionColor, brdf);
            s = vdivs(s, pdf);
            specularComponent = vadd(specularComponent, s);
            fresnel = FresnelReflectanceSchlick(rayDir, normal, v3(mp.specular.x, mp.specular.y, mp.specular.z));
        }
    }
    V3 albedo = v3(mp.albedo.x, mp.albedo.y, mp.albedo.z);
    V3 direct = vdivs(vmul(albedo, directContrib), ORC_M_PI);
    V3 spec = vmul(vscale(specularComponent, mp.reflectivity), fresnel);
    if (currentDepth == 0) {
        aov->albedo = albedo;
        aov->roughness = mp.roughness;
        aov->directLighting = direct;
        aov->indirectSpecular = spec;
    }
    return vadd(direct, spec);
}

/* PrimaryClosestHit / PrimaryMiss, RealtimeRaytracing.hlsl:105-126 */
static V3 traceRealtime(const PixelCtx &pc, const Ray &r, uint32_t flags, uint32_t depth, ShadingAOV *aov)
{
    Hit h = trace(pc, r, flags);
    if (h.inst == RT_NO_HIT) {
        V3 c = sampleEnvironment(pc, r.d);
        aov->directLighting = c;
        aov->indirectSpecular = v3(0, 0, 0);
        return c;
    }
    if (depth == 0) pc.st->primary_hits++; else pc.st->secondary_hits++;
    pc.st->shaded_hits++;
    const Scene &s = *pc.rc->scene;
    const Model &m = s.models[s.inst[h.inst].model];
    V3 n = normalize3(interpolateNormal(m, h.prim, h.u, h.v));
    V3 position = vadd(r.o, vscale(r.d, h.t));
    uint32_t mi = h.inst < pc.rc->nmats ? h.inst : pc.rc->nmats - 1;
    return shadeAOV(pc, pc.rc->mats[mi], position, n, r.d, depth, aov);
}

/* RayGen, RealtimeRaytracing.hlsl:22-46 (jitter scaled by 10, two AOV outputs, no accumulation) */
static inline void rayGenRealtime(const PixelCtx &pc, float direct[4], float indirect[4])
{
    const rt_camera_params &cp = pc.rc->pfc.cameraParams;
    float dimx = (float)pc.rc->width, dimy = (float)pc.rc->height;
    float dx = ((float)pc.px + 0.5f) / dimx; dx = dx * 2.0f; dx = dx - 1.0f;
    float dy = ((float)pc.py + 0.5f) / dimy; dy = dy * 2.0f; dy = dy - 1.0f;
    float jx = cp.jitters.x * 10.0f, jy = cp.jitters.y * 10.0f;
    Ray r;
    r.o = v3(cp.worldEyePos.x + jx, cp.worldEyePos.y + jy, cp.worldEyePos.z + 0.0f);
    float ndy = -dy;
    V3 dir = vscale(v3(cp.U.x, cp.U.y, cp.U.z), dx);
    dir = vadd(dir, vscale(v3(cp.V.x, cp.V.y, cp.V.z), ndy));
    dir = vadd(dir, v3(cp.W.x, cp.W.y, cp.W.z));
    r.d = normalize3(dir);
    r.tmin = 0.0f;
    r.tmax = ORC_RAY_MAX_T;
    pc.st->rays_primary++;
    ShadingAOV aov;
    aov.directLighting = v3(0, 0, 0); aov.indirectSpecular = v3(0, 0, 0);
    (void)traceRealtime(pc, r, RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES, 0, &aov);
    direct[0] = fmax_(aov.directLighting.x, 0.0f); direct[1] = fmax_(aov.directLighting.y, 0.0f);
    direct[2] = fmax_(aov.directLighting.z, 0.0f); direct[3] = 1.0f;
    indirect[0] = fmax_(aov.indirectSpecular.x, 0.0f); indirect[1] = fmax_(aov.indirectSpecular.y, 0.0f);
    indirect[2] = fmax_(aov.indirectSpecular.z, 0.0f); indirect[3] = 1.0f;
}

/* ---- DenoiseCompositor: assets/shaders/BilateralFilter.hlsli:14-118, DenoiseCommon.hlsli:18-77 ---- */

struct DenoiseParams {          /* cbuffer Params, DenoiseCommon.hlsli:18-26; defaults src/DenoiseCompositor.cpp:44-49 */
    float exposure, gamma;
    uint32_t tonemap, gammaCorrect;
    int32_t maxKernelSize;
    uint32_t debugVisualize;
};

struct Img { const float *p; int w, h; };

/* Texture2D load: out-of-bounds reads return 0 (D3D) */
static inline void texel(const Img &t, int x, int y, float out[4])
{
    if (x < 0 || y < 0 || x >= t.w || y >= t.h) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    const float *s = t.p + ((size_t)y * t.w + x) * 4;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; out[3] = s[3];
}

static inline float denoiseTapWeight(int i, float kernelRadius)
{
    /* BilateralFilter.hlsli:82-90: index into {1,1,.9,.75,.6,.5,0} */
    int a = i < 0 ? -i : i;
    int idx = (int)((float)(a * 5) / (0.001f + fabsf(kernelRadius * 0.8f)));
    idx = idx < 0 ? 0 : (idx > 6 ? 6 : idx);
    return idx < 2 ? 1.0f : (idx < 3 ? 0.9f : (idx < 4 ? 0.75f : (idx < 5 ? 0.6f : (idx < 6 ? 0.5f : 0.0f))));
}

/* filterKernel (BilateralFilter.hlsli:75-118).  The reference reads its taps from a 64+2*20 texel LDS
 * cache whose halo fill has an index-clamp race at cache slot 0; the cache is meant to equal the texture
 * with a zero border, which is what this computes.  |maxKernelSize| <= 20 (the cache's MAX_EXTENT). */
static inline void filterKernel(int pass, int kernelMaxSize, int x, int y, const Img &input, const Img &joint, float out[4])
{
    const int dx = pass == 0 ? 1 : 0, dy = pass == 0 ? 0 : 1;
    const float kernelRadius = (float)kernelMaxSize;
    float color[4] = {0, 0, 0, 0};
    float weight = 0.0f;
    float centerJoint[4];
    texel(joint, x, y, centerJoint);
    for (int i = -kernelMaxSize; i <= kernelMaxSize; ++i) {
        float s[4], sj[4];
        texel(input, x + dx * i, y + dy * i, s);
        texel(joint, x + dx * i, y + dy * i, sj);
        float gaussianWeight = denoiseTapWeight(i, kernelRadius);
        float dist = fabsf(sj[0] - centerJoint[0]);
        dist = dist + fabsf(sj[1] - centerJoint[1]);
        dist = dist + fabsf(sj[2] - centerJoint[2]);
        dist = dist * 10.0f;
        float colorWeight = 1.0f - fmin_(fmax_(dist, 0.0f), 1.0f);
        float bw = gaussianWeight * colorWeight;
        for (int k = 0; k < 4; k++) color[k] = color[k] + s[k] * bw;
        weight = weight + bw;
    }
    for (int k = 0; k < 4; k++) out[k] = color[k] / weight;
}

/* main() of pass 0 (H) and pass 1 (V), DenoiseCommon.hlsli:46-77 */
static inline void denoisePixel(int pass, const DenoiseParams &P, int x, int y, const Img &direct, const Img &input, float out[4])
{
    float c[4];
    if (P.debugVisualize == 2) texel(input, x, y, c);
    else filterKernel(pass, P.maxKernelSize, x, y, input, direct, c);
    if (pass == 1) {
        float d[4];
        texel(direct, x, y, d);
        if (P.debugVisualize == 0) { c[0] = c[0] + d[0]; c[1] = c[1] + d[1]; c[2] = c[2] + d[2]; }
        else if (P.debugVisualize == 3) { c[0] = d[0]; c[1] = d[1]; c[2] = d[2]; }
        c[0] = c[0] * P.exposure; c[1] = c[1] * P.exposure; c[2] = c[2] * P.exposure;
        if (P.tonemap) {
            float lum = c[0] * 0.299f;
            lum = lum + c[1] * 0.587f;
            lum = lum + c[2] * 0.114f;
            float reinhard = lum / (lum + 1.0f);
            float k = reinhard / lum;
            c[0] = fmax_(c[0] * k, 0.0f); c[1] = fmax_(c[1] * k, 0.0f); c[2] = fmax_(c[2] * k, 0.0f);
        }
        if (P.gammaCorrect) {
            float e = 1.0f / P.gamma;
            c[0] = saturate(pow_(c[0], e)); c[1] = saturate(pow_(c[1], e)); c[2] = saturate(pow_(c[2], e));
        }
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = 1.0f;
}

}  // namespace orc

#endif
