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
//
// This file: the stage kernels, the launch sequence of a frame (or of a batch of frames), the render calls and the work
// counters.  rt_shade.h: the shaders as device functions.  rt_pipeline_dev.h: PipeDev and the host object.
// rt_pipeline_host.hip: creation, setters, outputs, checkpoints, timing and statistics.
#include <hip/hip_fp16.h>

#include <array>
#include <new>
#include <utility>

#include "rt_shade.h"

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size);

using namespace rtd;

namespace {

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
            uint32_t ql;
            const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot_frame(pd, q, ql));
            const bool valid = pix_xy(pd, ql, px, py);
            r = primary_ray(pd, pd.pfcs[f].cameraParams, px, py);
            return valid;
        }
        const bool valid = pix_xy(pd, q, px, py);
        r = primary_ray(pd, px, py);
        return valid;
    }
};
typedef PrimarySrcT<true> PrimarySrc;        // (the counting kernels)
template <bool BATCH>
struct PrimarySinkT {
    const PipeDev &pd;
    RT_DEV void store(uint32_t q, const HitD &h, bool) const
    {
        const bool hit = h.inst != RT_NO_HIT;
        pd.lv[0].hit[q] = make_float4(hit ? h.t : HIT_MISS, h.u, h.v, __uint_as_float(h.prim));
        pd.lv[0].inst[q] = h.inst;
    }
};

// (see RT_LDS_STACK_ROWS_SETS: the single-level instantiations with that many stack rows are compiled for seven waves per SIMD,
// the 18-row ones for the six their LDS allows (the any-hit kernel with the shadow cache would take 81 registers otherwise);
// every other instantiation is left to the compiler -- a floor of 1 constrains nothing)
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU __attribute__((amdgpu_waves_per_eu(TWO_LEVEL ? 1 : STACK == RT_LDS_STACK_ROWS_SETS ? 7 : STACK == RT_LDS_STACK_ROWS ? 6 : 1)))
#endif
template <int STACK, bool TWO_LEVEL, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_primary(PipeDev pd)
{
    __shared__ int smem[(STACK + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    // the frame's counters and chunk pools start at zero: its first kernel clears them (nothing here uses them, every later
    // kernel of the frame does) instead of a 20-KB fill launch of its own
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)(POOL_OFFSET_WORDS + POOL_BYTES / 4); i += PBLOCK) pd.counters[i] = 0u;
    PrimarySrcT<BATCH> src = {pd};
    PrimarySinkT<BATCH> sink = {pd};
    // one 8x8 tile per wave, dealt by the hardware dispatcher -- or (experiment) a persistent launch that refills its lanes from a pool of tiles
    trace_wave<STACK, PBLOCK, TWO_LEVEL, 64u>(pd.sc, src, sink, pd.primary_persistent ? pd.pools + POOL_BYTES / 4 : nullptr, smem, nullptr);
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
    auto slot_of = [&](uint32_t idx) -> size_t { return L == 1 ? (size_t)(idx / n) * pd.lv[1].rstride + idx % n : (size_t)idx; };
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
    if (!PRIMARY && !emit_next && pd.shadow_compact && !pd.skip_unlit) {
        // The LAST radiance level (round 4): its hits spawn nothing, so all the emit pass has to leave is where their two light
        // rays start -- P = O + t D, the expression of closest_hit_aov -- and which of the two exist: both
        // (RaytracingCommon.hlsli:126-147 trace them whatever N.L is), or the one shade() draws in the one-light view
        // (ProgressiveRaytracing.hlsl:92-97: the first number of the pixel's sequence).  No normals record, no material, no
        // lobe sample: the resolve pass shades these hits anyway.  (With the reference's depth limits this is every secondary
        // hit: 3.7 M per 1080p frame, the emit kernel 0.065 -> 0.02 ms.)
        const f3 P = r.o + r.d * h.x;
        uint32_t mask = 3u;
        if (pd.kind != RT_PIPELINE_REALTIME && pd.pfc.options.debug == 2) {
            uint32_t seed = init_rand(px + py * pd.width, pd.pfc.cameraParams.frameCount);
            mask = next_rand(seed) < 0.5f ? 1u : 2u;
        }
        if ((uint32_t)L < pd.sh_levels) pd.sh_hits[pd.sh_cbase[L] + idx] = make_float4(P.x, P.y, P.z, __uint_as_float(mask | (frame << 8)));
        return;
    }
    EmitIO io(pd, L, idx, q, frame);
    (void)closest_hit(pd, io, r, h.x, h.y, h.z, __float_as_uint(h.w), pd.lv[L].inst[slot], (uint32_t)L, px + py * pd.width);
    io.finish_shadows(shadow_slots);
    if (emit_next) {
        if (L == 0) {
            for (uint32_t w = 0; w < 2; w++)
                if (!(io.sec_mask & (1u << w))) store_invalid(pd.lv[1].O, pd.lv[1].D, (size_t)w * pd.lv[1].rstride + idx);
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

// a ray queue of `batches` batches of *count rays; batch b lives at [b*stride, b*stride + *count): the radiance rays of a level
// (level 1: the diffuse and the specular batch of the primary hits)
struct QueueSrc {
    const float4 *O, *D;
    const uint32_t *count_ptr;
    uint32_t stride, batches, fl;
    RT_DEV uint32_t n() const { return *count_ptr; }
    RT_DEV uint32_t count() const { return n() * batches; }
    RT_DEV uint32_t flags() const { return fl; }
    // ray i of the queue = ray k of batch b.  At most two batches: a compare against the (wave-uniform) count instead of a
    // division and a remainder -- ~50 vector instructions per ray loaded and per result stored, in the part of the persistent
    // kernels that runs with the fewest lanes (round 4)
    RT_DEV size_t slot(uint32_t i) const
    {
        const uint32_t c = n(), b = i >= c ? 1u : 0u;
        return (size_t)b * stride + (i - b * c);
    }
    RT_DEV bool load(uint32_t i, RayD &r) const
    {
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
    l.point_free = 0.0f;
    return l;
}
static inline LightRays light_rays(const PipeDev &pd) { LightRays l = light_rays(pd.shadow_compact, pd.pfc); l.point_free = pd.point_free; return l; }
static inline LightRays no_light_rays()
{
    LightRays l;
    memset(&l, 0, sizeof l);
    return l;
}

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

// The shadow rays of the frame: ONE queue for the hits of every level (PipeDev::sh_*), ONE persistent any-hit launch.  Ray number i
// -> chunk i >> 6 -> the chunk's hits ((chunk >> log2) << 6 ...) and which of a hit's rays the chunk holds (chunk & (rays per
// hit - 1)): shifts and masks.  Normally the queue is COMPACT ("light rays"): both shadow rays of a shaded hit start at the hit
// point and go to the frame's two lights, so the emit pass stores one float4 per hit -- the point and, in the bits of w, which of
// the two rays exist (bits 0-1), which need not be traversed (bits 2-3) and the hit's frame of the set (bits 8..) -- and the loader
// rebuilds ray b of hit H with the very expressions of evaluateDirectionalLight / evaluatePointLight
// (RaytracingCommon.hlsli:126-147; directional_light / point_light in rt_shade.h): 16 B written and read per hit instead of
// 128 B.  The four rays of the ambient-occlusion view have random directions and keep the explicit origin / direction form.
struct ShadowQueue {
    const float4 *hits, *O, *D;
    uint32_t *vis;
    const uint32_t *nhit;               // counters + C_NHIT: the hits of every level
    uint32_t log2, fl;
    uint32_t lv_first, lv_count;        // the levels whose rays this launch covers (the frame's launch: all; the counting re-walks: 0 / the rest)
    uint32_t cbase[MAXD + 1], hstride[MAXD + 1];      // storage: PipeDev::sh_cbase, LevelDev::hstride
    LightRays lights;                   // the frame's light rays (lights.on: the compact form)
    const LightRays *frame_lights;      // device array [n_frames] (sets of frames)
    ShadowCacheDev cache;
    // the launch enumerates, level after level, round64(hits) rays of kind 0, then of kind 1, ... (wave-uniform: scalar loads and adds)
    RT_DEV uint32_t count() const
    {
        uint32_t b = 0;
#pragma unroll
        for (uint32_t k = 0; k <= (uint32_t)MAXD; k++) if (k >= lv_first && k < lv_first + lv_count) b += (nhit[k] + 63u) & ~63u;
        return b << log2;
    }
    RT_DEV uint32_t flags() const { return fl; }
};
template <bool BATCH>
struct ShadowSrcN : ShadowQueue {
    // Ray i of the launch -> (level, kind of ray b, hit j of the level) by compares against wave-uniform thresholds; `ticket` =
    // the ray's number in storage order, which is where the sink puts its result (the walk carries it in place of i).
    // BATCH: a set of frames -- the lights of frame f (bits 8.. of the hit's word).  The single-frame kernels pass a literal nullptr:
    // the branch folds away
    RT_DEV bool load(uint32_t i, RayD &r, uint32_t &ticket) const
    {
        uint32_t n = 0, R = 0, cb = 0, hs = 0, before = 0;
        ticket = 0;
#pragma unroll
        for (uint32_t k = 0; k <= (uint32_t)MAXD; k++) {
            if (k >= lv_first && k < lv_first + lv_count) {
                const uint32_t nk = nhit[k], Rk = (nk + 63u) & ~63u, start = before;
                before += Rk << log2;
                const bool here = i >= start;             // (the last level that says so wins)
                n = here ? nk : n; R = here ? Rk : R; cb = here ? cbase[k] : cb; hs = here ? hstride[k] : hs;
                if (here) ticket = start;                 // (queue index of the level's first ray, for now)
            }
        }
        const uint32_t local = i - ticket;
        uint32_t b = local >= R ? 1u : 0u;
        if (log2 > 1u) { b += local >= 2u * R ? 1u : 0u; b += local >= 3u * R ? 1u : 0u; }
        const uint32_t j = local - b * R;
        ticket = (cb << log2) + b * hs + j;
        r.o = mk3(0.0f, 0.0f, 0.0f);
        r.d = mk3(0.0f, 0.0f, 0.0f);
        r.tmin = 0.0f;
        r.tmax = -1.0f;                                      // no such ray: never traced
        if (j >= n) { ticket = RT_NO_HIT; return false; }   // (the entries that round a level up to 64: nothing to load, nothing to store)
        if (lights.on) {
            const v4f a = ldg16(hits, (size_t)(cb + j) * 16);
            const uint32_t bits = __float_as_uint(a.w);
            r.o = mk3(a.x, a.y, a.z);
            if (!((bits >> b) & 1u)) return false;
            if ((bits >> (2u + b)) & 1u) { r.tmax = RT_TMAX_SKIPPED; return false; }
            r.tmin = RAY_EPSILON;
            const LightRays *per_frame = BATCH ? frame_lights : nullptr;
            if (per_frame) {                                 // (three floats by hand: a struct copy ends up in scratch memory)
                const float *fl = b == 0u ? per_frame[(bits >> 8) & 0xffu].dir_to_light : per_frame[(bits >> 8) & 0xffu].point_pos;
                const f3 l = mk3(fl[0], fl[1], fl[2]);
                if (b == 0u) { r.d = l; r.tmax = RAY_MAX_T; }
                else {
                    const f3 path = l - r.o;
                    const float dist = length(path);
                    r.d = normalize(path);
                    r.tmax = dist - fmaxf(RAY_EPSILON, fl[3]);       // (fl[3]: point_free of that frame's lights)
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
                r.tmax = dist - fmaxf(RAY_EPSILON, lights.point_free);
            }
            return r.tmax > r.tmin;
        }
        const v4f a = ldg16(O, (size_t)ticket * 16), d = ldg16(D, (size_t)ticket * 16);
        r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
        r.d = mk3(d.x, d.y, d.z); r.tmax = d.w;
        return r.tmax > r.tmin;
    }
    // ---- the shadow cache (ShadowCacheDev): where this ray's entry lives, what it holds, what to put there ----
    RT_DEV uint32_t cache_slot(const RayD &r) const
    {
        const uint32_t res = cache.res;
        if (!(r.tmax < 1.0e37f)) {          // the directional light's rays run to RAY_MAX_T, the point light's to the light
            float u = r.o.x * cache.ua[0] + r.o.y * cache.ua[1] + r.o.z * cache.ua[2] + cache.ua[3];
            float v = r.o.x * cache.va[0] + r.o.y * cache.va[1] + r.o.z * cache.va[2] + cache.va[3];
            u = fminf(fmaxf(u, 0.0f), cache.res_f - 1.0f);
            v = fminf(fmaxf(v, 0.0f), cache.res_f - 1.0f);
            return (uint32_t)u * res + (uint32_t)v;
        }
        const float mx = r.o.x - cache.lp[0], my = r.o.y - cache.lp[1], mz = r.o.z - cache.lp[2];      // from the light to the point
        const float ax = fabsf(mx), ay = fabsf(my), az = fabsf(mz);
        uint32_t face;
        float ma, s, t;
        if (ax >= ay && ax >= az) { face = mx < 0.0f ? 1u : 0u; ma = ax; s = my; t = mz; }
        else if (ay >= az) { face = my < 0.0f ? 3u : 2u; ma = ay; s = mz; t = mx; }
        else { face = mz < 0.0f ? 5u : 4u; ma = az; s = mx; t = my; }
        const float inv = __builtin_amdgcn_rcpf(ma), half = 0.5f * cache.res_f;       // (where an entry lives need not be rounded correctly)
        const float cs = fminf(fmaxf((s * inv * 0.5f + 0.5f) * half, 0.0f), half - 1.0f);
        const float ct = fminf(fmaxf((t * inv * 0.5f + 0.5f) * half, 0.0f), half - 1.0f);
        const uint32_t r2 = res >> 1;
        return res * res + (face * r2 + (uint32_t)cs) * r2 + (uint32_t)ct;
    }
    // (the slot is computed once, when the ray is loaded; the walk keeps it in a spare row of the lane's LDS stack until a hit wants it)
    static constexpr bool has_first_candidates = true;
    template <bool TWO_LEVEL>
    RT_DEV uint32_t cached_leaf(uint32_t ticket, const RayD &r, uint32_t &slot, uint32_t &instance) const
    {
        slot = RT_NO_HIT;
        instance = 0u;
        if (!cache.table) return RT_NO_HIT;
        slot = RT_NO_HIT;
        if (cache.px_base && ticket < 2u * cache.hstride0) {          // a primary hit's ray: its pixel's entry
            const uint32_t b = ticket >= cache.hstride0 ? 1u : 0u;
            uint32_t q = cache.jlist0[ticket - b * cache.hstride0];
            if (cache.n_frames > 1u) q = (__umulhi(q >> 6, cache.frames_magic) << 6) | (q & 63u);      // tile-major slots of a set: chunk = tile * frames + frame
            if (q < cache.px_slots) slot = cache.px_base + 2u * q + b;
        }
        if (slot == RT_NO_HIT) slot = cache_slot(r);
        if (TWO_LEVEL) {
            const uint2 e = ((const uint2 *)cache.table)[slot];
            instance = e.y;
            return e.x;                                  // (checked against its instance's triangle count by the walk)
        }
        const uint32_t t = cache.table[slot];
        return t < cache.n_tris ? t : RT_NO_HIT;         // (an entry is only ever tested if it names a triangle that exists)
    }
    template <bool TWO_LEVEL>
    RT_DEV void remember(uint32_t slot, uint32_t sorted_triangle, uint32_t instance) const
    {
        // (a slot that the stack has overwritten in the meantime is some other number: inside the table it only makes a stale entry)
        if (!cache.table || slot >= cache.entries) return;
        if (TWO_LEVEL) ((uint2 *)cache.table)[slot] = make_uint2(sorted_triangle, instance);
        else cache.table[slot] = sorted_triangle;
    }
};
struct ShadowSinkN {      // ShadowMiss sets visibility 1 (ProgressiveRaytracing.hlsl:178-182)
    ShadowQueue s;
    RT_DEV void store(uint32_t ticket, const HitD &h, bool) const { if (ticket != RT_NO_HIT) s.vis[ticket] = h.inst == RT_NO_HIT ? 1u : 0u; }
};

template <int STACK, bool TWO_LEVEL, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_trace_shadow(SceneDev sc, ShadowQueue queues, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[(STACK + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    ShadowSrcN<BATCH> src;
    static_cast<ShadowQueue &>(src) = queues;
    ShadowSinkN sink = {queues};
    trace_wave<STACK, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, RT_SHADOW_UNORDERED != 0>(sc, src, sink, pool, smem, stat);
}

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_trace_secondary(SceneDev sc, QueueSrc src, float4 *hit1, uint32_t *inst1, uint32_t *pool, uint32_t *stat)
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
__global__ void __launch_bounds__(PBLOCK) k_walk_shadow(SceneDev sc, ShadowQueue queue, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    ShadowSrcN<true> src;
    static_cast<ShadowQueue &>(src) = queue;
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
RT_DEV float4 accumulate(const PipeDev &pd, const float4 prev, const Shaded &sh)
{
    const f3 c = sh.color;
    const float4 cur = make_float4(fmax2(c.x, 0.0f), fmax2(c.y, 0.0f), fmax2(c.z, 0.0f), 1.0f);
    if (pd.accum_mode == RT_ACCUM_SUM) return make_float4(prev.x + cur.x, prev.y + cur.y, prev.z + cur.z, prev.w + cur.w);
    const float n = (float)pd.pfc.cameraParams.accumCount;
    const float n1 = (float)(pd.pfc.cameraParams.accumCount + 1u);
    return make_float4((n * prev.x + cur.x) / n1, (n * prev.y + cur.y) / n1, (n * prev.z + cur.z) / n1, (n * prev.w + cur.w) / n1);
}
RT_DEV void write_aovs(const PipeDev &pd, size_t pixel, const Shaded &sh)          // two AOVs, no accumulation
{
    pd.aov_direct[pixel] = make_float4(fmax2(sh.aov_direct.x, 0.0f), fmax2(sh.aov_direct.y, 0.0f), fmax2(sh.aov_direct.z, 0.0f), 1.0f);
    pd.aov_indirect[pixel] = make_float4(fmax2(sh.aov_indirect.x, 0.0f), fmax2(sh.aov_indirect.y, 0.0f), fmax2(sh.aov_indirect.z, 0.0f), 1.0f);
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
    // (one bounce: the pixel's running value stays in registers from the first frame of the set to the last, resolve -4 %; the
    //  level-by-level form would pay for the four registers with its fifth wave: profiles/r03/resolve_registers.txt)
    const size_t pixel = (size_t)py * pd.width + px;
    const bool progressive = pd.kind != RT_PIPELINE_REALTIME;
#ifndef RT_RESOLVE_REG_ACC
#define RT_RESOLVE_REG_ACC 1
#endif
    constexpr bool KEEP = BATCH && !FLAT && RT_RESOLVE_REG_ACC;
    float4 acc = KEEP && progressive ? pd.accum[pixel] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t f = 0; f < (BATCH ? pd.n_frames : 1u); f++) {
        if (BATCH) pd.pfc = pd.pfcs[f];
        const uint32_t q = frame_slot(pd, f, ql);
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
        if (progressive) { acc = accumulate(pd, KEEP ? acc : pd.accum[pixel], sh); if (!KEEP) pd.accum[pixel] = acc; }
        else write_aovs(pd, pixel, sh);
    }
    if (KEEP && progressive) pd.accum[pixel] = acc;
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
template <class Src>
__global__ void __launch_bounds__(PBLOCK)
k_count_queue(SceneDev sc, Src src, unsigned long long *__restrict__ out)
{
    const uint32_t idx = blockIdx.x * PBLOCK + threadIdx.x;
    unsigned long long rays = 0, nodes = 0, tris = 0;
    if (idx < src.count()) {
        RayD r;
        uint32_t ticket;
        if (load_ray_of(src, idx, r, ticket, 0)) {
            uint32_t cn, ct;
            (void)trace_canonical(sc, r, src.flags(), cn, ct);
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

__global__ void k_debug_cube(PipeDev pd, const float *__restrict__ dirs, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 c = sample_cube(pd, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}


}  // namespace

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
        const uint32_t n = s->two_level ? (uint32_t)s->inst.size() : (s->inst.empty() || !s->inst[0].model ? 0u : s->inst[0].model->n_tris);
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
    if (want < 0) {                         // not set through the API: the environment, else by the size of the triangles
        const char *e = getenv("RT_SHADOW_CACHE_RES");
        want = e ? atoi(e) : -1;
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
    // RT_SHADOW_CACHE_PIXELS=0 / 1 overrides)
    static const int per_pixel_env = getenv("RT_SHADOW_CACHE_PIXELS") ? atoi(getenv("RT_SHADOW_CACHE_PIXELS")) : -1;
    const bool per_pixel = per_pixel_env < 0 ? s->two_level : per_pixel_env != 0;
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
    c.n_tris = s->two_level ? 0u : s->inst[0].model->n_tris;
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

// ---- queue memory ------------------------------------------------------------------------------------------------------
// Level l keeps, per RAY slot (l = 0: per pixel slot; no ray is stored there), the ray (32 B + 4 B pixel slot), its hit record
// (16 + 4 B) and the two compaction maps (4 + 4 B); per HIT of the level its shadow rays -- compact form: ONE float4 per hit +
// 4 B of visibility per shadow ray; explicit form (the ambient-occlusion view): 36 B per shadow ray; all levels in ONE queue -- and, for paths of more
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
int grow(DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return RT_OK;
    return b.reserve(bytes + bytes / 8);
}
// ray queue + hit records of level l for `slots` ray slots (level 0: pixel slots)
int reserve_level_rays(rt_pipeline *p, uint32_t l, size_t slots, bool deep)
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
int grow_keep(DevBuf &b, size_t bytes, size_t keep_bytes, hipStream_t st)
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
int reserve_shadows(rt_pipeline *p, size_t hits, uint32_t log2, bool compact, size_t keep_hits)
{
    hipStream_t st = p->ctx->stream;
    if (compact) RT_TRY(grow_keep(p->sh_hits, hits * 16, keep_hits * 16, st));
    else { RT_TRY(grow_keep(p->sh_O, (hits << log2) * 16, (keep_hits << log2) * 16, st)); RT_TRY(grow_keep(p->sh_D, (hits << log2) * 16, (keep_hits << log2) * 16, st)); }
    return grow_keep(p->sh_vis, (hits << log2) * 4, 0, st);         // (results: nothing is in there before the shadow launch)
}
// levels 0 .. n - 1 cast shadow rays: level 0 always has its entries (their masks are empty when no shadow ray is allowed at all)
inline uint32_t shadow_levels(uint32_t levels, uint32_t max_shadow) { return 1u + (max_shadow > 1u ? (levels < max_shadow - 1u ? levels : max_shadow - 1u) : 0u); }
inline size_t level_ray_bytes(uint32_t l, bool deep) { return (l > 0 ? 36u : 0u) + 28u + (l > 0 && deep ? 16u : 0u); }
inline size_t level_shadow_bytes(uint32_t shadow_slots, bool compact) { return compact ? 16u + 4u * shadow_slots : 36u * shadow_slots; }
size_t worst_case_queue_bytes(size_t cap, uint32_t levels, uint32_t max_shadow, uint32_t shadow_slots0, bool compact)
{
    const bool deep = levels > 1;
    size_t total = cap * level_ray_bytes(0, deep);
    for (uint32_t l = 1; l <= levels; l++) total += 2 * cap * level_ray_bytes(l, deep);
    return total + (cap + 2 * cap * (shadow_levels(levels, max_shadow) - 1u)) * level_shadow_bytes(shadow_slots0, compact);
}
size_t queue_budget(rt_pipeline *p)
{
    if (p->queue_budget) return p->queue_budget;
    static const char *const env = getenv("RT_QUEUE_BUDGET_MB");
    if (env) { const long long mb = atoll(env); if (mb > 0) return (size_t)mb << 20; }
    if (p->ctx->device_mem_total == 0) {         // asked once per context: hipMemGetInfo is a driver round trip, this runs per frame
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
        p->ctx->device_mem_total = total_b;
    }
    return p->ctx->device_mem_total / 4;
}
// everything a set of `cap` pixel slots can need at most, reserved now; the strides that go with it
int reserve_worst_case(rt_pipeline *p, size_t cap, uint32_t levels, uint32_t max_shadow, uint32_t shadow_slots0, bool compact)
{
    const bool deep = levels > 1;
    RT_TRY(reserve_level_rays(p, 0, cap, deep));
    for (uint32_t l = 1; l <= levels; l++) RT_TRY(reserve_level_rays(p, l, 2 * cap, deep));
    return reserve_shadows(p, cap + 2 * cap * (shadow_levels(levels, max_shadow) - 1u), shadow_slots0 > 2u ? 2u : 1u, compact, 0);
}
void bind_level(const rt_pipeline *p, PipeDev &pd, int l)
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

// The launches of one frame, or of one set of frames.  `counted`: the levels beyond the pixel slots are sized by what the
// compaction before them has counted (see "queue memory" above); otherwise the caller has reserved the worst case and set the
// strides.  pd is updated as the levels are bound and is what the counting re-walks replay (rt_pipeline::last_pd).
template <int STACK, bool TWO_LEVEL>
int launch_frame(rt_pipeline *p, PipeDev &pd, uint32_t shadow_slots, bool counted)
{
    hipError_t first_error = hipSuccess;
    auto record = [&](hipEvent_t e, hipStream_t s) { const hipError_t rc = hipEventRecord(e, s); if (first_error == hipSuccess) first_error = rc; };
    hipStream_t st = p->ctx->stream;
    const bool T = p->ring_frames > 0;
    const size_t ring_slot = T ? (size_t)(p->ring_pos % (uint64_t)p->ring_frames) : 0;
    hipEvent_t *ev = T ? &p->ring[ring_slot * EV_COUNT] : nullptr;
    const uint32_t cap = pd.cap;
    rt_context *ctx = p->ctx;
    const uint32_t levels = frame_levels(p);
    const bool deep = levels > 1, compact = pd.shadow_compact != 0;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    // counted queues: the hits the compaction of level l has just produced -> the shadow queue of level l and the ray queue of
    // level l + 1 get their sizes; `slots` = ray slots of the level whose launches come next (level 1: two batches)
    size_t hits_l = cap, slots = cap;
    size_t sh_total = 0;                        // entries of the shared shadow queue taken by the levels sized so far
    auto size_next = [&](uint32_t l, bool casts_shadows, bool spawns) -> int {
        if (counted) {
            HIP_TRY(hipMemcpyAsync(ctx->pinned, &pd.counters[C_NHIT + l], 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            hits_l = round64((size_t)ctx->pinned[0]);
            if (hits_l == 0) hits_l = 64;
        } else hits_l = l == 0 ? (size_t)cap : 2 * (size_t)cap;
        if (hits_l > 0xffffffc0ull / 2) { rt_set_error("render: more than 2^31 hits at radiance level %u", l); return RT_ERR_UNSUPPORTED; }
        pd.lv[l].hstride = (uint32_t)hits_l;
        if (casts_shadows) {
            if (((sh_total + hits_l) << pd.sh_log2) >= 0xffffffc0ull) { rt_set_error("render: more than 2^32 shadow rays in one set of launches"); return RT_ERR_UNSUPPORTED; }
            if (counted) RT_TRY(reserve_shadows(p, sh_total + hits_l, pd.sh_log2, compact, sh_total));
            pd.sh_cbase[l] = (uint32_t)sh_total;
            sh_total += hits_l;
        }
        bind_level(p, pd, (int)l);
        if (spawns) {
            slots = (l == 0 ? 2 : 1) * hits_l;
            if (l == 0) pd.lv[1].rstride = (uint32_t)hits_l;
            if (counted) RT_TRY(reserve_level_rays(p, l + 1, slots, deep));
            bind_level(p, pd, (int)l + 1);
        }
        return RT_OK;
    };
    if (T) record(ev[0], st);
    // primary rays are coherent: one 8x8 tile per wave, scheduled by the hardware dispatcher
    if (pd.primary_persistent) HIP_TRY(hipMemsetAsync(pd.pools + POOL_BYTES / 4, 0, PRIMARY_POOL_WORDS * 4, st));      // (its own first block cannot clear it: the others already draw from it)
    if (pd.n_frames > 1u) k_primary<STACK, TWO_LEVEL, true><<<pd.primary_persistent ? rt_persistent_grid(ctx, k_primary<STACK, TWO_LEVEL, true>, PBLOCK, cap) : blocks(cap), PBLOCK, 0, st>>>(pd);
    else k_primary<STACK, TWO_LEVEL, false><<<pd.primary_persistent ? rt_persistent_grid(ctx, k_primary<STACK, TWO_LEVEL, false>, PBLOCK, cap) : blocks(cap), PBLOCK, 0, st>>>(pd);
    k_compact_level<<<(cap + CTILES * CBLOCK - 1) / (CTILES * CBLOCK), CBLOCK, 0, st>>>(pd, 0);
    if (T) record(ev[1], st);
    RT_TRY(size_next(0, true, levels >= 1));
    const bool B = pd.n_frames > 1u;            // a batch of frames: the shading kernels pick the constants of every hit's frame
    if (B) k_shade_emit<true, true><<<blocks(hits_l), PBLOCK, 0, st>>>(pd, 0, shadow_slots, levels >= 1 ? 1u : 0u);
    else k_shade_emit<true, false><<<blocks(hits_l), PBLOCK, 0, st>>>(pd, 0, shadow_slots, levels >= 1 ? 1u : 0u);
    if (T) record(ev[2], st);
    const LightRays lr = light_rays(pd);
    for (uint32_t l = 1; l <= levels; l++) {
        // level 1: the diffuse and the specular batch of the primary hits; deeper: one ray per hit of level l-1
        const QueueSrc rays = {pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], pd.lv[1].rstride, l == 1 ? 2u : 1u, RT_RAY_FLAG_NONE};   // ProgressiveRaytracing.hlsl:53
        k_trace_secondary<STACK, TWO_LEVEL><<<rt_persistent_grid(ctx, k_trace_secondary<STACK, TWO_LEVEL>, PBLOCK, slots), PBLOCK, 0, st>>>(
            pd.sc, rays, pd.lv[l].hit, pd.lv[l].inst, pd.pools + (size_t)l * RT_POOL_GROUPS * RT_POOL_STRIDE, &pd.counters[C_SECONDARY]);
        k_compact_level<<<(unsigned)((slots + CTILES * CBLOCK - 1) / (CTILES * CBLOCK)), CBLOCK, 0, st>>>(pd, (int)l);
        if (T) record(ev[3 + 2 * (l - 1)], st);
        const bool casts_shadows = l < pd.sh_levels, spawns = l < levels;
        RT_TRY(size_next(l, casts_shadows, spawns));
        if (casts_shadows || spawns) {
            if (B) k_shade_emit<false, true><<<blocks(hits_l), PBLOCK, 0, st>>>(pd, (int)l, 2u, spawns ? 1u : 0u);
            else k_shade_emit<false, false><<<blocks(hits_l), PBLOCK, 0, st>>>(pd, (int)l, 2u, spawns ? 1u : 0u);
        }
        if (T) record(ev[4 + 2 * (l - 1)], st);
    }
    {   // every shadow ray of the frame (or set) in ONE persistent any-hit launch over the shared queue
        ShadowQueue sq;
        memset(&sq, 0, sizeof sq);
        sq.hits = pd.sh_hits; sq.O = pd.sh_O; sq.D = pd.sh_D; sq.vis = pd.sh_vis;
        sq.nhit = &pd.counters[C_NHIT];
        sq.log2 = pd.sh_log2; sq.fl = any;                                   // RaytracingCommon.hlsli:94
        sq.lv_first = 0; sq.lv_count = pd.sh_levels;
        for (int k = 0; k <= MAXD; k++) { sq.cbase[k] = pd.sh_cbase[k]; sq.hstride[k] = pd.lv[k].hstride; }
        sq.lights = lr;
        sq.frame_lights = B ? pd.frame_lights : nullptr;
        sq.cache = p->shadow_cache_dev;
        sq.cache.jlist0 = pd.lv[0].jlist;
        sq.cache.hstride0 = pd.lv[0].hstride;
        sq.cache.n_frames = pd.n_frames;
        sq.cache.frames_magic = (uint32_t)(0x100000000ull / (pd.n_frames ? pd.n_frames : 1u)) + 1u;
        if (sq.cache.px_slots > pd.fcap) sq.cache.px_slots = pd.fcap;
        const size_t rays_max = sh_total << pd.sh_log2;
        if (B) k_trace_shadow<STACK, TWO_LEVEL, true><<<rt_persistent_grid(ctx, k_trace_shadow<STACK, TWO_LEVEL, true>, PBLOCK, rays_max), PBLOCK, 0, st>>>(
            pd.sc, sq, pd.pools, &pd.counters[C_SHADOW]);
        else k_trace_shadow<STACK, TWO_LEVEL, false><<<rt_persistent_grid(ctx, k_trace_shadow<STACK, TWO_LEVEL, false>, PBLOCK, rays_max), PBLOCK, 0, st>>>(
            pd.sc, sq, pd.pools, &pd.counters[C_SHADOW]);
    }
    if (T) record(ev[EV_SHADOW], st);
    // (resolve: one thread per pixel slot of ONE frame; a batch's frames are accumulated in order inside the thread)
    if (levels <= 1) {                          // (level by level is slower here: 0.143 vs 0.118 ms at 1080p)
        if (B) k_resolve<false, true><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
        else k_resolve<false, false><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
    } else {
        for (uint32_t l = levels; l >= 1; l--) {
            const size_t n_l = pd.lv[l].hstride;       // (one thread per hit of the level)
            if (B) k_shade_level<true><<<blocks(n_l), PBLOCK, 0, st>>>(pd, (int)l);
            else k_shade_level<false><<<blocks(n_l), PBLOCK, 0, st>>>(pd, (int)l);
        }
        if (B) k_resolve<true, true><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
        else k_resolve<true, false><<<blocks(pd.fcap), PBLOCK, 0, st>>>(pd);
    }
    if (T) { record(ev[EV_RESOLVE], st); p->ring_levels[ring_slot] = (uint8_t)levels; p->ring_nframes[ring_slot] = (uint8_t)pd.n_frames; p->ring_pos++; }
    HIP_TRY(first_error);
    return RT_OK;
}

template <int STACK>
int launch_frame_any(rt_pipeline *p, PipeDev &pd, uint32_t shadow_slots, bool counted)
{
    return p->scene->two_level ? launch_frame<STACK, true>(p, pd, shadow_slots, counted) : launch_frame<STACK, false>(p, pd, shadow_slots, counted);
}

template <bool TWO_LEVEL>
static int count_walk_launch(rt_pipeline *p, unsigned long long *w)
{
    hipStream_t st = p->ctx->stream;
    const rt_context *ctx = p->ctx;
    const PipeDev &pd = p->last_pd;
    const uint32_t cap = pd.cap;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    k_walk_primary<TWO_LEVEL><<<rt_persistent_grid(ctx, k_walk_primary<TWO_LEVEL>, PBLOCK, cap), PBLOCK, 0, st>>>(pd, w + RT_WALK_WORDS * RT_STAGE_PRIMARY);
    const unsigned gq = rt_persistent_grid(ctx, k_walk_queue<TWO_LEVEL>, PBLOCK, (size_t)cap * 2);
    const unsigned gs = rt_persistent_grid(ctx, k_walk_shadow<TWO_LEVEL>, PBLOCK, (size_t)cap * 2);
    ShadowQueue sq;
    memset(&sq, 0, sizeof sq);
    sq.hits = pd.sh_hits; sq.O = pd.sh_O; sq.D = pd.sh_D; sq.vis = pd.sh_vis;
    sq.nhit = &pd.counters[C_NHIT];
    sq.log2 = pd.sh_log2; sq.fl = any;
    sq.lights = light_rays(pd);
    sq.frame_lights = pd.n_frames > 1u ? pd.frame_lights : nullptr;
    for (int k = 0; k <= MAXD; k++) { sq.cbase[k] = pd.sh_cbase[k]; sq.hstride[k] = pd.lv[k].hstride; }
    sq.lv_first = 0; sq.lv_count = 1;                       // the shadow rays of the primary hits ...
    k_walk_shadow<TWO_LEVEL><<<gs, PBLOCK, 0, st>>>(pd.sc, sq, w + RT_WALK_WORDS * RT_STAGE_SHADOW0);
    const uint32_t levels = pd.max_rad < (uint32_t)MAXD ? pd.max_rad : (uint32_t)MAXD;
    for (uint32_t l = 1; l <= levels; l++)
        k_walk_queue<TWO_LEVEL><<<gq, PBLOCK, 0, st>>>(pd.sc, QueueSrc{pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], pd.lv[1].rstride, l == 1 ? 2u : 1u, RT_RAY_FLAG_NONE},
                                                       w + RT_WALK_WORDS * RT_STAGE_SECONDARY);
    if (pd.sh_levels > 1u) {                                // ... and of all deeper hits
        sq.lv_first = 1; sq.lv_count = pd.sh_levels - 1u;
        k_walk_shadow<TWO_LEVEL><<<gs, PBLOCK, 0, st>>>(pd.sc, sq, w + RT_WALK_WORDS * RT_STAGE_SHADOW1);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
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
    static const bool seven_waves_always = getenv("RT_SEVEN_WAVES_ALWAYS") && atoi(getenv("RT_SEVEN_WAVES_ALWAYS")) != 0;      // (experiment: single frames on the sets' kernels)
    const bool set_rows = (n_frames > 1 || seven_waves_always) && !p->scene->two_level && ctx->lds_stack_rows != RT_LDS_STACK_ROWS_TEST && RT_LDS_STACK_ROWS_SETS != RT_LDS_STACK_ROWS;
    RT_TRY(rt_scene_dev_for_launch(ctx, p->scene, set_rows ? RT_LDS_STACK_ROWS_SETS : rt_lds_stack_rows(ctx), cap > ctx->cu_count * 16u * PBLOCK ? cap : ctx->cu_count * 16u * PBLOCK, &pd.sc));
    pd.pfc = frames[0];
    pd.n_frames = n_frames; pd.fcap = fcap;
    pd.pfcs = nullptr; pd.frame_lights = nullptr;
    pd.shadow_compact = ao_view ? 0u : 1u;          // the AO view's four rays have random directions
    // (the light buffer is keyed by the two lights: the AO view's random rays do not use it)
    const bool free_on = !(getenv("RT_FREE_RADIUS") && atoi(getenv("RT_FREE_RADIUS")) == 0);
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
    pd.skip_unlit = p->skip_unlit;
    pd.kind = p->kind;
    // The primary stage as a persistent launch that refills its lanes from a pool of tiles (instead of one tile per wave, dealt by the
    // hardware): pays where the rays of a tile part ways early -- two-level scenes, whose primary waves run at 0.48 of their lanes
    // (4096 instances at 4K: 1.39 -> 1.31 ms) -- and costs 13 % where they stay together (the single-level bench scene: 0.69 of the lanes
    // as it is).  RT_PRIMARY_PERSISTENT=0 / 1 overrides.  profiles/r04/c4_variants.txt, primary_persistent.txt
    static const int primary_persistent_env = getenv("RT_PRIMARY_PERSISTENT") ? atoi(getenv("RT_PRIMARY_PERSISTENT")) : -1;
    pd.primary_persistent = (primary_persistent_env < 0 ? p->scene->two_level : primary_persistent_env != 0) ? 1u : 0u;
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
    // 18 LDS stack rows + the 8-row top table = 26 KiB per 256-thread block = 6 resident blocks per CU, whatever
    // the depth of the tree; the rare deeper walk continues in global rows (rt_trace_wave.h)
    int launched;
    if (ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST) launched = launch_frame_any<RT_LDS_STACK_ROWS_TEST>(p, pd, shadow_slots, counted);
    else if (set_rows) launched = launch_frame<RT_LDS_STACK_ROWS_SETS, false>(p, pd, shadow_slots, counted);
    else launched = launch_frame_any<RT_LDS_STACK_ROWS>(p, pd, shadow_slots, counted);
    RT_TRY(launched);
    HIP_TRY(hipGetLastError());
    p->last_pd = pd;
    p->last_shadow_slots = shadow_slots;
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
    if (const char *e = getenv("RT_BATCH_MAX")) { const int v = atoi(e); if (v >= 1 && v <= (int)RT_MAX_BATCH) batch_max = (uint32_t)v; }
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

int rt_pipeline_flush_pending(rt_pipeline *p)
{
    if (!p || p->pending.empty()) return RT_OK;
    std::vector<rt_per_frame_constants> frames;
    frames.swap(p->pending);                    // (whatever happens, the frames are not rendered twice)
    std::vector<rt_pipeline *> &reg = p->ctx->deferred;
    for (size_t k = 0; k < reg.size(); k++) if (reg[k] == p) { reg.erase(reg.begin() + (long)k); break; }
    const rt_per_frame_constants keep = p->pfc;              // the constants of the last update(): a frame may have been updated and not rendered yet
    const int rc = render_frames(p, p->width, p->height, frames.data(), (uint32_t)frames.size(), 0, 0, 1);
    p->pfc = keep;
    return rc;
}

int rt_context_flush_deferred(rt_context *ctx)
{
    if (!ctx) return RT_OK;
    while (!ctx->deferred.empty()) RT_TRY(rt_pipeline_flush_pending(ctx->deferred.back()));      // (a flush takes the pipeline off the list)
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
    if (p->counters.bytes < POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4) {
        RT_TRY(p->counters.reserve(POOL_OFFSET_WORDS * 4 + POOL_BYTES + PRIMARY_POOL_WORDS * 4));
        HIP_TRY(hipMemsetAsync(p->counters.p, 0, p->counters.bytes, p->ctx->stream));
    }
    if (worst_case_queue_bytes(cap, levels_now, p->max_shadow, ao_view ? 4u : 2u, !ao_view) > queue_budget(p)) RT_TRY(reserve_level_rays(p, 0, cap, levels_now > 1));
    else RT_TRY(reserve_worst_case(p, cap, levels_now, p->max_shadow, ao_view ? 4u : 2u, !ao_view));
    if (frames > 1) RT_TRY(p->batch_consts.reserve((sizeof(rt_per_frame_constants) + sizeof(LightRays)) * RT_MAX_BATCH));
    if (p->scene) {                                    // the shadow cache's table as well
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
    const PipeDev &pd = p->last_pd;
    const uint32_t cap = pd.cap;
    const uint32_t any = RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH | RT_RAY_FLAG_SKIP_CLOSEST_HIT_SHADER;
    k_count_primary<<<blocks(cap), PBLOCK, 0, st>>>(pd, w + 3 * RT_STAGE_PRIMARY);
    ShadowSrcN<true> sq;
    memset(&sq, 0, sizeof sq);
    sq.hits = pd.sh_hits; sq.O = pd.sh_O; sq.D = pd.sh_D; sq.vis = pd.sh_vis;
    sq.nhit = &pd.counters[C_NHIT];
    sq.log2 = pd.sh_log2; sq.fl = any;
    sq.lights = light_rays(pd);
    sq.frame_lights = pd.n_frames > 1u ? pd.frame_lights : nullptr;
    for (int k = 0; k <= MAXD; k++) { sq.cbase[k] = pd.sh_cbase[k]; sq.hstride[k] = pd.lv[k].hstride; }
    sq.lv_first = 0; sq.lv_count = 1;
    k_count_queue<<<blocks((size_t)pd.lv[0].hstride << pd.sh_log2), PBLOCK, 0, st>>>(pd.sc, sq, w + 3 * RT_STAGE_SHADOW0);
    const uint32_t levels = pd.max_rad < (uint32_t)MAXD ? pd.max_rad : (uint32_t)MAXD;
    size_t deeper = 0;
    for (uint32_t l = 1; l <= levels; l++) {        // every secondary level adds into the same row
        const uint32_t batches = l == 1 ? 2u : 1u;
        const size_t slots = l == 1 ? 2 * (size_t)pd.lv[1].rstride : (size_t)pd.lv[l - 1].hstride;
        k_count_queue<<<blocks(slots), PBLOCK, 0, st>>>(pd.sc, QueueSrc{pd.lv[l].O, pd.lv[l].D, &pd.counters[C_NHIT + l - 1], pd.lv[1].rstride, batches, RT_RAY_FLAG_NONE},
                                                        w + 3 * RT_STAGE_SECONDARY);
        if (l < pd.sh_levels) deeper += pd.lv[l].hstride;
    }
    if (pd.sh_levels > 1u) {
        sq.lv_first = 1; sq.lv_count = pd.sh_levels - 1u;
        k_count_queue<<<blocks(deeper << pd.sh_log2), PBLOCK, 0, st>>>(pd.sc, sq, w + 3 * RT_STAGE_SHADOW1);
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
    RT_TRY(rt_pipeline_flush_pending(p));
    if (!p->rendered || !p->scene->built || p->scene->generation != p->last_scene_gen) { rt_set_error("count_walk: nothing rendered since the last change of scene, materials or output"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t st = p->ctx->stream;
    const size_t bytes = RT_STAGE_COUNT * RT_WALK_WORDS * sizeof(unsigned long long);
    RT_TRY(p->work.reserve(bytes));
    HIP_TRY(hipMemsetAsync(p->work.p, 0, bytes, st));
    unsigned long long *w = p->work.as<unsigned long long>();
    if (p->scene->two_level) RT_TRY(count_walk_launch<true>(p, w));
    else RT_TRY(count_walk_launch<false>(p, w));
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
    k_debug_cube<<<blocks(n), PBLOCK, 0, ctx->stream>>>(pd, sb[1].as<float>(), sb[2].as<float>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, sb[2].p, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

}  // extern "C"
