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
// This file: the stage kernels, the launch sequence of a frame (or of a set of frames) and the launches of the counting re-walks.
// rt_shade.h: the shaders as device functions.  rt_pipeline_dev.h: PipeDev and the host object.  rt_pipeline_queues.h: queue memory.
// rt_pipeline_render.hip: the render calls (regions, sets of frames, deferred mode), the shadow cache's and the free sphere's host side.
// rt_pipeline_host.hip: creation, setters, outputs, checkpoints, timing and statistics.
#include <hip/hip_fp16.h>

#include <array>
#include <new>
#include <utility>

#include "rt_shade.h"
#include "rt_trace_repack.h"
#include "rt_pipeline_queues.h"

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
    // (NO_DEEP launches, rt_trace_wave.h) the walk of pixel slot q would need a stack row beyond the LDS ones: k_primary_retry takes it from here
    RT_DEV void overflow(uint32_t q) const
    {
        const uint32_t k = atomicAdd(&pd.retry[0], 1u);
        if (k < pd.retry_cap) pd.retry[2u + k] = q;
    }
};
typedef PrimarySrcT<true> PrimarySrc;        // (the counting kernels)
// The pixel slots the one-tile-per-wave primary launch gave up (PipeDev::retry), as a ray source for a persistent launch WITH rows beyond
// LDS; a list that overflowed stands for every slot.  Slots come in any order here: every lane finds its own frame.
struct PrimaryRetrySrc {
    const PipeDev &pd;
    RT_DEV uint32_t count() const { const uint32_t n = pd.retry[0]; return n <= pd.retry_cap ? n : pd.cap; }
    RT_DEV uint32_t flags() const { return RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES; }
    RT_DEV bool load(uint32_t i, RayD &r, uint32_t &ticket) const
    {
        const uint32_t q = pd.retry[0] <= pd.retry_cap ? pd.retry[2u + i] : i;
        ticket = q;
        uint32_t px, py, ql;
        const uint32_t f = slot_frame(pd, q, ql);
        const bool valid = pix_xy(pd, ql, px, py);
        r = pd.n_frames > 1u ? primary_ray(pd, pd.pfcs[f].cameraParams, px, py) : primary_ray(pd, px, py);
        return valid;
    }
};
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
// the two-level ones for five -- they take 88 - 96 registers by themselves, except the primary stage of a set (99: a fourth wave lost
// for three registers; capped it fits 96 without scratch))
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU __attribute__((amdgpu_waves_per_eu(TWO_LEVEL ? 5 : RT_ROWS(STACK) == RT_LDS_STACK_ROWS_SETS ? 7 : RT_ROWS(STACK) == RT_LDS_STACK_ROWS ? 6 : 1)))
#endif
// PERSIST = false: one 8x8 tile per wave, dealt by the hardware dispatcher -- one thread per pixel slot, so this launch keeps NO stack rows
// beyond LDS (NO_DEEP: a ray that would need one goes to PipeDev::retry and k_primary_retry walks it); true: a persistent launch that refills
// its lanes from a pool of tiles (two-level scenes), with rows by resident thread like the other persistent launches
template <int STACK, bool TWO_LEVEL, bool BATCH, bool PERSIST>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_primary(PipeDev pd)
{
    __shared__ int smem[(RT_ROWS(STACK) + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    // the frame's counters and chunk pools start at zero: its first kernel clears them (nothing here uses them, every later
    // kernel of the frame does) instead of a 20-KB fill launch of its own
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < (uint32_t)(POOL_OFFSET_WORDS + POOL_BYTES / 4); i += PBLOCK) pd.counters[i] = 0u;
    PrimarySrcT<BATCH> src = {pd};
    PrimarySinkT<BATCH> sink = {pd};
    if (PERSIST) trace_wave<RT_ROWS(STACK), PBLOCK, TWO_LEVEL, 64u, false, false, false, RT_REFS(STACK)>(pd.sc, src, sink, pd.pools + POOL_BYTES / 4, smem, nullptr);
    else trace_wave<RT_ROWS(STACK), PBLOCK, TWO_LEVEL, 64u, false, false, true, RT_REFS(STACK)>(pd.sc, src, sink, nullptr, smem, nullptr);
}

// the pixel slots k_primary<.., PERSIST = false> gave up, walked from the start by a persistent launch with rows beyond LDS (static deal of
// the list: there is next to nothing on it -- on the bench scenes nothing: the launch returns before it touches its LDS)
template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_primary_retry(PipeDev pd)
{
    __shared__ int smem[(RT_ROWS(STACK) + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    if (pd.retry[0] == 0u) return;
    PrimaryRetrySrc src = {pd};
    PrimarySinkT<true> sink = {pd};
    trace_wave<RT_ROWS(STACK), PBLOCK, TWO_LEVEL, 64u, false, false, false, RT_REFS(STACK)>(pd.sc, src, sink, nullptr, smem, nullptr);
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
    if (L == 0 && pd.retry && blockIdx.x == 0 && threadIdx.x == 0) { pd.retry[1] = pd.retry[0]; pd.retry[0] = 0u; }      // (the retry launch in front of this one has used it;
                                                                                                                           //  [1]: the set's count, for the host's choice of primary launch)
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
    // Round 6: ONE BIT per shadow ray (bit ticket & 31 of word ticket >> 5, cleared before the launch), set by a no-return atomic when the ray
    // reaches its light.  A 4-byte store per ray cost 22 B of HBM writes per ray (scattered partial lines: 4.4 GB per 20-frame launch,
    // profiles/r05/c2s_write.md) to deliver that bit.  The 32 tickets of a word are consecutive rays of one (level, light) run and such runs
    // start at multiples of 64, so the word belongs to ONE chunk of the queue = one wave = one XCD's L2: no two L2s ever update the same word.
    RT_DEV void store(uint32_t ticket, const HitD &h, bool) const { if (ticket != RT_NO_HIT && h.inst == RT_NO_HIT) atomicOr(&s.vis[ticket >> 5], 1u << (ticket & 31u)); }
};

template <int STACK, bool TWO_LEVEL, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_trace_shadow(SceneDev sc, ShadowQueue queues, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[(RT_ROWS(STACK) + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    ShadowSrcN<BATCH> src;
    static_cast<ShadowQueue &>(src) = queues;
    ShadowSinkN sink = {queues};
    trace_wave<RT_ROWS(STACK), PBLOCK, TWO_LEVEL, (!TWO_LEVEL && RT_ROWS(STACK) == RT_LDS_STACK_ROWS_SETS) ? RT_POOL_CHUNK_SETS_ANYHIT : RT_POOL_CHUNK, RT_SHADOW_UNORDERED != 0, false, false, RT_REFS(STACK)>(sc, src, sink, pool, smem, stat);
}

// the shadow stage on the re-packed engine (rt_trace_repack.h; single-level scenes, option repack=1): one stack row less than the launch
// it replaces, so that the rings fit the same LDS budget
template <int STACK, bool BATCH>
__global__ void __launch_bounds__(PBLOCK) __attribute__((amdgpu_waves_per_eu(RT_ROWS(STACK) == RT_LDS_STACK_ROWS_SETS ? 7 : RT_ROWS(STACK) == RT_LDS_STACK_ROWS ? 6 : 1)))
k_trace_shadow_rp(SceneDev sc, ShadowQueue queues, uint32_t *pool, uint32_t *stat, char *records)
{
    constexpr int ROWS = RT_ROWS(STACK) - 1;
    __shared__ int smem[(ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK + RT_RP_LDS_EXTRA_INTS];
    ShadowSrcN<BATCH> src;
    static_cast<ShadowQueue &>(src) = queues;
    ShadowSinkN sink = {queues};
    trace_wave_repack<ROWS, PBLOCK, RT_ROWS(STACK) == RT_LDS_STACK_ROWS_SETS ? RT_POOL_CHUNK_SETS : RT_POOL_CHUNK, RT_REFS(STACK)>(sc, src, sink, pool, smem, stat, records);
}

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) RT_WAVES_PER_EU k_trace_secondary(SceneDev sc, QueueSrc src, float4 *hit1, uint32_t *inst1, uint32_t *pool, uint32_t *stat)
{
    __shared__ int smem[(RT_ROWS(STACK) + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    SecondarySink sink = {src, hit1, inst1};
    trace_wave<RT_ROWS(STACK), PBLOCK, TWO_LEVEL, (!TWO_LEVEL && RT_ROWS(STACK) == RT_LDS_STACK_ROWS_SETS) ? RT_POOL_CHUNK_SETS : RT_POOL_CHUNK, false, false, false, RT_REFS(STACK)>(sc, src, sink, pool, smem, stat);
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
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, 64u, false, true, false, true>(pd.sc, src, sink, nullptr, smem, nullptr, walk);
}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_walk_queue(SceneDev sc, QueueSrc src, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    NullSink sink;
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, false, true, false, true>(sc, src, sink, nullptr, smem, nullptr, walk);
}
template <bool TWO_LEVEL>
__global__ void __launch_bounds__(PBLOCK) k_walk_shadow(SceneDev sc, ShadowQueue queue, unsigned long long *walk)
{
    __shared__ int smem[(RT_LDS_STACK_ROWS + RT_TOP_ROWS(PBLOCK)) * PBLOCK];
    ShadowSrcN<true> src;
    static_cast<ShadowQueue &>(src) = queue;
    NullSink sink;
    trace_wave<RT_LDS_STACK_ROWS, PBLOCK, TWO_LEVEL, RT_POOL_CHUNK, RT_SHADOW_UNORDERED != 0, true, false, true>(sc, src, sink, nullptr, smem, nullptr, walk);
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
    float4 m = make_float4((n * prev.x + cur.x) / n1, (n * prev.y + cur.y) / n1, (n * prev.z + cur.z) / n1, (n * prev.w + cur.w) / n1);
    // the reference's RGBA16F texture (src/DXRExperimentsApp.cpp:28): what the next frame reads back is this mean rounded to fp16
    if (pd.accum_f16 == 1u) m = make_float4(__half2float(__float2half_rn(m.x)), __half2float(__float2half_rn(m.y)), __half2float(__float2half_rn(m.z)), __half2float(__float2half_rn(m.w)));
    else if (pd.accum_f16 == 2u) m = make_float4(__half2float(__float2half_rz(m.x)), __half2float(__float2half_rz(m.y)), __half2float(__float2half_rz(m.z)), __half2float(__float2half_rz(m.w)));
    return m;
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
    if (pd.primary_persistent) {
        if (pd.n_frames > 1u) k_primary<STACK, TWO_LEVEL, true, true><<<rt_persistent_grid(ctx, k_primary<STACK, TWO_LEVEL, true, true>, PBLOCK, cap), PBLOCK, 0, st>>>(pd);
        else k_primary<STACK, TWO_LEVEL, false, true><<<rt_persistent_grid(ctx, k_primary<STACK, TWO_LEVEL, false, true>, PBLOCK, cap), PBLOCK, 0, st>>>(pd);
    } else {
        if (pd.n_frames > 1u) k_primary<STACK, TWO_LEVEL, true, false><<<blocks(cap), PBLOCK, 0, st>>>(pd);
        else k_primary<STACK, TWO_LEVEL, false, false><<<blocks(cap), PBLOCK, 0, st>>>(pd);
        // the rays that would have needed a stack row beyond LDS (none on the bench scenes: the launch returns at once)
        if (pd.retry) k_primary_retry<STACK, TWO_LEVEL><<<rt_persistent_grid(ctx, k_primary_retry<STACK, TWO_LEVEL>, PBLOCK, cap), PBLOCK, 0, st>>>(pd);
    }
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
        HIP_TRY(hipMemsetAsync(pd.sh_vis, 0, ((rays_max + 31) / 32) * 4, st));      // the visibility bits: set by the rays that reach their light
        if (!TWO_LEVEL && ctx->opt_repack && RT_ROWS(STACK) != RT_LDS_STACK_ROWS_TEST) {
            if constexpr (!TWO_LEVEL && RT_ROWS(STACK) != RT_LDS_STACK_ROWS_TEST) {
                const unsigned grid = B ? rt_persistent_grid(ctx, k_trace_shadow_rp<STACK, true>, PBLOCK, rays_max) : rt_persistent_grid(ctx, k_trace_shadow_rp<STACK, false>, PBLOCK, rays_max);
                // (+ 64 B of tallies behind the records: rt_debug_repack_stats; a launch with another grid moves them, so they are cleared then)
                if (ctx->rp_grid != grid) {
                    RT_TRY(ctx->rp_records.reserve((size_t)grid * PBLOCK * RT_RP_SLOT_BYTES + 64));
                    HIP_TRY(hipMemsetAsync((char *)ctx->rp_records.p + (size_t)grid * PBLOCK * RT_RP_SLOT_BYTES, 0, 64, st));
                    ctx->rp_grid = grid;
                }
                if (B) k_trace_shadow_rp<STACK, true><<<grid, PBLOCK, 0, st>>>(pd.sc, sq, pd.pools, &pd.counters[C_SHADOW], (char *)ctx->rp_records.p);
                else k_trace_shadow_rp<STACK, false><<<grid, PBLOCK, 0, st>>>(pd.sc, sq, pd.pools, &pd.counters[C_SHADOW], (char *)ctx->rp_records.p);
            }
        } else
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

// (+ RT_STACK_REFS: the instantiations that look references up, for scenes that hold split triangles)
template <int STACK>
int launch_frame_any(rt_pipeline *p, PipeDev &pd, uint32_t shadow_slots, bool counted)
{
    if (p->scene->has_refs)
        return p->scene->two_level ? launch_frame<STACK + RT_STACK_REFS, true>(p, pd, shadow_slots, counted) : launch_frame<STACK + RT_STACK_REFS, false>(p, pd, shadow_slots, counted);
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

int rt_frame_launch(rt_pipeline *p, PipeDev &pd, uint32_t shadow_slots, bool counted, bool set_rows)
{
    // 18 LDS stack rows + the 8-row top table = 26 KiB per 256-thread block = 6 resident blocks per CU, whatever
    // the depth of the tree; the rare deeper walk continues in global rows (rt_trace_wave.h)
    if (p->ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST) return launch_frame_any<RT_LDS_STACK_ROWS_TEST>(p, pd, shadow_slots, counted);
    // (scene_for_set never asks for the sets' rows on a scene with split triangles: there the 18-row six-wave kernels are the faster ones --
    // 2.2 M-triangle stress scene 5.63 -> 4.89 ms per frame, 272 k 4.88 -> 4.86: profiles/r05/ref_rule.txt -- and the reference-aware
    // seven-wave instantiations, which spilled 36 - 72 B, are not compiled)
    if (set_rows) return launch_frame<RT_LDS_STACK_ROWS_SETS, false>(p, pd, shadow_slots, counted);
    return launch_frame_any<RT_LDS_STACK_ROWS>(p, pd, shadow_slots, counted);
}

int rt_frame_count_walk(rt_pipeline *p, unsigned long long *w)
{
    return p->scene->two_level ? count_walk_launch<true>(p, w) : count_walk_launch<false>(p, w);
}

int rt_frame_count_work(rt_pipeline *p, unsigned long long *w)
{
    hipStream_t st = p->ctx->stream;
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
    return RT_OK;
}

int rt_frame_debug_cube(hipStream_t st, const PipeDev &pd, const float *dirs, float *out, size_t n)
{
    k_debug_cube<<<blocks(n), PBLOCK, 0, st>>>(pd, dirs, out, n);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

