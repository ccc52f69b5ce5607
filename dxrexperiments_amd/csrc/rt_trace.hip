// rt_trace.hip -- batch TraceRay kernels behind rt_trace_batch().
#include "rt_trace_wave.h"

using namespace rtd;

namespace {

constexpr int TRACE_BLOCK = 256;

RT_DEV void store_hit(const TraceOut &out, size_t i, const HitD &h)
{
    const bool miss = h.inst == RT_NO_HIT;
    if (out.t) out.t[i] = miss ? -1.0f : h.t;
    if (out.u) out.u[i] = h.u;
    if (out.v) out.v[i] = h.v;
    if (out.prim) out.prim[i] = h.prim;
    if (out.inst) out.inst[i] = h.inst;
}

RT_DEV RayD load_ray(const float4 *__restrict__ o, const float4 *__restrict__ d, size_t i)
{
    const v4f a = ldg16(o, i * 16), b = ldg16(d, i * 16);
    RayD r;
    r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
    r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
    return r;
}

__global__ void __launch_bounds__(TRACE_BLOCK)
k_trace_canonical(SceneDev sc, const float4 *__restrict__ o, const float4 *__restrict__ d, size_t n, uint32_t flags, TraceOut out)
{
    const size_t i = (size_t)blockIdx.x * TRACE_BLOCK + threadIdx.x;
    if (i >= n) return;
    const RayD r = load_ray(o, d, i);
    uint32_t cn, ct;
    const HitD h = trace_canonical(sc, r, flags, cn, ct);
    store_hit(out, i, h);
    if (out.cnt_nodes) out.cnt_nodes[i] = cn;
    if (out.cnt_tris) out.cnt_tris[i] = ct;
}

struct BatchSrc {
    const float4 *o, *d;
    uint32_t n, fl;
    RT_DEV uint32_t count() const { return n; }
    RT_DEV uint32_t flags() const { return fl; }
    RT_DEV bool load(uint32_t i, RayD &r) const { r = load_ray(o, d, i); return true; }
};

struct BatchSink {
    TraceOut out;
    RT_DEV void store(uint32_t i, const HitD &h, bool) const { store_hit(out, i, h); }
};

template <int STACK, bool TWO_LEVEL>
__global__ void __launch_bounds__(TRACE_BLOCK) k_trace_fast(SceneDev sc, BatchSrc src, BatchSink sink, uint32_t *pool)
{
    __shared__ int smem[(RT_ROWS(STACK) + RT_TOP_ROWS(TRACE_BLOCK)) * TRACE_BLOCK];
    trace_wave<RT_ROWS(STACK), TRACE_BLOCK, TWO_LEVEL, RT_POOL_CHUNK, false, false, false, RT_REFS(STACK)>(sc, src, sink, pool, smem, nullptr);
}

template <int STACK>
void launch_fast(const rt_context *ctx, bool two_level, hipStream_t st, const SceneDev &sc, const BatchSrc &src, const BatchSink &sink, uint32_t *pool)
{
    if (two_level)
        k_trace_fast<STACK, true><<<rt_persistent_grid(ctx, k_trace_fast<STACK, true>, TRACE_BLOCK, src.n), TRACE_BLOCK, 0, st>>>(sc, src, sink, pool);
    else
        k_trace_fast<STACK, false><<<rt_persistent_grid(ctx, k_trace_fast<STACK, false>, TRACE_BLOCK, src.n), TRACE_BLOCK, 0, st>>>(sc, src, sink, pool);
}

}  // namespace

int rt_launch_trace(rt_context *ctx, const rt_scene *s, const float4 *o, const float4 *d, size_t n, uint32_t ray_flags,
                    uint32_t kernel, const TraceOut &out)
{
    if (n == 0) return RT_OK;
    SceneDev sc = s->dev();
    hipStream_t st = ctx->stream;
    if (kernel == RT_TRACE_CANONICAL) {
        const unsigned grid = (unsigned)((n + TRACE_BLOCK - 1) / TRACE_BLOCK);
        HIP_TRY(hipEventRecord(ctx->ev0, st));
        k_trace_canonical<<<grid, TRACE_BLOCK, 0, st>>>(sc, o, d, n, ray_flags, out);
    } else {
        if (n > 0xFFFFFF00ull) { rt_set_error("rt_trace_batch: more than 2^32 rays in one batch"); return RT_ERR_INVALID_ARG; }
        RT_TRY(ctx->pool.reserve(RT_POOL_GROUPS * RT_POOL_STRIDE * 4));
        HIP_TRY(hipMemsetAsync(ctx->pool.p, 0, RT_POOL_GROUPS * RT_POOL_STRIDE * 4, st));
        BatchSrc src = {o, d, (uint32_t)n, ray_flags};
        BatchSink sink = {out};
        HIP_TRY(hipEventRecord(ctx->ev0, st));
        uint32_t *pool = ctx->pool.as<uint32_t>();
        // a fixed number of LDS stack rows whatever the depth of the tree: deeper walks continue in global rows
        RT_TRY(rt_scene_dev_for_launch(ctx, s, rt_lds_stack_rows(ctx), (size_t)ctx->cu_count * 16 * TRACE_BLOCK, &sc));
        if (s->has_refs) {
            if (ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST) launch_fast<RT_LDS_STACK_ROWS_TEST + RT_STACK_REFS>(ctx, s->two_level, st, sc, src, sink, pool);
            else launch_fast<RT_LDS_STACK_ROWS + RT_STACK_REFS>(ctx, s->two_level, st, sc, src, sink, pool);
        } else if (ctx->lds_stack_rows == RT_LDS_STACK_ROWS_TEST) launch_fast<RT_LDS_STACK_ROWS_TEST>(ctx, s->two_level, st, sc, src, sink, pool);
        else launch_fast<RT_LDS_STACK_ROWS>(ctx, s->two_level, st, sc, src, sink, pool);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    return RT_OK;
}
