// rt_trace.hip -- batch TraceRay kernels behind rt_trace_batch().
#include "rt_trace_device.h"

using namespace rtd;

namespace {

constexpr int TRACE_BLOCK = 256;

RT_DEV RayD load_ray(const float4 *__restrict__ o, const float4 *__restrict__ d, size_t i)
{
    const float4 a = o[i], b = d[i];
    RayD r;
    r.o = mk3(a.x, a.y, a.z); r.tmin = a.w;
    r.d = mk3(b.x, b.y, b.z); r.tmax = b.w;
    return r;
}

RT_DEV void store_hit(const TraceOut &out, size_t i, const HitD &h)
{
    const bool miss = h.inst == RT_NO_HIT;
    if (out.t) out.t[i] = miss ? -1.0f : h.t;
    if (out.u) out.u[i] = h.u;
    if (out.v) out.v[i] = h.v;
    if (out.prim) out.prim[i] = h.prim;
    if (out.inst) out.inst[i] = h.inst;
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

template <int STACK>
__global__ void __launch_bounds__(TRACE_BLOCK)
k_trace_fast(SceneDev sc, const float4 *__restrict__ o, const float4 *__restrict__ d, size_t n, uint32_t flags, TraceOut out)
{
    __shared__ int smem[StackShape<STACK>::LDSN * TRACE_BLOCK];
    const size_t i = (size_t)blockIdx.x * TRACE_BLOCK + threadIdx.x;
    if (i >= n) return;
    const RayD r = load_ray(o, d, i);
    const HitD h = trace_fast<STACK, TRACE_BLOCK>(sc, r, flags, smem);
    store_hit(out, i, h);
}

}  // namespace

int rt_launch_trace(rt_context *ctx, const rt_scene *s, const float4 *o, const float4 *d, size_t n, uint32_t ray_flags,
                    uint32_t kernel, const TraceOut &out)
{
    if (n == 0) return RT_OK;
    const SceneDev sc = s->dev();
    const unsigned grid = (unsigned)((n + TRACE_BLOCK - 1) / TRACE_BLOCK);
    hipStream_t st = ctx->stream;
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    if (kernel == RT_TRACE_CANONICAL) {
        k_trace_canonical<<<grid, TRACE_BLOCK, 0, st>>>(sc, o, d, n, ray_flags, out);
    } else {
        const uint32_t need = s->stack_need;
        if (need <= 32) k_trace_fast<32><<<grid, TRACE_BLOCK, 0, st>>>(sc, o, d, n, ray_flags, out);
        else if (need <= 64) k_trace_fast<64><<<grid, TRACE_BLOCK, 0, st>>>(sc, o, d, n, ray_flags, out);
        else if (need <= 160) k_trace_fast<160><<<grid, TRACE_BLOCK, 0, st>>>(sc, o, d, n, ray_flags, out);
        else {
            rt_set_error("traversal stack need %u exceeds 160 entries", need);
            return RT_ERR_UNSUPPORTED;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    return RT_OK;
}
