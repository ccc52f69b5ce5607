// Microbenchmark: how fast can a wave fetch one 64-B BVH slab per lane from a cache-resident table on gfx950?
//   hipcc -O3 --offload-arch=gfx950 -w tools/microbench/slab_fetch.hip -o tools/microbench/slab_fetch
// Every lane draws a pseudo-random slab index per iteration (like an incoherent ray batch) and the wave
// fetches the 64 B of every lane's slab in one of these ways:
//   0  per lane: 4 x global_load_dwordx4 from the lane's own slab (what the traversal step does)
//   1  quad-cooperative: in round k = 0..3 the four lanes of a quad read the four 16-B pieces of quad-lane k's
//      slab (a wave instruction touches 16 whole 64-B segments instead of 64 separate 16-B pieces), then
//      four DPP quad permutes hand every lane its own slab
//   2  per lane: 2 x dwordx4 (a 32-B node: what a compressed node would cost)
//   3  per lane: 1 x dwordx4 (16 B)
//   4  per lane: 4 x dwordx4 from FOUR different slabs (same bytes as 0, four times the cache lines)
//   5  per lane: 8 x dwordx4 = all 128 B of ONE 128-B-aligned block (an eight-wide node that IS the L2 line / fabric request)
//   6  per lane: the two 64-B halves of one 128-B block, the second half addressed only after the first has returned
//      (what a node costs whose second line is a dependent fetch)
//   7  per lane: 6 x dwordx4 = the first 96 B of one 128-B-aligned block (origin + 48 plane bytes + 8 child codes)
//   8  per lane: 4 x dwordx4 = one 64-B slab, but at 128-B stride (a 64-B node in a 128-B slot: same lines per lane as 0,
//      tells whether 5 pays for bytes or for lines)
// Modes 5-8 index the table in 128-B blocks (half as many blocks in the same bytes).
// Reported: ns per wave-step per CU and lane-slabs per ns for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f *gptr4;

__device__ inline v4f ldg16(const void *base, unsigned byte_off) { return *(gptr4)((const char *)base + byte_off); }

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *table, unsigned n_slabs_mask, int iters, float *out)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    const unsigned lane = threadIdx.x & 63u, ql = lane & 3u;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        const unsigned idx = (s >> 8) & n_slabs_mask;
        v4f a, b, c, d;
        if (MODE == 0) {
            const unsigned off = idx << 6;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = ldg16(table, off + 32); d = ldg16(table, off + 48);
        } else if (MODE == 1) {
            // round k: every lane of the quad reads piece `ql` of the slab of quad-lane k
            const unsigned i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
            const unsigned i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);   // [1,1,1,1]
            const unsigned i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
            const unsigned i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);   // [3,3,3,3]
            const v4f r0 = ldg16(table, (i0 << 6) + (ql << 4));
            const v4f r1 = ldg16(table, (i1 << 6) + (ql << 4));
            const v4f r2 = ldg16(table, (i2 << 6) + (ql << 4));
            const v4f r3 = ldg16(table, (i3 << 6) + (ql << 4));
            // lane L of the quad owns slab L: piece p of it sits in lane p's r_L.  (A real kernel transposes with
            // 12 DPP moves per dword column; here only the cost of the loads matters, so just consume them.)
            a = r0; b = r1; c = r2; d = r3;
        } else if (MODE == 2) {
            const unsigned off = idx << 6;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = a; d = b;
        } else if (MODE == 3) {
            const unsigned off = idx << 6;
            a = ldg16(table, off); b = a; c = a; d = a;
        } else if (MODE == 5) {
            const unsigned off = (idx >> 1) << 7;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = ldg16(table, off + 32); d = ldg16(table, off + 48);
            const v4f e = ldg16(table, off + 64), f = ldg16(table, off + 80), g = ldg16(table, off + 96), h = ldg16(table, off + 112);
            a += e; b += f; c += g; d += h;
        } else if (MODE == 6) {
            const unsigned off = (idx >> 1) << 7;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = ldg16(table, off + 32); d = ldg16(table, off + 48);
            // the second half's address depends on the first half's data (always +64, but the hardware cannot know)
            const unsigned dep = (__float_as_uint(a.x) | __float_as_uint(d.w)) >> 31;        // 0 for this table (all values >= 0)
            const unsigned off2 = off + 64u;
            const v4f e = ldg16(table, off2 + dep), f = ldg16(table, off2 + 16 + dep), g = ldg16(table, off2 + 32 + dep), h = ldg16(table, off2 + 48 + dep);
            a += e; b += f; c += g; d += h;
        } else if (MODE == 7) {
            const unsigned off = (idx >> 1) << 7;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = ldg16(table, off + 32); d = ldg16(table, off + 48);
            const v4f e = ldg16(table, off + 64), f = ldg16(table, off + 80);
            a += e; b += f;
        } else if (MODE == 8) {
            const unsigned off = (idx >> 1) << 7;
            a = ldg16(table, off); b = ldg16(table, off + 16); c = ldg16(table, off + 32); d = ldg16(table, off + 48);
        } else {
            const unsigned off = idx << 6;
            a = ldg16(table, off); b = ldg16(table, (off + 0x40040u + 16) & ((n_slabs_mask << 6) | 63u));
            c = ldg16(table, (off + 0x80080u + 32) & ((n_slabs_mask << 6) | 63u)); d = ldg16(table, (off + 0xc00c0u + 48) & ((n_slabs_mask << 6) | 63u));
        }
        acc += a.x + b.y + c.z + d.w;
        s ^= __float_as_uint(acc) & 1u;          // next index depends on the data: one step at a time, like a walk
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, const float4 *table, unsigned mask, float *out, int blocks)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(table, mask, 100, out);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(table, mask, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_steps = (double)blocks * 4 * iters;
    printf("%-34s %8.3f ms  %7.1f ns per wave-step per CU  %7.1f lane-slabs/ns chip-wide\n", name, ms, ms * 1e6 / (wave_steps / 256.0),
           wave_steps * 64 / (ms * 1e6));
}

int main(int argc, char **argv)
{
    const unsigned n_slabs = argc > 1 ? (unsigned)atoi(argv[1]) : (1u << 17);      // 2^17 slabs = 8 MiB (a Sponza-class tree)
    const int per_cu = argc > 2 ? atoi(argv[2]) : 6;
    float4 *table;
    float *out;
    hipMalloc(&table, (size_t)n_slabs * 64);
    std::vector<float> h((size_t)n_slabs * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i & 1023) * 1e-3f;
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    printf("table %u slabs (%.1f MiB), %d blocks of 256 per CU\n", n_slabs, n_slabs * 64.0 / (1 << 20), per_cu);
    run<0>("0 per lane 4 x 16 B (one slab)", table, n_slabs - 1, out, blocks);
    run<1>("1 quad-cooperative 4 x 16 B", table, n_slabs - 1, out, blocks);
    run<2>("2 per lane 2 x 16 B", table, n_slabs - 1, out, blocks);
    run<3>("3 per lane 1 x 16 B", table, n_slabs - 1, out, blocks);
    run<4>("4 per lane 4 x 16 B (four slabs)", table, n_slabs - 1, out, blocks);
    run<5>("5 per lane 8 x 16 B (one 128-B block)", table, n_slabs - 1, out, blocks);
    run<6>("6 per lane 2 x 64 B, 2nd dependent", table, n_slabs - 1, out, blocks);
    run<7>("7 per lane 6 x 16 B (96 of 128 B)", table, n_slabs - 1, out, blocks);
    run<8>("8 per lane 4 x 16 B at 128-B stride", table, n_slabs - 1, out, blocks);
    return 0;
}
