// Microbenchmark (round 5): is the byte -> float conversion of the node step's 24 plane bytes available at the price of a
// multiplication?  v_cvt_f32_ubyteN is a "slow" instruction (1.85 ns per wave64 instruction per SIMD, valu_rate.hip) and it does not
// overlap with the other slow ones; v_mul_f32 is a "fast" one (1.05 ns alone, ~0.25 ns on top of a slow neighbour,
// valu_pairs.hip).  SDWA lets a VOP2 instruction read ONE BYTE of a source register, zero extended: as an f32 that is the
// denormal q * 2^-149, and  q * 2^-149 * 2^127 = q * 2^-22  exactly (gfx9 multiplies denormals at full rate when the kernel runs
// with IEEE denormals, which the product build does: -fno-gpu-flush-denormals-to-zero).  The plane distance
// fma(float(q), A, B) then is fma(q * 2^-22, A * 2^22, B): the same real product, one rounding -- the same bits.
// This program (1) checks the value for every byte and byte position, (2) times the instruction alone and in the pairs that
// matter.   hipcc -O3 --offload-arch=gfx950 -fno-gpu-flush-denormals-to-zero -w tools/microbench/sdwa_byte_mul.hip -o tools/microbench/sdwa_byte_mul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

__global__ void k_check(float *out)
{
    const unsigned q = threadIdx.x;              // 0..255
    const unsigned w = q | (q ^ 0x5au) << 8 | (q ^ 0xa5u) << 16 | (255u - q) << 24;
    const float big = 0x1p127f;
    float r0, r1, r2, r3;
    asm volatile("v_mul_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                 "v_mul_f32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                 "v_mul_f32_sdwa %2, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                 "v_mul_f32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(w), "v"(big));
    out[q * 4 + 0] = r0; out[q * 4 + 1] = r1; out[q * 4 + 2] = r2; out[q * 4 + 3] = r3;
}

#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define S(r) "v_mul_f32_sdwa %" #r ", %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define C(r) "v_cvt_f32_ubyte1 %" #r ", %9\n"
#define F(r) "v_fma_f32 %" #r ", %" #r ", %8, %8\n"
#define M(r) "v_max_f32 %" #r ", %" #r ", %8\n"
#define X(r) "v_max3_f32 %" #r ", %" #r ", %8, %8\n"
#define N(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, %10\n"
#define U(r) "v_mul_f32 %" #r ", %" #r ", %8\n"

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s, unsigned w)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long m = __builtin_amdgcn_read_exec() >> 1;
    const unsigned bytes = w + threadIdx.x;      // per-lane plane bytes
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : REGS : "v"(s), "v"(bytes), "s"(m));
        if (MODE == 1) asm volatile(C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) : REGS : "v"(s), "v"(bytes), "s"(m));
        if (MODE == 2) asm volatile(S(0) F(0) S(1) F(1) S(2) F(2) S(3) F(3) : REGS : "v"(s), "v"(bytes), "s"(m));        // a plane: byte -> float, fma (dependent)
        if (MODE == 3) asm volatile(C(0) F(0) C(1) F(1) C(2) F(2) C(3) F(3) : REGS : "v"(s), "v"(bytes), "s"(m));        // today's plane
        if (MODE == 4) asm volatile(S(0) M(1) S(2) M(3) S(4) M(5) S(6) M(7) : REGS : "v"(s), "v"(bytes), "s"(m));
        if (MODE == 5) asm volatile(S(0) X(1) S(2) X(3) S(4) X(5) S(6) X(7) : REGS : "v"(s), "v"(bytes), "s"(m));
        if (MODE == 6) asm volatile(S(0) N(1) S(2) N(3) S(4) N(5) S(6) N(7) : REGS : "v"(s), "v"(bytes), "s"(m));
        // one child of the node step, six planes + the min / max / compare: with SDWA multiplications, with conversions
        if (MODE == 7) asm volatile(S(0) F(0) S(1) F(1) S(2) F(2) S(3) F(3) S(4) F(4) S(5) F(5)
                                    "v_max3_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %8\n v_min3_f32 %3, %3, %4, %5\n v_min_f32 %3, %3, %8\n v_mul_f32 %3, 0x3f800080, %3\n v_cmp_le_f32_e64 s[20:21], %0, %3\n"
                                    : REGS : "v"(s), "v"(bytes), "s"(m) : "s20", "s21");
        if (MODE == 8) asm volatile(C(0) F(0) C(1) F(1) C(2) F(2) C(3) F(3) C(4) F(4) C(5) F(5)
                                    "v_max3_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %8\n v_min3_f32 %3, %3, %4, %5\n v_min_f32 %3, %3, %8\n v_mul_f32 %3, 0x3f800080, %3\n v_cmp_le_f32_e64 s[20:21], %0, %3\n"
                                    : REGS : "v"(s), "v"(bytes), "s"(m) : "s20", "s21");
        if (MODE == 9) asm volatile(U(0) U(1) U(2) U(3) U(4) U(5) U(6) U(7) : REGS : "v"(s), "v"(bytes), "s"(m));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
float run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters / 10, 0x1p127f, 0x11223344u);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0x1p127f, 0x11223344u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv)
{
    const int blocks = 256 * (argc > 1 ? atoi(argv[1]) : 7), iters = 100000;
    float *d; hipMalloc(&d, blocks * 256 * 4);
    k_check<<<1, 256>>>(d);
    float h[1024];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (unsigned q = 0; q < 256; q++) {
        const unsigned b[4] = {q, q ^ 0x5au, q ^ 0xa5u, 255u - q};
        for (int j = 0; j < 4; j++) if (h[q * 4 + j] != ldexpf((float)b[j], -22)) { if (bad < 8) printf("MISMATCH byte %u position %d: %a, expected %a\n", b[j], j, h[q * 4 + j], ldexpf((float)b[j], -22)); bad++; }
    }
    printf("value check: v_mul_f32_sdwa(byte k of w, 2^127) == byte * 2^-22 for all 256 values x 4 positions: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    const char *names[] = {"v_mul_f32_sdwa byte x 2^127 (denormal input)", "v_cvt_f32_ubyte1", "plane = sdwa mul ; fma (dependent)", "plane = cvt ; fma (dependent)", "sdwa mul | max", "sdwa mul | max3", "sdwa mul | cndmask",
                           "one child, sdwa planes (18)", "one child, cvt planes (18)", "v_mul_f32 (normal input)"};
    const int per[] = {8, 8, 8, 8, 8, 8, 8, 18, 18, 8};
    float ms[10] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters), run<5>(d, blocks, iters), run<6>(d, blocks, iters), run<7>(d, blocks, iters), run<8>(d, blocks, iters), run<9>(d, blocks, iters)};
    printf("# %d workgroups of 4 waves per CU\n", blocks / 256);
    for (int m = 0; m < 10; m++) printf("%-48s %8.3f ms  %.3f ns per wave64 instruction per SIMD  (%.2f ns per group of %d)\n", names[m], ms[m], ms[m] * 1e6 / ((double)blocks * 4 * iters * per[m] / 1024.0), ms[m] * 1e6 / ((double)blocks * 4 * iters / 1024.0), per[m]);
    return 0;
}
