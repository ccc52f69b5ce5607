// Microbenchmark: issue cost of the VALU instructions the traversal step is made of, on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -w tools/microbench/valu_rate.hip -o tools/microbench/valu_rate
// 2048 blocks x 256 threads (8 waves per SIMD), every wave runs `iters` iterations of 8 independent
// copies of one instruction; reported: ns per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const v2f s2 = {s, s};
    const unsigned long long m = __builtin_amdgcn_read_exec() >> 1;      // wave-uniform lane mask in an SGPR pair
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define OP(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 1) {
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(s2));
        } else if (MODE == 2) {
#define OP(n) "v_min_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 3) {
#define OP(n) "v_min3_f32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 4) {
#define OP(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
            asm volatile(R8(OP) : REGS : "v"(s) : "vcc");
#undef OP
        } else if (MODE == 5) {
#define OP(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %9\n"
            asm volatile(R8(OP) : REGS : "v"(s), "s"(m));
#undef OP
        } else if (MODE == 6) {
#define OP(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 7) {
#define OP(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 8) {
#define OP(n) "v_add_u32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 9) {
#define OP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s) : "vcc");
#undef OP
        } else if (MODE == 10) {
#define OP(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s) : "s20", "s21");
#undef OP
        } else if (MODE == 11) {
#define OP(n) "v_max_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 12) {
#define OP(n) "v_lshl_or_b32 %" #n ", %" #n ", 10, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 13) {
#define OP(n) "v_mov_b32 %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 14) {
#define OP(n) "v_med3_f32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 16) {      // the compiler's own select: (a < s) ? a : s  (v_cmp + v_cndmask)
            a0 = a0 < s ? a0 + 1.0f : s; a1 = a1 < s ? a1 + 1.0f : s; a2 = a2 < s ? a2 + 1.0f : s; a3 = a3 < s ? a3 + 1.0f : s;
            a4 = a4 < s ? a4 + 1.0f : s; a5 = a5 < s ? a5 + 1.0f : s; a6 = a6 < s ? a6 + 1.0f : s; a7 = a7 < s ? a7 + 1.0f : s;
            asm volatile("" : REGS);
        } else if (MODE == 15) {
#define OP(n) "v_mul_f32 %" #n ", 0x3f800080, %" #n "\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 17) {
#define OP(n) "v_fmac_f32 %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 18) {
            asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(s2));
        } else if (MODE == 19) {
#define OP(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 20) {
#define OP(n) "v_add_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 21) {
#define OP(n) "v_and_b32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 22) {
#define OP(n) "v_max3_f32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 23) {
#define OP(n) "v_min_u32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 25) {
#define OP(n) "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 26) {      // f16 sources converted inside the fma: src0 = low half of the register, src1 / src2 f32
#define OP(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %8 op_sel_hi:[1,0,0]\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 27) {
#define OP(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 28) {
#define OP(n) "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 29) {
#define OP(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 30) {
#define OP(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 31) {
#define OP(n) "v_lshrrev_b32 %" #n ", 8, %" #n "\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 32) {      // mixed streams: are the costs additive?  4 x (v_fma_f32 ; v_mul_f32)
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_fma_f32 %4, %4, %8, %8\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %8\n v_mul_f32 %7, %7, %8\n" : REGS : "v"(s));
        } else if (MODE == 33) {      // 4 x (v_cvt_f32_ubyte1 ; v_fma_f32)
            asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_cvt_f32_ubyte1 %2, %2\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_cvt_f32_ubyte1 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_cvt_f32_ubyte1 %6, %6\n v_fma_f32 %7, %7, %8, %8\n" : REGS : "v"(s));
        } else if (MODE == 34) {      // 4 x (v_cndmask_b32 sgpr ; v_max3_f32)
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %8\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %8\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %8\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %8\n" : REGS : "v"(s), "s"(m));
        } else if (MODE == 35) {      // the step's own mix, roughly: cvt fma cvt fma min max3 cmp cndmask
            asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_cvt_f32_ubyte2 %2, %2\n v_fma_f32 %3, %3, %8, %8\n"
                         "v_min_f32 %4, %4, %8\n v_max3_f32 %5, %5, %8, %8\n v_cmp_le_f32_e64 s[20:21], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, %9\n" : REGS : "v"(s), "s"(m) : "s20", "s21");
        } else if (MODE == 36) {      // dependent chain: each instruction reads the one before (what ONE wave's step mostly is)
            asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n"
                         "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %0, %0, %8, %8\n" : REGS : "v"(s));
        } else if (MODE == 37) {      // dependent chain of 2-cycle instructions
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n"
                         "v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n" : REGS : "v"(s));
        } else if (MODE == 38) {      // a stream with the class shares of the traversal kernels (profiles/r03/c2_mix.md: add 9, mul 10, fma 18, cvt 12,
                                      // int 15, min / max / compare / select 36 %), 32 independent instructions over 8 registers
            asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_sub_f32 %2, %2, %8\n v_max3_f32 %3, %3, %8, %8\n"
                         "v_cmp_le_f32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_mul_f32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         "v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_min_f32 %2, %2, %8\n v_lshl_or_b32 %3, %3, 10, %8\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %8\n v_mul_f32 %6, %6, %8\n v_cmp_ne_u32_e64 s[22:23], %7, %8\n"
                         "v_cvt_f32_ubyte2 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_add_f32 %2, %2, %8\n v_min3_f32 %3, %3, %8, %8\n"
                         "v_cmp_lt_f32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_mul_f32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                         "v_cvt_f32_ubyte3 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_max_f32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_cndmask_b32_e64 %4, %4, %8, %9\n v_fmac_f32 %5, %8, %8\n v_sub_f32 %6, %6, %8\n v_lshlrev_b32 %7, 6, %7\n"
                         : REGS : "v"(s), "s"(m) : "s20", "s21", "s22", "s23");
        } else if (MODE == 39) {      // round 5: the packed-f16 candidates for the node step's culling planes (VERDICT r4 task 1)
#define OP(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 40) {
#define OP(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 41) {
#define OP(n) "v_pk_min_f16 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 42) {
#define OP(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 43) {
#define OP(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 44) {      // low half + high half of one register (the (lo, -hi) pair of a child): SDWA word selects
#define OP(n) "v_add_f16_sdwa %" #n ", %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
            asm volatile(R8(OP) : REGS : "v"(s));
#undef OP
        } else if (MODE == 45) {
#define OP(n) "v_cmp_le_f16_e64 s[20:21], %" #n ", %8\n"
            asm volatile(R8(OP) : REGS : "v"(s) : "s20", "s21");
#undef OP
        } else if (MODE == 46) {      // 4 x (v_perm_b32 ; v_pk_fma_f16): two planes per pair of instructions
            asm volatile("v_perm_b32 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n"
                         "v_perm_b32 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n" : REGS : "v"(s));
        } else if (MODE == 47) {      // one child of a packed-f16 step: 3 perm, 3 pk_fma, 3 pk_max, sdwa add, cmp (11 instructions)
            asm volatile("v_perm_b32 %0, %0, %8, %8\n v_pk_fma_f16 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n"
                         "v_perm_b32 %2, %2, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_max_f16 %3, %0, %1\n v_pk_max_f16 %4, %2, %8\n v_pk_max_f16 %5, %3, %4\n"
                         "v_add_f16_sdwa %6, %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_cmp_le_f16_e64 s[20:21], %6, %8\n"
                         : REGS : "v"(s) : "s20", "s21");
        } else if (MODE == 48) {      // the same child in today's f32 form: 6 cvt, 6 fma, max3, max, min3, min, mul, cmp (18 instructions)
            asm volatile("v_cvt_f32_ubyte0 %0, %7\n v_fma_f32 %0, %0, %8, %8\n v_cvt_f32_ubyte1 %1, %7\n v_fma_f32 %1, %1, %8, %8\n v_cvt_f32_ubyte2 %2, %7\n v_fma_f32 %2, %2, %8, %8\n"
                         "v_cvt_f32_ubyte3 %3, %7\n v_fma_f32 %3, %3, %8, %8\n v_cvt_f32_ubyte0 %4, %7\n v_fma_f32 %4, %4, %8, %8\n v_cvt_f32_ubyte1 %5, %7\n v_fma_f32 %5, %5, %8, %8\n"
                         "v_max3_f32 %0, %0, %1, %2\n v_max_f32 %0, %0, %8\n v_min3_f32 %3, %3, %4, %5\n v_min_f32 %3, %3, %8\n v_mul_f32 %3, 0x3f800080, %3\n v_cmp_le_f32_e64 s[20:21], %0, %3\n"
                         : REGS : "v"(s) : "s20", "s21");
        } else if (MODE == 24) {
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(s2));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
float run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0000001f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char **argv)
{
    // argv[1]: workgroups per CU (each 4 waves = one per SIMD): 8 (default) = eight waves per SIMD, 6 = the traversal kernels' occupancy, 1 = one wave alone
    const int blocks = 256 * (argc > 1 ? atoi(argv[1]) : 8), iters = 100000;
    float *d; hipMalloc(&d, blocks * 256 * 4);
    const char *names[] = {"v_mul_f32", "v_pk_mul_f32", "v_min_f32", "v_min3_f32", "v_cndmask_b32 vcc", "v_cndmask_b32 sgpr", "v_sub_f32", "v_fma_f32",
                           "v_add_u32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_max_f32", "v_lshl_or_b32", "v_mov_b32", "v_med3_f32", "v_mul_f32 literal", "cmp+add+cndmask (x3)", "v_fmac_f32 (VOP2)", "v_pk_fma_f32", "v_cvt_f32_ubyte1", "v_add_f32", "v_and_b32",
                           "v_max3_f32", "v_min_u32", "v_pk_add_f32", "v_perm_b32", "v_fma_mix_f32", "v_bfe_u32", "v_and_or_b32", "v_lshl_add_u32", "v_cvt_f32_u32", "v_lshrrev_b32",
                           "mixed fma+mul", "mixed cvt+fma", "mixed cndmask+max3", "mixed step-like", "dependent fma chain", "dependent mul chain", "traversal-kernel mix",
                           "v_pk_fma_f16", "v_pk_max_f16", "v_pk_min_f16", "v_pk_add_f16", "v_cvt_pkrtz_f16_f32", "v_add_f16 sdwa w0+w1", "v_cmp_le_f16 sgpr", "mixed perm+pk_fma_f16", "one child, packed f16 (11)", "one child, f32 (18)"};
    float ms[49] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters),
                    run<5>(d, blocks, iters), run<6>(d, blocks, iters), run<7>(d, blocks, iters), run<8>(d, blocks, iters), run<9>(d, blocks, iters),
                    run<10>(d, blocks, iters), run<11>(d, blocks, iters), run<12>(d, blocks, iters), run<13>(d, blocks, iters), run<14>(d, blocks, iters),
                    run<15>(d, blocks, iters), run<16>(d, blocks, iters), run<17>(d, blocks, iters), run<18>(d, blocks, iters), run<19>(d, blocks, iters),
                    run<20>(d, blocks, iters), run<21>(d, blocks, iters), run<22>(d, blocks, iters), run<23>(d, blocks, iters), run<24>(d, blocks, iters),
                    run<25>(d, blocks, iters), run<26>(d, blocks, iters), run<27>(d, blocks, iters), run<28>(d, blocks, iters), run<29>(d, blocks, iters), run<30>(d, blocks, iters), run<31>(d, blocks, iters),
                    run<32>(d, blocks, iters), run<33>(d, blocks, iters), run<34>(d, blocks, iters), run<35>(d, blocks, iters), run<36>(d, blocks, iters), run<37>(d, blocks, iters), run<38>(d, blocks, iters),
                    run<39>(d, blocks, iters), run<40>(d, blocks, iters), run<41>(d, blocks, iters), run<42>(d, blocks, iters), run<43>(d, blocks, iters), run<44>(d, blocks, iters), run<45>(d, blocks, iters), run<46>(d, blocks, iters), run<47>(d, blocks, iters), run<48>(d, blocks, iters)};
    for (int m = 0; m < 49; m++) {
        const int per = (m == 1 || m == 18 || m == 24) ? 4 : (m == 38 ? 32 : m == 47 ? 11 : m == 48 ? 18 : 8);
        const double wave_insts = (double)blocks * 4 * iters * per;
        printf("%-20s %8.3f ms  %.3f ns per wave64 instruction per SIMD\n", names[m], ms[m], ms[m] * 1e6 / (wave_insts / 1024.0));
    }
    return 0;
}
