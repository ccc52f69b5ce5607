// Microbenchmark (round 5): which VALU instruction classes of gfx950 overlap?  valu_rate.hip prices single instructions: "fast"
// ones (v_mul / v_add / v_sub / v_mov: ~1.05 - 1.17 ns per wave64 instruction per SIMD) and "slow" ones (everything else the
// node step is made of: ~1.8 - 1.9 ns), and shows that v_cvt_f32_ubyte and v_fma_f32 ALTERNATING cost 1.18 ns each -- two
// half-rate pipes working side by side.  This one runs every pair (X, Y) of a list of instructions as X Y X Y X Y X Y over
// eight independent registers and prints ns per instruction: ~1.1 = the pair overlaps (different pipes, issue bound),
// ~1.9 = it does not (same pipe).
//   hipcc -O3 --offload-arch=gfx950 -w tools/microbench/valu_pairs.hip -o tools/microbench/valu_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
// instruction n of the list, on register r (operand 8 = a VGPR constant, 9 = an SGPR pair mask)
#define I0(r) "v_fma_f32 %" #r ", %" #r ", %8, %8\n"
#define I1(r) "v_mul_f32 %" #r ", %" #r ", %8\n"
#define I2(r) "v_cvt_f32_ubyte1 %" #r ", %" #r "\n"
#define I3(r) "v_max_f32 %" #r ", %" #r ", %8\n"
#define I4(r) "v_max3_f32 %" #r ", %" #r ", %8, %8\n"
#define I5(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, %9\n"
#define I6(r) "v_cmp_le_f32_e64 s[20:21], %" #r ", %8\n"
#define I7(r) "v_perm_b32 %" #r ", %" #r ", %8, %8\n"
#define I8(r) "v_add_u32 %" #r ", %" #r ", %8\n"
#define I9(r) "v_lshl_or_b32 %" #r ", %" #r ", 10, %8\n"
#define I10(r) "v_pk_fma_f16 %" #r ", %" #r ", %8, %8\n"
#define I11(r) "v_pk_max_f16 %" #r ", %" #r ", %8\n"
#define I12(r) "v_and_b32 %" #r ", %" #r ", %8\n"
#define I13(r) "v_add_f32 %" #r ", %" #r ", %8\n"
#define NI 14

#define PAIR(X, Y) X(0) Y(1) X(2) Y(3) X(4) Y(5) X(6) Y(7)

template <int A, int B>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long m = __builtin_amdgcn_read_exec() >> 1;
    for (int i = 0; i < iters; i++) {
#define CASE(X, Y) if (A == X && B == Y) asm volatile(PAIR(I##X, I##Y) : REGS : "v"(s), "s"(m) : "s20", "s21");
#define ROW(X) CASE(X, 0) CASE(X, 1) CASE(X, 2) CASE(X, 3) CASE(X, 4) CASE(X, 5) CASE(X, 6) CASE(X, 7) CASE(X, 8) CASE(X, 9) CASE(X, 10) CASE(X, 11) CASE(X, 12) CASE(X, 13)
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int A, int B>
float run(float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<A, B><<<blocks, 256>>>(d, iters / 10, 1.0000001f);
    hipEventRecord(e0);
    k<A, B><<<blocks, 256>>>(d, iters, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int A, int B = 0>
struct Row {
    static void go(float *d, int blocks, int iters, float (*t)[NI])
    {
        if (B >= A) t[A][B] = run<A, B>(d, blocks, iters);
        if constexpr (B + 1 < NI) Row<A, B + 1>::go(d, blocks, iters, t);
        else if constexpr (A + 1 < NI) Row<A + 1, 0>::go(d, blocks, iters, t);
    }
};

int main(int argc, char **argv)
{
    const int blocks = 256 * (argc > 1 ? atoi(argv[1]) : 7), iters = 30000;
    float *d; hipMalloc(&d, blocks * 256 * 4);
    static float t[NI][NI];
    Row<0>::go(d, blocks, iters, t);
    const char *names[NI] = {"fma_f32", "mul_f32", "cvt_ubyte", "max_f32", "max3_f32", "cndmask", "cmp_f32", "perm_b32", "add_u32", "lshl_or", "pk_fma_f16", "pk_max_f16", "and_b32", "add_f32"};
    printf("# ns per wave64 instruction per SIMD of the stream X Y X Y X Y X Y (eight independent registers), %d workgroups of 4 waves per CU; diagonal = X alone\n", blocks / 256);
    printf("%-11s", "");
    for (int b = 0; b < NI; b++) printf(" %10s", names[b]);
    printf("\n");
    for (int a = 0; a < NI; a++) {
        printf("%-11s", names[a]);
        for (int b = 0; b < NI; b++) {
            const float ms = b >= a ? t[a][b] : t[b][a];
            printf(" %10.3f", ms * 1e6 / ((double)blocks * 4 * iters * 8 / 1024.0));
        }
        printf("\n");
    }
    return 0;
}
