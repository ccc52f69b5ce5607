// rt_level_scan.h -- prefix sums over the workgroups of ONE launch, for the level-by-level builders.
//
// The PLOC rounds (rt_bvh_ploc.hip) and the collapse levels (rt_bvh_wide.hip) each need "how many outputs do the
// workgroups before me produce" and "how many are there in all" before the next step can run.  Round 2 got both from a
// library scan plus a 4-byte copy to the host per round: ~12 launches and one host round trip for ~15 us of work.  Here
// every workgroup publishes its tally, the LAST one to arrive turns the tallies into exclusive offsets in place and writes
// the size of the next round to device memory, and the host launches a batch of rounds without looking.  Order is by
// workgroup index, not by arrival, so node numbers stay run-to-run deterministic.
//
// gfx950 has one L2 per XCD and they are not coherent with each other: tallies are stored and loaded at agent scope and
// bracketed by __threadfence() (write-back before the arrival counter is bumped, invalidate before the last workgroup reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt_scan {

__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= (uint32_t)o) v += u;
    }
    return v;
}

// exclusive prefix of v over the B threads of the workgroup; total = sum over the workgroup.  lds: B / 64 words.
template <unsigned B> __device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t *lds, uint32_t &total)
{
    constexpr unsigned WAVES = B / 64;
    const uint32_t inc = wave_inclusive(v), wave = threadIdx.x >> 6;
    __syncthreads();                                    // (lds may still be read from an earlier call)
    if ((threadIdx.x & 63u) == 63u) lds[wave] = inc;
    __syncthreads();
    uint32_t off = 0, all = 0;
    for (unsigned w = 0; w < WAVES; w++) {
        const uint32_t s = lds[w];
        if (w < wave) off += s;
        all += s;
    }
    total = all;
    return off + inc - v;
}

// Publishes this workgroup's tally and reports whether it is the last of nblocks to do so (uniform over the workgroup).
template <class T> __device__ __forceinline__ bool publish_and_arrive(T *tally, T v, uint32_t *counter, uint32_t nblocks, uint32_t *lds_flag)
{
    if (threadIdx.x == 0) {
        __hip_atomic_store(&tally[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        *lds_flag = atomicAdd(counter, 1u) == nblocks - 1u ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *lds_flag != 0u;
    if (last) __threadfence();
    return last;
}

// Called by every thread of the last workgroup: tally[0..nblocks) -> exclusive prefix in place; returns the total.
// T is uint32_t, or uint64_t carrying two independent 32-bit sums.  lds: B entries of T.
template <class T, unsigned B> __device__ __forceinline__ T scan_tallies(T *tally, uint32_t nblocks, T *lds)
{
    const uint32_t t = threadIdx.x, chunk = (nblocks + B - 1) / B;
    const uint32_t lo = t * chunk < nblocks ? t * chunk : nblocks, hi = lo + chunk < nblocks ? lo + chunk : nblocks;
    T s = 0;
    for (uint32_t i = lo; i < hi; i++) s += __hip_atomic_load(&tally[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    lds[t] = s;
    __syncthreads();
    for (unsigned off = 1; off < B; off <<= 1) {
        const T u = t >= off ? lds[t - off] : (T)0;
        __syncthreads();
        lds[t] += u;
        __syncthreads();
    }
    T run = lds[t] - s;
    const T total = lds[B - 1];
    for (uint32_t i = lo; i < hi; i++) {
        const T v = __hip_atomic_load(&tally[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tally[i] = run;
        run += v;
    }
    return total;
}

}  // namespace rt_scan
