#!/bin/bash
# shadow cache: coarser cells for the rays of levels >= 1 (the incoherent ones): RT_SHADOW_CACHE_DEEP_SHIFT=s masks the low s bits of their
# cell coordinates, so that they share 4^-s of the table's lines; with the table at 4096 (automatic) and at 8192 cells per side
mkdir -p gpurun_out/r46
{
STEPS=60 WARM=30 BATCH=32 HBM=16 tools/bench_env.sh "RT_SHADOW_CACHE_DEEP_SHIFT=0" "RT_SHADOW_CACHE_DEEP_SHIFT=1" "RT_SHADOW_CACHE_DEEP_SHIFT=2" "RT_SHADOW_CACHE_DEEP_SHIFT=3" "RT_SHADOW_CACHE_DEEP_SHIFT=4"
STEPS=60 WARM=30 BATCH=32 tools/bench_env.sh "RT_SHADOW_CACHE_RES=8192 RT_SHADOW_CACHE_DEEP_SHIFT=0" "RT_SHADOW_CACHE_RES=8192 RT_SHADOW_CACHE_DEEP_SHIFT=1" "RT_SHADOW_CACHE_RES=8192 RT_SHADOW_CACHE_DEEP_SHIFT=2" "RT_SHADOW_CACHE_RES=8192 RT_SHADOW_CACHE_DEEP_SHIFT=3"
STEPS=30 WARM=10 BATCH=1 tools/bench_env.sh "RT_SHADOW_CACHE_DEEP_SHIFT=0" "RT_SHADOW_CACHE_DEEP_SHIFT=2"
} > gpurun_out/r46/deep_shift.txt 2>&1
