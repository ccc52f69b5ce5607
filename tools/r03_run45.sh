#!/bin/bash
# the GPU parity suites under unusual settings: a shadow cache so coarse that most entries are wrong, sets cut to 3 frames with
# the free sphere off, two workgroups per CU
mkdir -p gpurun_out/r45
{
echo "== RT_SHADOW_CACHE_RES=16"; RT_SHADOW_CACHE_RES=16 python -m pytest tests -q -m gpu 2>&1 | tail -6
echo "== RT_BATCH_MAX=3 RT_FREE_RADIUS=0"; RT_BATCH_MAX=3 RT_FREE_RADIUS=0 python -m pytest tests -q -m gpu 2>&1 | tail -6
echo "== RT_PERSISTENT_BLOCKS_PER_CU=2"; RT_PERSISTENT_BLOCKS_PER_CU=2 python -m pytest tests -q -m gpu 2>&1 | tail -6
} > gpurun_out/r45/suites.txt 2>&1
