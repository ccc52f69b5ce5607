#!/bin/bash
# Runs on the GPU box (via gpurun): for each workload (c2 = the bench default, c5 = the HBM-bound 10 M-triangle one)
# a kernel-trace pass and separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ and TCC
# sets in passes of their own), each around the same bench command.
# usage: tools/run_profiles.sh <tag> [workloads...]    -> gpurun_out/<tag>/<workload>_<pass>.md (+ .log)
# workloads: c2 (bench default), c5 (10 M triangles 4K), c4 (4096 instances 4K realtime + denoiser: tools/profile_c4.py)
TAG=${1:-r02}; shift
WL=${@:-c2 c5}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  PROG=$R/bench.py
  if [ $w = c2 ]; then ARGS="--steps 8 --warmup 2 --cpu-seconds 0 --hbm-frames 0 --no-live-pmc"; elif [ $w = c5 ]; then ARGS="--workload c5 --hbm-frames 4 --no-live-pmc"; else PROG=$R/tools/profile_c4.py; ARGS="4"; fi
  run() {   # name, rocprof options...
    n=$1; shift
    timeout 600 rocprofv3 "$@" -d $OUT/${w}_$n -o p --output-format csv -- python3 $PROG $ARGS > $OUT/${w}_$n.log 2>&1
    python3 $R/tools/profile_summary.py $OUT/${w}_$n "rocprofv3 $* -- python3 $(basename $PROG) $ARGS" > $OUT/${w}_$n.md
    rm -rf $OUT/${w}_$n
    tail -n 1 $OUT/${w}_$n.log | cut -c 1-300
  }
  run kt --kernel-trace --stats
  run fetch --kernel-trace --pmc FETCH_SIZE
  run write --kernel-trace --pmc WRITE_SIZE
  run tcc --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
  run ea --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
  run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
done
