#!/bin/bash
# Runs on the GPU box (via gpurun): for each workload (c2 = the bench default, c5 = the HBM-bound 10 M-triangle one)
# a kernel-trace pass and separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ and TCC
# sets in passes of their own), each around the same bench command.
# usage: tools/run_profiles.sh <tag> [workloads...]    -> gpurun_out/<tag>/<workload>_<pass>.md (+ .log)
# workloads: c2h (THE bench default: `python3 bench.py`, whatever it runs), c2d (the driver's form: --gpus 1 --steps 20 --warmup 5), c2 (the bench scene frame by frame), c5 (10 M triangles 4K), c4 (4096 instances 4K realtime + denoiser: tools/profile_c4.py),
#            c2b / c5b (the same two with 8 frames per set of launches, rt_pipeline_render_batch); PASSES="kt ea write" limits the passes
TAG=${1:-r02}; shift
WL=${@:-c2 c5}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  PROG=$R/bench.py
  if [ $w = c2h ]; then ARGS="--cpu-seconds 0 --no-live-pmc";
  elif [ $w = c2d ]; then ARGS="--gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-live-pmc";       # the driver's form of the command
  elif [ $w = c2 ]; then ARGS="--steps 8 --warmup 2 --batch 1 --cpu-seconds 0 --hbm-frames 0 --no-live-pmc";
  elif [ $w = c2b ]; then ARGS="--steps 60 --warmup 30 --cpu-seconds 0 --hbm-frames 0 --no-live-pmc --no-frame-by-frame --no-roofline";       # the headline launches only: sets of 30 frames
  elif [ $w = c5 ]; then ARGS="--workload c5 --hbm-frames 4 --no-live-pmc";
  elif [ $w = c5b ]; then ARGS="--workload c5 --hbm-frames 16 --batch 16 --no-live-pmc --no-roofline";
  elif [ $w = c2s ]; then ARGS="--steps 20 --warmup 20 --batch 20 --cpu-seconds 0 --hbm-frames 0 --no-live-pmc --no-frame-by-frame --no-roofline";      # sets of 20: what the driver's --steps 20 renders
  else PROG=$R/tools/profile_c4.py; ARGS="4"; fi
  run() {   # name, rocprof options...
    n=$1; shift
    timeout 300 rocprofv3 "$@" -d $OUT/${w}_$n -o p --output-format csv -- python3 $PROG $ARGS > $OUT/${w}_$n.log 2>&1
    python3 $R/tools/profile_summary.py $OUT/${w}_$n "rocprofv3 $* -- python3 $(basename $PROG) $ARGS" > $OUT/${w}_$n.md
    rm -rf $OUT/${w}_$n
    tail -n 1 $OUT/${w}_$n.log | cut -c 1-300
  }
  want() { [ -z "$PASSES" ] || echo " $PASSES " | grep -q " $1 "; }
  want kt && run kt --kernel-trace --stats
  [ -n "$PASSES" ] && { want ea && run ea --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum; want write && run write --kernel-trace --pmc WRITE_SIZE; want sq && run sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE; want fetch && run fetch --kernel-trace --pmc FETCH_SIZE; want tcp && run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum; want lanes && run lanes --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE; want tcc && run tcc --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum; want mix && run mix --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE; want mix2 && run mix2 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE; continue; }
  run fetch --kernel-trace --pmc FETCH_SIZE
  run write --kernel-trace --pmc WRITE_SIZE
  run tcc --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
  run ea --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
  run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  # the vector-memory path between the lanes and the L2 (round 3): L1 tag accesses, L1 -> L2 read requests and their summed
  # latency, cycles the L1 waits for L2 data, address-unit busy cycles and the wave instructions it took
  run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  # (the TA set -- TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum -- aborts rocprofv3 7.2 on
  # this image with signal 6 and is not collected)
  # the vector instructions by class (round 3): with the per-instruction issue costs of tools/microbench/valu_rate they give the
  # cycles the SIMDs' vector pipes were actually taken
  run mix --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE
  run mix2 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
  run ta2 --kernel-trace --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
done
if [ -n "$CALIBRATE" ]; then
  # the same counters on tools/microbench/slab_fetch (known lines per lane and launch): counter units per 64-B line / 128-B block
  for n in tcp ta2; do
    case $n in
      tcp) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum";;
      ta) C="TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum";;
      ta2) C="TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE";;
    esac
    for sz in 131072 8388608; do
      timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/slab_${sz}_$n -o p --output-format csv -- $R/tools/microbench/slab_fetch $sz 6 > $OUT/slab_${sz}_$n.log 2>&1
      python3 $R/tools/profile_summary.py $OUT/slab_${sz}_$n "rocprofv3 --pmc $C -- tools/microbench/slab_fetch $sz 6 (launches: 100 warm-up iterations, then 2000 timed; 1536 workgroups of 256; one block index per lane and iteration)" > $OUT/slab_${sz}_$n.md
      rm -rf $OUT/slab_${sz}_$n
    done
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/slab_kt -o p --output-format csv -- $R/tools/microbench/slab_fetch 131072 6 > $OUT/slab_kt.log 2>&1
  python3 $R/tools/profile_summary.py $OUT/slab_kt "rocprofv3 --kernel-trace --stats -- tools/microbench/slab_fetch 131072 6" > $OUT/slab_kt.md; rm -rf $OUT/slab_kt
fi
