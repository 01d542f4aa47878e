#!/bin/bash
# the GPU suite under each documented opt-out knob: the fallbacks must stay green.
# usage (on the GPU box): bash tools/knob_matrix.sh <outdir>
out=gpurun_out/$1
mkdir -p $out
for knob in DANCE_HEAD_WGS=2 DANCE_HEAD_SHAPE=32 DANCE_FUSE16_SHAPE=32 DANCE_TAIL16=0 DANCE_INIT16=0 DANCE_HEAD_HALVES=0 DANCE_HEAD_SMALL_N=0 \
            DANCE_FUSE16_TAIL_SPLIT=0 DANCE_FUSE16_SLOTS=0 DANCE_PEEL=0 DANCE_TAIL16_GROUP=4 DANCE_TAIL16_GROUP=1 DANCE_STACK_MID=0 DANCE_HEAD_HALVES_MID=0 DANCE_STACK_ODD_SPLIT=0 DANCE_STREAMS=2 DANCE_SYNC_CHECKS=lazy DANCE_SYNC_CHECKS=0; do
  echo "== $knob" | tee -a $out/summary.txt
  env $knob timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -x \
      > $out/$knob.log 2>&1
  rc=$?
  tail -1 $out/$knob.log | tee -a $out/summary.txt
  # a fault or a hang ends the matrix: nothing more on the GPU after it
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: rc=$rc" | tee -a $out/summary.txt; exit $rc; fi
done
