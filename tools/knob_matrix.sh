#!/bin/bash
# the GPU suite under each documented environment knob of the Python layer (the library itself
# reads no environment -- its kernel-form choices are the dance_policy argument, exercised by the tests): 6 runs.
# usage (on the GPU box): bash tools/knob_matrix.sh <outdir>
out=gpurun_out/$1
mkdir -p $out
for knob in DANCE_PRECISION=fp32 DANCE_STREAMS=2 DANCE_STREAMS=1 DANCE_PEEL=0 DANCE_SYNC_CHECKS=lazy DANCE_FEATPREP=torch; do
  echo "== $knob" | tee -a $out/summary.txt
  env $knob timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread \
      > $out/$knob.log 2>&1
  rc=$?
  tail -1 $out/$knob.log | tee -a $out/summary.txt
  # a fault or a hang ends the matrix: nothing more on the GPU after it
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: rc=$rc" | tee -a $out/summary.txt; exit $rc; fi
done
