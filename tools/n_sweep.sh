#!/bin/bash
# stage times against the contour count (16 images x <per-image>): how much a kernel loses to the
# last, partly filled round of workgroups.  usage: n_sweep.sh <outdir> <per-image counts...>
out=$1; shift
mkdir -p gpurun_out/$out
for per in "$@"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --per-image $per --no-extras --no-cpu-baseline > gpurun_out/$out/per_$(printf %03d $per).json 2>> gpurun_out/$out/err.log || exit 1
done
