#!/bin/bash
# usage: ab.sh <outdir> <variant names...>; "base" = the in-tree library
out=$1; shift
mkdir -p gpurun_out/$out
for rep in 1 2; do
for vname in "$@"; do
  if [ "$vname" == "base" ]; then unset DANCE_HIP_LIB; else export DANCE_HIP_LIB=$PWD/dance_amd/_variants/lib_$vname.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/$out/${vname}_$rep.json 2>> gpurun_out/$out/err.log || exit 1
done
done
