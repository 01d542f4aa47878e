#!/bin/bash
# the Dsnake two-stream probe (tools/dsnake_streams_probe.py) once per build of the resampler's lerp
# (tools/build_variant.py lerpN exact_ops.hip -DDANCE_EXP_LERP=N --allow-packed): which ingredient of the packed
# fp32 form makes lanes 48..63 go wrong beside the other stream's MFMA kernels.  ONE pass, no retries.
out=gpurun_out/$1; mkdir -p $out
for v in ${LERP_VARIANTS:-base lerp5 lerp1 lerp2 lerp3 lerp4}; do
  if [ "$v" == "base" ]; then unset DANCE_HIP_LIB; else export DANCE_HIP_LIB=$PWD/dance_amd/_variants/lib_$v.so; fi
  echo "== $v" >> $out/probe.txt
  timeout -k 10 120 python3 tools/dsnake_streams_probe.py >> $out/probe.txt 2>> $out/err.log || { echo "rc=$?" >> $out/probe.txt; exit 1; }
done
cat $out/probe.txt
