#!/bin/bash
# every BASELINE configuration on ONE box (box-to-box spread is 5-10 %: only same-box figures compare):
#   bash tools/all_configs.sh <outdir>      -> gpurun_out/<outdir>/{cfg2,cfg3,v196,v256,cfg5}.json, dsnake.log, pipe.log
out=gpurun_out/$1
mkdir -p $out
B="python bench.py --steps 20 --warmup 5"
timeout -k 10 300 $B > $out/cfg2.json 2>> $out/err.log || exit 1
timeout -k 10 200 $B --workload cfg3 --no-extras --no-cpu-baseline > $out/cfg3.json 2>> $out/err.log || exit 1
timeout -k 10 200 $B --vertices 196 --no-extras --no-cpu-baseline > $out/v196.json 2>> $out/err.log || exit 1
timeout -k 10 200 $B --vertices 256 --no-extras --no-cpu-baseline > $out/v256.json 2>> $out/err.log || exit 1
timeout -k 10 300 python bench.py --workload cfg5 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $out/cfg5.json 2>> $out/err.log || exit 1
timeout -k 10 200 python tools/dsnake_bench.py > $out/dsnake.log 2>> $out/err.log || exit 1
timeout -k 10 200 python tools/pipeline_bench.py > $out/pipe.log 2>> $out/err.log || exit 1
for f in cfg2 cfg3 v196 v256 cfg5; do echo "== $f"; python tools/show_bench.py $out/$f.json; done
tail -4 $out/dsnake.log; tail -4 $out/pipe.log
