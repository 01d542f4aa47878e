#!/bin/bash
# kernel trace of the Dsnake module forward (tools/dsnake_bench.py, NUM_CONVS=0 leg only): bash tools/dsnake_profile.sh <tag>
TAG=$1; ROOT=$(pwd); OUT=/tmp/dsp_$TAG; KEEP=$ROOT/gpurun_out/dsp_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
DSNAKE_BENCH_ONLY=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o $TAG -- python3 $ROOT/tools/dsnake_bench.py > $KEEP/bench.log 2> $KEEP/trace.err
cd $ROOT
python3 tools/prof_summary.py $(find $OUT/trace -name "*kernel_trace.csv") > $KEEP/summary.txt
head -40 $KEEP/summary.txt
