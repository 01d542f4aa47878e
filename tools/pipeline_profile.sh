#!/bin/bash
# kernel trace of the one-image pipeline (tools/pipeline_bench.py): bash tools/pipeline_profile.sh <tag>
TAG=$1; ROOT=$(pwd); OUT=/tmp/pp_$TAG; KEEP=$ROOT/gpurun_out/pp_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o $TAG -- python3 $ROOT/tools/pipeline_bench.py > $KEEP/bench.log 2> $KEEP/trace.err
cd $ROOT
python3 tools/prof_summary.py $(find $OUT/trace -name "*kernel_trace.csv") > $KEEP/summary.txt
head -45 $KEEP/summary.txt; tail -4 $KEEP/bench.log
