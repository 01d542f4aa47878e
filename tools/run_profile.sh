# usage (on the GPU box, from the repo root): bash tools/run_profile.sh <tag>
# kernel trace + stats of the default bench, then two PMC passes (own runs, kernel-trace only)
set -e
TAG=$1
ROOT=$(pwd)
OUT=/tmp/prof_$TAG
KEEP=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/trace.err
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc1 -o $TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc1.json 2> $OUT/pmc1.err
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc2 -o $TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc2.json 2> $OUT/pmc2.err
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc3 -o $TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc3.json 2> $OUT/pmc3.err
cd $ROOT
# only the small summaries travel back (gpurun_out is capped at 64 MiB)
for d in trace pmc1 pmc2 pmc3; do
    for f in $(find $OUT/$d -name "*kernel_stats.csv" -o -name "*kernel_trace.csv" -o -name "*counter_collection.csv"); do
        cp $f $KEEP/${d}_$(basename $f)
    done
done
cp $OUT/*.json $OUT/*.err $KEEP/ 2>/dev/null || true
python3 tools/prof_summary.py $KEEP/trace_*kernel_trace.csv > $KEEP/summary_trace.txt
python3 tools/pmc_summary.py $KEEP/pmc1_*counter_collection.csv $KEEP/pmc1_*kernel_trace.csv > $KEEP/summary_pmc1.txt
python3 tools/pmc_summary.py $KEEP/pmc2_*counter_collection.csv $KEEP/pmc2_*kernel_trace.csv > $KEEP/summary_pmc2.txt
python3 tools/pmc_summary.py $KEEP/pmc3_*counter_collection.csv $KEEP/pmc3_*kernel_trace.csv > $KEEP/summary_pmc3.txt
python3 tools/traffic_json.py $KEEP/pmc2_*counter_collection.csv $KEEP/pmc3_*counter_collection.csv $TAG $KEEP/hbm_traffic.json > /dev/null
# the raw per-dispatch tables stay on the box: gpurun_out travels back only below 64 MiB
rm -f $KEEP/*_counter_collection.csv $KEEP/*_kernel_trace.csv
ls -la $KEEP
