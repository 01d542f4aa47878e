# usage (GPU box, repo root): bash tools/pmc_stack.sh <tag> <n> <v> <coop>  -- two counter passes over tools/stack_probe.py
set -e
TAG=$1; N=$2; V=$3; COOP=$4
ROOT=$(pwd)
OUT=/tmp/pmcs_$TAG
KEEP=$ROOT/gpurun_out/pmcs_$TAG
mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o $TAG -- python3 $ROOT/tools/stack_probe.py $N $V 10 $COOP > $OUT/a.log 2>&1
timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_ANY SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_SALU TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/b -o $TAG -- python3 $ROOT/tools/stack_probe.py $N $V 10 $COOP > $OUT/b.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $(find $OUT/a -name "*counter_collection.csv") $(find $OUT/a -name "*kernel_trace.csv") > $KEEP/a.txt
python3 tools/pmc_summary.py $(find $OUT/b -name "*counter_collection.csv") $(find $OUT/b -name "*kernel_trace.csv") > $KEEP/b.txt
grep "k_res_stack" $KEEP/a.txt $KEEP/b.txt
