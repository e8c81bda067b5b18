#!/bin/bash
# The same evidence as tools/measure.sh for a program that is not bench.py (the program itself goes directly after rocprofv3's `--`):
#   bash tools/measure_cmd.sh <tag> <name> python3 tools/bench_big.py 10000 1000 64 700
# -> profiles/<tag>_{kernel_stats,traffic,pmc}_<name>.* through tools/measure_summary.py (per dispatch: the program should make few calls)
export TMPDIR=/tmp; R=$PWD; TAG=$1; NAME=$2; shift 2
O=$R/gpurun_out/measure/${TAG}_${NAME}; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- "$@" > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- "$@" > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- "$@" > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/pmcA -- "$@" > $O/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB -- "$@" > $O/pmcB.log 2>&1
cd $R
python3 tools/measure_summary.py $TAG $NAME
