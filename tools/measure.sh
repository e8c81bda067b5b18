#!/bin/bash
# Everything the measurement claims of DESIGN.md / bench.py rest on, in one run on the GPU box (from the repo root):
#   bash tools/measure.sh <tag> <workload> [bench args]
# -> gpurun_out/measure/<tag>_<workload>/: bench.json (the driver's line, with cpu_baseline), per-kernel times
#    (rocprofv3 --kernel-trace --stats), HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, as
#    MI355X_MICROARCH.md "HBM" prescribes) and the SQ counters behind the stall / LDS figures.  tools/measure_summary.py
#    (run HERE, in the same tree) turns them into the small tracked files under profiles/.
#   NAME=lfmmi_den_peaky bash tools/measure.sh <tag> lfmmi_den --emissions peaky     (NAME: what the tracked files are called)
export TMPDIR=/tmp; R=$PWD; TAG=$1; WL=$2; shift 2
NAME=${NAME:-$WL}
PW=${PMC_WARMUP:-0}   # (calls before the one the counters are read for: LAST_CALL=1 makes tools/measure_summary.py take the last call alone)
O=$R/gpurun_out/measure/${TAG}_${NAME}; rm -rf $O; mkdir -p $O
timeout 600 python3 $R/bench.py --workload $WL --steps 20 --warmup 5 "$@" > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-sharp "$@" > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --workload $WL --steps 1 --warmup $PW --no-cpu-baseline --no-sharp "$@" > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --workload $WL --steps 1 --warmup $PW --no-cpu-baseline --no-sharp "$@" > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/pmcA -- python3 $R/bench.py --workload $WL --steps 1 --warmup $PW --no-cpu-baseline --no-sharp "$@" > $O/pmcA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB -- python3 $R/bench.py --workload $WL --steps 1 --warmup $PW --no-cpu-baseline --no-sharp "$@" > $O/pmcB.log 2>&1
cd $R
python3 tools/measure_summary.py $TAG $NAME
