#!/bin/bash
# per-kernel times of a bench run: bash tools/prof_trace.sh <tag> [bench args]  -> gpurun_out/trace_<tag>/ + summary on stdout
export TMPDIR=/tmp; R=$PWD; T=$1; shift; cd /tmp
rm -rf $R/gpurun_out/trace_$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$T -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/trace_$T.log 2>&1
tail -1 $R/gpurun_out/trace_$T.log | cut -c1-200
f=$(ls $R/gpurun_out/trace_$T/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-200
