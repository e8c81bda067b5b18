#!/bin/bash
# HBM traffic of one pdfposteriors launch from the TCC counters (separate --pmc passes, as
# MI355X_MICROARCH.md "HBM" prescribes).  Run on the GPU box from the repo root:
#   bash tools/traffic.sh [bench args]        -> gpurun_out/traffic/{fetch,write}
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/traffic
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/traffic/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/traffic_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/traffic/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/traffic_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/traffic/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/traffic_trace.log 2>&1
tail -1 $R/gpurun_out/traffic_trace.log | cut -c1-300
