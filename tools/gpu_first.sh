#!/bin/bash
# contact of a new kernel with the GPU: debug case, forced-kernel parity tests, the whole GPU suite, short traces
mkdir -p gpurun_out
python tools/dev/dbg1.py 2>&1 | grep -c nan
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/t1.log 2>&1; echo "t1 rc=$?" >> gpurun_out/t1.log
tail -5 gpurun_out/t1.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t2.log 2>&1; echo "t2 rc=$?" >> gpurun_out/t2.log
tail -8 gpurun_out/t2.log
bash tools/prof_trace.sh row
