#!/bin/bash
# all tracked workloads of one round: bash tools/measure_all.sh <tag>   (on the GPU box, from the repo root)
for w in lfmmi_den wsj_den wsj_num lexicon5000 ergodic64 l2r3 lfmmi_den4000; do
  bash tools/measure.sh "$1" $w > gpurun_out/measure_$w.log 2>&1
  tail -2 gpurun_out/measure_$w.log
done
python3 bench.py --emissions peaky --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/measure/peaky_lfmmi_den.json
python3 bench.py --emissions peaky_offset --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/measure/peaky_offset_lfmmi_den.json
python3 bench.py --workload wsj_den --emissions peaky --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/measure/peaky_wsj_den.json
