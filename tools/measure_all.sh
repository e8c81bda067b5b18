#!/bin/bash
# all tracked workloads of one round: bash tools/measure_all.sh <tag>   (on the GPU box, from the repo root)
for w in lfmmi_den wsj_den wsj_num lexicon5000 ergodic64 l2r3 lfmmi_den4000 lfmmi_den6000 lfmmi_den_p400; do
  bash tools/measure.sh "$1" $w > gpurun_out/measure_$w.log 2>&1
  tail -2 gpurun_out/measure_$w.log
done
P=gpurun_out/measure/profiles
# sharp emissions (a trained acoustic model's outputs): the float64 exact kernels' regime
timeout 300 python3 bench.py --emissions peaky --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_peaky.json
timeout 300 python3 bench.py --emissions peaky_offset --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_peaky_offset.json
timeout 300 python3 bench.py --workload wsj_den --emissions peaky --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_wsj_den_peaky.json
timeout 300 python3 bench.py --workload lfmmi_den4000 --emissions peaky --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den4000_peaky.json
timeout 600 python3 tools/sharpness.py > $P/$1_sharpness.txt 2>/dev/null
# the N > 1 path of bench.py on this box's one GPU (two ranks over gloo)
MM_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_2ranks_one_gpu_gloo.json
# graphs beyond config 3's size: more pdfs, more states (teams of 4 / 8), beyond every fast path
timeout 600 python3 tools/bench_big.py $P/$1_bench_big.json > /dev/null 2>&1
# host cost of a batch of new numerator graphs
timeout 300 python3 tools/host_cost.py $P/$1_host_cost.json > /dev/null 2>&1
# per-step cycle stamps (diagnostic build, if it was made: make -C markovmodels.jl_amd/csrc stamps)
if [ -f gpurun_stamps/libmarkovmodels_amd_stamps.so ]; then
  timeout 300 python3 tools/stamps_pairs.py > $P/$1_stamps_lfmmi_den.txt 2>/dev/null
  timeout 300 python3 tools/stamps_lane.py > $P/$1_stamps_ergodic64.txt 2>/dev/null
fi
ls $P | wc -l
