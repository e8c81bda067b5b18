#!/bin/bash
# all tracked workloads of one round: bash tools/measure_all.sh <tag>   (on the GPU box, from the repo root)
for w in lfmmi_den wsj_den wsj_num lexicon5000 ergodic64 l2r3 lfmmi_den4000 lfmmi_den6000 lfmmi_den_p400; do
  bash tools/measure.sh "$1" $w > gpurun_out/measure_$w.log 2>&1
  tail -2 gpurun_out/measure_$w.log
done
P=gpurun_out/measure/profiles
# sharp emissions (a trained acoustic model's outputs): the wide-exponent kernels' regime -- bench line, rocprofv3 kernel statistics, SQ
# counters and TCC traffic like the randn workloads (round 6); "consistent" = sharp ALONG a path sampled from the graph.  The counters
# are read for the THIRD call: the first runs the float32 kernels and the float64 ones behind them, the later ones the wide kernels alone
PMC_WARMUP=2 LAST_CALL=1 NAME=lfmmi_den_peaky bash tools/measure.sh "$1" lfmmi_den --emissions peaky --no-cpu-baseline > gpurun_out/measure_lfmmi_den_peaky.log 2>&1
PMC_WARMUP=2 LAST_CALL=1 NAME=wsj_den_peaky bash tools/measure.sh "$1" wsj_den --emissions peaky --no-cpu-baseline > gpurun_out/measure_wsj_den_peaky.log 2>&1
PMC_WARMUP=2 LAST_CALL=1 NAME=lfmmi_den_consistent bash tools/measure.sh "$1" lfmmi_den --emissions consistent --no-cpu-baseline > gpurun_out/measure_lfmmi_den_consistent.log 2>&1
timeout 300 python3 bench.py --emissions peaky_offset --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_peaky_offset.json
timeout 300 python3 bench.py --workload lfmmi_den4000 --emissions peaky --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den4000_peaky.json
timeout 900 python3 tools/sharpness.py > $P/$1_sharpness.txt 2>/dev/null
# SURVEY 8(d)'s second run of config 3: lengths U[750, 1500] (frames/s counts the frames inside the lengths)
timeout 300 python3 bench.py --varlen --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_varlen.json
timeout 300 python3 bench.py --workload wsj_den --varlen --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_wsj_den_varlen.json
# config 4's global batch (B = 2048) on this one GPU: what eight GPUs would have to beat
timeout 600 python3 bench.py --batch 2048 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_B2048_one_gpu.json
timeout 600 python3 bench.py --batch 2048 --varlen --no-cpu-baseline --no-sharp --steps 5 --warmup 2 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_B2048_varlen_one_gpu.json
# the caller's step (examples/test_cuda.jl:128-152): numerator + denominator + gradient, T = 700 and 150
timeout 300 python3 bench.py --workload lfmmi_step --frames 700 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_step.json
timeout 300 python3 bench.py --workload lfmmi_step --frames 150 --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_step_T150.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/measure/$1_lfmmi_step/trace -- python3 $OLDPWD/bench.py --workload lfmmi_step --frames 700 --steps 5 --warmup 2 > $OLDPWD/gpurun_out/measure_lfmmi_step_trace.log 2>&1
cd $OLDPWD
python3 tools/measure_summary.py "$1" lfmmi_step > /dev/null 2>&1
# the N > 1 path of bench.py on this box's one GPU (two ranks over gloo)
MM_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $P/$1_bench_lfmmi_den_2ranks_one_gpu_gloo.json
# graphs beyond config 3's size: more pdfs, more states (teams of 4 / 8), beyond every fast path; the stream kernels under the profiler
timeout 600 python3 tools/bench_big.py $P/$1_bench_big.json > /dev/null 2>&1
bash tools/measure_cmd.sh "$1" big10000 python3 $PWD/tools/bench_big.py 10000 1000 64 700 > gpurun_out/measure_big10000.log 2>&1
# the reference's linear algebra at the boundary (mm_spmv / mm_spmm), the alpha / beta export, ProbSemiring entries
timeout 300 python3 tools/bench_linalg.py $P/$1_bench_linalg.json > /dev/null 2>&1
bash tools/measure_cmd.sh "$1" linalg python3 $PWD/tools/bench_linalg.py > gpurun_out/measure_linalg.log 2>&1
timeout 300 python3 tools/bench_export.py $P/$1_bench_export.json > /dev/null 2>&1
timeout 300 python3 tools/bench_prob_mfma.py $P/$1_prob_mfma.json > /dev/null 2>&1
# host cost of a batch of new numerator graphs
timeout 300 python3 tools/host_cost.py $P/$1_host_cost.json > /dev/null 2>&1
# per-step cycle stamps (diagnostic build, if it was made: make -C markovmodels.jl_amd/csrc stamps)
if [ -f gpurun_stamps/libmarkovmodels_amd_stamps.so ]; then
  timeout 300 python3 tools/stamps_pairs.py > $P/$1_stamps_lfmmi_den.txt 2>/dev/null
  SHARP=1 timeout 300 python3 tools/stamps_pairs.py > $P/$1_stamps_wide.txt 2>/dev/null
  WL=wsj_den timeout 300 python3 tools/stamps_pairs.py > $P/$1_stamps_wsj_den.txt 2>/dev/null
  timeout 300 python3 tools/stamps_lane.py > $P/$1_stamps_ergodic64.txt 2>/dev/null
fi
ls $P | wc -l
