# GPU clock and package power while bench.py runs a long timed loop (is a kernel power-limited?):
#   bash tools/dev/smi_during_bench.sh <steps> [bench.py arguments]      (GPU box; rocm-smi read-only)
STEPS=$1; shift
python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-sharp "$@" > /tmp/smi_bench.log 2>&1 &
BP=$!
sleep 20   # (import torch, build the batch; the timed loop should run ~40 s)
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "sclk|Max Graphics|Current Socket" | sed -e 's/GPU\[0\]\t*: //' | tr '\n' ';'; echo; sleep 1
done
wait $BP; tail -1 /tmp/smi_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench:', d['config']['workload'][:60], 'ms_per_step', round(d['ms_per_step'],3), 'redo', d.get('redo_utterances'))"
