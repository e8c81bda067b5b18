for B in 32 64 128; do
  for K in auto split; do
    if [ $K = auto ]; then unset MM_DEBUG MM_KERNEL; else export MM_DEBUG=1 MM_KERNEL=split; fi
    echo "B=$B K=$K: $(python bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["redo_utterances"], d["roofline"]["kernel"][:60])')"
  done
done
unset MM_DEBUG MM_KERNEL
echo "ergodic64: $(python bench.py --workload ergodic64 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200)"
