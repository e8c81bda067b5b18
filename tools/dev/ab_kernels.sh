# per-kernel times (rocprofv3 averages) of bench.py on several builds of the library on ONE box: tools/dev/ab_kernels.sh <lib> ... [-- bench.py arguments]
# (exact kernels off -- MM_NO_REDO, MM_EXACT_FIRST=0 -- so that experiment builds with wrong results time their float32 kernels alone)
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$(realpath $1)"); shift; done; [ "$1" = "--" ] && shift
R=$PWD; cd /tmp; export TMPDIR=/tmp MM_DEBUG=1 MM_NO_REDO=1 MM_EXACT_FIRST=0 MM_BENCH_NOCHECK=1
for i in 1 2; do
  for L in "${LIBS[@]}"; do
    rm -rf /tmp/abk; MM_AMD_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/bench.py --steps 5 --warmup 2 --no-sharp --no-cpu-baseline "$@" > /tmp/abk.log 2>&1
    echo "== $(basename $L)"
    python3 - <<'PY'
import csv,glob
for f in glob.glob('/tmp/abk/*/*kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if 'mm_fb' in r[0] or 'mm_wave' in r[0] or 'mm_stream_k' in r[0]: print('   ', r[0][:52], r[1], round(float(r[3])/1e3,1), 'us')
PY
  done
done
