# team kernels with every poll of the mates' rows switched off (wrong results: the work alone), exact kernels off, per-kernel times
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in base nopoll; do
  export MM_DEBUG=1 MM_NO_REDO=1 MM_EXACT_FIRST=0
  if [ $cfg = nopoll ]; then export MM_SPLIT_SLEEP=520; else unset MM_SPLIT_SLEEP; fi
  for wl in wsj_den lfmmi_den4000 lfmmi_den6000; do
    rm -rf /tmp/tr_${cfg}_$wl
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_${cfg}_$wl -- python3 $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-sharp > /tmp/log_${cfg}_$wl 2>&1
    echo "== $cfg $wl"
    python3 - /tmp/tr_${cfg}_$wl <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'):
    for r in csv.reader(open(f)):
        if 'mm_fb' in r[0]: print('   ',r[0][:50], r[1], float(r[3])/1e3,'us')
PY
    tail -1 /tmp/log_${cfg}_$wl | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done
