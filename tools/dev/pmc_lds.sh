#!/bin/bash
# LDS counters of the pair kernels with / without MM_BANKOPT (one box): bash tools/dev/pmc_lds.sh
R=$(pwd); O=$R/gpurun_out/pmc_lds; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for V in 0 1; do
  if [ $V = 1 ]; then export MM_DEBUG=1 MM_BANKOPT=1; fi
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/v$V -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sharp > $O/v$V.log 2>&1
  python3 - $O/v$V <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "fbp" in k: print(sys.argv[1][-2:], k, {c: round(v / 1e6, 2) for c, v in d.items()})
PY
done
