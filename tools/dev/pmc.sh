#!/bin/bash
# bash tools/dev/pmc.sh <tag> "<counters>" [bench args]: one --pmc pass over a 1-step bench run, per-kernel means on stdout
export TMPDIR=/tmp; R=$PWD; T=$1; C=$2; shift; shift; cd /tmp
rm -rf $R/gpurun_out/pmc_$T
rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$T -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$T.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$R/gpurun_out/pmc_$T/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mm_" not in k: continue
        k = k.replace("void ", "").replace("(mm::RunParams)", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, " ".join(f"{c}={v / len(n[k]):.4g}" for c, v in sorted(acc[k].items())))
PY
