#!/usr/bin/env python3
"""Sum the counters of tools/pmc_cmd.sh per kernel name (substring filter as argv[1])."""
import collections, csv, glob, os, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for d in ("pmcA", "pmcB"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[-1])):
        if pat in r["Kernel_Name"]:
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], k) not in seen:
                seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
    for k, v in acc.items():
        print(d, k, "dispatches", cnt[k])
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {x / cnt[k]:.4g} per dispatch")
