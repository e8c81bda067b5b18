#!/usr/bin/env python3
"""Turn gpurun_out/traffic/{fetch,write,trace} into profiles/<tag>_traffic.json + a trimmed kernel-stats csv.
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM): the read side is doubled, WRITE_SIZE is taken as is.
usage: traffic_summary.py <tag> <workload> [B N]   (B, N: the configuration measured, default 256 1500)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
B, N = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (256, 1500)


def counter(sub, name):
    """Per kernel: KiB per dispatch.  One pdfposteriors call = one dispatch of each kernel (forward, backward)."""
    per_kernel = {}
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "traffic", sub, "*", "*counter_collection.csv")):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if "mm_" in r["Kernel_Name"] and r["Counter_Name"] == name:
                key = (r["Kernel_Name"], r["Dispatch_Id"])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
        for (k, _), v in per_dispatch.items():
            per_kernel.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in per_kernel.items()}, sum(len(v) for v in per_kernel.values())


fetch_k, nf = counter("fetch", "FETCH_SIZE")
write_k, nw = counter("write", "WRITE_SIZE")
fetch_kib, write_kib = sum(fetch_k.values()), sum(write_k.values())
per_kernel = {
    k[:60]: {"read_bytes_corrected": 2.0 * fetch_k.get(k, 0.0) * 1024, "write_bytes": write_k.get(k, 0.0) * 1024}
    for k in sorted(set(fetch_k) | set(write_k))
}
out = {
    "workload": workload,
    "B": B,
    "N": N,
    "kernel_dispatches_measured": [nf, nw],
    "per_kernel": per_kernel,
    "FETCH_SIZE_KiB_per_call": fetch_kib,
    "WRITE_SIZE_KiB_per_call": write_kib,
    "read_bytes_per_launch_corrected": 2.0 * fetch_kib * 1024,
    "write_bytes_per_launch": write_kib * 1024,
    "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024 + write_kib * 1024,
    "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is",
}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_traffic_{workload}.json"), "w"), indent=1)
print(json.dumps(out))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "traffic", "trace", "*", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{workload}.csv"), "w", newline="") as g:
        w = csv.writer(g)
        for r in rows:
            r[0] = r[0][:100]
            w.writerow(r)
    print("\n".join(",".join(r[:4]) for r in rows[:3]))
