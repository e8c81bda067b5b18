#!/usr/bin/env python3
"""Turn gpurun_out/measure/<tag>_<workload>/ (tools/measure.sh) into the tracked evidence:
    profiles/<tag>_bench_<workload>.json         the bench line of the same build
    profiles/<tag>_kernel_stats_<workload>.csv   rocprofv3 --kernel-trace --stats, engine kernels only
    profiles/<tag>_traffic_<workload>.json       HBM bytes per launch from the TCC counters, per kernel
    profiles/<tag>_pmc_<workload>.json           SQ counters per kernel + the derived stall / LDS ratios
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM): the read side is doubled, WRITE_SIZE is taken as is.
Every file carries the hash of the kernel sources it was measured on (tools/srchash.py)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from srchash import source_hash  # noqa: E402

tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "measure", f"{tag}_{workload}")
dst = os.path.join(ROOT, "gpurun_out", "measure", "profiles")  # merged back by gpurun; copied to profiles/ at home
os.makedirs(dst, exist_ok=True)
sha = source_hash()


def short(k):
    return k.replace("void mm::", "").replace("void ", "").replace("(mm::RunParams)", "").replace("(anonymous namespace)::", "")


def ours(k):
    """the engine's kernels (namespace mm, or the global-namespace launch wrappers mm_*)"""
    return "mm::mm_" in k or "mm::(anonymous namespace)::mm_" in k or k.startswith("void mm_") or k.startswith("mm_")


LAST_CALL = os.environ.get("LAST_CALL") == "1"


def counters(sub, last_call=False):
    """{kernel: {counter: mean per dispatch}} for the engine's kernels.  last_call: only the dispatches of the run's LAST engine call
    (it starts at the last dispatch of the kernel the run's first call started with) -- for inputs whose first call differs from the
    steady state: on sharp emissions the first call runs the float32 kernels and then the float64 ones for what they marked, every
    later call the wide kernels alone (PMC_WARMUP=2 LAST_CALL=1 in tools/measure_all.sh)."""
    acc, disp = {}, {}
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if ours(r["Kernel_Name"])]
        if last_call and rows:
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            start = max(int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == rows[0]["Kernel_Name"])
            rows = [r for r in rows if int(r["Dispatch_Id"]) >= start]
        for r in rows:
            k = short(r["Kernel_Name"])
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp.setdefault(k, set()).add(r["Dispatch_Id"])
    return {k: {c: v / len(disp[k]) for c, v in cs.items()} for k, cs in acc.items()}, {k: len(v) for k, v in disp.items()}


# (tools/measure_cmd.sh profiles a tool that is not bench.py: no bench line, the counters and the kernel statistics only)
bench = None
try:
    bench = json.loads(open(os.path.join(src, "bench.json")).read())
    bench["source_hash"] = sha
    json.dump(bench, open(os.path.join(dst, f"{tag}_bench_{workload}.json"), "w"), indent=1)
except (OSError, ValueError):
    pass

stats = {}
for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, f"{tag}_kernel_stats_{workload}.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(rows[0] + [f"source_hash={sha}"])
        for r in rows[1:]:
            if ours(r[0]):
                w.writerow(r)
                stats[short(r[0])] = {"calls": int(r[1]), "avg_ns": float(r[3])}

fetch, nf = counters("fetch", LAST_CALL)
write, nw = counters("write", LAST_CALL)
per_kernel = {}
for k in sorted(set(fetch) | set(write)):
    rd = 2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    per_kernel[k] = {"read_bytes_corrected": rd, "write_bytes": wr, "dispatches_measured": [nf.get(k, 0), nw.get(k, 0)]}
cfg = bench["config"] if bench else {}
traffic = {
    "workload": workload,
    "source_hash": sha,
    "B": cfg.get("global_batch"),
    "N": cfg.get("seq_len"),
    "per_kernel": per_kernel,
    "hbm_bytes_per_launch": sum(v["read_bytes_corrected"] + v["write_bytes"] for v in per_kernel.values()),
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"] if bench and "roofline" in bench else None,
    "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; one "
                  "pdfposteriors call = one dispatch of each kernel" + (" (the dispatches of the run's last call only: the steady state)" if LAST_CALL else ""),
}
if per_kernel:
    json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic_{workload}.json"), "w"), indent=1)

pa, _ = counters("pmcA")
pb, _ = counters("pmcB")
pmc = {"workload": workload, "source_hash": sha, "per_kernel": {}}
for k in sorted(set(pa) | set(pb)):
    c = dict(pa.get(k, {}))
    c.update(pb.get(k, {}))
    d = {}
    if c.get("SQ_WAVE_CYCLES"):
        d["wait_any_per_wave_cycle"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        d["wait_inst_any_per_wave_cycle"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_per_idx_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVES") and k in stats:
        d["avg_kernel_ns"] = stats[k]["avg_ns"]
    pmc["per_kernel"][k] = {"counters_per_dispatch": c, "derived": d}
if pmc["per_kernel"]:
    json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_{workload}.json"), "w"), indent=1)

print(json.dumps({"source_hash": sha, "kernel_avg_ns": {k: v["avg_ns"] for k, v in stats.items()},
                  "hbm_bytes_per_launch": traffic["hbm_bytes_per_launch"], "bench_ms": bench["ms_per_step"] if bench else None,
                  "frac": bench["roofline"]["frac"] if bench and "roofline" in bench else None}))
