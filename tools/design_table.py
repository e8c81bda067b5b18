#!/usr/bin/env python3
"""Prints the measurement table of DESIGN.md section 7 from the tracked profiles: python tools/design_table.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = [("lfmmi_den", "config 3 `lfmmi_den` (pair kernels; S = 2000, T = 1500, B = 256)"),
        ("wsj_den", "reference's WSJ denominator, 3032 states, B = 128, T = 700 (split pair kernels, 256 workgroups)"),
        ("wsj_num", "reference's WSJ numerator ×128 (T = 700; wave kernel)"),
        ("lexicon5000", "config 5 `lexicon5000` (Viterbi, T = 1000, B = 128; row-lane kernels)"),
        ("ergodic64", "config 2 `ergodic64` (dense 64-state HMM, T = 500, B = 32; pair kernels)"),
        ("l2r3", "config 1 `l2r3` (3-state left-to-right HMM, T = 100, B = 1: the reference's CPU-runnable plumbing case; wave kernel, one workgroup -- a latency)"),
        ("lfmmi_den4000", "`lfmmi_den4000`: config 3's family, 4000 states, 65 k arcs, B = 128, T = 700 (teams of 4)"),
        ("lfmmi_den6000", "`lfmmi_den6000`: 6000 states, 97 k arcs, 300 pdfs, B = 128, T = 700 (teams of 8, 5 pdf passes)"),
        ("lfmmi_den_p400", "`lfmmi_den_p400`: 2000 states, 33 k arcs, 400 pdfs, B = 256, T = 1500 (pair kernels, 8 pdf passes)")]
print("| workload | ms / call | frames/s | `roofline.frac` | HBM traffic (TCC counters) vs algorithmic | CPU port, all host cores / 1 thread | source hash |")
print("|---|---|---|---|---|---|---|")
for w, name in rows:
    b = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{w}.json")))
    t = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic_{w}.json")))
    c = b.get("cpu_baseline", {})
    fr = b['roofline']['frac']
    small = t['hbm_bytes_per_launch'] < 1e7
    div, unit = (1e3, "KB") if small else (1e9, "GB")
    print(f"| {name} | {b['ms_per_step']:.2f} | {b['value']:.3g} | {(f'{fr:.3f}' if fr >= 1e-3 else f'{fr:.0e}')} | "
          f"{t['hbm_bytes_per_launch'] / div:.2f} vs {t['algorithmic_bytes_per_launch'] / div:.2f} {unit} | "
          f"{c.get('value', 0):.3g} ({c.get('cores', '?')} cores) / {c.get('one_thread', {}).get('value', 0):.3g} | `{b['source_hash']}` |")
