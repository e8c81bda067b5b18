#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv (kernel names can be kilobytes long)
into a committed summary.  usage: summarize.py <kernel_stats.csv> <out.csv>"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:100]
        w.writerow(r)
