#!/usr/bin/env python3
"""Hash of the kernel sources: profiles/*.json carry it, bench.py quotes a measured figure only when it matches
the sources it runs (a stale profile must never describe a newer kernel)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "markovmodels.jl_amd", "csrc", "*"))):
        if f.endswith((".hip", ".cpp", ".h")) or os.path.basename(f) == "Makefile":
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
