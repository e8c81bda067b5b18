#!/usr/bin/env python3
"""which forms does a graph of config 3's family get?  S P [B] (MM_DEBUG=1 MM_VERBOSE=1 prints the packer's lines)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_DEBUG", "1")
os.environ.setdefault("MM_VERBOSE", "1")
import __graft_entry__ as ge
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
S, P = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
g = wl.lfmmi_denominator(S, P, seed=1)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
print(bf.kernels())
