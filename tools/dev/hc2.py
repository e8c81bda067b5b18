import importlib, os, sys, time
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
sm = mm.statemap(g.state2pdf, g.P)
torch.zeros(1, device="cuda")
for rep in range(4):
    fs = [wl.to_fsm(mm, g) for _ in range(128)]
    t0=time.perf_counter(); cfs = mm.compile_many(fs, sm); t1=time.perf_counter()
    print("compile_many total %.2f ms" % (1e3*(t1-t0)), flush=True)
