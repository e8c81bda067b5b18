import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
den = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz"))
num = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
B=128
P=den.P
print("den.P", den.P, "num.P", num.P)
for which in ("numP","denP","after_den"):
    if which=="after_den":
        cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P)); bden = mm.batch(*([cden]*B))
    PP = num.P if which=="numP" else P
    for rep in range(4):
        nfs=[wl.to_fsm(mm,num) for _ in range(B)]
        t0=time.perf_counter(); c=mm.compile_many(nfs, mm.statemap(num.state2pdf, PP)); t1=time.perf_counter(); b=mm.batch(*c); t2=time.perf_counter()
        print(which, rep, "compile %.2f batch %.2f"%(1e3*(t1-t0),1e3*(t2-t1)), b.kernels()[:30])
