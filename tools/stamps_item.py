"""Diagnostic: per-phase cycle sums of the item kernel forward loop on the WSJ numerator graph (needs the -DMM_STAMPS build)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
os.environ["MM_AMD_LIB"]=os.path.join(ROOT,"gpurun_stamps","libmarkovmodels_amd_stamps.so")
import __graft_entry__ as ge, torch
mm=ge.load_package(); wl=importlib.import_module(mm.__name__+".workloads"); L=importlib.import_module(mm.__name__+"._lib")
g=wl.load_npz_graph(os.path.join(ROOT,"tests","golden","num_fsm_wsj.npz")); B,N=128,700
if os.environ.get("S"):
    g=wl.lfmmi_denominator(int(os.environ["S"]), 84, seed=1); B,N=32,200
    os.environ["MM_DEBUG"]="1"; os.environ["MM_KERNEL"]="item"
cf=mm.compile(wl.to_fsm(mm,g), mm.statemap(g.state2pdf,g.P)); bf=mm.batch(*([cf]*B))
V=torch.randn(B,N,g.P,device="cuda")
print(bf.kernels())
bf.pdfposteriors(V); bf.pdfposteriors(V); torch.cuda.synchronize()
n=B*16*16; out=np.zeros(n,dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes=[C.c_void_p,C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data,n)==0
s=out.reshape(B,16,2,8).astype(np.float64)/N
for k,nm in enumerate(["partmax(M)","em/alpha store","for_items","part_put","barrier"]):
    x=s[:,:,0,k]; print(f"{nm:16s} per wave:", " ".join(f"{v:.0f}" for v in x.mean(0)))
