import sys, time, importlib; sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge, numpy as np, torch
mm=ge.load_package(); wl=importlib.import_module(mm.__name__+'.workloads')
g=wl.lexicon_fsm(5000,84,seed=0); B,N=128,1000
cf=mm.compile(wl.to_fsm(mm,g,semiring="tropical"), mm.statemap(g.state2pdf,g.P))
bf=mm.batch(*([cf]*B))
V=torch.randn(B,N,g.P,device="cuda")
for _ in range(2): bf.viterbi(V)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(3): p,s=bf.viterbi(V)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
print("viterbi cfg5: %.2f ms/call, %.3g frames/s"%(dt*1e3, B*N/dt), "paths ok", bool((p>=0).all()), float(s.mean()))
# numerator-style: distinct small graphs
gs=[wl.random_fsm(400+7*i, 84, 2.3, seed=i) for i in range(64)]
cfs=[mm.compile(wl.to_fsm(mm,x), mm.statemap(x.state2pdf,x.P)) for x in gs]
bn=mm.batch(*cfs); Vn=torch.randn(64,700,84,device="cuda")
for _ in range(2): bn.pdfposteriors(Vn)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(3): bn.pdfposteriors(Vn)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
print("numerator-style 64 distinct graphs ~600 states, N=700: %.2f ms/call, %.3g frames/s"%(dt*1e3, 64*700/dt))
g=wl.load_npz_graph('/root/repo/tests/golden/num_fsm_wsj.npz')
cf=mm.compile(wl.to_fsm(mm,g), mm.statemap(g.state2pdf,g.P)); bn=mm.batch(*([cf]*128)); Vn=torch.randn(128,700,84,device="cuda")
for _ in range(2): bn.pdfposteriors(Vn)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(3): gam,ttl=bn.pdfposteriors(Vn)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/3
print("wsj numerator graph x128, N=700: %.2f ms/call, %.3g frames/s"%(dt*1e3, 128*700/dt), float(ttl.mean()))
