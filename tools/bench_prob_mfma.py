#!/usr/bin/env python3
"""The ProbSemiring emission GEMM on the matrix cores (mm_prob_emission_mfma_kernel) against the recursion kernel gathering
the emissions itself: a dense mixture state map C_hat (S1 x P1), B utterances of N frames, float32.

    python tools/bench_prob_mfma.py [out.json]                      # with the GEMM
    MM_GENERIC_NO_MFMA=1 python tools/bench_prob_mfma.py [out.json] # without (the switch is read once per process)
"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np
import scipy.sparse as sp
import torch
import ctypes as C
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")
inf = importlib.import_module(mm.__name__ + ".inference")
S, P, B, N = 500, 255, 64, 300
g = wl.random_fsm(S, P, 3.0, seed=1)
g.init_w, g.w, g.final_w = np.exp(g.init_w), np.exp(g.w), np.exp(g.final_w)
S1, P1, N1 = S + 1, P + 1, N + 1
rng = np.random.default_rng(0)
Cd = rng.random((S1, P1))
Cd[:S, P] = 0.0
Cd[S, :] = 0.0
Cd[S, P] = 1.0
Cd /= Cd.sum(1, keepdims=True)
Cm = mm.GeneralStateMap(sp.csr_matrix(Cd), "prob")
cf = mm.compile(wl.to_fsm(mm, g, "prob", np.float32), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.exp(0.3 * torch.randn(B, N1, P1, device="cuda"))
hm = C.c_void_p()
ip, ix, dv = (np.ascontiguousarray(Cm.indptr, dtype=np.int64), np.ascontiguousarray(Cm.indices, dtype=np.int64), np.ascontiguousarray(Cm.data, dtype=np.float64))
inf.check(L.lib.mm_statemap_create(inf.SEMIRING_ID["prob"], S1, P1, ix.shape[0], 8, 0, 8, ip.ctypes.data, ix.ctypes.data, dv.ctypes.data, C.byref(hm)))
maps = (C.c_void_p * B)(*([hm] * B))
for _ in range(3):
    gam, ttl = bf.pdfposteriors_ex(V, maps)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 10
for _ in range(K):
    gam, ttl = bf.pdfposteriors_ex(V, maps)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
flops = 2.0 * N1 * S1 * P1 * B
res = dict(what=f"ProbSemiring float32, S1 = {S1}, P1 = {P1} (dense C_hat), N1 = {N1}, B = {B}", kernels=bf.kernels_generic(), ms_per_call=ms,
           emission_gemm_flops=flops, finite=bool(torch.isfinite(gam).all().item()))
# ... and the same ProbSemiring{Float32} FSMs with their OWN one-hot state map and likelihoods of expand()'s form: the fast entry
# (mm_pdfposteriors_f32 on the library's log twins, round 6) instead of the generic one
Vf = torch.exp(0.3 * torch.randn(B, N, P, device="cuda"))
lens = torch.full((B,), N, dtype=torch.int32, device="cuda")
for _ in range(3):
    bf.pdfposteriors(Vf, lens)
torch.cuda.synchronize()
e0.record()
for _ in range(K):
    g2, t2 = bf.pdfposteriors(Vf, lens)
e1.record()
torch.cuda.synchronize()
Vh = torch.zeros(B, N1, P1, device="cuda")
Vh[:, :N, :P] = Vf
Vh[:, N, P] = 1.0
for _ in range(2):
    g3, t3 = bf.pdfposteriors_ex(Vh, None)
torch.cuda.synchronize()
e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e2.record()
for _ in range(3):
    g3, t3 = bf.pdfposteriors_ex(Vh, None)
e3.record()
torch.cuda.synchronize()
res["one_hot_map"] = dict(what="the FSMs' own one-hot state map, V_hat of expand()'s form", fast_entry_ms=e0.elapsed_time(e1) / K, fast_entry_kernels=bf.kernels(),
                          generic_entry_ms=e2.elapsed_time(e3) / 3, max_abs_diff_gamma=float((g2 - g3).abs().max()),
                          max_rel_diff_ttl=float(((t2 - t3).abs() / t3.abs().clamp_min(1e-30)).max()))
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
L.lib.mm_statemap_destroy(hm)
