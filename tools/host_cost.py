#!/usr/bin/env python3
"""Host cost of a batch of NEW numerator graphs (examples/test_cuda.jl:74-78 builds one every training step): 128 graphs of
the reference's WSJ numerator's size -> FSM objects -> compile -> batch -> the call.  `compile.(fsms)` graph by graph
(mm_fsm_create, one upload per form) against compile_many (mm_fsm_create_many: host threads, one allocation, one copy).

    python tools/host_cost.py [out.json]      # on the GPU box; the JSON goes to profiles/ (tools/measure_all.sh)
"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
B, N = 128, 700
V = torch.randn(B, N, g.P, device="cuda")
sm = mm.statemap(g.state2pdf, g.P)
res = {}
for mode in ("single", "many"):
    rows = []
    for rep in range(5):
        t0 = time.perf_counter()
        fs = [wl.to_fsm(mm, g) for _ in range(B)]
        t1 = time.perf_counter()
        cfs = mm.compile_many(fs, sm) if mode == "many" else [mm.compile(f, sm) for f in fs]
        t2 = time.perf_counter()
        bf = mm.batch(*cfs)
        t3 = time.perf_counter()
        gam, ttl = bf.pdfposteriors(V)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        gam, ttl = bf.pdfposteriors(V)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        bf2 = mm.batch(*cfs)
        t6 = time.perf_counter()
        rows.append(dict(fsm_objects_ms=1e3 * (t1 - t0), compile_ms=1e3 * (t2 - t1), batch_ms=1e3 * (t3 - t2), first_call_ms=1e3 * (t4 - t3),
                         second_call_ms=1e3 * (t5 - t4), batch_of_known_fsms_ms=1e3 * (t6 - t5)))
        del bf, bf2, cfs, fs
    best = {k: min(r[k] for r in rows[1:]) for k in rows[0]}
    best["compile_plus_batch_ms"] = min(r["compile_ms"] + r["batch_ms"] for r in rows[1:])
    res[mode] = best
    print(mode, {k: round(v, 2) for k, v in best.items()}, flush=True)
res["what"] = f"{B} new graphs of the reference's WSJ numerator ({g.S} states, {g.n_arcs} arcs), T = {N}; best of 4 repetitions after a warm-up; host = the GPU box"
res["kernels"] = "mm_wave_kernel"
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
