#!/usr/bin/env python3
"""bench.py -- pdfposteriors() frames/s on synthetic lattices (BASELINE.json metric).

One "step" = one pdfposteriors() call of the HIP engine over one batch of
utterances whose log-likelihoods are already resident in HBM.  Default workload
= BASELINE.json configs[2]: the LF-MMI denominator FSM (S = 2000, ~34 k arcs,
P = 84), T = 1500 frames, B = 256 utterances per GPU (weak scaling: configs[3]
is B = 2048 over 8 GPUs = 256 per GPU).  At N > 1 GPUs the utterances are
sharded (they are independent: block-diagonal batch) and each step ends with
the one real exchange of the path, the total-log-likelihood all-reduce (RCCL).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...        # N > 1 without a launcher: spawns one rank per GPU itself (self_launch)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    MM_BENCH_BACKEND=gloo python bench.py --gpus 2 ...   # ranks share the visible GPUs, scalars travel over gloo (tests)

Prints ONE JSON line on rank 0 (see the task contract), including
  "roofline":     algorithmic bytes per launch / measured kernel time vs 8 TB/s HBM
  "cpu_baseline": the C oracle (a port of the reference's CPU operation order;
                  the Julia reference cannot run here) timed on the host cores
                  on a bounded sample of the same workload.
  "sharp":        the same workload on the inputs of a trained acoustic model (log-softmax of 10 x N(0,1)), measured in the
                  same process after the timed region: ms_per_step, frac, redo_utterances, which kernels ran
  "per_rank":     (N > 1) every rank's kernel time per call and the time of its log Z all-reduce
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def make_workload(wl, name):
    """(graph, frames, utterances per GPU, semiring).  lfmmi_den = BASELINE.json configs[2] (the metric's
    configuration); lexicon5000 = configs[4] (Viterbi); ergodic64 = configs[1]; l2r3 = configs[0]; wsj_den / wsj_num = the reference's
    own benchmark graphs (misc/benchmark/README.md: T = 700, B = 128); lfmmi_den4000 / lfmmi_den6000 / lfmmi_den_p400 = graphs of
    config 3's family beyond config 3's size (teams of 4, teams of 8, 400 pdfs)."""
    if name == "lfmmi_den":
        return wl.lfmmi_denominator(2000, 84, seed=0), 1500, 256, "log"
    if name == "ergodic64":
        return wl.dense_ergodic(64, seed=0), 500, 32, "log"
    if name == "l2r3":  # configs[0]: the reference's CPU-runnable plumbing case (one utterance: a latency, not a throughput)
        return wl.l2r_hmm(3), 100, 1, "log"
    if name == "wsj_den":
        return wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 700, 128, "log"
    if name == "wsj_num":
        return wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz")), 700, 128, "log"
    if name == "lexicon5000":
        return wl.lexicon_fsm(5000, 84, seed=0), 1000, 128, "tropical"
    if name == "lfmmi_den4000":  # (config 3's graph family beyond the teams of two: teams of four workgroups)
        return wl.lfmmi_denominator(4000, 84, seed=1), 700, 128, "log"
    if name == "lfmmi_den6000":  # (6000 states, 97 k arcs, 300 pdfs: teams of eight workgroups, five pdf passes)
        return wl.lfmmi_denominator(6000, 300, seed=1), 700, 128, "log"
    if name == "lfmmi_den_p400":  # (config 3's size with 400 pdfs: the pair kernels' instances of eight pdf passes)
        return wl.lfmmi_denominator(2000, 400, seed=1), 1500, 256, "log"
    raise SystemExit(f"unknown workload {name}")


def measured_traffic(workload, B, N):
    """HBM bytes per launch from the TCC PMC counters (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction of
    MI355X_MICROARCH.md), collected by tools/measure.sh in separate --pmc passes and committed under
    profiles/ (rocprofv3 cannot wrap a bench run from inside).  Only quoted when the profile was taken on
    exactly the kernel sources that run now (tools/srchash.py) and on this configuration; otherwise null."""
    import glob

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from srchash import source_hash

    sha = source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{workload}.json")), reverse=True):
        t = json.load(open(f))
        if t.get("source_hash") == sha and (B, N) == (t.get("B"), t.get("N")):
            return t["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def algorithmic_bytes(g, B, N, lens_sum, semiring="log"):
    """SURVEY.md 8(d).  Forward-backward: alpha written once and read once, emissions read twice, posteriors
    written once, graph read once per pass (shared).  Viterbi: emissions read once, int32 back-pointers written
    once ((N+1)(4(P+1) + 4(S+1)) per utterance), the path written once (4N)."""
    S1, P1, A = g.S + 1, g.P + 1, g.n_arcs
    if semiring == "tropical":
        return B * ((N + 1) * (4 * P1 + 4 * S1) + 4 * N) + 4 * B + (8 * A + 4 * (g.S + 2))
    per_frame = 8 * S1 + 8 * P1
    return B * (N + 1) * per_frame + 4 * lens_sum * g.P + 4 * B + 2 * (8 * A + 4 * (g.S + 2))


def cpu_baseline(g, N, threads, budget_utts):
    """The oracle (kind "port"): C restatement of the reference's CPU path
    (CSC scatter SpMV, one logaddexp per arc, alpha and beta materialised,
    float32), OpenMP over utterances.  The per-thread scratch is allocated and
    first-touched outside the timed region (oracle/mm_oracle.c thread_ws)."""
    o, oc = ge.load_oracle()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import graphs

    of = graphs.to_oracle(o, g, "log", np.float32)
    rng = np.random.default_rng(123)
    V = rng.standard_normal((budget_utts, N, g.P)).astype(np.float32)
    oc.warm(g.S + 1, g.P, N, threads)
    oc.batch_shared(of, g.state2pdf, g.P, V[:1, : min(N, 20)], None, dtype=np.float32, nthreads=1)  # code warm
    t0 = time.perf_counter()
    oc.batch_shared(of, g.state2pdf, g.P, V, None, dtype=np.float32, nthreads=threads)
    dt = time.perf_counter() - t0
    return budget_utts * N / dt, dt


def julia_reference(g, N, B):
    """BASELINE.md plan item 1: the reference ITSELF as the CPU baseline when the box can run it -- `julia` on PATH with
    MarkovModels.jl (and its Semirings.jl / CUDA.jl dependencies) installed.  Returns (frames/s, seconds, note) or None;
    this image has no Julia, so the C port below is what normally runs (kind "port")."""
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("julia")
    if not exe:
        return None
    script = r"""
using MarkovModels, Semirings, SparseArrays, DelimitedFiles
K = LogSemiring{Float32}
d = ARGS[1]
arcs = readdlm(joinpath(d, "arcs.txt")); init = readdlm(joinpath(d, "init.txt")); fin = readdlm(joinpath(d, "final.txt"))
s2p = vec(Int.(readdlm(joinpath(d, "s2p.txt")))); S, P, N, B = parse.(Int, ARGS[2:5])
fsm = FSM([Int(r[1]) => K(r[2]) for r in eachrow(init)], [(Int(r[1]), Int(r[2])) => K(r[3]) for r in eachrow(arcs)],
          [Int(r[1]) => K(r[2]) for r in eachrow(fin)], collect(1:S))
C = sparse(1:S+1, vcat(s2p, P + 1), fill(one(K), S + 1), S + 1, P + 1)
fsms = rawunion([fsm for _ in 1:B]...)
Vs = [expand(K.(randn(Float32, P, N)), N) for _ in 1:B]
pdfposteriors(rawunion(fsm), [Vs[1][:, 1:21]], [C])
t = @elapsed pdfposteriors(fsms, Vs, [C for _ in 1:B])
println("SECONDS ", t)
"""
    try:
        with tempfile.TemporaryDirectory() as d:
            np.savetxt(os.path.join(d, "arcs.txt"), np.c_[g.src + 1, g.dst + 1, g.w])
            np.savetxt(os.path.join(d, "init.txt"), np.c_[g.init_idx + 1, g.init_w])
            np.savetxt(os.path.join(d, "final.txt"), np.c_[g.final_idx + 1, g.final_w])
            np.savetxt(os.path.join(d, "s2p.txt"), np.asarray(g.state2pdf) + 1, fmt="%d")
            open(os.path.join(d, "run.jl"), "w").write(script)
            r = subprocess.run([exe, os.path.join(d, "run.jl"), d, str(g.S), str(g.P), str(N), str(B)], capture_output=True, text=True,
                               timeout=600)
        for ln in r.stdout.splitlines():
            if ln.startswith("SECONDS "):
                dt = float(ln.split()[1])
                return B * N / dt, dt, f"julia {exe}: MarkovModels.pdfposteriors on the host, {B} utterances x {N} frames, Float32"
    except Exception as e:  # (no MarkovModels.jl in the depot, a different GraphSpec layout, a time-out: the port runs instead)
        print(f"[bench] julia found but the reference did not run: {e}", file=sys.stderr)
    return None


def host_cores():
    """CPU cores this process can actually use: the affinity mask, capped by the cgroup's CPU quota (a container
    can see 256 hardware threads and be allowed the time of 8: more threads than that only take turns)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): spawn the N ranks here -- one child
    process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what torch.distributed.run would set --
    BEFORE this process has touched the GPU (nothing below imports torch), relay the children's output and make rank
    0's JSON line the last line of stdout.  Exit code: the first non-zero one of the ranks."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    import threading

    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (RCCL across processes needs dmabuf IPC on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves the others waiting in a collective: end them (our own children, by PID) and report
    failed = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad and not failed:
            failed = bad[0]
            deadline = time.time() + 10.0
        if failed and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(10.0)
    codes = [p.returncode for p in procs]
    out0 = (chunks[0] if chunks else b"").decode(errors="replace")
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    json_line = None
    for ln in reversed(lines):
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            json_line = ln
            break
    for ln in lines:
        if ln is not json_line:
            print(ln, file=sys.stderr)
    rc = failed or next((c for c in codes if c), 0)
    if json_line is None and rc == 0:
        rc = 1
    sys.stderr.flush()
    if json_line is not None and rc == 0:
        print(json_line, flush=True)
    sys.exit(rc)


def reduce_over_ranks(dist, torch, rdev, elapsed, frames_local, kernel_ms, allreduce_ms):
    """What rank 0 reports for a run of `world` ranks: the step time is the MAX over ranks of the barrier-bracketed elapsed time,
    the work the SUM of the ranks' frames (value = sum frames x steps / max elapsed), plus every rank's own kernel time per
    call and the time of its log Z all-reduce so that a bad scaling curve can be attributed (a slow rank, a slow collective)."""
    world = dist.get_world_size()
    mine = torch.tensor([elapsed, float(frames_local), kernel_ms, allreduce_ms], device=rdev, dtype=torch.float64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    tab = torch.stack(rows).cpu().numpy()
    return {
        "elapsed": float(tab[:, 0].max()),
        "frames_total": int(round(tab[:, 1].sum())),
        "kernel_ms": [float(x) for x in tab[:, 2]],
        "allreduce_ms": [float(x) for x in tab[:, 3]],
        "elapsed_per_rank": [float(x) for x in tab[:, 0]],
    }


def sharp_record(torch, bf, g, B, N, gamma, lens, abytes, rank, Vs=None, label="log-softmax(10 x N(0,1))", workload=None, suffix="_peaky"):
    """The same workload on the inputs a TRAINED acoustic model produces, measured in this process AFTER the timed headline with
    the engine's own policy (mm_batch_set_exact_policy auto: the second call on hard inputs skips the float32 kernels).  Two
    flavours (examples/test_cuda.jl:124-143 feeds network outputs):
      "sharp"             log-softmax of 10 x N(0,1): sharp and INCONSISTENT with the graph (a random arg-max sequence is not a
                          path) -- the forward and the backward mass of a frame overlap ~160 log2 below their maxima, beyond
                          float32's exponent range: the wide-exponent kernels do the work;
      "sharp_consistent"  log-softmax(10 x (onehot(pdf of a path sampled from the graph) + 0.3 N(0,1))): sharp ALONG a path, as a
                          trained model's outputs are -- the two masses meet on the path."""
    if Vs is None:
        gen = torch.Generator(device="cuda").manual_seed(5000 + rank)
        Vs = torch.log_softmax(10.0 * torch.randn(B, N, g.P, device="cuda", generator=gen), dim=-1)
    for _ in range(4):
        bf.pdfposteriors(Vs, lens, out=gamma)
        torch.cuda.synchronize()  # (the policy reads the last FINISHED call's marks)
    K = 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        ttl = bf.pdfposteriors(Vs, lens, out=gamma)[1]
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K
    kms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    traffic, traffic_src = measured_traffic(workload + suffix, B, N) if workload else (None, None)
    return {
        "emissions": label,
        "steps": K,
        "ms_per_step": 1e3 * wall,
        "kernel_ms": kms,
        "value": float(lens.sum().item()) / wall,
        "unit": "frames/s",
        "frac": abytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_src,
        "redo_utterances": bf.last_redo_count(),
        "fallback_utterances": bf.last_fallback_count(),
        "exact_first": bool(bf.last_exact_first()),
        "kernels": bf.kernels("log"),
        "finite": bool(torch.isfinite(ttl).all().item()),
    }


def lfmmi_step_main(args):
    """`--workload lfmmi_step`: the caller's step (examples/test_cuda.jl:128-152) -- the reference's WSJ denominator x B + B WSJ
    numerators (one compiled graph per utterance, as a training step brings them), `lfmmi_loss` forward + backward on device-resident
    log-likelihoods -- next to its parts: the denominator call alone, the numerator call alone, the three assemblies of
    gamma_den - gamma_num (lfmmi.py: fused / serial / concurrent) and the fused step replayed from ONE hipGraph."""
    import torch

    torch.cuda.set_device(0)
    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    lf = importlib.import_module(mm.__name__ + ".lfmmi")
    den = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz"))
    num = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
    N = args.frames or 700
    B = args.batch or 128
    P = den.P
    cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))
    bden = mm.batch(*([cden] * B))
    # B handles, one graph per utterance -- compiled five times from new FSM objects: the first calls of a process pay its one-off costs
    # (the packer's thread pool, first-touch pages of the pinned staging buffer, the allocator's first device blocks: 190, 17, 60 ms
    # measured), the later ones are what every step of a training run pays (tools/dev/host_cost_probe.py; tools/host_cost.py)
    # (the graphs of the step before are released before the clock starts, as a training loop's are when its step ends: the library
    # then reuses their device blocks -- with all five sets kept alive every call allocates fresh device memory: 11-15 ms)
    host_ms = []
    cnums = bnum = None
    for _ in range(5):
        nfs = [wl.to_fsm(mm, num) for _ in range(B)]
        cnums = bnum = None
        t0 = time.perf_counter()
        cnums = mm.compile_many(nfs, mm.statemap(num.state2pdf, P))
        bnum = mm.batch(*cnums)
        host_ms.append(1e3 * (time.perf_counter() - t0))
    if args.posterior_floor > 0:
        bden.set_posterior_floor(args.posterior_floor)
    bden.reserve(N)
    bnum.reserve(N)
    gen = torch.Generator(device="cuda").manual_seed(1000)
    V = torch.randn(B, N, P, device="cuda", generator=gen)
    if args.emissions != "randn":
        V = torch.log_softmax(10.0 * V, dim=-1)
    lens = torch.full((B,), N, device="cuda", dtype=torch.int32)
    buf = torch.empty(B, N, P, device="cuda")
    frames = B * N

    def timed(fn, K=args.steps, W=args.warmup):
        for _ in range(W):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / K
        return wall, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    Vg = V.clone().requires_grad_(True)

    def full_step():
        Vg.grad = None
        loss, _, _ = mm.lfmmi_loss(Vg, bnum, bden, lens)
        loss.backward()

    den_ms = timed(lambda: bden.pdfposteriors(V, lens, out=buf))
    num_ms = timed(lambda: bnum.pdfposteriors(V, lens, out=buf))
    parts = {m: timed(lambda m=m: lf.posteriors_difference(V, bnum, bden, lens, m, out=buf)) for m in ("fused", "serial", "concurrent")}
    step_ms = timed(full_step)
    # the fused step from one hipGraph
    bden.set_exact_policy("f32_first")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        lf.posteriors_difference(V, bnum, bden, lens, "fused", out=buf)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lf.posteriors_difference(V, bnum, bden, lens, "fused", out=buf)
    graph_ms = timed(graph.replay)
    f32_ms = timed(lambda: lf.posteriors_difference(V, bnum, bden, lens, "fused", out=buf))  # (the same launches, eager)
    bden.set_exact_policy("auto")
    out = {
        "metric": "lfmmi_step_frames_per_sec",
        "value": frames / (step_ms[0] * 1e-3),
        "unit": "frames/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_ms[0],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if args.emissions == "randn" else f"synthetic ({args.emissions} emissions)",
        "config": {"workload": f"lfmmi_step: WSJ denominator (S={den.S}, {den.n_arcs} arcs) x {B} + {B} WSJ numerators (S={num.S}, one compiled graph per "
                               f"utterance), P={P}, T={N}, lfmmi_loss forward + backward (examples/test_cuda.jl:128-152)",
                   "global_batch": B, "seq_len": N, "parallelism": "one GPU"},
        # wall = host clock per call (launch-bound steps show here), kernel = HIP events around the call on its stream
        "step": {"wall_ms": step_ms[0], "kernel_ms": step_ms[1], "what": "lfmmi_loss(mode='auto') forward + loss.backward()"},
        "den_ms": {"wall_ms": den_ms[0], "kernel_ms": den_ms[1], "kernels": bden.kernels()},
        "num_ms": {"wall_ms": num_ms[0], "kernel_ms": num_ms[1], "kernels": bnum.kernels()},
        "difference_ms": {m: {"wall_ms": v[0], "kernel_ms": v[1]} for m, v in parts.items()},
        "grad_ms": {"kernel_ms": step_ms[1] - parts["fused"][1], "what": "forward + backward minus the fused difference: the loss scalar and grad * grad_output"},
        "hipgraph": {"replay_wall_ms": graph_ms[0], "replay_kernel_ms": graph_ms[1], "eager_wall_ms": f32_ms[0], "eager_kernel_ms": f32_ms[1],
                     "what": "posteriors_difference(mode='fused'), exact policy f32_first, captured once and replayed"},
        "ratio_step_to_den": step_ms[1] / den_ms[1],
        "ratio_fused_difference_to_den": parts["fused"][1] / den_ms[1],
        "redo_utterances": bden.last_redo_count(),
        "host_ms_compile_and_batch_numerators": min(host_ms[2:]),
        "host_ms_compile_and_batch_numerators_first_calls_of_the_process": host_ms[:2],
    }
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="lfmmi_den")
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--varlen", action="store_true", help="lengths U[N/2, N] instead of all N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharp", action="store_true", help="skip the `sharp` sub-record (the same workload on sharp emissions, after the timed region)")
    ap.add_argument("--emissions", default="randn", choices=["randn", "peaky", "peaky_offset", "consistent"],
                    help="randn: N(0,1) log-likelihoods (default); peaky: log-softmax of sigma x N(0,1) (sharp, inconsistent with the graph); "
                         "peaky_offset: the same shifted by -300 nats (GMM-like scores); consistent: log-softmax(sigma x (onehot(pdf of a "
                         "path sampled from the graph) + 0.3 N(0,1))) (sharp along a path: a trained acoustic model)")
    ap.add_argument("--sigma", type=float, default=10.0, help="sharpness of the peaky / consistent emissions")
    ap.add_argument("--posterior-floor", type=float, default=0.0,
                    help="mm_batch_set_posterior_floor (default: the library's 1e-30); 1e-12 keeps sharp emissions on the fast kernels")
    args = ap.parse_args()
    if args.workload == "lfmmi_step":
        return lfmmi_step_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # (does not return)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # MM_BENCH_BACKEND=gloo: the ranks share whatever GPUs are visible (two ranks on the one GPU of a test box) and the
    # scalar exchange goes over gloo -- the N > 1 code path without N GPUs (tests/test_dist_gloo.py).  Default: RCCL.
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("MM_BENCH_FORCE_DIST") == "1"  # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    g, N, B, semiring = make_workload(wl, args.workload)
    N = args.frames or N
    B = args.batch or B
    cf = mm.compile(wl.to_fsm(mm, g, semiring), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    if args.posterior_floor > 0 and semiring == "log":
        bf.set_posterior_floor(args.posterior_floor)
    gen = torch.Generator(device="cuda").manual_seed(1000 + rank)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    if args.emissions == "consistent":
        V = torch.from_numpy(wl.path_consistent_emissions(g, B, N, args.sigma, seed=1000 + rank)).cuda()
    elif args.emissions != "randn":
        V = torch.log_softmax(args.sigma * V, dim=-1) - (300.0 if args.emissions == "peaky_offset" else 0.0)
    if args.varlen:
        lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    else:
        lens = torch.full((B,), N, device="cuda", dtype=torch.int32)
    frames_local = int(lens.sum().item())
    gamma = torch.empty(B, N, g.P, device="cuda") if semiring == "log" else None

    def call():
        """one pass of the hot path over the batch: pdfposteriors (log semiring) or bestpath (tropical)"""
        if semiring == "log":
            return bf.pdfposteriors(V, lens, out=gamma)[1]
        return bf.viterbi(V, lens)[1]

    def step():
        ttl = call()
        return mm.dist.allreduce_logz(ttl) if use_dist else ttl

    if semiring == "log":
        bf.team_xcd_stats()  # (team kernels: switches their same-XCD counter on; read after the timed steps)
    for _ in range(args.warmup):
        step()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        ttl = call()
        ev[i][1].record()
        if use_dist:
            mm.dist.allreduce_logz(ttl)
            ev[i][2].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    # (over gloo the reduction is a host round trip inside allreduce_logz: the events then bracket the device's idle time)
    allreduce_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if use_dist else 0.0
    frames_total = frames_local
    per_rank = None
    if use_dist:
        per_rank = reduce_over_ranks(dist, torch, "cpu" if backend == "gloo" else "cuda", elapsed, frames_local, kernel_ms, allreduce_ms)
        elapsed, frames_total = per_rank["elapsed"], per_rank["frames_total"]
    assert os.environ.get("MM_BENCH_NOCHECK") or torch.isfinite(ttl).all(), "non-finite log-likelihoods"
    # utterances of the last call that the fast kernels handed to the exact ones (flag and redo): they were computed twice
    redo = bf.last_redo_count() if semiring == "log" else 0
    # team kernels: how many of the launches' workgroups found their whole team on one XCD (warm-up + timed steps)
    teams_xcd = bf.team_xcd_stats() if semiring == "log" else (0, 0)

    sharp = sharp_consistent = None
    if semiring == "log" and args.emissions == "randn" and not args.no_sharp and args.posterior_floor <= 0:
        ab = algorithmic_bytes(g, B, N, frames_local, semiring)
        wname = None if (args.varlen or args.batch or args.frames) else args.workload
        sharp = sharp_record(torch, bf, g, B, N, gamma, lens, ab, rank, workload=wname)
        try:  # (left-to-right graphs have no accepting path of an arbitrary length to sample: no record for them)
            Vc = torch.from_numpy(wl.path_consistent_emissions(g, B, N, 10.0, seed=7000 + rank)).cuda()
        except ValueError:
            Vc = None
        if Vc is not None and not args.varlen:
            sharp_consistent = sharp_record(torch, bf, g, B, N, gamma, lens, ab, rank, Vs=Vc, workload=wname, suffix="_consistent",
                                            label="log-softmax(10 x (onehot(pdf of a sampled path) + 0.3 N(0,1)))")
            # ... and what an LF-MMI step pays on those inputs: to a gradient, posteriors below 1e-12 are zero (lfmmi.py:
            # mm_batch_set_posterior_floor(1e-12)) -- a fresh batch with that floor keeps them on the float32 kernels
            bfl = mm.batch(*([cf] * B))
            bfl.set_posterior_floor(1e-12)
            rec = sharp_record(torch, bfl, g, B, N, gamma, lens, ab, rank, Vs=Vc, label="the same, mm_batch_set_posterior_floor(1e-12)")
            sharp_consistent["posterior_floor_1e-12"] = {k: rec[k] for k in ("ms_per_step", "kernel_ms", "frac", "redo_utterances", "exact_first")}
            del bfl
    if rank == 0:
        abytes = algorithmic_bytes(g, B, N, frames_local, semiring)
        achieved = abytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, None) if args.varlen else measured_traffic(args.workload, B, N)
        out = {
            "metric": "pdfposteriors_frames_per_sec" if semiring == "log" else "bestpath_frames_per_sec",
            "value": frames_total * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic" if args.emissions == "randn" else f"synthetic ({args.emissions} emissions, sigma {args.sigma:g})")
                    + (f", posterior floor {args.posterior_floor:g}" if args.posterior_floor > 0 else ""),
            "redo_utterances": redo,
            **({"teams_on_one_xcd": {"workgroups": teams_xcd[0], "of": teams_xcd[1], "share": teams_xcd[0] / teams_xcd[1]}} if teams_xcd[1] else {}),
            "config": {
                "workload": f"{g.name}: S={g.S} states, {g.n_arcs} arcs, P={g.P} pdfs, T={N} frames, "
                            f"B={B} utterances/GPU, {semiring} semiring, shared graph"
                            + (", lengths U[T/2,T]" if args.varlen else ""),
                "global_batch": B * world,
                "seq_len": N,
                "parallelism": f"utterance-sharded x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": bf.kernels(semiring) + " (all launches of one call; HIP events around the call)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": abytes,
                "kernel_ms": kernel_ms,
            },
        }
        if sharp is not None:
            out["sharp"] = sharp
        if sharp_consistent is not None:
            out["sharp_consistent"] = sharp_consistent
        if per_rank is not None:
            # every rank's own numbers: kernel time per call (HIP events), the log Z all-reduce behind it, the barrier-bracketed wall
            out["per_rank"] = {
                "kernel_ms": per_rank["kernel_ms"], "kernel_ms_min": min(per_rank["kernel_ms"]), "kernel_ms_max": max(per_rank["kernel_ms"]),
                "allreduce_ms": per_rank["allreduce_ms"], "allreduce_ms_max": max(per_rank["allreduce_ms"]),
                "elapsed_s": per_rank["elapsed_per_rank"], "backend": backend,
            }
        if semiring == "tropical":
            # SURVEY 8(d)'s figure counts int32 back-pointers; the row-lane kernels write ONE byte per state and frame (rows padded to
            # 256 bytes) and read it back once: what the kernel actually moves, next to what the problem is priced at
            kbytes = B * ((N + 1) * (4 * (g.P + 1) + 2 * (((g.S + 2) + 255) // 256 * 256)) + 4 * N)
            out["roofline"]["kernel_bytes_per_launch"] = kbytes
            out["roofline"]["frac_of_peak_on_kernel_bytes"] = kbytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["note"] = ("frac = SURVEY 8(d)'s algorithmic bytes (int32 back-pointers) / time / peak; frac_of_peak_on_kernel_bytes = "
                                       "what the byte-back-pointer kernels move (emissions in, one byte per state and frame out and back in, the path)")
        # what actually bounds the kernels: one random 4-byte LDS read per arc and pass (32 lanes per clock
        # and CU without bank conflicts; MI355X_MICROARCH.md: 256 CUs, 128 B/clk/CU LDS, 2.4 GHz)
        lds_bytes = (2 if semiring == "log" else 1) * 4 * g.n_arcs * frames_local
        out["lds_gather_roofline"] = {
            "bound": "lds",
            "achieved": lds_bytes / (kernel_ms * 1e-3) / 1e9,
            "peak": 256 * 128 * 2.4,
            "unit": "GB/s",
            "frac": lds_bytes / (kernel_ms * 1e-3) / 1e9 / (256 * 128 * 2.4),
            "note": "gathered LDS bytes only (arcs x 4 B x 2 passes); informative, the graded roofline is the HBM one",
        }
        if not args.no_cpu_baseline:
            # (1) one thread: faithful to the reference, whose CPU path is serial (src/inference.jl:62-110);
            # (2) all host hardware threads, OpenMP over utterances: the "whole node" denominator.
            cores = host_cores()
            n1 = max(1, min(4, int(round(6000 / max(N, 1)))))  # ~10 s at ~600 frames/s
            v1, dt1 = cpu_baseline(g, N, 1, n1)
            per_thread = max(2, int(round(15.0 * v1 / max(N, 1))))  # ~15 s of work per thread at the one-thread rate
            nb = per_thread * cores
            v, dt = cpu_baseline(g, N, cores, nb)
            ref = julia_reference(g, N, n1) if semiring == "log" else None
            if ref is not None:
                out["cpu_baseline_reference"] = {"value": ref[0], "unit": "frames/s", "cores": 1, "kind": "reference", "sample": ref[2]}
            out["cpu_baseline"] = {
                "value": v,
                "unit": "frames/s",
                "cores": cores,
                "kind": "port",
                "julia": "ran: see cpu_baseline_reference" if ref is not None else "not on this box (shutil.which('julia') is None) or MarkovModels.jl missing: the C port",
                "sample": f"{nb} utterances x {N} frames of the same workload, float32, OpenMP over utterances "
                          f"({per_thread} per thread), {dt:.1f} s",
                "one_thread": {
                    "value": v1,
                    "unit": "frames/s",
                    "cores": 1,
                    "sample": f"{n1} utterances x {N} frames, float32, serial like the reference, {dt1:.1f} s",
                },
                # per-thread throughput at full width relative to one thread alone (SMT siblings share a core)
                "per_thread_ratio": v / cores / v1,
            }
            if v < 0.5 * cores * v1:
                print(f"[bench] warning: all-cores CPU baseline scales to {v / v1:.1f}x on {cores} threads",
                      file=sys.stderr)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line of stdout: RCCL writes a banner through C stdio, which is
        # block-buffered on a pipe and would otherwise be flushed at exit, after our line
        import ctypes

        sys.stdout.flush()
        sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
