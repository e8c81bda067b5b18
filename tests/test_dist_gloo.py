"""The N > 1 path on CPU: world_size 2 over gloo.  Utterances shard across ranks
with no data-path collective; the one exchange is the total log-likelihood
all-reduce / ttl all-gather (markovmodels.jl_amd/dist.py).  The per-rank
"engine" here is the oracle (this is a test of the sharding + collectives, not
of the kernels)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import __graft_entry__ as ge
    import graphs

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    o, oc = ge.load_oracle()
    g = wl.random_fsm(15, 4, 2.5, seed=3)
    B, N = 7, 12
    rng = np.random.default_rng(0)
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = rng.integers(3, N + 1, B).astype(np.int32)
    lo, hi = mm.dist.shard_range(B, rank, world)
    _, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[lo:hi], lens[lo:hi], dtype=np.float64)
    ttl_local = torch.from_numpy(ttl).float()
    total = mm.dist.allreduce_logz(ttl_local)
    sizes = [mm.dist.shard_range(B, r, world)[1] - mm.dist.shard_range(B, r, world)[0] for r in range(world)]
    allttl = mm.dist.allgather_ttl(ttl_local, sizes)
    q.put((rank, float(total), allttl.numpy().tolist()))
    dist.destroy_process_group()


def _agg_worker(rank, world, port, q):
    """one rank of a `bench.py --gpus 4` run as far as its end-of-run reductions go (the engine needs a GPU; the numbers a
    rank brings to them do not)"""
    sys.path.insert(0, ROOT)
    import bench

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, N, steps = 256, 1500, 20
    elapsed = 0.060 + 0.004 * rank  # rank 3 is the slowest
    r = bench.reduce_over_ranks(dist, torch, "cpu", elapsed, B * N, 2.9 + 0.1 * rank, 0.05 * (rank + 1))
    q.put((rank, r, B * world, r["frames_total"] * steps / r["elapsed"]))
    dist.destroy_process_group()


def test_four_rank_bench_reductions():
    """What rank 0 of `bench.py --gpus 4` prints: global_batch = 4 B, value = (sum of the ranks' frames) x steps / (MAX of the
    ranks' elapsed times), and every rank's own kernel / all-reduce time next to it -- the reductions themselves, on four
    gloo ranks (the kernels need a GPU: tests/test_dist_gloo.py::test_bench_launches_its_own_ranks runs the whole program)."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_agg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, r, gb, value in got:
        assert gb == 4 * 256
        assert r["frames_total"] == 4 * 256 * 1500
        assert abs(r["elapsed"] - 0.072) < 1e-12 and len(r["kernel_ms"]) == 4
        assert np.allclose(r["kernel_ms"], [2.9, 3.0, 3.1, 3.2]) and np.allclose(r["allreduce_ms"], [0.05, 0.1, 0.15, 0.2])
        assert np.allclose(r["elapsed_per_rank"], [0.060, 0.064, 0.068, 0.072])
        assert abs(value - 4 * 256 * 1500 * 20 / 0.072) < 1e-3


def test_shard_helpers():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    mm = ge.load_package()
    for B, W in ((7, 2), (256, 8), (5, 8), (2048, 8)):
        rs = [mm.dist.shard_range(B, r, W) for r in range(W)]
        assert rs[0][0] == 0 and rs[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
    parts = mm.dist.shard_by_length([10, 1, 9, 2, 8, 3], 2)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([10, 1, 9, 2, 8, 3][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 1


def test_two_rank_logz_allreduce():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import __graft_entry__ as ge
    import graphs

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    # single-process reference
    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    o, oc = ge.load_oracle()
    g = wl.random_fsm(15, 4, 2.5, seed=3)
    rng = np.random.default_rng(0)
    V = rng.standard_normal((7, 12, g.P)).astype(np.float32)
    lens = rng.integers(3, 13, 7).astype(np.int32)
    _, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    for rank, total, allttl in res:
        assert np.isclose(total, ttl.sum(), rtol=1e-6)
        assert np.allclose(allttl, ttl, rtol=1e-6)


def _gpu_worker(rank, world, port, q):
    """One rank of a sharded pdfposteriors step with the HIP engine: its shard of the batch on the GPU (both ranks share the
    one GPU of the test box), the total log-likelihood over gloo."""
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as ge

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    g = wl.lfmmi_denominator(600, 40, seed=5)
    B, N = 11, 60
    rng = np.random.default_rng(0)
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = rng.integers(20, N + 1, B).astype(np.int32)
    lo, hi = mm.dist.shard_range(B, rank, world)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * (hi - lo)))
    gam, ttl = bf.pdfposteriors(torch.from_numpy(V[lo:hi]).cuda(), torch.from_numpy(lens[lo:hi]).cuda())
    total = mm.dist.allreduce_logz(ttl.cpu())
    sizes = [mm.dist.shard_range(B, r, world)[1] - mm.dist.shard_range(B, r, world)[0] for r in range(world)]
    allttl = mm.dist.allgather_ttl(ttl.cpu(), sizes)
    q.put((rank, float(total), allttl.numpy().tolist(), bf.kernels()[:20], float(gam.sum())))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_with_the_hip_engine():
    """The N > 1 path with the product on the device: world size 2 over gloo, every rank computes its shard with the HIP
    engine (pair kernels), the scalar exchange is the same code the RCCL job runs (dist.allreduce_logz / allgather_ttl).
    Against the oracle on the whole batch."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import __graft_entry__ as ge
    import graphs

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    o, oc = ge.load_oracle()
    g = wl.lfmmi_denominator(600, 40, seed=5)
    rng = np.random.default_rng(0)
    V = rng.standard_normal((11, 60, g.P)).astype(np.float32)
    lens = rng.integers(20, 61, 11).astype(np.int32)
    g_ref, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    for rank, total, allttl, kernels, gsum in res:
        assert "mm_fbp_kernel" in kernels
        assert np.isclose(total, ttl.sum(), rtol=1e-5)
        assert np.allclose(allttl, ttl, rtol=1e-5, atol=1e-4)
    assert np.isclose(sum(r[4] for r in res), g_ref.sum(), rtol=1e-4)


def _run_bench(extra_env, *argv, timeout=600):
    import subprocess

    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (what the driver's command looks like): the parent spawns the two
    ranks before it touches the GPU, both run the HIP engine on the one GPU of the box, the scalars go over gloo, and
    rank 0's JSON line is the last line of stdout -- BASELINE.json configs[3] (the batch sharded over ranks, logZ
    all-reduce) end to end, incl. the MAX / SUM reductions of the elapsed time and the frame count."""
    import json

    r = _run_bench({"MM_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2
    assert out["config"]["global_batch"] == 512
    assert out["scaling"] == "weak"
    # 2 ranks x 256 utterances x 1500 frames per step
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * 256 * 1500) < 1.0
    assert "roofline" in out and out["roofline"]["frac"] > 0


def test_bench_self_launch_reports_a_failed_rank():
    """No GPU here: both ranks die at their first device call.  The parent must come back with their exit status (not
    hang in a rendezvous, not print a JSON line)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run_bench({"MM_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=120)
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
