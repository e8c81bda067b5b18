"""Multi-GPU: utterances are independent (batching is block diagonal:
src/fsmops.jl:28-36, src/inference.jl:28-36), so a batch shards over ranks with
no data-path collective.  The only exchange is the tiny total-log-likelihood
reduction the LF-MMI loss consumes (examples/test_cuda.jl:140-152 computes
ttl_num / ttl_den per utterance): an all-gather of ttl[B_local] and/or an
all-reduce of its sum, over torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).  One process per GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of B utterances: [lo, hi) for this rank (sizes differ by <= 1)."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_by_length(lens: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced split (longest-processing-time first): the work of an
    utterance is proportional to its number of frames."""
    order = sorted(range(len(lens)), key=lambda i: -int(lens[i]))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += int(lens[i])
    return [sorted(x) for x in out]


def allreduce_logz(ttl_local, group=None):
    """Sum over ALL utterances of log Z (accumulated in float64): one scalar
    all-reduce.  Returns a 0-dim float64 tensor on ttl_local's device."""
    import torch
    import torch.distributed as dist

    s = ttl_local.double().sum().reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s[0]


def allgather_ttl(ttl_local, sizes: Sequence[int], group=None):
    """Every rank gets the ttl of every utterance (ranks may hold different
    numbers of utterances: pad to the largest shard)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return ttl_local.clone()
    m = max(sizes)
    pad = torch.full((m,), float("-inf"), dtype=ttl_local.dtype, device=ttl_local.device)
    pad[: ttl_local.numel()] = ttl_local
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)])
