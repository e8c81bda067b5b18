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
        if s.is_cuda and dist.get_backend(group) == "gloo":  # (gloo reduces host tensors: the scalar takes the detour)
            h = s.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            s = h.to(s.device)
        else:
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
    dev = ttl_local.device
    host = ttl_local.is_cuda and dist.get_backend(group) == "gloo"  # (gloo gathers host tensors)
    pad = torch.full((m,), float("-inf"), dtype=ttl_local.dtype, device="cpu" if host else dev)
    pad[: ttl_local.numel()] = ttl_local
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)]).to(dev)


class RcclComm:
    """An RCCL communicator of this process for the C-ABI collectives (include/markovmodels_amd.h:
    mm_allreduce_logz / mm_allgather_ttl) -- the path a host binding without torch.distributed uses
    (julia/MarkovModelsAMD.jl).  The communicator is made with the RCCL that PyTorch has loaded already
    (ctypes on the process's own symbols), so the library, which resolves RCCL the same way, talks to the same one.

        uid = RcclComm.unique_id() on rank 0, sent to the other ranks by any means (from_torch() uses the
        initialised torch.distributed group); comm = RcclComm(world, rank, uid)
    """

    UID_BYTES = 128  # rccl.h: NCCL_UNIQUE_ID_BYTES

    @staticmethod
    def _rccl():
        import ctypes as C

        import torch  # noqa: F401  (loads librccl.so into the process)

        lib = C.CDLL(None)
        if not hasattr(lib, "ncclCommInitRank"):
            import os

            lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
        return lib

    @classmethod
    def unique_id(cls) -> bytes:
        import ctypes as C

        buf = C.create_string_buffer(cls.UID_BYTES)
        rc = cls._rccl().ncclGetUniqueId(buf)
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId failed ({rc})")
        return buf.raw

    def __init__(self, world: int, rank: int, uid: bytes):
        import ctypes as C

        class Uid(C.Structure):
            _fields_ = [("internal", C.c_char * self.UID_BYTES)]

        self._lib = self._rccl()
        from . import _lib

        _lib.check(_lib.lib.mm_set_rccl(C.c_void_p(self._lib._handle)))  # the engine's collectives use THIS RCCL
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        self.world, self.rank = int(world), int(rank)
        u = Uid()
        C.memmove(C.byref(u), uid, self.UID_BYTES)
        h = C.c_void_p()
        rc = self._lib.ncclCommInitRank(C.byref(h), self.world, u, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")
        self.handle = h

    @classmethod
    def from_torch(cls, group=None):
        """The same ranks as the initialised torch.distributed group (the id travels through that group)."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0])

    def close(self):
        import ctypes as C

        if getattr(self, "handle", None):
            self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self._lib.ncclCommDestroy(self.handle)
            self.handle = None

    def allreduce_logz(self, ttl_local):
        """sum over all ranks and utterances of ttl (float64, 0-dim tensor on the device): mm_allreduce_logz"""
        import torch

        from . import _lib

        t = ttl_local.contiguous().float()
        out = torch.empty(1, dtype=torch.float64, device=t.device)
        _lib.check(_lib.lib.mm_allreduce_logz(self.handle, t.data_ptr(), t.numel(), out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
        return out[0]

    def allgather_ttl(self, ttl_local, sizes: Sequence[int]):
        """the ttl of every utterance of every rank (shards of different sizes are padded to the largest): mm_allgather_ttl"""
        import torch

        from . import _lib

        m = max(sizes)
        pad = torch.full((m,), float("-inf"), dtype=torch.float32, device=ttl_local.device)
        pad[: ttl_local.numel()] = ttl_local
        out = torch.empty(self.world * m, dtype=torch.float32, device=ttl_local.device)
        _lib.check(_lib.lib.mm_allgather_ttl(self.handle, pad.data_ptr(), m, out.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
        return torch.cat([out[r * m : r * m + n] for r, n in enumerate(sizes)])
