"""LF-MMI loss on top of the engine: the step right after the hot path in the reference's
caller (examples/test_cuda.jl:140-152): two pdfposteriors calls per mini-batch -- numerator
graphs (one per utterance) and the shared denominator graph -- and

    loss      = - sum_b (log Z_num[b] - log Z_den[b])
    d loss / d V[b, n, p] = gamma_den[b, n, p] - gamma_num[b, n, p]

(the derivative of log Z w.r.t. a log-likelihood is the pdf posterior).  `lfmmi_loss` is a
torch.autograd.Function over the device-resident log-likelihoods; in a data-parallel job the
per-rank losses are summed with `dist.allreduce_logz` (one scalar over RCCL).

How the step is put together (`mode`):
  "fused"       denominator call into the gradient buffer, then the numerator call with
                mm_batch_set_gamma_mode(accumulate, scale = -1) INTO the same buffer: gamma_den - gamma_num is there when the
                numerator kernel ends, no third pass (numerator batches of the wave kernel: LF-MMI numerators are)
  "concurrent"  the numerator call on a side stream beside the denominator call, each into its own buffer, one in-place
                subtraction behind the join -- pays where the denominator's grid leaves compute units free (small batches)
  "serial"      both calls on the caller's stream, then the subtraction (what round 5 did; any numerator batch)
  "auto"        "fused" when the numerator batch supports it, else "serial"
Everything is launches on the caller's stream (and, for "concurrent", one side stream forked from and joined into it by events):
the whole step can be captured in a hipGraph once both workspaces are sized (`reserve`).
"""
from __future__ import annotations

from typing import Optional

_SIDE = {}


def _side_stream(torch, device):
    s = _SIDE.get(device)
    if s is None:
        s = _SIDE[device] = torch.cuda.Stream(device=device)
    return s


def posteriors_difference(V, num_batch, den_batch, lens=None, mode: str = "auto", out=None):
    """(gamma_den - gamma_num)[B, N, P], ttl_num[B], ttl_den[B] for device-resident V: the two engine calls of an LF-MMI step and
    the gradient they give, without autograd."""
    import torch

    from ._lib import MarkovModelsAMDError

    if mode not in ("auto", "fused", "concurrent", "serial"):
        raise ValueError(f"mode {mode!r}")
    B, N, P = V.shape
    grad = out if out is not None else torch.empty((B, N, P), dtype=torch.float32, device=V.device)
    fused = mode in ("auto", "fused")
    if fused:
        try:
            num_batch.set_gamma_mode(True, -1.0)
        except MarkovModelsAMDError:
            if mode == "fused":
                raise
            fused = False
    if fused:
        try:
            _, t_den = den_batch.pdfposteriors(V, lens, out=grad)
            _, t_num = num_batch.pdfposteriors(V, lens, out=grad)  # grad += -1 * gamma_num
        finally:
            num_batch.set_gamma_mode(False, 1.0)
        return grad, t_num, t_den
    if mode == "concurrent":
        cur = torch.cuda.current_stream(V.device)
        side = _side_stream(torch, V.device)
        side.wait_stream(cur)  # (V is produced on the caller's stream)
        with torch.cuda.stream(side):
            g_num, t_num = num_batch.pdfposteriors(V, lens)
        _, t_den = den_batch.pdfposteriors(V, lens, out=grad)
        cur.wait_stream(side)
        g_num.record_stream(cur)
        t_num.record_stream(cur)
        grad.sub_(g_num)
        return grad, t_num, t_den
    g_num, t_num = num_batch.pdfposteriors(V, lens)
    _, t_den = den_batch.pdfposteriors(V, lens, out=grad)
    grad.sub_(g_num)
    return grad, t_num, t_den


def _function():
    import torch

    class _LFMMI(torch.autograd.Function):
        @staticmethod
        def forward(ctx, V, num, den, lens, mode):
            grad, t_num, t_den = posteriors_difference(V.detach(), num, den, lens, mode)
            ctx.save_for_backward(grad)
            ctx.mark_non_differentiable(t_num, t_den)
            loss = -(t_num.double() - t_den.double()).sum()
            return loss.to(V.dtype), t_num, t_den

        @staticmethod
        def backward(ctx, gl, _gn, _gd):
            (grad,) = ctx.saved_tensors
            return grad * gl, None, None, None, None

    return _LFMMI


def lfmmi_loss(V, num_batch, den_batch, lens: Optional["torch.Tensor"] = None, mode: str = "auto"):
    """V: [B, N, P] float32 log-likelihoods on the HIP device (requires_grad as needed);
    num_batch / den_batch: BatchedFSM of B utterances each (log semiring).
    Returns (loss, ttl_num[B], ttl_den[B]); utterances without an accepting numerator or
    denominator path have ttl = -inf and must be filtered by the caller.

    The gradient is a difference of posteriors: one below 1e-12 changes nothing.  A training loop says so once --
    `den_batch.set_posterior_floor(1e-12)` -- and the denominator stays on the float32 kernels when the model's outputs get
    sharp (the default floor, 1e-30, sends such utterances to the wide-exponent kernels: ~1.4x the time of a call; with the
    floor, emissions as sharp as log-softmax(10 x) along a path of the graph mark nothing: profiles/r06_sharpness_floor1e-12.txt).

    Host cost: a batch of 128 numerator graphs that are new to the engine takes ~3.8 ms to compile and batch
    (`compile_many`: packed on the host's cores, one allocation, one copy; profiles/r04_host_cost.json), the numerator call
    itself 0.44 ms at T = 700: keep the CompiledFSM of an utterance across epochs -- a batch of known FSMs only assembles
    descriptors (0.1 ms).  Measured step: `bench.py --workload lfmmi_step` (profiles/r06_bench_lfmmi_step*.json)."""
    return _function().apply(V, num_batch, den_batch, lens, mode)
