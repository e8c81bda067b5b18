"""LF-MMI loss on top of the engine: the step right after the hot path in the reference's
caller (examples/test_cuda.jl:140-152): two pdfposteriors calls per mini-batch -- numerator
graphs (one per utterance) and the shared denominator graph -- and

    loss      = - sum_b (log Z_num[b] - log Z_den[b])
    d loss / d V[b, n, p] = gamma_den[b, n, p] - gamma_num[b, n, p]

(the derivative of log Z w.r.t. a log-likelihood is the pdf posterior).  `lfmmi_loss` is a
torch.autograd.Function over the device-resident log-likelihoods; in a data-parallel job the
per-rank losses are summed with `dist.allreduce_logz` (one scalar over RCCL).
"""
from __future__ import annotations

from typing import Optional


def _function():
    import torch

    class _LFMMI(torch.autograd.Function):
        @staticmethod
        def forward(ctx, V, num, den, lens):
            g_num, t_num = num.pdfposteriors(V.detach(), lens)
            g_den, t_den = den.pdfposteriors(V.detach(), lens)
            ctx.save_for_backward(g_den - g_num)
            ctx.mark_non_differentiable(t_num, t_den)
            loss = -(t_num.double() - t_den.double()).sum()
            return loss.to(V.dtype), t_num, t_den

        @staticmethod
        def backward(ctx, gl, _gn, _gd):
            (grad,) = ctx.saved_tensors
            return grad * gl, None, None, None

    return _LFMMI


def lfmmi_loss(V, num_batch, den_batch, lens: Optional["torch.Tensor"] = None):
    """V: [B, N, P] float32 log-likelihoods on the HIP device (requires_grad as needed);
    num_batch / den_batch: BatchedFSM of B utterances each (log semiring).
    Returns (loss, ttl_num[B], ttl_den[B]); utterances without an accepting numerator or
    denominator path have ttl = -inf and must be filtered by the caller.

    The gradient is a difference of posteriors: one below 1e-12 changes nothing.  A training loop says so once --
    `den_batch.set_posterior_floor(1e-12)` -- and the denominator stays on the fast kernels when the model's outputs get
    sharp (the default floor, 1e-30, sends such utterances to the exact kernels: 3-6x the time of a call).

    Host cost: a batch of 128 numerator graphs that are new to the engine takes ~30 ms to compile and batch (their kernel
    forms are packed on the host's cores), the call itself 0.4 ms: keep the CompiledFSM of an utterance across epochs --
    a batch of known FSMs only assembles descriptors."""
    return _function().apply(V, num_batch, den_batch, lens)
