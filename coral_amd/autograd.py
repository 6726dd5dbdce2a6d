"""Autograd glue: `model(**batch).loss.backward()` drives the engine's hand-written backward.

The contract `transformers.Trainer` relies on ($TF/trainer.py:2005,2038) is that
`model(**batch)["loss"]` is a scalar tensor with autograd.  The engines compute their own backward
(a fixed sequence of HIP kernels into the flat fp32 gradient buffer), so the loss is tied to a
`torch.autograd.Function` whose backward calls `engine.backward(loss_scale=<incoming gradient>)`:
`loss.backward()`, `(loss / accum).backward()` and `accelerator.backward(loss)` all work, the
incoming gradient staying on the device (no host synchronisation).  PyTorch is plumbing here: no
arithmetic of the hot path runs through autograd.

The native training loop (`coral_amd.trainer.DataParallelTrainer`) calls `engine.backward(...)`
directly (it passes per-bucket hooks for the gradient all-reduce); both routes fill the same buffer.
"""

from __future__ import annotations

import torch


class _EngineBackward(torch.autograd.Function):
    """loss (no graph) -> loss (with graph): the backward runs the owner's engine backward once."""

    @staticmethod
    def forward(ctx, anchor, loss, owner):
        ctx.owner = owner
        ctx.token = owner._autograd_token
        return loss.detach().view(loss.shape)

    @staticmethod
    def backward(ctx, grad_out):
        owner = ctx.owner
        if ctx.token != owner._autograd_token:
            raise RuntimeError("loss.backward() called for a forward pass that is no longer the engine's last one "
                               "(the engines keep the activations of ONE forward)")
        kw = dict(getattr(owner, "backward_kwargs", None) or {})
        owner.engine.backward(loss_scale=grad_out, **kw)
        owner._autograd_token += 1  # a second backward through the same graph must not run the kernels again
        return None, None, None


def attach_backward(owner, loss: torch.Tensor) -> torch.Tensor:
    """Return `loss` as a tensor whose `.backward()` runs `owner.engine.backward`.  `owner` is the HF-shaped model
    wrapper (it holds `.engine`; an optional dict `owner.backward_kwargs` is forwarded, e.g. `bucket_done` hooks or
    `overwrite_matrices`).  Without grad mode (`torch.no_grad()`, evaluation) the loss is returned unchanged."""
    if not torch.is_grad_enabled():
        return loss
    anchor = getattr(owner, "_autograd_anchor", None)
    if anchor is None or anchor.device != loss.device:
        anchor = torch.zeros((), device=loss.device, requires_grad=True)
        owner._autograd_anchor = anchor
        owner._autograd_token = 0
    owner._autograd_token += 1
    return _EngineBackward.apply(anchor, loss, owner)
