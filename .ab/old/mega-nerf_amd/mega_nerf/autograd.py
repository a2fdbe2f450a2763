"""Autograd node for a direct ``NeRF.forward(x)`` call with gradients enabled (reference nerf.py:115-160 under
torch autograd): parameters are differentiable, the sample coordinates ``x`` are not (the reference never
back-propagates into them either: importance samples are detached, rendering.py:215)."""
from __future__ import annotations

from typing import Optional

import torch

from mega_nerf import _native as N


class _MlpFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, sigma_noise, *params):
        x = x.contiguous().float()
        B, ncol = x.shape
        out = torch.empty(B, model.rgb_dim + 1, device=x.device, dtype=torch.float32)
        dirs = idx = None
        if model.has_dir:
            # x[:, -4:-1] (nerf.py:146): the view direction when an appearance column follows it, else quirk Q8's
            # [last xyz coordinate, d_x, d_y]
            dirs = x[:, ncol - 4:]
        if model.embedding_a is not None:
            idx = x[:, ncol - 1:]
        noise = sigma_noise.contiguous().float().view(-1) if sigma_noise is not None else None
        if model.rgb_dim > 3:
            # raw SH coefficients (the colour epilogue belongs to rendering.py:300-305): layer by layer, wide output rows
            from mega_nerf.models.layerwise import LayerwiseTape
            ctx.tape = LayerwiseTape(model, x, ncol, dirs, ncol, 1, idx, ncol, 1, B, out, model.rgb_dim + 1, noise, False, -1, True)
        else:
            ctx.tape = model.train_eval(x, ncol, dirs, ncol, 1, idx, ncol, 1, B, out, noise, -1, None, 0)
        ctx.model, ctx.keep = model, (x, noise)
        ctx.names = [k for k, _ in model.named_parameters()]
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, d_out):
        from mega_nerf.training import _zero_grads
        d_out = d_out.contiguous().float()
        grads = _zero_grads(ctx.names, ctx.params)
        ctx.tape.backward(d_out, d_out.shape[1], grads)
        return (None, None, None) + tuple(grads[k] for k in ctx.names)


def mlp_forward_with_grad(model, x: torch.Tensor, sigma_only: bool, sigma_noise: Optional[torch.Tensor]) -> torch.Tensor:
    N.require_device(x, 'x')
    if sigma_only:
        raise NotImplementedError('sigma_only evaluations are inference-only on the MI355X path (use torch.no_grad())')
    if x.shape[0] == 0:
        return torch.empty(0, model.rgb_dim + 1, device=x.device, dtype=torch.float32)
    return _MlpFunction.apply(model, x, sigma_noise, *[p for _, p in model.named_parameters()])
