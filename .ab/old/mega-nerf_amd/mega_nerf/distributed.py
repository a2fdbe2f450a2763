"""Multi-GPU plumbing for the submodule-per-GPU layout.

The hot path shards by spatial submodule with NO collective in the data path.  The two exchanges that exist
replace filesystem hand-offs of the reference:

* :func:`all_reduce_metrics` -- one ``all_reduce(SUM)`` of a packed fp64 vector ``[sum_0 .. sum_k, count]``
  (reference: per-image ``tmp_val_metrics/*.pt`` files + barriers, runner.py:422-448, 495-510);
* :func:`gather_submodule_weights` -- one ``all_gather`` of flat fp32 weight buffers (reference: checkpoints on disk
  collected by scripts/merge_submodules.py:33-68).

Works with ``nccl`` (= RCCL on ROCm, device tensors) and ``gloo`` (CPU tensors, used by the unit tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist


def assign_submodules(n_submodules: int, world_size: int) -> List[List[int]]:
    """Submodule j -> rank j % world_size (25 cells on 8 GPUs -> 4,3,3,3,3,3,3,3)."""
    return [[j for j in range(n_submodules) if j % world_size == r] for r in range(world_size)]


def images_for_rank(n_images: int, rank: int, world_size: int) -> List[int]:
    """Validation image i is rendered by rank i % world_size (runner.py:396)."""
    return list(range(rank, n_images, world_size))


def all_reduce_metrics(sums: Dict[str, float], count: int, device: torch.device) -> Tuple[Dict[str, float], int]:
    """Global (sum, count) of per-rank metric sums with a single collective.  Keys must match on all ranks."""
    keys = sorted(sums)
    packed = torch.tensor([float(sums[k]) for k in keys] + [float(count)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    vals = packed.tolist()
    return {k: v for k, v in zip(keys, vals[:-1])}, int(round(vals[-1]))


def average_gradients(params: Sequence[torch.nn.Parameter]) -> None:
    """Data-parallel training of ONE submodule on several ranks (the reference's DDP mode, runner.py:120-129): replace every
    parameter's gradient by its mean over the ranks so that identical optimiser steps keep the replicas bit-identical.  ONE
    all_reduce per step over a flat buffer of all gradients (~5 MB for fg + bg: far below where bucketing pays on xGMI); missing
    gradients count as zero.  When the gradients already are views of one contiguous buffer (training.FusedTrainStep's gradient
    area, or the flat buffer of training._zero_grads) the collective runs in place on that buffer, without a copy.  No-op without an
    initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    params = list(params)
    if not params:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    flat = _as_one_buffer(grads)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def _as_one_buffer(grads):
    """The contiguous 1-D tensor the gradients are views of (padding between them included), or None."""
    try:
        base = grads[0].untyped_storage()
        if any(g.untyped_storage().data_ptr() != base.data_ptr() or not g.is_contiguous() or g.dtype != grads[0].dtype for g in grads):
            return None
        lo = min(g.storage_offset() for g in grads)
        hi = max(g.storage_offset() + g.numel() for g in grads)
        if hi - lo > 2 * sum(g.numel() for g in grads) + 1024:
            return None                           # views of something much larger (e.g. a whole workspace): copy instead
        return torch.as_strided(grads[0], (hi - lo,), (1,), lo)
    except (RuntimeError, AttributeError):
        return None


def any_rank(flag: bool, device: torch.device) -> bool:
    """Logical OR of a per-rank flag (e.g. "this batch had background rays": runner.py:269-272 must decide alike everywhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def flatten_state(state: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, List[Tuple[str, torch.Size]]]:
    """Deterministic (name-sorted) fp32 flattening of a state_dict."""
    spec = [(k, state[k].shape) for k in sorted(state)]
    flat = torch.cat([state[k].detach().reshape(-1).float() for k, _ in spec]) if spec else torch.zeros(0)
    return flat, spec


def unflatten_state(flat: torch.Tensor, spec: Sequence[Tuple[str, torch.Size]]) -> Dict[str, torch.Tensor]:
    out, o = {}, 0
    for k, shape in spec:
        n = int(torch.Size(shape).numel())
        out[k] = flat[o:o + n].reshape(shape).clone()
        o += n
    return out


def gather_submodule_weights(state: Dict[str, torch.Tensor]) -> List[Dict[str, torch.Tensor]]:
    """Every rank contributes one submodule's state_dict (same architecture everywhere); returns the list of all
    state_dicts in rank order on every rank.  One all_gather of ~5 MB per rank (fg+bg of a 256-wide model)."""
    flat, spec = flatten_state(state)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [unflatten_state(flat, spec)]
    bufs = [torch.empty_like(flat) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, flat.contiguous())
    return [unflatten_state(b, spec) for b in bufs]
