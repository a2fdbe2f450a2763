"""Merging trained submodules into one container (reference: scripts/merge_submodules.py:22-79).

Two sources for the per-cell weights:
  * :func:`merge_from_checkpoints` -- the reference's file hand-off: ``<ckpt_prefix><i>/<version>/models/<iters>.pt``;
  * :func:`merge_in_job`           -- straight from the trainers' memory at the end of a one-submodule-per-GPU job:
    ONE all_gather (RCCL over xGMI with the ``nccl`` backend; ``gloo`` in the CPU tests) of the flat fp32 weight buffers,
    rank 0 writes the archive.  Cell j lives on rank j % world_size (mega_nerf.distributed.assign_submodules).
The archive itself is written by mega_nerf.models.export (TorchScript MegaNeRFContainer)."""
from __future__ import annotations

from argparse import Namespace
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn
from torch.nn.modules.utils import consume_prefix_in_state_dict_if_present

from mega_nerf.distributed import flatten_state, unflatten_state
from mega_nerf.models.export import build_container, save_container
from mega_nerf.models.mega_nerf_container import MegaNeRFContainer
from mega_nerf.models.model_utils import get_bg_nerf, get_nerf


def _single_model_hparams(hparams: Namespace) -> Namespace:
    hp = Namespace(**vars(hparams))
    hp.container_path, hp.ckpt_path, hp.train_mega_nerf = None, None, None
    return hp


def find_checkpoint(ckpt_prefix: Path, cell: int, train_iterations: int) -> Path:
    """Newest experiment version of cell ``cell`` that holds ``models/<train_iterations>.pt`` (:35-47)."""
    cell_dir = ckpt_prefix.parent / '{}{}'.format(ckpt_prefix.name, cell)
    if not cell_dir.exists():
        raise Exception('{} not found'.format(cell_dir))
    for version in sorted((int(x.name) for x in cell_dir.iterdir()), reverse=True):
        candidate = cell_dir / str(version) / 'models' / '{}.pt'.format(train_iterations)
        if candidate.exists():
            return candidate
    raise Exception('Could not find {}.pt in {}'.format(train_iterations, cell_dir))


def _load_into(model: nn.Module, state: Dict[str, torch.Tensor]) -> nn.Module:
    consume_prefix_in_state_dict_if_present(state, prefix='module.')
    merged = model.state_dict()
    merged.update(state)
    model.load_state_dict(merged)
    return model


def models_from_checkpoint(hparams: Namespace, checkpoint: Path) -> Tuple[nn.Module, Optional[nn.Module]]:
    loaded = torch.load(checkpoint, map_location='cpu', weights_only=False)
    fg_state = loaded['model_state_dict']
    consume_prefix_in_state_dict_if_present(fg_state, prefix='module.')
    count = len(fg_state['embedding_a.weight']) if hparams.appearance_dim > 0 else 0
    hp = _single_model_hparams(hparams)
    fg = _load_into(get_nerf(hp, count), fg_state)
    bg = _load_into(get_bg_nerf(hp, count), loaded['bg_model_state_dict']) if 'bg_model_state_dict' in loaded else None
    return fg, bg


def container_from_models(hparams: Namespace, fg: List[nn.Module], bg: List[nn.Module], centroid_metadata: dict) -> MegaNeRFContainer:
    return build_container(fg, bg, centroid_metadata, hparams.pos_dir_dim > 0, hparams.appearance_dim > 0)


def merge_from_checkpoints(hparams: Namespace) -> MegaNeRFContainer:
    centroid_metadata = torch.load(hparams.centroid_path, map_location='cpu', weights_only=False)
    fg, bg = [], []
    for i in range(len(centroid_metadata['centroids'])):
        f, b = models_from_checkpoint(hparams, find_checkpoint(Path(hparams.ckpt_prefix), i, hparams.train_iterations))
        fg.append(f)
        if b is not None:
            bg.append(b)
    return container_from_models(hparams, fg, bg, centroid_metadata)


def merge_in_job(hparams: Namespace, local: Dict[int, Tuple[nn.Module, Optional[nn.Module]]], centroid_metadata: dict,
                 device: Optional[torch.device] = None) -> Optional[MegaNeRFContainer]:
    """``local``: the cells this rank trained, ``{cell index: (nerf, bg_nerf or None)}`` with cell j on rank j % world.
    Returns the container on rank 0 (None elsewhere).  One all_gather; every cell must share one architecture."""
    n_cells = len(centroid_metadata['centroids'])
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = [j for j in range(n_cells) if j % world == rank]
    if sorted(local) != mine:
        raise Exception('rank {} must hold cells {} (got {})'.format(rank, mine, sorted(local)))
    has_bg = bool(local) and all(b is not None for _, b in local.values())
    spec_fg = spec_bg = None
    flats = []
    for j in mine:
        f, b = local[j]
        ff, spec_fg = flatten_state(f.state_dict())
        flats.append(ff)
        if has_bg:
            fb, spec_bg = flatten_state(b.state_dict())
            flats.append(fb)
    slots = (n_cells + world - 1) // world
    per_cell = sum(x.numel() for x in flats) // len(mine) if mine else 0
    if world > 1:
        # small metadata handshake (architecture agreement), then the one data collective
        info = [None] * world
        dist.all_gather_object(info, (per_cell, has_bg) if mine else None)
        seen = {x for x in info if x is not None}
        if len(seen) != 1:
            raise Exception('submodules differ in architecture across ranks: {}'.format(sorted(seen)))
        per_cell, has_bg = next(iter(seen))
        dev = device if device is not None else (flats[0].device if flats else torch.device('cpu'))
        buf = torch.zeros(slots * per_cell, dtype=torch.float32, device=dev)
        if flats:
            mine_flat = torch.cat([x.to(dev) for x in flats])
            buf[:mine_flat.numel()] = mine_flat
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
    else:
        gathered = [torch.cat(flats) if flats else torch.zeros(0)]
    if rank != 0:
        return None
    # every rank ran the same model code, so rank 0's specs describe all cells
    hp = _single_model_hparams(hparams)
    n_fg = sum(int(torch.Size(s).numel()) for _, s in spec_fg)
    count = dict(spec_fg)['embedding_a.weight'][0] if hparams.appearance_dim > 0 else 0
    fg, bg = [], []
    for j in range(n_cells):
        chunk = gathered[j % world].cpu()[(j // world) * per_cell:(j // world + 1) * per_cell]
        fg.append(_load_into(get_nerf(hp, count), unflatten_state(chunk[:n_fg], spec_fg)))
        if has_bg:
            bg.append(_load_into(get_bg_nerf(hp, count), unflatten_state(chunk[n_fg:], spec_bg)))
    return container_from_models(hparams, fg, bg, centroid_metadata)


def single_cell_container(hparams: Namespace, checkpoint: Path) -> MegaNeRFContainer:
    """One trained model wrapped as a 1-cell container (reference scripts/convert_to_container.py:20-51): centroid at
    the origin, unit bounds, 3-D clustering -- so that single-model runs feed the same downstream tools as merged ones."""
    fg, bg = models_from_checkpoint(hparams, checkpoint)
    clustering = {'centroids': torch.zeros(1, 3), 'grid_dim': [1, 1], 'min_position': torch.zeros(3), 'max_position': torch.ones(3),
                  'cluster_2d': False}
    return container_from_models(hparams, [fg], [bg] if bg is not None else [], clustering)


def check_container_on_device(hparams: Namespace, path: str) -> None:
    """Read an archive back the way eval.py does and evaluate one sample per branch on the device (the smoke check at the
    end of merge_submodules.py:82-100 / convert_to_container.py:53-72); skipped with a notice on a host without a HIP device."""
    archive = torch.jit.load(path, map_location='cpu')
    has_bg = any(name.startswith('bg_sub_module_') for name, _ in archive.named_children())
    if not torch.cuda.is_available():
        print('container written to {}; skipping the test evaluation (no HIP device)'.format(path))
        return
    device = torch.device('cuda')
    hp = Namespace(**vars(hparams))
    hp.container_path, hp.ckpt_path = path, None
    width = 3 + (3 if hparams.pos_dir_dim > 0 else 0) + (1 if hparams.appearance_dim > 0 else 0)
    print('fg test eval: {}'.format(get_nerf(hp, 0).to(device).eval()(torch.ones(1, width, device=device))))
    if has_bg:
        print('bg test eval: {}'.format(get_bg_nerf(hp, 0).to(device).eval()(torch.ones(1, width + 4, device=device))))


__all__ = ['merge_from_checkpoints', 'merge_in_job', 'save_container', 'find_checkpoint', 'models_from_checkpoint',
           'single_cell_container', 'check_container_on_device']
