"""The merged model as it goes to disk: every cell's foreground / background NeRF plus the clustering metadata needed
to route between them.  ``torch.jit.script`` of this module IS the container file format (reference
mega_nerf/models/mega_nerf_container.py:7-25, written by merge_submodules.py:70-79), so the attribute names below are a
contract with the viewer, create_octree.py and ``--container_path``:

    sub_module_<i>, bg_sub_module_<i>          child modules, one per cell
    centroids (n, 3), grid_dim (2,) int32, min_position (3,), max_position (3,)     tensors (plain attributes, not buffers)
    need_viewdir, need_appearance_embedding, cluster_2d                              bools
"""
from typing import Dict, List, Sequence, Union

import torch
from torch import nn

FIELDS = ('centroids', 'grid_dim', 'min_position', 'max_position', 'need_viewdir', 'need_appearance_embedding', 'cluster_2d')


class MegaNeRFContainer(nn.Module):
    def __init__(self, sub_modules: List[nn.Module], bg_sub_modules: List[nn.Module], centroids: torch.Tensor,
                 grid_dim: torch.Tensor, min_position: torch.Tensor, max_position: torch.Tensor, need_viewdir: bool,
                 need_appearance_embedding: bool, cluster_2d: bool):
        super().__init__()
        self._add_cells('sub_module_{}', sub_modules)
        self._add_cells('bg_sub_module_{}', bg_sub_modules)
        values: Dict[str, Union[torch.Tensor, bool]] = dict(zip(FIELDS, (centroids, grid_dim, min_position, max_position,
                                                                         bool(need_viewdir), bool(need_appearance_embedding),
                                                                         bool(cluster_2d))))
        for name in FIELDS:
            setattr(self, name, values[name])

    def _add_cells(self, pattern: str, cells: Sequence[nn.Module]) -> None:
        for number, cell in enumerate(cells):
            self.add_module(pattern.format(number), cell)
