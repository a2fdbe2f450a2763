"""Bag of trained submodules + clustering metadata (reference: mega_nerf/models/mega_nerf_container.py:7-25).
Attribute names are the on-disk contract of merged containers (merge_submodules.py:70-78)."""
from typing import List

import torch
from torch import nn


class MegaNeRFContainer(nn.Module):
    def __init__(self, sub_modules: List[nn.Module], bg_sub_modules: List[nn.Module], centroids: torch.Tensor,
                 grid_dim: torch.Tensor, min_position: torch.Tensor, max_position: torch.Tensor, need_viewdir: bool,
                 need_appearance_embedding: bool, cluster_2d: bool):
        super().__init__()
        for i, m in enumerate(sub_modules):
            setattr(self, 'sub_module_{}'.format(i), m)
        for i, m in enumerate(bg_sub_modules):
            setattr(self, 'bg_sub_module_{}'.format(i), m)
        self.centroids = centroids
        self.grid_dim = grid_dim
        self.min_position = min_position
        self.max_position = max_position
        self.need_viewdir = need_viewdir
        self.need_appearance_embedding = need_appearance_embedding
        self.cluster_2d = cluster_2d
