"""Coarse/fine model pair (reference: mega_nerf/models/cascade.py:7-18)."""
from typing import Optional

import torch
from torch import nn


class Cascade(nn.Module):
    def __init__(self, coarse: nn.Module, fine: nn.Module):
        super().__init__()
        self.coarse = coarse
        self.fine = fine

    def forward(self, use_coarse: bool, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        return (self.coarse if use_coarse else self.fine)(x, sigma_only, sigma_noise)
