"""Two NeRFs behind one module: a coarse one for the stratified pass and a fine one for the importance-sampled pass
(reference: mega_nerf/models/cascade.py:7-18; selected by ``--use_cascade``).  The checkpoint keys are
``coarse.*`` / ``fine.*``; rendering.py / training.py pick the sub-model per pass through :meth:`select`."""
from typing import Optional

import torch
from torch import nn


class Cascade(nn.Module):
    def __init__(self, coarse: nn.Module, fine: nn.Module):
        super().__init__()
        self.add_module('coarse', coarse)
        self.add_module('fine', fine)

    def select(self, use_coarse: bool) -> nn.Module:
        return self._modules['coarse' if use_coarse else 'fine']

    def forward(self, use_coarse: bool, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        model = self.select(use_coarse)
        return model(x, sigma_only=sigma_only, sigma_noise=sigma_noise)
