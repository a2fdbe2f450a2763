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

    def extra_repr(self) -> str:
        return 'passes: coarse -> stratified samples, fine -> stratified + importance samples'

    def sub_models(self):
        """(name, module) pairs in checkpoint-key order."""
        return [('coarse', self._modules['coarse']), ('fine', self._modules['fine'])]

    @property
    def training_paths(self):
        """Which training kernels each pass will use (diagnostics): 'fused' or 'layerwise' per sub-model."""
        return {name: ('fused' if getattr(m, 'fused_train_supported', lambda: False)() else 'layerwise')
                for name, m in self.sub_models()}

    def select(self, use_coarse: bool) -> nn.Module:
        return self._modules['coarse' if use_coarse else 'fine']

    def forward(self, use_coarse: bool, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        model = self.select(use_coarse)
        return model(x, sigma_only=sigma_only, sigma_noise=sigma_noise)
