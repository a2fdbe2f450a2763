"""TorchScript export of trained submodules -- the *on-disk format* between training and the consumers of a merged
model (reference: scripts/merge_submodules.py:70-79 writes ``torch.jit.script(MegaNeRFContainer(...))``; the viewer,
create_octree.py and ``--container_path`` read it back with ``torch.jit.load`` and call
``sub_module_i(x, sigma_only, sigma_noise)``).

The MI355X ``NeRF`` module evaluates through a C ABI and cannot be scripted, so the archive holds
:class:`PortableNeRF` twins: same parameter names / shapes, and a plain-torch ``forward`` with the semantics of
reference nerf.py:115-160 so that *other* tools can run the archive anywhere.  This package never evaluates a
PortableNeRF itself: ``get_nerf(container_path=...)`` rebuilds native modules from the state_dicts
(model_utils.nerf_from_scripted)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from mega_nerf.models.mega_nerf_container import MegaNeRFContainer


class ShiftedSoftplus(nn.Module):
    """softplus(x - 1); the class name is part of the format (readers tell it from ReLU by name)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.softplus(x - 1.0, 1.0, 20.0)


class PortableNeRF(nn.Module):
    def __init__(self, pos_xyz_dim: int, pos_dir_dim: int, layers: int, skip_layers: List[int], layer_dim: int,
                 appearance_dim: int, affine_appearance: bool, appearance_count: int, rgb_dim: int, xyz_dim: int,
                 shifted_softplus: bool):
        super().__init__()
        self.xyz_dim, self.pos_xyz_dim, self.pos_dir_dim = xyz_dim, pos_xyz_dim, pos_dir_dim
        self.skip_layers = list(skip_layers)
        self.has_dir, self.has_app = pos_dir_dim > 0, appearance_dim > 0
        in_xyz = xyz_dim * (1 + 2 * pos_xyz_dim)
        in_dir = 3 * (1 + 2 * pos_dir_dim) if pos_dir_dim > 0 else 0
        self.xyz_encodings = nn.ModuleList(
            nn.Sequential(nn.Linear(in_xyz if i == 0 else layer_dim + (in_xyz if i in self.skip_layers else 0), layer_dim), nn.ReLU())
            for i in range(layers))
        self.has_affine = bool(affine_appearance)                 # nerf.py:87-89: appearance enters as a 3x4 colour transform
        self.has_final = self.has_dir or (self.has_app and not self.has_affine)
        self.embedding_a = nn.Embedding(appearance_count, appearance_dim) if self.has_app else None
        self.affine = nn.Linear(appearance_dim, 12) if self.has_affine else None
        self.xyz_encoding_final = nn.Linear(layer_dim, layer_dim) if self.has_final else None
        self.dir_a_encoding = nn.Sequential(nn.Linear(layer_dim + in_dir + (0 if self.has_affine else appearance_dim), layer_dim // 2),
                                            nn.ReLU()) if self.has_final else None
        self.sigma = nn.Linear(layer_dim, 1)
        self.sigma_activation = ShiftedSoftplus() if shifted_softplus else nn.ReLU()
        self.rgb = nn.Linear(layer_dim // 2 if self.has_final else layer_dim, rgb_dim)
        self.rgb_sigmoid = rgb_dim == 3

    def _encode(self, v: torch.Tensor, bands: int) -> torch.Tensor:
        parts = [v]
        for k in range(bands):
            parts.append(torch.sin(v * float(2 ** k)))
            parts.append(torch.cos(v * float(2 ** k)))
        return torch.cat(parts, -1)

    def forward(self, x: torch.Tensor, sigma_only: bool = False, sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        expected = self.xyz_dim
        if not sigma_only:
            expected += (3 if self.has_dir else 0) + (1 if self.has_app else 0)
        if x.shape[1] != expected:
            raise Exception('Unexpected input shape: {} (expected: {}, xyz_dim: {})'.format(x.shape, expected, self.xyz_dim))
        pts = self._encode(x[:, :self.xyz_dim], self.pos_xyz_dim)
        h = pts
        for i, layer in enumerate(self.xyz_encodings):
            if i in self.skip_layers:
                h = torch.cat([pts, h], -1)
            h = layer(h)
        density = self.sigma(h)
        if sigma_noise is not None:
            density = density + sigma_noise
        density = self.sigma_activation(density)
        if sigma_only:
            return density
        if self.xyz_encoding_final is not None and self.dir_a_encoding is not None:
            feats = [self.xyz_encoding_final(h)]
            if self.has_dir:
                feats.append(self._encode(x[:, -4:-1], self.pos_dir_dim))        # nerf.py:146 (quirk Q8 included)
            if self.embedding_a is not None and self.affine is None:
                feats.append(self.embedding_a(x[:, -1].long()))
            h = self.dir_a_encoding(torch.cat(feats, -1))
        colour = self.rgb(h)
        if self.affine is not None and self.embedding_a is not None:                 # nerf.py:156-158
            t = self.affine(self.embedding_a(x[:, -1].long())).view(-1, 3, 4)
            colour = (torch.matmul(t[:, :, :3], colour.unsqueeze(-1)) + t[:, :, 3:]).squeeze(-1)
        if self.rgb_sigmoid:
            colour = torch.sigmoid(colour)
        return torch.cat([colour, density], -1)


def to_portable(model) -> PortableNeRF:
    """PortableNeRF carrying the parameters of a native ``mega_nerf.models.nerf.NeRF`` (CPU copy)."""
    from mega_nerf.models.nerf import ShiftedSoftplus as NativeSoftplus
    p = PortableNeRF(model.pos_xyz_dim, model.pos_dir_dim, model.layers, model.skip_layers, model.layer_dim, model.appearance_dim,
                     model.affine is not None, model.appearance_count, model.rgb_dim, model.xyz_dim,
                     isinstance(model.sigma_activation, NativeSoftplus))
    sd = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
    own = p.state_dict()
    for k in own:
        if k in sd:
            own[k] = sd[k]
    p.load_state_dict(own)
    return p.eval()


def build_container(sub_modules, bg_sub_modules, centroid_metadata: dict, need_viewdir: bool,
                    need_appearance_embedding: bool) -> MegaNeRFContainer:
    """MegaNeRFContainer of portable twins + the clustering metadata written by create_cluster_masks.py (params.pt)."""
    return MegaNeRFContainer([to_portable(m) for m in sub_modules], [to_portable(m) for m in bg_sub_modules],
                             centroid_metadata['centroids'], torch.IntTensor(centroid_metadata['grid_dim']),
                             centroid_metadata['min_position'], centroid_metadata['max_position'], need_viewdir,
                             need_appearance_embedding, bool(centroid_metadata['cluster_2d']))


def save_container(container: MegaNeRFContainer, path) -> None:
    torch.jit.save(torch.jit.script(container.eval()), str(path))
