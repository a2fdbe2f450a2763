"""Spatial router over per-cell NeRFs (reference: mega_nerf/models/mega_nerf.py:7-61).

Routing weights follow the reference exactly: hard argmin for ``boundary_margin == 1`` and
inverse-distance blending of every cell within ``boundary_margin * d_min`` otherwise.  Each cell's
samples are evaluated by the fused MLP kernel of that cell's weights.
"""
from typing import List, Optional

import torch
from torch import nn

from mega_nerf import _native as N


class MegaNeRF(nn.Module):
    def __init__(self, sub_modules: List[nn.Module], centroids: torch.Tensor, boundary_margin: float, xyz_real: bool,
                 cluster_2d: bool, joint_training: bool = False):
        super().__init__()
        assert boundary_margin >= 1
        self.sub_modules = nn.ModuleList(sub_modules)
        self.register_buffer('centroids', centroids)
        self.boundary_margin = boundary_margin
        self.xyz_real = xyz_real
        self.cluster_dim_start = 1 if cluster_2d else 0
        self.joint_training = joint_training

    # attributes rendering.py reads from a plain NeRF
    @property
    def has_dir(self):
        return self.sub_modules[0].has_dir

    @property
    def embedding_a(self):
        return self.sub_modules[0].embedding_a

    def _route(self, pos: torch.Tensor):
        """(n_sub, B) blend weights; zero = not routed (mega_nerf.py:21-31)."""
        d = torch.cdist(pos[:, self.cluster_dim_start:3], self.centroids[:, self.cluster_dim_start:].to(pos.device))
        if self.boundary_margin > 1:
            inv = 1 / (d + 1e-8)
            inv[d > self.boundary_margin * d.min(dim=1, keepdim=True)[0]] = 0
            return (inv / inv.sum(dim=-1, keepdim=True)).t().contiguous()
        w = torch.zeros_like(d)
        w.scatter_(1, d.argmin(dim=1, keepdim=True), 1.0)
        return w.t().contiguous()

    def _run(self, x_in: torch.Tensor, pos: torch.Tensor, dirs_rows, idx_rows, out: torch.Tensor, sigma_only: bool,
             noise: Optional[torch.Tensor], sh_deg: int):
        w = self._route(pos)
        out.zero_()
        for i, child in enumerate(self.sub_modules):
            rows = torch.nonzero(w[i] > 0, as_tuple=False).view(-1)      # host sync, like mega_nerf.py:38
            if rows.numel() == 0:
                continue
            xi = x_in.index_select(0, rows)
            di = dirs_rows.index_select(0, rows) if dirs_rows is not None else None
            ii = idx_rows.index_select(0, rows) if idx_rows is not None else None
            ni = noise.index_select(0, rows) if noise is not None else None
            sub = torch.empty(rows.numel(), out.shape[1], device=out.device, dtype=torch.float32)
            child.evaluate(xi, xi.shape[1], di, 3, ii, 1, 1, rows.numel(), sub, ni, sigma_only, sh_deg)
            if self.boundary_margin == 1:
                out.index_copy_(0, rows, sub)
            else:
                out.index_add_(0, rows, sub * w[i].index_select(0, rows).unsqueeze(-1))

    def evaluate_routed(self, xyz: torch.Tensor, part, S: int, out: torch.Tensor, noise, sh_deg: int):
        """Render-path entry: xyz [n, S, 3|4|7], per-ray dirs/idx in ``part``."""
        n = xyz.shape[0]
        if part.n_units is not None:
            n = min(n, int(part.n_units.item()))                         # compacted background rays
            if n == 0:
                return
        x = xyz[:n].reshape(n * S, xyz.shape[-1])
        pos = x[:, :3]
        x_in = x[:, 3:].contiguous() if self.xyz_real else x
        child0 = self.sub_modules[0]
        need_dir = child0.has_dir or sh_deg >= 0
        dirs_rows = part.dirs[:n].unsqueeze(1).expand(n, S, 3).reshape(n * S, 3) if need_dir else None
        idx_rows = None
        if child0.embedding_a is not None:
            idx_rows = part.idx[:n].unsqueeze(1).expand(n, S).reshape(n * S).contiguous()
        self._run(x_in, pos, dirs_rows.contiguous() if dirs_rows is not None else None, idx_rows,
                  out.view(-1, out.shape[-1])[:n * S], False, noise[:n * S] if noise is not None else None, sh_deg)

    def forward(self, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        N.require_device(x, 'x')
        child0 = self.sub_modules[0]
        x = x.contiguous().float()
        pos = x[:, :3]
        x_in = x[:, 3:].contiguous() if self.xyz_real else x
        D = child0.xyz_dim
        dirs_rows = idx_rows = None
        if not sigma_only:
            if child0.has_dir:
                dirs_rows = x_in[:, x_in.shape[1] - 4:x_in.shape[1] - 1].contiguous()
            if child0.embedding_a is not None:
                idx_rows = x_in[:, -1].contiguous()
        out = torch.empty(x.shape[0], 1 if sigma_only else child0.rgb_dim + 1, device=x.device, dtype=torch.float32)
        self._run(x_in[:, :D].contiguous() if not sigma_only else x_in, pos, dirs_rows, idx_rows, out, sigma_only,
                  sigma_noise.view(-1) if sigma_noise is not None else None, -1)
        return out
