"""Per-cell ray masks on the MI355X: the inner loop of scripts/create_cluster_masks.py (reference :104-210).

``min_dist_ratios`` replaces the chunked ``cdist`` / ``min`` cascade (:157-187) with one kernel launch per image
(mnr_cluster_min_ratios); ``cell_centroids`` restates the grid set-up (:66-81) and ``write_mask`` the on-disk format
every trainer reads back through ImageMetadata.load_mask (a ZIP holding one torch-saved bool[H, W], :203-210)."""
import zipfile
from pathlib import Path
from typing import Optional, Sequence, Tuple
from zipfile import ZipFile

import torch

from mega_nerf import _native as N

_Z_STEPS = {}


def z_steps(ray_samples: int, device: torch.device) -> torch.Tensor:
    """torch.linspace(0, 1, S) as the CPU computes it (the sample table is data), cached per device."""
    key = (ray_samples, str(device))
    if key not in _Z_STEPS:
        _Z_STEPS[key] = torch.linspace(0, 1, ray_samples).to(device)
    return _Z_STEPS[key]


def cell_centroids(camera_positions: torch.Tensor, grid_dim: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Cell centres of a grid_dim[0] x grid_dim[1] lattice over the camera footprint (altitude axis pinned to 0).
    Returns (centroids [n, 3], min_position, max_position); same fp32 arithmetic as the reference (:66-81)."""
    lo, hi = camera_positions.min(dim=0)[0], camera_positions.max(dim=0)[0]
    span = hi[1:] - lo[1:]
    g0, g1 = int(grid_dim[0]), int(grid_dim[1])
    along = [torch.arange(g) * span[i] / g + span[i] / (g * 2) for i, g in enumerate((g0, g1))]
    cells = torch.zeros(g0, g1, 3)
    cells[:, :, 1] = lo[1]
    cells[:, :, 2] = lo[2]
    cells[:, :, 1] += along[0].unsqueeze(1)
    cells[:, :, 2] += along[1]
    return cells.view(-1, 3), lo, hi


def min_dist_ratios(rays: torch.Tensor, centroids: torch.Tensor, ray_samples: int, cluster_2d: bool,
                    boundary_margin: Optional[float] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """rays (..., 8) on the HIP device, centroids (n, 3) -> (ratios (..., n) f32, masks (n, ...) bool or None).

    ratios[r, j] = min over the ray's samples of dist(sample, centroid_j) / (min_k dist(sample, centroid_k) + 1e-8)."""
    N.require_device(rays, 'rays')
    lead = rays.shape[:-1]
    r = rays.reshape(-1, 8).contiguous().float()
    cen = centroids.to(r.device, torch.float32).contiguous()
    n = cen.shape[0]
    ratios = torch.empty(r.shape[0], n, device=r.device, dtype=torch.float32)
    masks = torch.empty(n, r.shape[0], device=r.device, dtype=torch.uint8) if boundary_margin is not None else None
    t = z_steps(ray_samples, r.device)
    with torch.cuda.device(r.device):
        N.check(N.lib().mnr_cluster_min_ratios(ratios.data_ptr(), masks.data_ptr() if masks is not None else None,
                                               r.data_ptr(), r.shape[0], t.data_ptr(), int(ray_samples), cen.data_ptr(), n,
                                               int(bool(cluster_2d)),
                                               float(boundary_margin if boundary_margin is not None else 0.0),
                                               N.stream_ptr()))
    return ratios.view(*lead, n), (masks.view(n, *lead).bool() if masks is not None else None)


def write_mask(path: Path, mask: torch.Tensor) -> None:
    with ZipFile(path, compression=zipfile.ZIP_DEFLATED, mode='w') as zf:
        with zf.open(path.name, 'w') as f:
            torch.save(mask, f)


def read_mask(path: Path) -> torch.Tensor:
    with ZipFile(path) as zf:
        with zf.open(path.name) as f:
            return torch.load(f, map_location='cpu')
