"""Ray generation on the MI355X (reference: mega_nerf/ray_utils.py:6-84), same signatures.

``directions``/``c2w`` must live on the HIP device; outputs are fresh device tensors.  Entry points only
enqueue on the *current* stream of the calling thread, so the dataset prefetch thread of the reference
(filesystem_dataset.py:70-77) can call them concurrently with training.
"""
import ctypes as C
from typing import List, Optional

import torch

from mega_nerf import _native as N


def get_ray_directions(W: int, H: int, fx: float, fy: float, cx: float, cy: float, center_pixels: bool,
                       device: torch.device) -> torch.Tensor:
    device = torch.device(device)
    if device.type != 'cuda':
        raise N.NativeError('get_ray_directions needs a HIP device (got {}); there is no CPU fallback'.format(device))
    with torch.cuda.device(device):
        out = torch.empty(H, W, 3, device=device, dtype=torch.float32)
        N.check(N.lib().mnr_ray_directions(out.data_ptr(), W, H, float(fx), float(fy), float(cx), float(cy),
                                           int(bool(center_pixels)), N.stream_ptr()))
    return out


def _alt(ray_altitude_range: Optional[List[float]]):
    if ray_altitude_range is None:
        return None
    return (C.c_float * 2)(float(ray_altitude_range[0]), float(ray_altitude_range[1]))


def get_rays(directions: torch.Tensor, c2w: torch.Tensor, near: float, far: float,
             ray_altitude_range: List[float]) -> torch.Tensor:
    """directions (H, W, 3), c2w (3, 4) -> (H, W, 8) = [origin, direction, near, far]."""
    N.require_device(directions, 'directions')
    c2w = c2w.to(directions.device, torch.float32).contiguous()
    d = directions.contiguous().float()
    P = d.numel() // 3
    out = torch.empty(*d.shape[:-1], 8, device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        N.check(N.lib().mnr_get_rays(out.data_ptr(), d.data_ptr(), P, 1, c2w.data_ptr(), 1, float(near), float(far),
                                     _alt(ray_altitude_range), N.stream_ptr()))
    return out


def get_rays_batch(directions: torch.Tensor, c2w: torch.Tensor, near: float, far: float,
                   ray_altitude_range: List[float]) -> torch.Tensor:
    """directions (n, P, 3), c2w (n, 3, 4) -> (n, P, 8)."""
    N.require_device(directions, 'directions')
    c2w = c2w.to(directions.device, torch.float32).contiguous()
    d = directions.contiguous().float()
    n, P = d.shape[0], d.shape[1]
    out = torch.empty(n, P, 8, device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        N.check(N.lib().mnr_get_rays(out.data_ptr(), d.data_ptr(), P, n, c2w.data_ptr(), n, float(near), float(far),
                                     _alt(ray_altitude_range), N.stream_ptr()))
    return out


def get_rays_indexed(directions: torch.Tensor, pixel_indices: torch.Tensor, c2ws: torch.Tensor, img_indices: torch.Tensor,
                     near: float, far: float, ray_altitude_range: List[float]) -> torch.Tensor:
    """Rays of the (image, pixel) pairs of a training chunk: directions (P, 3) shared by all images, c2ws (n, 3, 4),
    int32 index vectors (M,) -> (M, 8).  Replaces the unique/gather dance of filesystem_dataset.py:103-121."""
    N.require_device(directions, 'directions')
    dev = directions.device
    d = directions.contiguous().float()
    poses = c2ws.to(dev, torch.float32).contiguous()
    pix = pixel_indices.to(dev, torch.int32).contiguous()
    img = img_indices.to(dev, torch.int32).contiguous()
    if pix.shape != img.shape or pix.dim() != 1:
        raise N.NativeError('pixel_indices and img_indices must be 1-D and of equal length')
    out = torch.empty(pix.shape[0], 8, device=dev, dtype=torch.float32)
    err = torch.zeros(1, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        N.check(N.lib().mnr_get_rays_indexed(out.data_ptr(), d.data_ptr(), d.shape[0], pix.data_ptr(), poses.data_ptr(),
                                             poses.shape[0], img.data_ptr(), pix.shape[0], float(near), float(far),
                                             _alt(ray_altitude_range), err.data_ptr(), N.stream_ptr()))
    out._mnr_index_error = err            # checked lazily by the dataset (no sync here)
    return out
