"""Validation metrics on the MI355X (reference: mega_nerf/metrics.py:8-10 PSNR, :51-121 SSIM).

The reference moves both images to the host and evaluates them with ATen CPU kernels (runner.py:399-436); here one
kernel pass over the device-resident images (csrc/metrics.hip) accumulates the squared error and the SSIM map, and only
two doubles travel to the host.  LPIPS needs pretrained networks that are not available here and stays out of scope."""
from typing import Dict, Tuple

import torch

from mega_nerf import _native as N

_filters: Dict[Tuple[int, float, str], torch.Tensor] = {}


def _gaussian(filter_size: int, filter_sigma: float, device: torch.device) -> torch.Tensor:
    """Normalised 1-D blur taps with the reference's fp32 arithmetic (metrics.py:77-82), cached on the device."""
    key = (filter_size, float(filter_sigma), str(device))
    if key not in _filters:
        half = filter_size // 2
        shift = (2 * half - filter_size + 1) / 2
        taps = torch.exp(-0.5 * ((torch.arange(filter_size) - half + shift) / filter_sigma) ** 2)
        _filters[key] = (taps / torch.sum(taps)).float().to(device)
    return _filters[key]


def _accumulate(pred: torch.Tensor, target: torch.Tensor, max_val: float, filter_size: int, filter_sigma: float, k1: float,
                k2: float) -> Tuple[float, float, int]:
    """(sum of squared errors, sum of the SSIM map, number of values) over [..., H, W, 3] image pairs."""
    N.require_device(pred, 'rgbs')
    N.require_device(target, 'target_rgbs')
    if pred.shape != target.shape or pred.shape[-1] != 3 or pred.dim() < 3:
        raise N.NativeError('image metrics expect two [..., H, W, 3] tensors of equal shape (got {} and {})'.format(
            tuple(pred.shape), tuple(target.shape)))
    H, W = pred.shape[-3], pred.shape[-2]
    p = pred.detach().reshape(-1, H, W, 3).float()
    t = target.detach().reshape(-1, H, W, 3).float()
    taps = _gaussian(filter_size, filter_sigma, p.device)
    acc = torch.zeros(2, dtype=torch.float64, device=p.device)
    with torch.cuda.device(p.device):
        for i in range(p.shape[0]):
            a, b = p[i], t[i]
            if a.stride(2) != 1 or a.stride(1) != 3:
                a = a.contiguous()
            if b.stride(2) != 1 or b.stride(1) != 3 or b.stride(0) != a.stride(0):
                a, b = a.contiguous(), b.contiguous()
            N.check(N.lib().mnr_image_metrics(a.data_ptr(), b.data_ptr(), H, W, a.stride(0), taps.data_ptr(), filter_size,
                                              float(max_val), float(k1), float(k2), acc.data_ptr(), N.stream_ptr()))
    se, ss = acc.tolist()                                  # the one host read
    return se, ss, p.shape[0] * H * W * 3


def psnr(rgbs: torch.Tensor, target_rgbs: torch.Tensor) -> float:
    """-10 log10(mean squared error) over all elements (metrics.py:8-10); any [..., 3] shape."""
    import math
    flat_p, flat_t = rgbs.reshape(1, -1, 3), target_rgbs.reshape(1, -1, 3)
    if flat_p.shape[1] == 0:
        return math.nan                                  # mean over nothing (torch.mean of an empty tensor is nan, too)
    se, _, n = _accumulate(flat_p, flat_t, 1.0, 1, 1.0, 0.01, 0.03)
    mse = se / n
    return -10 * math.log10(mse) if mse > 0 else math.inf


def ssim(rgbs: torch.Tensor, target_rgbs: torch.Tensor, max_val: float, filter_size: int = 11, filter_sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> float:
    """Mean SSIM of [..., H, W, 3] images, same signature and definition as the reference (metrics.py:51-121)."""
    _, ss, n = _accumulate(rgbs, target_rgbs, max_val, filter_size, filter_sigma, k1, k2)
    return ss / n


def psnr_ssim(rgbs: torch.Tensor, target_rgbs: torch.Tensor, max_val: float = 1.0) -> Tuple[float, float]:
    """Both validation metrics of runner.py:416,427 from ONE pass over the image pair."""
    import math
    se, ss, n = _accumulate(rgbs, target_rgbs, max_val, 11, 1.5, 0.01, 0.03)
    mse = se / n
    return (-10 * math.log10(mse) if mse > 0 else math.inf), ss / n
