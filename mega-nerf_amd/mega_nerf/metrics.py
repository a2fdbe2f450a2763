"""PSNR (reference: mega_nerf/metrics.py:8-10).  SSIM/LPIPS are out of scope."""
import torch


def psnr(rgbs: torch.Tensor, target_rgbs: torch.Tensor) -> float:
    mse = torch.mean((rgbs - target_rgbs) ** 2)
    return -10 * torch.log10(mse).item()
