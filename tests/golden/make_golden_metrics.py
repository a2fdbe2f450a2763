#!/usr/bin/env python3
"""Golden values for the validation metrics from the REAL reference (mega_nerf/metrics.py psnr / ssim, torch CPU).
``lpips`` (absent here) is imported at the top of that module but only used by its ``lpips()`` function, so an empty
stand-in module is registered for the import.  Build container only."""
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.modules.setdefault('lpips', types.ModuleType('lpips'))
from mega_nerf import metrics as M  # noqa: E402  (reference)

f32 = np.float32


def images(rng, H, W, noise):
    y, x = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing='ij')
    base = np.stack([0.5 + 0.4 * np.sin(7 * x + 3 * y), 0.5 + 0.4 * np.cos(5 * x * y + 1), x * y], -1)
    gt = np.clip(base + 0.05 * rng.standard_normal(base.shape), 0, 1).astype(f32)
    pred = np.clip(gt + noise * rng.standard_normal(base.shape), 0, 1).astype(f32)
    return pred, gt


def main():
    rng = np.random.default_rng(11)
    out = {}
    for i, (H, W, noise) in enumerate([(37, 53, 0.05), (64, 96, 0.2), (16, 32, 0.01), (9, 7, 0.1), (50, 40, 0.0)]):
        pred, gt = images(rng, H, W, noise)
        if noise == 0.0:
            pred = gt.copy()
            pred[3, 4, 1] += 0.25          # a single differing value keeps the PSNR finite
        out['pred_%d' % i], out['gt_%d' % i] = pred, gt
        out['psnr_%d' % i] = M.psnr(torch.from_numpy(pred).view(-1, 3), torch.from_numpy(gt).view(-1, 3))
        out['ssim_%d' % i] = M.ssim(torch.from_numpy(pred), torch.from_numpy(gt), 1)
        # the right-half evaluation of runner.py:413-427
        half = W // 2
        out['ssim_half_%d' % i] = M.ssim(torch.from_numpy(pred[:, half:].copy()), torch.from_numpy(gt[:, half:].copy()), 1)
        print(i, H, W, out['psnr_%d' % i], out['ssim_%d' % i], out['ssim_half_%d' % i])
    out['n'] = 5
    np.savez_compressed(HERE / 'metrics.npz', **out)


if __name__ == '__main__':
    main()
