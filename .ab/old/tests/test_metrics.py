"""Validation metrics: oracle pinned to values computed by the reference's metrics.py
(tests/golden/make_golden_metrics.py), device kernel checked against both."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

G = dict(np.load(Path(__file__).resolve().parent / 'golden' / 'metrics.npz'))
CASES = list(range(int(G['n'])))


@pytest.mark.parametrize('i', CASES)
def test_oracle_matches_reference_metrics(i):
    pred, gt = G['pred_%d' % i], G['gt_%d' % i]
    assert abs(O.psnr(pred.reshape(-1, 3), gt.reshape(-1, 3)) - float(G['psnr_%d' % i])) < 1e-4
    assert abs(O.ssim(pred, gt, 1.0) - float(G['ssim_%d' % i])) < 2e-6
    half = pred.shape[1] // 2
    assert abs(O.ssim(pred[:, half:], gt[:, half:], 1.0) - float(G['ssim_half_%d' % i])) < 2e-6


def test_cpu_tensors_are_refused():
    from mega_nerf import _native as N
    from mega_nerf.metrics import ssim
    with pytest.raises(N.NativeError):
        ssim(torch.zeros(8, 8, 3), torch.zeros(8, 8, 3), 1)


@pytest.mark.gpu
@pytest.mark.parametrize('i', CASES)
def test_gpu_metrics_match_reference(i):
    from mega_nerf.metrics import psnr, psnr_ssim, ssim
    pred, gt = torch.from_numpy(G['pred_%d' % i]).cuda(), torch.from_numpy(G['gt_%d' % i]).cuda()
    assert abs(psnr(pred.view(-1, 3), gt.view(-1, 3)) - float(G['psnr_%d' % i])) < 1e-4
    assert abs(ssim(pred, gt, 1) - float(G['ssim_%d' % i])) < 5e-6
    half = pred.shape[1] // 2
    p, s = psnr_ssim(pred[:, half:], gt[:, half:], 1.0)                      # strided views, as the Runner passes them
    assert abs(s - float(G['ssim_half_%d' % i])) < 5e-6
    with np.errstate(divide='ignore'):
        want = O.psnr(G['pred_%d' % i][:, half:].reshape(-1, 3), G['gt_%d' % i][:, half:].reshape(-1, 3))
    assert (p == want) if np.isinf(want) else abs(p - want) < 1e-4          # identical halves: PSNR = inf on both sides


@pytest.mark.gpu
def test_gpu_metrics_large_image_and_batch():
    """4608 x 3456 image: SSIM of an image with itself is exactly 1, PSNR of a constant offset is exact, a leading
    batch dimension averages like the reference (mean over all values)."""
    import math
    from mega_nerf.metrics import psnr, ssim
    g = torch.Generator(device='cuda').manual_seed(0)
    img = torch.rand(3456, 4608, 3, device='cuda', generator=g)
    assert abs(ssim(img, img, 1) - 1.0) < 1e-6
    assert abs(psnr(img, img + 0.125) - (-10 * math.log10(0.125 ** 2))) < 1e-6
    a, b = torch.from_numpy(G['pred_0']).cuda(), torch.from_numpy(G['gt_0']).cuda()
    both = ssim(torch.stack([a, b]), torch.stack([b, b]), 1)
    assert abs(both - (float(G['ssim_0']) + 1.0) / 2) < 5e-6
