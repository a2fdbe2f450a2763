"""North star: "PSNR within 0.05 dB".  A student field is trained for a few Adam steps in TRAINING mode (stratified jitter,
sigma noise, random u) on the MI355X path and, with the identical injected random numbers, on the torch-CPU restatement
of the reference (oracle/torch_oracle.py, pinned to the goldens); both are then evaluated on held-out rays against the
same teacher field.  bench.py runs the long form (24 steps x 192 rays: 0.0007 dB apart); this is the short form."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / 'mega-nerf_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

pytestmark = pytest.mark.gpu


def test_student_psnr_matches_cpu_restatement(monkeypatch):
    import bench as b
    import synthetic_scene as S
    from mega_nerf import ray_utils
    from mega_nerf.opts import get_opts_base
    monkeypatch.setattr(b, 'PSNR_STEPS', 8)
    monkeypatch.setattr(b, 'PSNR_BATCH', 96)
    monkeypatch.setattr(b, 'PSNR_TEST_RAYS', 192)
    dev = torch.device('cuda:0')
    hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    s = S.SCENE
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
    prob = b.psnr_problem(hp, rays.cpu().numpy())
    psnr_here, tgt_train, tgt_test = b.psnr_gpu(hp, prob, dev)
    psnr_ref = b.psnr_cpu(hp, prob, tgt_train, tgt_test)
    assert np.isfinite(psnr_here) and 5.0 < psnr_here < 60.0            # a real image, neither garbage nor a trivial one
    assert abs(psnr_here - psnr_ref) < 0.05, (psnr_here, psnr_ref)      # tolerance of BASELINE.json's north star
