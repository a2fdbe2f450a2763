#!/usr/bin/env python3
"""GPU-box half of the fixture ``render_overfit_hip_eval``: overfit one batch with THIS implementation's one-call training step
(bench.py's protocol: 30 steps of mnr_train_step on the batch that is rendered afterwards, training-mode randomness, random target
colours) and write the weight displacement, int8-quantised per tensor exactly like make_golden.run_overfit does, to
``gpurun_out/hip_overfit_deltas.npz``.  The build-container half (``make_golden.py render_overfit_hip_eval``) hands the decoded
weights to the REAL reference, which renders them in fp32 and in fp64.

    gpurun -- python tests/golden/export_hip_overfit.py
"""
import sys
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
for p in (ROOT, ROOT / 'mega-nerf_amd', HERE):
    sys.path.insert(0, str(p))

import common  # noqa: E402
from oracle.nerf_oracle import make_hparams  # noqa: E402  (only the hparams field list)

SEED_RAYS, SEED_FG, SEED_BG, STEPS, N = 7, 1000, 1500, 30, 1024


def main():
    from mega_nerf import ray_utils
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    from mega_nerf.training import FusedTrainStep
    dev = torch.device('cuda')
    s = common.SCENE
    hp = make_hparams(coarse_samples=64, fine_samples=128)
    A = s['appearance_count']
    cfgs = (common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256))
    inits = (common.make_weights(cfgs[0], A, SEED_FG), common.make_weights(cfgs[1], A, SEED_BG))

    def native(cfg, w):
        m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim, False, A, 3, cfg.xyz_dim,
                 ShiftedSoftplus())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m.to(dev).train()

    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    rays_all = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8).cpu().numpy()
    rays, idx = common.pick_rays(rays_all, N, SEED_RAYS)
    nf, nb = native(cfgs[0], inits[0]), native(cfgs[1], inits[1])
    step = FusedTrainStep([(nf, nb)], Namespace(**vars(hp)), torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev),
                          N, seed=11)
    gen = torch.Generator(device='cpu').manual_seed(3)
    batch = (torch.from_numpy(rays).to(dev), torch.from_numpy(idx.astype(np.int32)).to(dev), torch.rand(N, 3, generator=gen).to(dev))
    losses = [float(step([batch])[0][0]) for _ in range(STEPS)]
    torch.cuda.synchronize()
    out = dict(rays=rays, idx=idx.astype(np.int32), seed_fg=SEED_FG, seed_bg=SEED_BG, steps=STEPS, losses=np.array(losses, np.float32))
    for tag, m, init in (('fg', nf, inits[0]), ('bg', nb, inits[1])):
        for k, v in m.state_dict().items():
            dlt = v.detach().cpu().numpy().astype(np.float64) - init[k].astype(np.float64)
            scale = np.float32(max(float(np.abs(dlt).max()), 1e-30) / 127.0)
            out['dq_%s_%s' % (tag, k)] = np.clip(np.rint(dlt / scale), -127, 127).astype(np.int8)
            out['ds_%s_%s' % (tag, k)] = scale
    dst = ROOT / 'gpurun_out' / 'hip_overfit_deltas.npz'
    dst.parent.mkdir(exist_ok=True)
    np.savez_compressed(dst, **out)
    print('wrote', dst, 'loss %.5f -> %.5f' % (losses[0], losses[-1]))


if __name__ == '__main__':
    main()
