import sys
from argparse import Namespace
from pathlib import Path
import numpy as np, torch
ROOT = Path('/root/repo')
for p in (ROOT, ROOT / 'mega-nerf_amd', ROOT / 'tests' / 'golden'):
    sys.path.insert(0, str(p))
import common
from oracle.nerf_oracle import make_hparams
from mega_nerf import ray_utils
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
from mega_nerf.training import FusedTrainStep, CellTrainer
import mega_nerf.training as TR
dev = torch.device('cuda'); s = common.SCENE
hp = make_hparams(coarse_samples=64, fine_samples=128); A = s['appearance_count']
cfgs = (common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256))
def native(cfg, w, train):
    m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim, False, A, 3, cfg.xyz_dim, ShiftedSoftplus())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.to(dev); m.train(train); return m
d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
rays_all = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8).cpu().numpy()
sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
for seeds in ((1000, 1500, 7), (31000, 31500, 31)):
    rays, idx = common.pick_rays(rays_all, 1024, seeds[2])
    gen = torch.Generator(device='cpu').manual_seed(3)
    batch = (torch.from_numpy(rays).to(dev), torch.from_numpy(idx.astype(np.int32)).to(dev), torch.rand(1024, 3, generator=gen).to(dev))
    for mode in ('fused-train', 'fused-eval', 'autograd-eval'):
        import os
        if mode.startswith('autograd'): os.environ['MNR_NO_FUSED_STEP'] = '1'
        else: os.environ.pop('MNR_NO_FUSED_STEP', None)
        train = mode.endswith('train')
        nf = native(cfgs[0], common.make_weights(cfgs[0], A, seeds[0]), train); nb = native(cfgs[1], common.make_weights(cfgs[1], A, seeds[1]), train)
        tr = CellTrainer(nf, nb, Namespace(**vars(hp)), sc, sr, seed=11)
        L = [float(tr.step(*batch)[0]) for _ in range(40)]
        print(seeds, mode, 'fused' if tr.fused is not None else 'torch', np.round(L[::3], 5).tolist(), 'max', max(L), flush=True)
