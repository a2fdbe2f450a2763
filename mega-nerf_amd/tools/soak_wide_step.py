#!/usr/bin/env python3
"""Soak of the one-call training step: N cells of width W (default: two 512-wide Building cells) stepped 300 times on fresh batches towards
a learnable target (colour = f(direction)); prints the per-cell loss every 50 steps and whether the weights stayed finite.  (The cell
seeded 1000 collapses to the constant solution at every width and cell count -- its "sharpened" initial field (synthetic_scene.make_weights)
dies under Adam at lr 5e-4; the cell seeded 2000 reaches 1e-4.  Identical behaviour at 256 / 512 and 1 / 2 cells is the point.)
    python mega-nerf_amd/tools/soak_wide_step.py [width] [cells]"""
import sys, json, time
sys.path.insert(0, '.'); sys.path.insert(0, 'mega-nerf_amd')
import torch, bench
import synthetic_scene as S
from mega_nerf import ray_utils
from mega_nerf.opts import get_opts_base
from mega_nerf.training import FusedTrainStep
dev = torch.device('cuda:0')
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NCELL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128', '--layer_dim', str(W)])
s = S.SCENE
sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
all_rays = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
cells, batches = [], []
for c in range(NCELL):
    (fg, _, _), (bg, _, _) = bench.build_models(hp, dev, 1000 * (c + 1), W)
    fg.train(), bg.train()
    cells.append((fg, bg))
g = torch.Generator(device='cpu').manual_seed(1)
fs = FusedTrainStep(cells, hp, sc, sr, 1024)
losses = []
for it in range(300):
    bs = []
    for c in range(NCELL):
        sel = torch.randperm(all_rays.shape[0], generator=g)[:1024].to(dev)
        r = all_rays[sel].contiguous()
        tgt = torch.sigmoid(r[:, 3:6] * 3)          # a learnable target: colour = f(direction)
        bs.append((r, torch.randint(0, 100, (1024,), generator=g).float().to(dev), tgt))
    loss, n_bg, err = fs(bs)
    if it % 50 == 0 or it == 299:
        losses.append([round(float(v), 5) for v in loss])
fs.health()
w = [float(p.detach().abs().max()) for m in cells[0] for p in m.parameters()]
print(json.dumps({'what': '%d cells of width %d, 300 one-call steps on fresh batches' % (NCELL, W), 'loss_every_50': losses, 'finite': all(v == v and v < 1e6 for v in w)}))
