#!/usr/bin/env python3
"""Race hunt for the software-pipelined forward kernels (chunk barriers taken two batches early, LDS-DMA'd bias rows, LDS stashes):
evaluation renders are deterministic, so ANY run-to-run difference in their outputs is an LDS / barrier hazard.  Renders the benchmark batch
N times per configuration (default architecture, SH head, 512-wide pair kernel, routed 8-cell container) and counts renders whose outputs
are not bit-identical to the first one; a ragged second batch size in between perturbs the launch shapes.  One JSON line per configuration.

    python mega-nerf_amd/tools/stress_determinism.py [--iters 2000]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'mega-nerf_amd'))

import bench  # noqa: E402  (build_models, the synthetic scene)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2000)
    a = ap.parse_args()
    import synthetic_scene as S
    from mega_nerf import ray_utils
    from mega_nerf.models.mega_nerf import MegaNeRF
    from mega_nerf.opts import get_opts_base
    from mega_nerf.rendering import render_rays_async
    dev = torch.device('cuda')
    s = S.SCENE
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    rays_all = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
    g = torch.Generator(device='cpu').manual_seed(5)
    sel = torch.randperm(rays_all.shape[0], generator=g)
    cases = [('default 8x256', [], 256, 0), ('sh_deg 2', ['--sh_deg', '2', '--pos_dir_dim', '0'], 256, 0), ('fg 8x512 (pair kernel)', [], 512, 0),
             ('8-cell container', [], 256, 8)]
    for name, flags, width, cells in cases:
        hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128', '--layer_dim', str(width)] + flags)
        if cells:
            cent = torch.stack([torch.zeros(cells), torch.linspace(-.45, .45, 2).repeat_interleave(4), torch.linspace(-.45, .45, 4).repeat(2)], 1)
            sub = [bench.build_models(hp, dev, 1000 + 7 * j, width) for j in range(cells)]
            fg = MegaNeRF([c[0][0] for c in sub], cent, hp.boundary_margin, False, False).to(dev).eval()
            bg = MegaNeRF([c[1][0] for c in sub], cent, hp.boundary_margin, True, False).to(dev).eval()
            hp.container_path = 'stress'
        else:
            (fg, _, _), (bg, _, _) = bench.build_models(hp, dev, 1000, width)
            fg.eval(), bg.eval()
        batches = []
        for n in (1024, 777):
            r = rays_all[sel[:n].to(dev)].contiguous()
            batches.append((r, torch.randint(0, s['appearance_count'], (n,), generator=g).float().to(dev)))
        keys = ('rgb_fine', 'depth_fine', 'bg_lambda_fine')
        first, bad = [], torch.zeros((), device=dev, dtype=torch.int64)
        with torch.no_grad():
            for it in range(a.iters):
                b = it % 2 if it % 5 == 4 else 0                     # mostly the benchmark batch, every fifth render the ragged one
                res = render_rays_async(fg, bg, batches[b][0], batches[b][1], hp, sc, sr, True, False, True)[0]
                out = torch.cat([res[k].reshape(-1) for k in keys])
                if len(first) <= b or first[b] is None:
                    while len(first) <= b:
                        first.append(None)
                    first[b] = out.clone()
                else:
                    bad += (out.view(torch.int32) != first[b].view(torch.int32)).any().long()
        torch.cuda.synchronize()
        print(json.dumps({'config': name, 'renders': a.iters, 'renders_differing_from_the_first': int(bad),
                          'finite': bool(torch.isfinite(first[0]).all())}), flush=True)
        del fg, bg
        torch.cuda.empty_cache()
    # the TRAINING forward (tape-writing kernels, feature-split tail included): the same step -- same weights (no optimiser), same Philox
    # counter -- again and again; its colours and depth variances depend on the forward only and must not move by a bit
    from mega_nerf.training import FusedTrainStep
    hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    (fg, _, _), (bg, _, _) = bench.build_models(hp, dev, 1000, 256)
    fg.train(), bg.train()
    fs = FusedTrainStep([(fg, bg)], hp, sc, sr, 1024)
    r = rays_all[sel[:1024].to(dev)].contiguous()
    batch = (r, torch.randint(0, s['appearance_count'], (1024,), generator=g).float().to(dev), torch.rand(1024, 3, generator=g).to(dev))
    first, bad = None, torch.zeros((), device=dev, dtype=torch.int64)
    for it in range(a.iters):
        fs.step_count = 0
        fs([batch], optimize=False)
        out = torch.cat([fs.rgb.reshape(-1), fs.depth_variance.reshape(-1)])
        if first is None:
            first = out.clone()
        else:
            bad += (out.view(torch.int32) != first.view(torch.int32)).any().long()
    torch.cuda.synchronize()
    print(json.dumps({'config': 'default 8x256, training forward (mnr_train_step without the optimiser)', 'steps': a.iters,
                      'steps_differing_from_the_first': int(bad), 'finite': bool(torch.isfinite(first).all())}), flush=True)


if __name__ == '__main__':
    main()
