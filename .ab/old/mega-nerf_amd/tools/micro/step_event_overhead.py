#!/usr/bin/env python3
"""What the step's own HIP events cost: bench.py times its steps WITH mnr_step_profile on (18 event records per step, the population the
`roofline` numbers come from).  Here the same step, alternately with and without them, K steps per region, R regions each.

    python mega-nerf_amd/tools/micro/step_event_overhead.py [--steps 60] [--regions 4]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'mega-nerf_amd'))

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--regions', type=int, default=4)
    a = ap.parse_args()
    import synthetic_scene as S
    from mega_nerf import ray_utils
    from mega_nerf.opts import get_opts_base
    from mega_nerf.training import FusedTrainStep
    dev = torch.device('cuda')
    s = S.SCENE
    sc, sr = torch.from_numpy(s['sphere_center']).to(dev), torch.from_numpy(s['sphere_radius']).to(dev)
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, dev)
    rays_all = ray_utils.get_rays(d, torch.from_numpy(s['c2w']).to(dev), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8)
    g = torch.Generator(device='cpu').manual_seed(42)
    sel = torch.randperm(rays_all.shape[0], generator=g)[:1024].to(dev)
    batch = (rays_all[sel].contiguous(), torch.randint(0, s['appearance_count'], (1024,), generator=g).float().to(dev),
             torch.rand(1024, 3, generator=g).to(dev))
    hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    (fg, _, _), (bg, _, _) = bench.build_models(hp, dev, 1000, 256)
    fg.train(), bg.train()
    fs = FusedTrainStep([(fg, bg)], hp, sc, sr, 1024)
    for _ in range(10):
        fs([batch])
    out = {'with_events': [], 'without_events': []}
    for r in range(2 * a.regions):
        on = r % 2 == 0
        fs.profile(a.steps if on else 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fs([batch])
        torch.cuda.synchronize()
        out['with_events' if on else 'without_events'].append(round((time.perf_counter() - t0) / a.steps * 1e3, 4))
    fs.profile(0)
    out['mean_ms'] = {k: round(sum(v) / len(v), 4) for k, v in out.items()}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
