#!/usr/bin/env python3
"""Times mnr_cluster_min_ratios on a Rubble-sized image (4608 x 3456 rays x 1000 samples x 8 cells) with HIP events.
Prints one JSON line; the kernel is VALU-bound (IEEE sqrt + divide per (sample, cell) pair), so the figure of merit is
(sample, cell) pairs per second."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mega_nerf.cluster_masks import min_dist_ratios          # noqa: E402
from mega_nerf.ray_utils import get_ray_directions, get_rays  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--width', type=int, default=4608)
    ap.add_argument('--height', type=int, default=3456)
    ap.add_argument('--cells', type=int, nargs=2, default=[2, 4])
    ap.add_argument('--samples', type=int, default=1000)
    ap.add_argument('--cluster_2d', action='store_true')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda')
    W, H = a.width, a.height
    c2w = torch.tensor([[.6, 0, -.8, -.3], [0, 1, 0, .1], [.8, 0, .6, .05]])
    dirs = get_ray_directions(W, H, W * 0.8, W * 0.8, W / 2, H / 2, True, dev)
    rays = get_rays(dirs, c2w.to(dev), 0.01, 2.0, [-0.5, 0.2])
    g0, g1 = a.cells
    cen = torch.stack([torch.zeros(g0 * g1), torch.linspace(-.4, .4, g0).repeat_interleave(g1),
                       torch.linspace(-.4, .4, g1).repeat(g0)], 1)
    min_dist_ratios(rays, cen, a.samples, a.cluster_2d, 1.15)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        min_dist_ratios(rays, cen, a.samples, a.cluster_2d, 1.15)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    pairs = W * H * a.samples * g0 * g1
    print(json.dumps({'kernel': 'k_cluster_ratios', 'image': [W, H], 'cells': g0 * g1, 'samples': a.samples,
                      'cluster_2d': a.cluster_2d, 'ms_per_image': round(best, 3), 'gpairs_per_s': round(pairs / best / 1e6, 2),
                      'rays_per_s': round(W * H / best * 1e3)}))


if __name__ == '__main__':
    main()
