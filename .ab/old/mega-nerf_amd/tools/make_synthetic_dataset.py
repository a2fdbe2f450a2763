#!/usr/bin/env python3
"""Write a small synthetic dataset in the Mega-NeRF on-disk layout (README.md:79-88 of the reference):

    <out>/coordinates.pt                      {'origin_drb': (3,), 'pose_scale_factor': float}
    <out>/{train,val}/metadata/<stem>.pt      {'W','H','intrinsics': [fx,fy,cx,cy], 'c2w': (3,4)}
    <out>/{train,val}/rgbs/<stem>.png

The images are renders of a seeded random "teacher" field (fg + bg NeRF with default init, sharpened density)
through the MI355X renderer, so the views are mutually consistent and a student can be trained on them.
Nothing in the reference produces synthetic data; BASELINE config 1 needs this.
"""
import argparse
import sys
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf import ray_utils  # noqa: E402
from mega_nerf.models.nerf import NeRF, ShiftedSoftplus  # noqa: E402
from mega_nerf.opts import get_opts_base  # noqa: E402
from mega_nerf.rendering import render_rays  # noqa: E402


def look_at(pos: np.ndarray, target: np.ndarray) -> np.ndarray:
    """c2w (3,4) in the dataset's axis convention (x = down, y = right, z = back)."""
    back = pos - target
    back /= np.linalg.norm(back)
    down = np.array([1.0, 0, 0])
    right = np.cross(back, down)
    right /= np.linalg.norm(right)
    down = np.cross(right, back)
    return np.stack([down, right, back, pos], 1).astype(np.float32)


def _mk(p: Path) -> Path:
    p.mkdir(parents=True, exist_ok=True)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--val_every', type=int, default=8)
    ap.add_argument('--size', type=int, default=400)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--samples', type=int, nargs=2, default=[64, 128])
    args = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(args.seed)
    rng = np.random.default_rng(args.seed)
    hp = get_opts_base().parse_args(['--coarse_samples', str(args.samples[0]), '--fine_samples', str(args.samples[1])])
    hp = Namespace(**vars(hp))

    def teacher(xyz_dim):
        m = NeRF(12, 4, 8, [4], 256, 48, False, args.images, 3, xyz_dim, ShiftedSoftplus())
        with torch.no_grad():
            m.sigma.weight *= 40
            m.sigma.bias += 2
        return m.to(dev).eval()
    fg, bg = teacher(3), teacher(4)

    out = Path(args.out)
    W = H = args.size
    f = 0.75 * W
    intr = torch.tensor([f, f, W / 2, H / 2])
    altitude = [-0.5, 0.2]                                   # normalised units; x points down
    sphere_c = torch.tensor([-0.15, 0.0, 0.0], device=dev)
    sphere_r = torch.tensor([0.6, 1.2, 1.2], device=dev)
    torch.save({'origin_drb': torch.zeros(3), 'pose_scale_factor': 1.0}, _mk(out) / 'coordinates.pt')
    dirs = ray_utils.get_ray_directions(W, H, f, f, W / 2, H / 2, True, dev)
    from PIL import Image
    for i in range(args.images):
        ang = 2 * np.pi * i / args.images
        pos = np.array([-0.3 + 0.05 * rng.standard_normal(), 0.35 * np.cos(ang), 0.35 * np.sin(ang)])
        c2w = torch.from_numpy(look_at(pos, np.array([0.1, 0.0, 0.0])))
        split = 'val' if i % args.val_every == args.val_every - 1 else 'train'
        stem = '%06d' % i
        rays = ray_utils.get_rays(dirs, c2w.to(dev), 0.01, 1e5, altitude).view(-1, 8)
        idx = torch.full((rays.shape[0],), float(i), device=dev)
        rgb = []
        with torch.no_grad():
            for s in range(0, rays.shape[0], 65536):
                res, _ = render_rays(fg, bg, rays[s:s + 65536], idx[s:s + 65536], hp, sphere_c, sphere_r, False, False,
                                     False)
                rgb.append(res['rgb_fine'])
        img = (torch.cat(rgb).clamp(0, 1).view(H, W, 3) * 255).round().byte().cpu().numpy()
        Image.fromarray(img).save(_mk(out / split / 'rgbs') / (stem + '.png'))
        torch.save({'W': W, 'H': H, 'intrinsics': intr.clone(), 'c2w': c2w}, _mk(out / split / 'metadata') / (stem + '.pt'))
    print('wrote', args.images, 'images to', out, '(ray_altitude_range', altitude, ', near 0.01)')


if __name__ == '__main__':
    main()
