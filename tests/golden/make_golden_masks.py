#!/usr/bin/env python3
"""Golden fixtures for the cluster-mask step, produced by running the REAL reference script
(/root/reference/scripts/create_cluster_masks.py, torch CPU) on tiny synthetic datasets.

Build container only.  ``configargparse`` (absent here) is only needed by the reference's CLI parser, which this
script bypasses by handing ``main`` a Namespace, so an empty stand-in module is registered for the import.
Stored per case: the dataset inputs (poses, intrinsics, coordinates), the flags, params.pt's contents and every mask
(bit-packed).  Nothing from the reference is copied -- only what it computes is recorded."""
import importlib.util
import sys
import tempfile
import types
from argparse import Namespace
from pathlib import Path
from zipfile import ZipFile

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.modules.setdefault('configargparse', types.ModuleType('configargparse'))
spec = importlib.util.spec_from_file_location('ref_create_cluster_masks', '/root/reference/scripts/create_cluster_masks.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

f32 = np.float32

CASES = {
    # name: (grid_dim, cluster_2d, boundary_margin, ray_samples, W, H)
    'masks_2x2_3d': ([2, 2], False, 1.15, 1000, 40, 30),
    'masks_2x4_2d': ([2, 4], True, 1.15, 1000, 40, 30),
    'masks_3x3_hard': ([3, 3], False, 1.0, 257, 24, 18),
    'masks_5x5_wide': ([5, 5], True, 2.0, 1100, 16, 12),
}


def poses(n, rng):
    """Drone-like cameras: x is 'down', looking obliquely at the ground from slightly different spots."""
    out = []
    for i in range(n):
        yaw = rng.uniform(-0.6, 0.6)
        tilt = rng.uniform(0.5, 1.1)
        # camera axes (columns): right, up, back  (rays leave along -back)
        back = np.array([-np.sin(tilt), -np.cos(tilt) * np.sin(yaw), -np.cos(tilt) * np.cos(yaw)])
        right = np.cross([1.0, 0.0, 0.0], back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        pos = np.array([rng.uniform(-0.35, -0.2), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)])
        out.append(np.concatenate([np.stack([right, up, back], 1), pos[:, None]], 1).astype(f32))
    return out


def run_case(name, grid_dim, cluster_2d, margin, ray_samples, W, H, seed):
    rng = np.random.default_rng(seed)
    n_train, n_val = 3, 1
    c2ws = poses(n_train + n_val, rng)
    intr = np.array([W * 0.8, W * 0.8, W / 2.0, H / 2.0], f32)
    origin_drb = np.array([10.0, 2.0, -3.0], f32)
    psf = 50.0
    with tempfile.TemporaryDirectory() as tmp:
        data, out = Path(tmp) / 'data', Path(tmp) / 'masks'
        names = []
        for i, c2w in enumerate(c2ws):
            sub = 'train' if i < n_train else 'val'
            (data / sub / 'metadata').mkdir(parents=True, exist_ok=True)
            stem = '{:06d}'.format(i)
            names.append((sub, stem))
            torch.save({'W': W, 'H': H, 'c2w': torch.from_numpy(c2w), 'intrinsics': torch.from_numpy(intr)},
                       data / sub / 'metadata' / (stem + '.pt'))
        torch.save({'origin_drb': torch.from_numpy(origin_drb), 'pose_scale_factor': psf}, data / 'coordinates.pt')
        hp = Namespace(ray_altitude_range=[-15.0, 20.0], output=str(out), resume=False, dataset_path=str(data), grid_dim=grid_dim,
                       near=0.5, far=None, cluster_2d=cluster_2d, ray_samples=ray_samples, center_pixels=True,
                       ray_chunk_size=48 * 1024, dist_chunk_size=64 * 1024 * 1024, segmentation_path=None,
                       boundary_margin=margin)
        ref.main(hp)
        params = torch.load(out / 'params.pt', map_location='cpu', weights_only=False)
        n_cells = params['centroids'].shape[0]
        masks = np.zeros((len(names), n_cells, H, W), bool)
        for i, (_, stem) in enumerate(names):
            for j in range(n_cells):
                with ZipFile(out / str(j) / (stem + '.pt')) as zf, zf.open(stem + '.pt') as f:
                    m = torch.load(f, map_location='cpu')
                assert m.dtype == torch.bool and tuple(m.shape) == (H, W)
                masks[i, j] = m.numpy()
    np.savez_compressed(HERE / (name + '.npz'), c2w=np.stack(c2ws), intr=intr, W=W, H=H, origin_drb=origin_drb, psf=psf,
                        n_train=n_train, grid_dim=np.array(grid_dim), cluster_2d=cluster_2d, margin=margin, ray_samples=ray_samples,
                        hp_near=0.5, hp_altitude=np.array([-15.0, 20.0]),
                        centroids=params['centroids'].numpy(), min_position=params['min_position'].numpy(),
                        max_position=params['max_position'].numpy(), near=float(params['near']), far=float(params['far']),
                        ray_altitude_range=np.array([float(x) for x in params['ray_altitude_range']], f32),
                        masks=np.packbits(masks), masks_shape=np.array(masks.shape),
                        names=np.array([s + '/' + n for s, n in names]))
    print(name, 'cells', n_cells, 'coverage per cell', masks.mean((0, 2, 3)).round(3))


if __name__ == '__main__':
    torch.manual_seed(0)
    for k, (name, cfg) in enumerate(CASES.items()):
        run_case(name, *cfg, seed=100 + k)
