"""Cluster-mask step: oracle pinned to masks written by the reference's own
scripts/create_cluster_masks.py (tests/golden/make_golden_masks.py), HIP path checked against both."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

GOLD = Path(__file__).resolve().parent / 'golden'
CASES = ['masks_2x2_3d', 'masks_2x4_2d', 'masks_3x3_hard', 'masks_5x5_wide']
ROOT = Path(__file__).resolve().parent.parent


def load(name):
    g = dict(np.load(GOLD / (name + '.npz')))
    g['masks'] = np.unpackbits(g['masks'])[:int(np.prod(g['masks_shape']))].reshape(g['masks_shape']).astype(bool)
    return g


def oracle_rays(g, i):
    W, H = int(g['W']), int(g['H'])
    fx, fy, cx, cy = [float(v) for v in g['intr']]
    dirs = O.get_ray_directions(W, H, fx, fy, cx, cy, True)
    return O.get_rays(dirs, g['c2w'][i], float(g['near']), float(g['far']), [float(v) for v in g['ray_altitude_range']])


@pytest.mark.parametrize('name', CASES)
def test_oracle_reproduces_reference_masks(name):
    g = load(name)
    t = O.linspace01(int(g['ray_samples']))
    for i in range(g['c2w'].shape[0]):
        rays = oracle_rays(g, i).reshape(-1, 8)
        ratios = O.cluster_min_dist_ratios(rays, g['centroids'], t, bool(g['cluster_2d']))
        got = (ratios <= np.float32(g['margin'])).T.reshape(g['masks'].shape[1:])
        assert np.array_equal(got, g['masks'][i]), (name, i, int((got != g['masks'][i]).sum()))


@pytest.mark.parametrize('name', CASES)
def test_cell_centroids_match_reference(name):
    from mega_nerf.cluster_masks import cell_centroids
    g = load(name)
    cen, lo, hi = cell_centroids(torch.from_numpy(g['c2w'][:, :3, 3]), [int(v) for v in g['grid_dim']])
    assert np.array_equal(cen.numpy(), g['centroids'])
    assert np.array_equal(lo.numpy(), g['min_position']) and np.array_equal(hi.numpy(), g['max_position'])


def test_mask_file_roundtrip(tmp_path):
    from mega_nerf.cluster_masks import read_mask, write_mask
    from mega_nerf.image_metadata import ImageMetadata
    m = torch.rand(7, 9) > 0.5
    write_mask(tmp_path / '000001.pt', m)
    assert torch.equal(read_mask(tmp_path / '000001.pt'), m)
    md = ImageMetadata(Path('x.jpg'), torch.eye(4)[:3], 9, 7, torch.ones(4), 0, tmp_path / '000001.pt', False)
    assert torch.equal(md.load_mask(), m)


# ------------------------------------------------------------------ GPU parity (through the C ABI)

@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_gpu_ratios_bit_exact_vs_oracle(name):
    from mega_nerf.cluster_masks import min_dist_ratios
    g = load(name)
    t = O.linspace01(int(g['ray_samples']))
    for i in range(2):
        rays = oracle_rays(g, i).reshape(-1, 8)
        want = O.cluster_min_dist_ratios(rays, g['centroids'], t, bool(g['cluster_2d']))
        got, masks = min_dist_ratios(torch.from_numpy(rays).cuda(), torch.from_numpy(g['centroids']), int(g['ray_samples']),
                                     bool(g['cluster_2d']), float(g['margin']))
        assert np.array_equal(got.cpu().numpy(), want)          # fp32 values, bit for bit
        assert np.array_equal(masks.cpu().numpy().reshape(g['masks'].shape[1:]), g['masks'][i])


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES[:2])
def test_gpu_script_writes_reference_files(name, tmp_path):
    """The mirrored scripts/create_cluster_masks.py on a dataset in the reference's layout: params.pt and every mask file
    equal to what the reference script wrote for the same dataset."""
    import importlib.util
    from argparse import Namespace
    from mega_nerf.cluster_masks import read_mask
    g = load(name)
    data, out = tmp_path / 'data', tmp_path / 'masks'
    names = [str(s) for s in g['names']]
    for i, nm in enumerate(names):
        sub, stem = nm.split('/')
        (data / sub / 'metadata').mkdir(parents=True, exist_ok=True)
        torch.save({'W': int(g['W']), 'H': int(g['H']), 'c2w': torch.from_numpy(g['c2w'][i]),
                    'intrinsics': torch.from_numpy(g['intr'])}, data / sub / 'metadata' / (stem + '.pt'))
    torch.save({'origin_drb': torch.from_numpy(g['origin_drb']), 'pose_scale_factor': float(g['psf'])}, data / 'coordinates.pt')
    spec = importlib.util.spec_from_file_location('create_cluster_masks', ROOT / 'mega-nerf_amd' / 'scripts' / 'create_cluster_masks.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(Namespace(ray_altitude_range=[float(v) for v in g['hp_altitude']], output=str(out), resume=False, dataset_path=str(data),
                       grid_dim=[int(v) for v in g['grid_dim']], near=float(g['hp_near']), far=None, cluster_2d=bool(g['cluster_2d']),
                       ray_samples=int(g['ray_samples']), center_pixels=True, segmentation_path=None,
                       boundary_margin=float(g['margin'])))
    params = torch.load(out / 'params.pt', map_location='cpu', weights_only=False)
    assert np.array_equal(params['centroids'].numpy(), g['centroids'])
    assert float(params['near']) == float(g['near']) and float(params['far']) == float(g['far'])
    assert [float(x) for x in params['ray_altitude_range']] == [float(x) for x in g['ray_altitude_range']]
    for i, nm in enumerate(names):
        stem = nm.split('/')[1]
        for j in range(g['masks'].shape[1]):
            m = read_mask(out / str(j) / (stem + '.pt'))
            assert m.dtype == torch.bool and np.array_equal(m.numpy(), g['masks'][i, j]), (nm, j)
    # --resume leaves complete outputs alone
    mod.main(Namespace(ray_altitude_range=[float(v) for v in g['hp_altitude']], output=str(out), resume=True, dataset_path=str(data),
                       grid_dim=[int(v) for v in g['grid_dim']], near=float(g['hp_near']), far=None, cluster_2d=bool(g['cluster_2d']),
                       ray_samples=int(g['ray_samples']), center_pixels=True, segmentation_path=None,
                       boundary_margin=float(g['margin'])))


@pytest.mark.gpu
def test_gpu_cluster_masks_large_image_properties():
    """Full-size image (4608 x 3456 would take the oracle hours): size-independent properties instead --
    every ray belongs to its nearest cell (ratio <= 1 for at least one cell), ratios >= ~1 never below
    dmin/(dmin+1e-8), a margin-1.15 mask contains the margin-1.0 mask, and a random subset equals the oracle."""
    from mega_nerf.cluster_masks import min_dist_ratios
    from mega_nerf.ray_utils import get_ray_directions, get_rays
    g = load('masks_2x4_2d')
    dev = torch.device('cuda')
    W, H = 1152, 864
    dirs = get_ray_directions(W, H, 900.0, 900.0, W / 2, H / 2, True, dev)
    rays = get_rays(dirs, torch.from_numpy(g['c2w'][0]).to(dev), float(g['near']), float(g['far']),
                    [float(v) for v in g['ray_altitude_range']])
    cen = torch.from_numpy(g['centroids'])
    r115, m115 = min_dist_ratios(rays, cen, 1000, True, 1.15)
    r100, m100 = min_dist_ratios(rays, cen, 1000, True, 1.0)
    assert torch.equal(r115, r100)
    assert bool((r115.min(-1)[0] <= 1.0).all())
    assert bool((m115 | ~m100).all()) and bool(m100.any(0).all())
    pick = torch.randperm(W * H, generator=torch.Generator().manual_seed(3))[:512]
    sub = rays.view(-1, 8)[pick.to(dev)].cpu().numpy()
    want = O.cluster_min_dist_ratios(sub, g['centroids'], O.linspace01(1000), True)
    assert np.array_equal(r115.view(-1, 8)[pick.to(dev)].cpu().numpy(), want)


@pytest.mark.gpu
def test_gpu_script_segmentation_masks_and_resume(tmp_path):
    """--segmentation_path ANDs an external per-image mask into every cell mask (create_cluster_masks.py:194-208);
    --resume regenerates only images whose mask files are missing or unreadable (:118-139)."""
    import importlib.util
    from argparse import Namespace
    from mega_nerf.cluster_masks import read_mask, write_mask
    g = load('masks_2x2_3d')
    data, out, seg = tmp_path / 'data', tmp_path / 'masks', tmp_path / 'seg'
    seg.mkdir()
    names = [str(s) for s in g['names']]
    rng = np.random.default_rng(0)
    seg_masks = {}
    for i, nm in enumerate(names):
        sub, stem = nm.split('/')
        (data / sub / 'metadata').mkdir(parents=True, exist_ok=True)
        torch.save({'W': int(g['W']), 'H': int(g['H']), 'c2w': torch.from_numpy(g['c2w'][i]), 'intrinsics': torch.from_numpy(g['intr'])},
                   data / sub / 'metadata' / (stem + '.pt'))
        seg_masks[stem] = torch.from_numpy(rng.uniform(size=(int(g['H']), int(g['W']))) > 0.4)
        write_mask(seg / (stem + '.pt'), seg_masks[stem])
    torch.save({'origin_drb': torch.from_numpy(g['origin_drb']), 'pose_scale_factor': float(g['psf'])}, data / 'coordinates.pt')
    spec = importlib.util.spec_from_file_location('create_cluster_masks', ROOT / 'mega-nerf_amd' / 'scripts' / 'create_cluster_masks.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def hp(resume):
        return Namespace(ray_altitude_range=[float(v) for v in g['hp_altitude']], output=str(out), resume=resume, dataset_path=str(data),
                         grid_dim=[int(v) for v in g['grid_dim']], near=float(g['hp_near']), far=None, cluster_2d=bool(g['cluster_2d']),
                         ray_samples=int(g['ray_samples']), center_pixels=True, segmentation_path=str(seg),
                         boundary_margin=float(g['margin']))

    mod.main(hp(False))
    for i, nm in enumerate(names):
        stem = nm.split('/')[1]
        for j in range(g['masks'].shape[1]):
            want = np.logical_and(g['masks'][i, j], seg_masks[stem].numpy())
            assert np.array_equal(read_mask(out / str(j) / (stem + '.pt')).numpy(), want), (nm, j)
    # damage one file, delete another: --resume repairs exactly those images
    stem0, stem1 = names[0].split('/')[1], names[1].split('/')[1]
    (out / '1' / (stem0 + '.pt')).write_bytes(b'not a zip')
    (out / '2' / (stem1 + '.pt')).unlink()
    keep = (out / '0' / (names[2].split('/')[1] + '.pt')).stat().st_mtime_ns
    mod.main(hp(True))
    assert np.array_equal(read_mask(out / '1' / (stem0 + '.pt')).numpy(), np.logical_and(g['masks'][0, 1], seg_masks[stem0].numpy()))
    assert (out / '2' / (stem1 + '.pt')).exists()
    assert (out / '0' / (names[2].split('/')[1] + '.pt')).stat().st_mtime_ns == keep          # untouched
