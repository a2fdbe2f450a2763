"""Spatial-cell masks for per-submodule training -- same flags, outputs and on-disk layout as the reference's
scripts/create_cluster_masks.py (:19-210): ``<output>/params.pt`` and ``<output>/<cell>/<image>.pt`` (ZIP with one
torch-saved bool[H, W]).  One process per GPU, images striped ``rank::world_size`` (no data-path collective);
ray generation and the sample/centroid distance loop run as HIP kernels (mega_nerf.cluster_masks)."""
import datetime
import os
import sys
import traceback
from argparse import Namespace
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf import _native as N                                                          # noqa: E402
from mega_nerf.cluster_masks import cell_centroids, min_dist_ratios, read_mask, write_mask   # noqa: E402
from mega_nerf.misc_utils import main_print, main_tqdm                                       # noqa: E402
from mega_nerf.opts import get_opts_base                                                     # noqa: E402
from mega_nerf.ray_utils import get_ray_directions, get_rays                                 # noqa: E402


def _get_mask_opts() -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--dataset_path', type=str, required=True)
    parser.add_argument('--segmentation_path', type=str, default=None)
    parser.add_argument('--output', type=str, required=True)
    parser.add_argument('--grid_dim', nargs='+', type=int, required=True)
    parser.add_argument('--ray_samples', type=int, default=1000)
    parser.add_argument('--ray_chunk_size', type=int, default=48 * 1024, help='accepted for compatibility (unused)')
    parser.add_argument('--dist_chunk_size', type=int, default=64 * 1024 * 1024, help='accepted for compatibility (unused)')
    parser.add_argument('--resume', default=False, action='store_true')
    return parser.parse_known_args()[0]


def _already_done(output_path: Path, filename: str, n_cells: int) -> bool:
    for j in range(n_cells):
        mask_path = output_path / str(j) / filename
        if not mask_path.exists():
            return False
        try:
            read_mask(mask_path)
        except Exception:
            traceback.print_exc()
            return False
    return True


@torch.inference_mode()
def main(hparams: Namespace) -> None:
    assert hparams.ray_altitude_range is not None
    if not torch.cuda.is_available():
        raise N.NativeError('create_cluster_masks needs a HIP device; there is no CPU fallback')
    output_path = Path(hparams.output)
    distributed = 'RANK' in os.environ
    if distributed:
        torch.cuda.set_device(int(os.environ['LOCAL_RANK']))
        dist.init_process_group(backend='nccl', timeout=datetime.timedelta(0, hours=24))
        rank, world_size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
        if rank == 0:
            output_path.mkdir(parents=True, exist_ok=hparams.resume)
        dist.barrier()
    else:
        rank, world_size = 0, 1
        output_path.mkdir(parents=True, exist_ok=hparams.resume)
    device = torch.device('cuda', torch.cuda.current_device())

    dataset_path = Path(hparams.dataset_path)
    coordinate_info = torch.load(dataset_path / 'coordinates.pt', map_location='cpu')
    origin_drb, pose_scale_factor = coordinate_info['origin_drb'], coordinate_info['pose_scale_factor']
    ray_altitude_range = [(x - origin_drb[0]) / pose_scale_factor for x in hparams.ray_altitude_range]

    metadata_paths = list((dataset_path / 'train' / 'metadata').iterdir()) + list((dataset_path / 'val' / 'metadata').iterdir())
    camera_positions = torch.stack([torch.load(x, map_location='cpu')['c2w'][:3, 3] for x in metadata_paths])
    main_print('Number of images in dir: {}'.format(camera_positions.shape))
    centroids, min_position, max_position = cell_centroids(camera_positions, hparams.grid_dim)
    main_print('Coord range: {} {}'.format(min_position, max_position))
    main_print('Centroids: {}'.format(centroids))

    near = hparams.near / pose_scale_factor
    far = hparams.far / pose_scale_factor if hparams.far is not None else 2

    if rank == 0:
        torch.save({'origin_drb': origin_drb, 'pose_scale_factor': pose_scale_factor, 'ray_altitude_range': ray_altitude_range,
                    'near': near, 'far': far, 'centroids': centroids, 'grid_dim': (hparams.grid_dim),
                    'min_position': min_position, 'max_position': max_position, 'cluster_2d': hparams.cluster_2d},
                   output_path / 'params.pt')
        if not hparams.resume:
            for i in range(centroids.shape[0]):
                (output_path / str(i)).mkdir(parents=True)
    if distributed:
        dist.barrier()

    centroids_dev = centroids.to(device)
    n_cells = centroids.shape[0]
    for subdir in ['train', 'val']:
        metadata_paths = list((dataset_path / subdir / 'metadata').iterdir())
        for i in main_tqdm(range(rank, len(metadata_paths), world_size)):
            metadata_path = metadata_paths[i]
            filename = metadata_path.stem + '.pt'
            if hparams.resume and _already_done(output_path, filename, n_cells):
                continue
            metadata = torch.load(metadata_path, map_location='cpu')
            intrinsics = metadata['intrinsics']
            directions = get_ray_directions(metadata['W'], metadata['H'], intrinsics[0], intrinsics[1], intrinsics[2],
                                            intrinsics[3], hparams.center_pixels, device)
            rays = get_rays(directions, metadata['c2w'].to(device), near, far, ray_altitude_range)
            _, masks = min_dist_ratios(rays, centroids_dev, hparams.ray_samples, hparams.cluster_2d, hparams.boundary_margin)
            masks = masks.cpu()                                   # (cells, H, W) bool, one device->host copy per image
            segmentation_mask = None
            if hparams.segmentation_path is not None:
                segmentation_mask = read_mask(Path(hparams.segmentation_path) / filename)
            for j in range(n_cells):
                cell_mask = masks[j].clone()
                if segmentation_mask is not None:
                    cell_mask = torch.logical_and(cell_mask, segmentation_mask)
                write_mask(output_path / str(j) / filename, cell_mask)


if __name__ == '__main__':
    main(_get_mask_opts())
