"""Render a list of poses with a trained model (checkpoint or merged container): same flags, input files and output tree as the
reference's scripts/render_images.py (:19-144) --

    <input>/poses.txt        one c2w per line, 12 floats (3 x 4, row-major)
    <input>/intrinsics.txt   W H fx fy cx cy per line (divided by --val_scale_factor)
    <input>/embeddings.txt   appearance index per line
    <output>/rgbs/%06d.jpg, depths/%06d.jpg (log-depth heat map), cells/%06d.jpg (render tinted by the nearest centroid of every
    pixel's surface point), depths_npz/%06d.npy (metric depth, with --save_depth_npz)

Every image is one ``Runner.render_image`` call = ray generation + render_rays on the device (csrc/raygen.hip, csrc/step.hip); poses are
striped over the ranks (pose i -> rank i % world, :81).  The surface points and their nearest centroids stay on the device (the reference
moves rays and depth to the host and runs cdist there, :125-129).  Colour maps: OpenCV is not part of this image; the depth ramp is
``Runner.visualize_scalars`` and the cell tint is the plain hue wheel (hue = cell / n_cells), where the reference uses OpenCV's
COLORMAP_INFERNO / COLORMAP_HSV tables.
"""
import os
import sys
import traceback
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.image_metadata import ImageMetadata   # noqa: E402
from mega_nerf.misc_utils import main_tqdm           # noqa: E402
from mega_nerf.opts import get_opts_base             # noqa: E402
from mega_nerf.runner import Runner                  # noqa: E402


def _get_render_opts(argv=None) -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--input', type=str, required=True)
    parser.add_argument('--output', type=str, required=True)
    parser.add_argument('--dataset_path', type=str, required=True)
    parser.add_argument('--centroids_path', type=str, required=True)
    parser.add_argument('--save_depth_npz', default=False, action='store_true')
    parser.add_argument('--resume', default=False, action='store_true')
    return parser.parse_args(argv)


def _rows(path: Path):
    with path.open() as f:
        return [line.strip().split() for line in f if line.strip()]


def _hue_wheel(h: torch.Tensor) -> torch.Tensor:
    """hue in [0, 1) -> fully saturated RGB in [0, 255] (..., 3)."""
    k = (h.unsqueeze(-1) * 6 + torch.tensor([5., 3., 1.], device=h.device)) % 6
    return (1 - torch.clamp(torch.minimum(k, 4 - k), 0, 1)) * 255


@torch.inference_mode()
def _render_images(hparams: Namespace) -> None:
    from PIL import Image
    runner = Runner(hparams, False)
    inp, output = Path(hparams.input), Path(hparams.output)
    centroids = torch.load(hparams.centroids_path, map_location='cpu', weights_only=False)['centroids'].float().to(runner.device)
    c2ws = [torch.tensor([float(x) for x in row]).view(3, 4) for row in _rows(inp / 'poses.txt')]
    intrinsics = [[float(x) / hparams.val_scale_factor for x in row] for row in _rows(inp / 'intrinsics.txt')]
    embeddings = [int(row[0]) for row in _rows(inp / 'embeddings.txt')]

    rank = int(os.environ.get('RANK', '0'))
    if rank == 0:
        for sub in ('rgbs', 'depths', 'cells') + (('depths_npz',) if hparams.save_depth_npz else ()):
            (output / sub).mkdir(parents=True, exist_ok=hparams.resume)
    world_size = 1
    if runner.distributed:
        dist.barrier()
        world_size = int(os.environ['WORLD_SIZE'])

    runner.nerf.eval()
    if runner.bg_nerf is not None:
        runner.bg_nerf.eval()

    for i in main_tqdm(np.arange(rank, len(c2ws), world_size)):
        cell_path = output / 'cells' / '{0:06d}.jpg'.format(i)
        if hparams.resume and cell_path.exists():
            try:
                np.array(Image.open(cell_path))          # the last file written for a pose: readable = the pose is complete
                continue
            except Exception:
                traceback.print_exc()
        W, H = int(intrinsics[i][0]), int(intrinsics[i][1])
        results, rays = runner.render_image(ImageMetadata(Path(''), c2ws[i], W, H, torch.tensor(intrinsics[i][2:]), embeddings[i], None, False))
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        rgbs = (results[f'rgb_{typ}'].view(H, W, 3) * 255).byte()
        Image.fromarray(rgbs.cpu().numpy()).save(output / 'rgbs' / '{0:06d}.jpg'.format(i))

        depth = torch.nan_to_num(results[f'depth_{typ}']).view(H, W)
        if hparams.save_depth_npz:
            np.save(str(output / 'depths_npz' / '{0:06d}.npy'.format(i)), (depth * runner.pose_scale_factor).cpu().numpy())
        if f'bg_depth_{typ}' in results:
            # background depths are inverse-sphere quantities of size 1e7-1e8: clamp to the foreground's 95 % quantile
            to_use = torch.nan_to_num(results[f'fg_depth_{typ}']).view(-1)
            while to_use.shape[0] > 2 ** 24:
                to_use = to_use[::2]
            depth = depth.clamp_max(torch.quantile(to_use, 0.95))
        Image.fromarray(Runner.visualize_scalars(torch.log(depth + 1e-8))).save(output / 'depths' / '{0:06d}.jpg'.format(i))

        rays = rays.view(H, W, -1)
        locations = rays[..., :3] + rays[..., 3:6] * depth.unsqueeze(-1)
        cells = torch.cdist(locations.view(-1, 3), centroids).argmin(dim=1).view(H, W).float() / len(centroids)
        tint = _hue_wheel((cells * 255).byte().float() / 256.0)
        blend = (rgbs.float() * 0.7 + tint * 0.3 + 0.5).clamp(0, 255).byte()
        Image.fromarray(blend.cpu().numpy()).save(cell_path)


def main(hparams: Namespace) -> None:
    assert hparams.ckpt_path is not None or hparams.container_path is not None
    if hparams.detect_anomalies:
        with torch.autograd.detect_anomaly():
            _render_images(hparams)
    else:
        _render_images(hparams)


if __name__ == '__main__':
    main(_get_render_opts())
