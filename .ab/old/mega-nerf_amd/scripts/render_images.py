"""Render a list of camera poses with a trained model (a checkpoint or a merged container).

Drop-in for the reference's scripts/render_images.py (:19-144): same command line, same input files, same output tree --

    <input>/poses.txt        one camera-to-world matrix per line: 12 floats, 3 x 4 row-major
    <input>/intrinsics.txt   "W H fx fy cx cy" per line (every number is divided by --val_scale_factor)
    <input>/embeddings.txt   one appearance index per line
    <output>/rgbs/NNNNNN.jpg        the render
    <output>/depths/NNNNNN.jpg      heat map of log depth (background depths clamped to the foreground's 95 % quantile)
    <output>/cells/NNNNNN.jpg       the render tinted by the centroid nearest to every pixel's surface point
    <output>/depths_npz/NNNNNN.npy  depth in scene units (--save_depth_npz)

What happens per pose is ONE ``Runner.render_image`` call: ray generation (csrc/raygen.hip) and render_rays (csrc/step.hip) on the
device.  Surface points and the nearest-centroid search stay on the device as well (the reference moves rays and depth to the host
for them, :125-129).  Poses are striped over the ranks of a multi-process launch (pose i belongs to rank i mod world, :81).
OpenCV is not part of this image: the depth heat map is ``Runner.visualize_scalars`` and the cell tint a plain hue wheel
(hue = cell / number of cells) where the reference goes through OpenCV's COLORMAP_INFERNO / COLORMAP_HSV tables.
"""
import os
import sys
import traceback
from argparse import Namespace
from pathlib import Path
from typing import List, NamedTuple

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.image_metadata import ImageMetadata   # noqa: E402
from mega_nerf.misc_utils import main_tqdm           # noqa: E402
from mega_nerf.opts import get_opts_base             # noqa: E402
from mega_nerf.runner import Runner                  # noqa: E402

# the script's own flags on top of the base set: (name, is a switch)
_FLAGS = (('input', False), ('output', False), ('dataset_path', False), ('centroids_path', False), ('save_depth_npz', True), ('resume', True))
_SUBDIRS = ('rgbs', 'depths', 'cells')


def _get_render_opts(argv=None) -> Namespace:
    p = get_opts_base()
    for name, switch in _FLAGS:
        if switch:
            p.add_argument('--' + name, default=False, action='store_true')
        else:
            p.add_argument('--' + name, type=str, required=True)
    return p.parse_args(argv)


class Pose(NamedTuple):
    c2w: torch.Tensor          # (3, 4)
    width: int
    height: int
    pinhole: torch.Tensor      # fx, fy, cx, cy
    appearance: int


def read_poses(folder: Path, scale: float) -> List[Pose]:
    """The three text files of a pose list, line k of each describing pose k."""
    def table(name):
        return [ln.split() for ln in (folder / name).read_text().splitlines() if ln.strip()]

    cams, pins, apps = table('poses.txt'), table('intrinsics.txt'), table('embeddings.txt')
    if not (len(cams) == len(pins) == len(apps)):
        raise ValueError('poses.txt, intrinsics.txt and embeddings.txt of {} disagree in length: {} / {} / {}'.format(
            folder, len(cams), len(pins), len(apps)))
    poses = []
    for cam, pin, app in zip(cams, pins, apps):
        vals = [float(v) / scale for v in pin]
        poses.append(Pose(torch.tensor([float(v) for v in cam]).view(3, 4), int(vals[0]), int(vals[1]), torch.tensor(vals[2:6]), int(app[0])))
    return poses


def hue_wheel(hue: torch.Tensor) -> torch.Tensor:
    """Fully saturated colours of hue in [0, 1): (..., 3) floats in [0, 255]."""
    k = (hue.unsqueeze(-1) * 6 + torch.tensor([5., 3., 1.], device=hue.device)) % 6
    return (1 - torch.clamp(torch.minimum(k, 4 - k), 0, 1)) * 255


_hue_wheel = hue_wheel


def _finished(marker: Path) -> bool:
    """--resume: the cell overlay is the last file written for a pose, so a readable one means the pose is complete."""
    from PIL import Image
    if not marker.exists():
        return False
    try:
        np.asarray(Image.open(marker))
        return True
    except Exception:
        traceback.print_exc()
        return False


def _write_pose(k: int, pose: Pose, out_dir: Path, runner: Runner, centroids: torch.Tensor, save_npz: bool) -> None:
    from PIL import Image
    stem = '{0:06d}'.format(k)
    meta = ImageMetadata(Path(''), pose.c2w, pose.width, pose.height, pose.pinhole, pose.appearance, None, False)
    out, rays = runner.render_image(meta)
    level = 'fine' if 'rgb_fine' in out else 'coarse'
    H, W = pose.height, pose.width

    colour = (out['rgb_' + level].view(H, W, 3) * 255).byte()
    Image.fromarray(colour.cpu().numpy()).save(out_dir / 'rgbs' / (stem + '.jpg'))

    depth = torch.nan_to_num(out['depth_' + level]).view(H, W)
    if save_npz:
        np.save(str(out_dir / 'depths_npz' / (stem + '.npy')), (depth * runner.pose_scale_factor).cpu().numpy())
    if 'bg_depth_' + level in out:
        # background depths are inverse-sphere quantities of size 1e7 - 1e8: clamp them for display and for the
        # surface points below, as the reference does, to the 95 % quantile of the foreground depths (subsampled to 2^24 values)
        fg = torch.nan_to_num(out['fg_depth_' + level]).reshape(-1)
        halvings = 0
        while (fg.numel() + (1 << halvings) - 1) >> halvings > 2 ** 24:
            halvings += 1
        fg = fg[::1 << halvings]
        depth = depth.clamp_max(torch.quantile(fg, 0.95))
    Image.fromarray(Runner.visualize_scalars(torch.log(depth + 1e-8))).save(out_dir / 'depths' / (stem + '.jpg'))

    grid = rays.view(H, W, -1)
    surface = torch.addcmul(grid[..., 0:3], grid[..., 3:6], depth.unsqueeze(-1))
    nearest = torch.cdist(surface.reshape(-1, 3), centroids).argmin(dim=1).view(H, W)
    share = ((nearest.float() / centroids.shape[0]) * 255).byte().float() / 256.0          # 8-bit cell level, as the reference quantises it
    overlay = (colour.float() * 0.7 + hue_wheel(share) * 0.3 + 0.5).clamp(0, 255).byte()
    Image.fromarray(overlay.cpu().numpy()).save(out_dir / 'cells' / (stem + '.jpg'))


def render_pose_list(hparams: Namespace) -> None:
    runner = Runner(hparams, False)
    out_dir = Path(hparams.output)
    poses = read_poses(Path(hparams.input), float(hparams.val_scale_factor))
    centroids = torch.load(hparams.centroids_path, map_location='cpu', weights_only=False)['centroids'].float().to(runner.device)

    rank, world = int(os.environ.get('RANK', '0')), 1
    if rank == 0:
        for sub in _SUBDIRS + (('depths_npz',) if hparams.save_depth_npz else ()):
            (out_dir / sub).mkdir(parents=True, exist_ok=hparams.resume)       # an existing tree is refused unless --resume
    if runner.distributed:
        torch.distributed.barrier()
        world = int(os.environ['WORLD_SIZE'])

    for m in (runner.nerf, runner.bg_nerf):
        if m is not None:
            m.eval()                                                           # deterministic renders of BOTH models (:73-75)
    with torch.inference_mode():
        for k in main_tqdm(range(rank, len(poses), world)):
            if hparams.resume and _finished(out_dir / 'cells' / '{0:06d}.jpg'.format(k)):
                continue
            _write_pose(k, poses[k], out_dir, runner, centroids, hparams.save_depth_npz)


def main(hparams: Namespace) -> None:
    if hparams.ckpt_path is None and hparams.container_path is None:
        raise AssertionError('render_images needs --ckpt_path or --container_path')
    if hparams.detect_anomalies:
        with torch.autograd.detect_anomaly():
            render_pose_list(hparams)
        return
    render_pose_list(hparams)


if __name__ == '__main__':
    main(_get_render_opts())
