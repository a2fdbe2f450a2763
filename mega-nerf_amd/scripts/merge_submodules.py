"""Collect the per-cell checkpoints of a Mega-NeRF run into one TorchScript container -- same flags and output as the
reference's scripts/merge_submodules.py (:13-79).  (The in-job variant that gathers the weights over RCCL instead of
the filesystem is mega_nerf.merge.merge_in_job.)"""
import sys
from argparse import Namespace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.merge import merge_from_checkpoints, save_container   # noqa: E402
from mega_nerf.models.model_utils import get_bg_nerf, get_nerf       # noqa: E402
from mega_nerf.opts import get_opts_base                             # noqa: E402


def _get_merge_opts() -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--ckpt_prefix', type=str, required=True)
    parser.add_argument('--centroid_path', type=str, required=True)
    parser.add_argument('--output', type=str, required=True)
    return parser.parse_known_args()[0]


@torch.inference_mode()
def main(hparams: Namespace) -> None:
    container = merge_from_checkpoints(hparams)
    save_container(container, hparams.output)
    n_bg = sum(1 for name, _ in container.named_children() if name.startswith('bg_sub_module_'))
    # read the archive back the way eval.py does and evaluate one sample per branch (:82-100)
    if not torch.cuda.is_available():
        print('container written to {}; skipping the test evaluation (no HIP device)'.format(hparams.output))
        return
    device = torch.device('cuda')
    hp = Namespace(**vars(hparams))
    hp.container_path, hp.ckpt_path = hparams.output, None
    width = 3 + (3 if hparams.pos_dir_dim > 0 else 0) + (1 if hparams.appearance_dim > 0 else 0)
    nerf = get_nerf(hp, 0).to(device).eval()
    print('fg test eval: {}'.format(nerf(torch.ones(1, width, device=device))))
    if n_bg > 0:
        bg_nerf = get_bg_nerf(hp, 0).to(device).eval()
        print('bg test eval: {}'.format(bg_nerf(torch.ones(1, width + 4, device=device))))


if __name__ == '__main__':
    main(_get_merge_opts())
