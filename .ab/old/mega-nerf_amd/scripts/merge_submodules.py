"""Collect the per-cell checkpoints of a Mega-NeRF run into one TorchScript container -- same flags and output as the
reference's scripts/merge_submodules.py (:13-79).  (The in-job variant that gathers the weights over RCCL instead of
the filesystem is mega_nerf.merge.merge_in_job.)"""
import sys
from argparse import Namespace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.merge import merge_from_checkpoints, save_container, check_container_on_device   # noqa: E402
from mega_nerf.opts import get_opts_base                             # noqa: E402


def _get_merge_opts() -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--ckpt_prefix', type=str, required=True)
    parser.add_argument('--centroid_path', type=str, required=True)
    parser.add_argument('--output', type=str, required=True)
    return parser.parse_known_args()[0]


@torch.inference_mode()
def main(hparams: Namespace) -> None:
    save_container(merge_from_checkpoints(hparams), hparams.output)
    check_container_on_device(hparams, hparams.output)          # read back + one sample per branch (:82-100)


if __name__ == '__main__':
    main(_get_merge_opts())
