"""Wrap ONE trained checkpoint (``--ckpt_path``) as a single-cell container so that it can be fed to the tools that
expect a merged model -- same flags and output as the reference's scripts/convert_to_container.py (:13-72)."""
import sys
from argparse import Namespace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from mega_nerf.merge import save_container, single_cell_container, check_container_on_device   # noqa: E402
from mega_nerf.opts import get_opts_base                                                     # noqa: E402


def _get_merge_opts() -> Namespace:
    parser = get_opts_base()
    parser.add_argument('--output', type=str, required=True)
    return parser.parse_known_args()[0]


@torch.inference_mode()
def main(hparams: Namespace) -> None:
    if hparams.ckpt_path is None:
        raise Exception('--ckpt_path is required')
    save_container(single_cell_container(hparams, Path(hparams.ckpt_path)), hparams.output)
    check_container_on_device(hparams, hparams.output)


if __name__ == '__main__':
    main(_get_merge_opts())
