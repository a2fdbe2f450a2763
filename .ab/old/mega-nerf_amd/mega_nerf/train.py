"""Training entry point: ``python -m mega_nerf.train --config_file ... --exp_name ... --dataset_path ...``.
Same flags and behaviour as the reference's mega_nerf/train.py; both entry points share mega_nerf.runner.run_cli."""
from argparse import Namespace

from mega_nerf.runner import cli_options, run_cli


def _get_train_opts() -> Namespace:
    return cli_options()


def main(hparams: Namespace) -> None:
    run_cli(hparams, 'train')


if __name__ == '__main__':
    main(_get_train_opts())
