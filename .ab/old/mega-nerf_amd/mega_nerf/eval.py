"""Evaluation entry point: ``python -m mega_nerf.eval --ckpt_path | --container_path ... --exp_name ... --dataset_path ...``.
Same flags and behaviour as the reference's mega_nerf/eval.py; both entry points share mega_nerf.runner.run_cli."""
from argparse import Namespace

from mega_nerf.runner import cli_options, run_cli


def _get_eval_opts() -> Namespace:
    return cli_options()


def main(hparams: Namespace) -> None:
    if hparams.ckpt_path is None and hparams.container_path is None:
        raise AssertionError('evaluation needs --ckpt_path or --container_path')
    run_cli(hparams, 'eval')


if __name__ == '__main__':
    main(_get_eval_opts())
