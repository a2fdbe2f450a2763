"""Hyper-parameters with the reference's flag names and defaults (mega_nerf/opts.py:4-103).

The reference uses ``configargparse`` (absent here); this is plain argparse plus ``--config_file <yaml>`` whose
keys are flag names (the 1-6 line files under ``configs/`` of the reference work unchanged: bare ``flag: true``
entries switch store_true/store_false flags on)."""
import argparse
import sys


class _Parser(argparse.ArgumentParser):
    def parse_known_args(self, args=None, namespace=None):
        args = list(sys.argv[1:] if args is None else args)
        cfg = None
        for i, a in enumerate(args):
            if a == '--config_file' and i + 1 < len(args):
                cfg = args[i + 1]
            elif a.startswith('--config_file='):
                cfg = a.split('=', 1)[1]
        if cfg is not None:
            import yaml
            with open(cfg) as f:
                items = yaml.safe_load(f) or {}
            extra = []
            for k, v in items.items():
                if isinstance(v, bool):
                    if v:
                        extra.append('--' + k)
                elif isinstance(v, (list, tuple)):
                    extra += ['--' + k] + [str(x) for x in v]
                else:
                    extra += ['--' + k, str(v)]
            args = extra + args          # command line wins over the file
        return super().parse_known_args(args, namespace)


def get_opts_base() -> argparse.ArgumentParser:
    p = _Parser()
    a = p.add_argument
    a('--config_file', type=str, default=None)
    a('--dataset_type', type=str, default='memory', choices=['filesystem', 'memory'],
      help='"memory": the whole ray set resident in HBM (default here; the reference defaults to "filesystem"); '
           '"filesystem": the reference\'s parquet chunk directories (--chunk_paths), one chunk resident at a time')
    a('--chunk_paths', type=str, nargs='+', default=None)
    a('--num_chunks', type=int, default=200)
    a('--disk_flush_size', type=int, default=10000000)
    a('--train_every', type=int, default=1)
    a('--cluster_mask_path', type=str, default=None)
    a('--ckpt_path', type=str, default=None)
    a('--container_path', type=str, default=None)
    a('--near', type=float, default=1)
    a('--far', type=float, default=None)
    a('--ray_altitude_range', nargs='+', type=float, default=None)
    a('--coarse_samples', type=int, default=256)
    a('--fine_samples', type=int, default=512)
    a('--train_scale_factor', type=int, default=1)
    a('--val_scale_factor', type=int, default=4)
    a('--pos_xyz_dim', type=int, default=12)
    a('--pos_dir_dim', type=int, default=4)
    a('--layers', type=int, default=8)
    a('--skip_layers', type=int, nargs='+', default=[4])
    a('--layer_dim', type=int, default=256)
    a('--bg_layer_dim', type=int, default=256)
    a('--appearance_dim', type=int, default=48)
    a('--affine_appearance', default=False, action='store_true')
    a('--use_cascade', default=False, action='store_true')
    a('--train_mega_nerf', type=str, default=None)
    a('--boundary_margin', type=float, default=1.15)
    a('--all_val', default=False, action='store_true')
    a('--cluster_2d', default=False, action='store_true')
    a('--sh_deg', type=int, default=None)
    a('--no_center_pixels', dest='center_pixels', default=True, action='store_false')
    a('--no_shifted_softplus', dest='shifted_softplus', default=True, action='store_false')
    a('--batch_size', type=int, default=1024)
    a('--image_pixel_batch_size', type=int, default=64 * 1024)
    a('--model_chunk_size', type=int, default=32 * 1024)
    a('--perturb', type=float, default=1.0)
    a('--noise_std', type=float, default=1.0)
    a('--lr', type=float, default=5e-4)
    a('--lr_decay_factor', type=float, default=0.1)
    a('--no_bg_nerf', dest='bg_nerf', default=True, action='store_false')
    a('--ellipse_scale_factor', type=float, default=1.1)
    a('--no_ellipse_bounds', dest='ellipse_bounds', default=True, action='store_false')
    a('--train_iterations', type=int, default=500000)
    a('--val_interval', type=int, default=500001)
    a('--ckpt_interval', type=int, default=10000)
    a('--no_resume_ckpt_state', dest='resume_ckpt_state', default=True, action='store_false')
    a('--no_amp', dest='amp', default=True, action='store_false',
      help='accepted for compatibility: the MI355X kernels always compute in fp32 (exact fp32 MFMA)')
    a('--detect_anomalies', default=False, action='store_true')
    a('--random_seed', type=int, default=42)
    return p
