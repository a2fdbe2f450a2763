"""End-to-end drop-in check on the GPU: synthetic dataset in the reference's on-disk layout -> train.main ->
checkpoint with the reference's keys -> eval.main -> metrics.txt."""
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_train_then_eval_entry_points(tmp_path):
    from mega_nerf import eval as ev
    from mega_nerf import train as tr
    from mega_nerf.opts import get_opts_base
    data = tmp_path / 'data'
    subprocess.run([sys.executable, str(ROOT / 'mega-nerf_amd' / 'tools' / 'make_synthetic_dataset.py'), '--out', str(data),
                    '--images', '8', '--val_every', '4', '--size', '32', '--samples', '32', '64'], check=True)
    assert (data / 'coordinates.pt').exists() and len(list((data / 'val' / 'metadata').iterdir())) == 2
    common = ['--dataset_path', str(data), '--coarse_samples', '32', '--fine_samples', '64', '--near', '0.01',
              '--ray_altitude_range', '-0.5', '0.2', '--val_scale_factor', '1', '--batch_size', '512']

    def parse(extra):
        p = get_opts_base()
        p.add_argument('--exp_name', type=str, required=True)
        p.add_argument('--dataset_path', type=str, required=True)
        return p.parse_args(common + extra)

    exp = tmp_path / 'exp'
    tr.main(parse(['--exp_name', str(exp), '--train_iterations', '40', '--ckpt_interval', '20']))
    run0 = exp / '0'
    ck = torch.load(run0 / 'models' / '40.pt', map_location='cpu', weights_only=False)
    for key in ('model_state_dict', 'bg_model_state_dict', 'optimizers', 'iteration', 'torch_random_state',
                'np_random_state', 'random_state', 'dataset_index', 'scaler'):
        assert key in ck, key
    assert ck['iteration'] == 40 and 'xyz_encodings.0.0.weight' in ck['model_state_dict']
    assert (run0 / 'models' / '20.pt').exists()
    m_train = (run0 / 'metrics.txt').read_text()
    assert 'Average val/psnr' in m_train
    for f in ('hparams.txt', 'command.txt', 'image_indices.txt'):
        assert (run0 / f).exists()
    # evaluation of the checkpoint through eval.py reproduces the validation PSNR written after training
    ev.main(parse(['--exp_name', str(exp), '--ckpt_path', str(run0 / 'models' / '40.pt')]))
    m_eval = (exp / '1' / 'metrics.txt').read_text()
    def metric(text, key):
        return float([ln for ln in text.splitlines() if ln.startswith('Average ' + key)][0].split(':')[1])

    a, b = metric(m_train, 'val/psnr'), metric(m_eval, 'val/psnr')
    assert abs(a - b) < 0.05, (a, b)            # north star: PSNR within 0.05 dB
    assert a > 5.0
    assert abs(metric(m_train, 'val/ssim') - metric(m_eval, 'val/ssim')) < 1e-3 and 0.0 < metric(m_eval, 'val/ssim') <= 1.0
