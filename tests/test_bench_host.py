"""Host-side pieces of bench.py that run without a GPU: usable-thread detection and the time-bounded CPU baseline leg."""
import importlib.util
import time
from pathlib import Path

import numpy as np

import common
from test_oracle_golden import load

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', ROOT / 'bench.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_usable_cores_is_bounded():
    b = _bench()
    n = b.usable_cores()
    assert 1 <= n <= 32
    assert b.usable_cores(cap=2) <= 2


def test_cpu_baseline_is_time_bounded_and_well_formed():
    """The baseline leg sizes its sample from a 32-ray probe, so even a slow host finishes in seconds."""
    b = _bench()
    from mega_nerf.opts import get_opts_base
    hp = get_opts_base().parse_args(['--coarse_samples', '16', '--fine_samples', '16'])
    fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
    fw, bw = common.make_weights(fcfg, 100, 1), common.make_weights(bcfg, 100, 2)
    g = load('render_fgbg_train')
    rays = np.tile(g['rays'], (2, 1))[:64]
    idx = np.tile(g['idx'], 2)[:64].astype(np.float32)
    tgt = np.random.default_rng(0).uniform(0, 1, (64, 3)).astype(np.float32)
    for mode in ('eval', 'train'):
        t0 = time.time()
        out = b.cpu_baseline(hp, rays, idx, tgt, fw, bw, fcfg, bcfg, 64, mode)
        assert time.time() - t0 < 60
        assert out['unit'] == 'rays/s' and out['kind'] == 'port' and out['value'] > 0 and 1 <= out['cores'] <= 32
        assert 'rays' in out['sample']


def test_reference_config_files_parse():
    """The 1-6 line yaml files under the reference's configs/ (contents restated here) map onto the flag set."""
    import tempfile
    from mega_nerf.opts import get_opts_base
    cases = {
        'ray_altitude_range: [11, 38]\n': dict(ray_altitude_range=[11.0, 38.0], layer_dim=256, use_cascade=False),
        'ray_altitude_range: [11, 38]\nsh_deg: 2\npos_dir_dim: 0\n': dict(sh_deg=2, pos_dir_dim=0),
        'ray_altitude_range: [11, 38]\nappearance_dim: 0\nuse_cascade: true\nlayer_dim: 2048\nno_bg_nerf: true\n':
            dict(appearance_dim=0, use_cascade=True, layer_dim=2048, bg_nerf=False),
        'ray_altitude_range: [14, 30]\ncluster_2d: true\nno_ellipse_bounds: true\n': dict(cluster_2d=True, ellipse_bounds=False),
    }
    for text, want in cases.items():
        with tempfile.NamedTemporaryFile('w', suffix='.yaml', delete=False) as f:
            f.write(text)
        hp = get_opts_base().parse_args(['--config_file', f.name, '--batch_size', '2048'])
        for k, v in want.items():
            assert getattr(hp, k) == v, (text, k, getattr(hp, k))
        assert hp.batch_size == 2048                       # command line wins over the file
