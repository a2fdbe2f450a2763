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


def test_submodule_to_rank_mapping_of_the_strong_scaling_mode():
    """bench.py --submodules S deals cell j to rank j % world (parscripts/run_8.txt: one per GPU at 8 GPUs; Building's 25 cells
    on 8 GPUs -> 4,3,3,3,3,3,3,3), every cell exactly once, and a cell's seeds (weights 1000 (c+1), batch 42 + c) do not
    depend on the world size -- so the total work of a step is the same at every N."""
    from mega_nerf.distributed import assign_submodules
    for n_cells, world in ((8, 1), (8, 2), (8, 4), (8, 8), (25, 8)):
        parts = assign_submodules(n_cells, world)
        assert len(parts) == world and sorted(c for p in parts for c in p) == list(range(n_cells))
        assert all(c % world == r for r, p in enumerate(parts) for c in p)
    assert [len(p) for p in assign_submodules(25, 8)] == [4, 3, 3, 3, 3, 3, 3, 3]
    assert assign_submodules(8, 8) == [[i] for i in range(8)]


def test_psnr_protocol_problem_is_seeded_and_disjoint():
    """The north-star PSNR check (bench.py: psnr_problem / psnr_gpu / psnr_cpu) trains both implementations on the same
    batches with the same random numbers: the problem must be reproducible and its test rays held out."""
    b = _bench()
    from mega_nerf.opts import get_opts_base
    hp = get_opts_base().parse_args(['--coarse_samples', '64', '--fine_samples', '128'])
    rays = np.random.default_rng(5).uniform(-1, 1, (20000, 8)).astype(np.float32)
    p1, p2 = b.psnr_problem(hp, rays), b.psnr_problem(hp, rays)
    assert len(p1['batches']) == b.PSNR_STEPS and p1['test'].shape == (b.PSNR_TEST_RAYS, 8)
    for (r1, d1), (r2, d2) in zip(p1['batches'], p2['batches']):
        assert np.array_equal(r1, r2) and all(np.array_equal(d1[k], d2[k]) for k in d1)
        assert d1['fg_perturb'].shape == (b.PSNR_BATCH, 64) and d1['bg_u'].shape == (b.PSNR_BATCH, 64) and d1['fg_noise_fine'].shape == (b.PSNR_BATCH * 128,)
    train_rows = {r.tobytes() for batch, _ in p1['batches'] for r in batch}
    assert not any(r.tobytes() in train_rows for r in p1['test'])
    assert not np.array_equal(p1['teacher'][0]['sigma.weight'], p1['student'][0]['sigma.weight'])


def test_torch_oracle_accepts_injected_randoms():
    """oracle/torch_oracle.render_rays with ``randoms`` is deterministic in training mode and differs from another draw."""
    import torch
    from oracle import torch_oracle as TO
    from mega_nerf.opts import get_opts_base
    hp = get_opts_base().parse_args(['--coarse_samples', '16', '--fine_samples', '16'])
    fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
    fg, bg = TO.make_models(hp, fcfg, common.make_weights(fcfg, 100, 1), bcfg, common.make_weights(bcfg, 100, 2), 100)
    fg.train(), bg.train()
    g = load('render_fgbg_train')
    rays, idx = torch.from_numpy(g['rays'][:16]), torch.from_numpy(g['idx'][:16].astype(np.float32))
    sc, sr = torch.from_numpy(common.SCENE['sphere_center']), torch.from_numpy(common.SCENE['sphere_radius'])

    def rnd(seed):
        r = np.random.default_rng(seed)
        return {k: torch.from_numpy(r.random(shape, dtype=np.float32)) for k, shape in
                (('fg_perturb', (16, 16)), ('fg_noise_coarse', (256,)), ('fg_u', (16, 16)), ('fg_noise_fine', (256,)),
                 ('bg_perturb', (16, 8)), ('bg_noise_coarse', (128,)), ('bg_u', (16, 8)), ('bg_noise_fine', (128,)))}
    with torch.no_grad():
        a = TO.render_rays(fg, bg, rays, idx, hp, sc, sr, rnd(1))['rgb_fine']
        b = TO.render_rays(fg, bg, rays, idx, hp, sc, sr, rnd(1))['rgb_fine']
        c = TO.render_rays(fg, bg, rays, idx, hp, sc, sr, rnd(2))['rgb_fine']
    assert torch.equal(a, b) and not torch.equal(a, c)


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


REFERENCE_FLAGS = ['--config_file', '--dataset_type', '--chunk_paths', '--num_chunks', '--disk_flush_size', '--train_every',
                   '--cluster_mask_path', '--ckpt_path', '--container_path', '--near', '--far', '--ray_altitude_range', '--coarse_samples',
                   '--fine_samples', '--train_scale_factor', '--val_scale_factor', '--pos_xyz_dim', '--pos_dir_dim', '--layers',
                   '--skip_layers', '--layer_dim', '--bg_layer_dim', '--appearance_dim', '--affine_appearance', '--use_cascade',
                   '--train_mega_nerf', '--boundary_margin', '--all_val', '--cluster_2d', '--sh_deg', '--no_center_pixels',
                   '--no_shifted_softplus', '--batch_size', '--image_pixel_batch_size', '--model_chunk_size', '--perturb', '--noise_std',
                   '--lr', '--lr_decay_factor', '--no_bg_nerf', '--ellipse_scale_factor', '--no_ellipse_bounds', '--train_iterations',
                   '--val_interval', '--ckpt_interval', '--no_resume_ckpt_state', '--no_amp', '--detect_anomalies', '--random_seed']


def test_flag_set_is_the_reference_flag_set():
    """Every flag of the reference's opts.get_opts_base() (names restated above) exists here, and nothing else does, so
    command lines and config files carry over; the defaults that define "the Rubble config" are the reference's."""
    from mega_nerf.opts import get_opts_base
    parser = get_opts_base()
    mine = sorted(s for a in parser._actions for s in a.option_strings if s.startswith('--') and s != '--help')
    assert mine == sorted(REFERENCE_FLAGS)
    hp = parser.parse_args([])
    assert (hp.coarse_samples, hp.fine_samples, hp.layer_dim, hp.bg_layer_dim, hp.layers, hp.skip_layers) == (256, 512, 256, 256, 8, [4])
    assert (hp.pos_xyz_dim, hp.pos_dir_dim, hp.appearance_dim, hp.batch_size, hp.boundary_margin) == (12, 4, 48, 1024, 1.15)
    assert hp.bg_nerf and hp.ellipse_bounds and hp.center_pixels and hp.shifted_softplus and hp.sh_deg is None
    assert (hp.lr, hp.lr_decay_factor, hp.train_iterations, hp.random_seed, hp.perturb) == (5e-4, 0.1, 500000, 42, 1.0)


def test_script_flags_and_depth_ramp():
    """scripts/render_images.py takes the reference script's flags on top of the base set (scripts/render_images.py:19-29 of the
    reference), and Runner.visualize_scalars maps near -> bright, far -> dark between the 5 % / 95 % quantiles (runner.py:598-610)."""
    import importlib.util
    import numpy as np
    import torch
    from pathlib import Path
    from mega_nerf.runner import Runner
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location('render_images', root / 'mega-nerf_amd' / 'scripts' / 'render_images.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hp = mod._get_render_opts(['--input', 'i', '--output', 'o', '--dataset_path', 'd', '--centroids_path', 'c'])
    assert (hp.input, hp.output, hp.dataset_path, hp.centroids_path, hp.save_depth_npz, hp.resume) == ('i', 'o', 'd', 'c', False, False)
    import pytest
    with pytest.raises(SystemExit):
        mod._get_render_opts(['--input', 'i'])                         # the other three are required, as in the reference
    with pytest.raises(AssertionError):
        mod.main(hp)                                                   # neither --ckpt_path nor --container_path (:137)
    v = Runner.visualize_scalars(torch.linspace(0, 1, 40 * 50).view(40, 50))
    assert v.shape == (40, 50, 3) and v.dtype == np.uint8
    lum = v.astype(np.float64).sum(-1).reshape(-1)
    assert (np.diff(lum) <= 0).all() and lum[0] > 600 and lum[-1] < 10             # monotone ramp, inverted
    assert (v.reshape(-1, 3)[:100] == v[0, 0]).all() and (v.reshape(-1, 3)[-100:] == v[-1, -1]).all()   # clamped outside the quantiles
    hue = mod._hue_wheel(torch.tensor([0.0, 1 / 3, 2 / 3]))
    np.testing.assert_allclose(hue.numpy(), [[255, 0, 0], [0, 255, 0], [0, 0, 255]], atol=1e-3)


def test_filesystem_dataset_refuses_a_cpu_device(tmp_path):
    import pytest
    import torch
    from mega_nerf import _native as N
    from mega_nerf.datasets.filesystem_dataset import FilesystemDataset
    with pytest.raises(N.NativeError, match='no CPU fallback'):
        FilesystemDataset([], 0.1, 1.0, None, True, torch.device('cpu'), [tmp_path / 'c'], 1, 1, 10)
