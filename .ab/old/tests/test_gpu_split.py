"""Opt-in split-precision inference (csrc/mlp_fwd_h2.hip: f16 hi/lo operand halves, three 16-bit MFMA products per layer, fp32
accumulation) against the reference's outputs, against fp64, and against the fp32 kernels it may stand in for."""
import ctypes as C
from argparse import Namespace

import numpy as np
import pytest
import torch

import common
import fp64_ref
from test_gpu_parity import DEV, T, native_models, native_nerf
from test_oracle_golden import load, mlp_variant

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture
def split_precision():
    from mega_nerf import rendering as R
    R.SPLIT_PRECISION = True
    yield
    R.SPLIT_PRECISION = False


def _h2_forward(m, cfg, x, noise=None):
    """x [B, xyz + 3 + 1] (reference input layout) through mnr_mlp_forward_multi_h2, one segment."""
    from mega_nerf import _native as N
    xt = T(x)
    B, ncol = x.shape
    out = torch.empty(B, 4, device=DEV)
    nz = T(noise.reshape(-1)) if noise is not None else None
    io = m.mlp_io(xt, ncol, xt[:, ncol - 4:], ncol, xt[:, ncol - 1:], ncol, 1, B, out, nz)
    desc, packed = m.packed_h2()
    seg = (N.MlpLaunch * 1)()
    seg[0].packed_dev, seg[0].desc, seg[0].io = packed.data_ptr(), C.pointer(desc), C.pointer(io)
    N.check(N.lib().mnr_mlp_forward_multi_h2(seg, 1, None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('name', ['fg', 'bg'])
def test_split_precision_mlp_matches_reference(name):
    """NeRF.forward on the reference's golden batch (200 rows: a ragged last 128-row workgroup), with and without sigma noise:
    the reference's outputs to 1e-4 relative (the north-star tolerance; measured ~1e-6), the fp32 kernel's to 2e-5."""
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    x = g[name + '_x']
    for noise, key in ((None, '_out'), (g[name + '_noise'], '_out_noise')):
        got = _h2_forward(m, cfg, x, noise)
        np.testing.assert_allclose(got, g[name + key], rtol=1e-4, atol=2e-6)
        with torch.no_grad():
            ref32 = m(T(x), sigma_noise=T(noise) if noise is not None else None).cpu().numpy()
        np.testing.assert_allclose(got, ref32, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['fg', 'bg'])
def test_split_precision_is_fp32_class_against_fp64(name):
    """20 000 rows of a sharpened model (peaky sigma, like a trained field) against the fp64 restatement: the split-precision
    kernel's error is of the fp32 kernel's order (<= 4x + 1e-6 of the output scale) -- "fp32-equivalent", measured not assumed."""
    hp, cfg, _ = mlp_variant(name)
    w = common.make_weights(cfg, 100, 4321, sharpen=True)
    m = native_nerf(cfg, w)
    rng = np.random.default_rng(17)
    B = 20000
    dirs = rng.standard_normal((B, 3))
    x = np.concatenate([rng.uniform(-1, 1, (B, cfg.xyz_dim)), dirs / np.linalg.norm(dirs, axis=-1, keepdims=True),
                        rng.integers(0, 100, (B, 1))], 1).astype(f32)
    wt = {k: torch.tensor(v, dtype=torch.float64) for k, v in w.items()}
    with torch.no_grad():
        ref = fp64_ref.nerf_forward64(wt, cfg, torch.tensor(x, dtype=torch.float64)).numpy()
        k32 = m(T(x)).cpu().numpy()
    h2 = _h2_forward(m, cfg, x)
    for c, nm in ((slice(0, 3), 'rgb'), (slice(3, 4), 'sigma')):
        scale = float(np.abs(ref[:, c]).max())
        e32, e16 = float(np.abs(k32[:, c] - ref[:, c]).max()) / scale, float(np.abs(h2[:, c] - ref[:, c]).max()) / scale
        print(name, nm, 'fp32 kernel %.2e  split %.2e' % (e32, e16))
        assert e16 <= 4 * e32 + 1e-6, (nm, e16, e32)


@pytest.mark.parametrize('name', ['render_fgbg_eval', 'render_fgonly_eval', 'render_q13_eval', 'render_default_samples_eval'])
def test_split_precision_render_goldens(name, split_precision):
    """render_rays with every MLP pass on the split-precision kernel against the reference's outputs: same tolerances as the
    fp32 path (1e-4 relative on rgb / depth)."""
    from mega_nerf.rendering import render_rays
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    flags = [bool(v) for v in g['flags']]
    with torch.no_grad():
        res, present = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), Namespace(**vars(hp)),
                                   T(s['sphere_center']) if bg_nerf is not None else None, T(s['sphere_radius']) if bg_nerf is not None else None,
                                   *flags, _randoms=rnd)
    assert present == bool(g['present'])
    for k in sorted(k[4:] for k in g if k.startswith('res_')):
        a, b = res[k].cpu().numpy(), g['res_' + k]
        tol = dict(rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max()))) if 'variance' in k else dict(rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(a, b, err_msg=k, **tol)


def test_split_precision_benchmark_shape_all_rays(split_precision):
    """The benchmark's 1024 rays x (64 + 128) samples on the split-precision kernel: ALL rays within the north-star bound of the
    numpy oracle (same assertion as the fp32 path's test)."""
    from test_gpu_parity_extra import _benchmark_shape_check
    _benchmark_shape_check()


def test_split_precision_benchmark_shape_all_rays_after_training_steps(split_precision):
    """... and on weights 25 training steps old (the state bench.py's `eval_split_precision.rgb_difference_to_f32_kernels` is taken in):
    rays that miss the bound are rays with a fine sample that sits elsewhere than the oracle's, as for the fp32 kernels."""
    from test_gpu_parity_extra import _benchmark_shape_check
    _benchmark_shape_check(train_steps=25, max_offenders=80)
    _benchmark_shape_check(train_steps=30, max_offenders=100, same_batch=True)


@pytest.mark.parametrize('fixture', ['render_container8_eval', 'render_container_2d_eval'])
def test_split_precision_routed_container(split_precision, fixture):
    """A merged container (MegaNeRF router, boundary margin 1.15; 8 cells clustered in 3-D, and 4 cells clustered in 2-D with the background
    routed per sample on its true far-away point) with every cell's rows on the split-precision kernel in ONE launch per pass
    (mnr_mlp_forward_cells_h2): the reference's outputs at the fp32 path's tolerances."""
    from test_gpu_parity_extra import test_new_render_goldens
    from mega_nerf.models.mega_nerf import MegaNeRF
    test_new_render_goldens(fixture)
    hp, nerf, bg_nerf = native_models(fixture)
    assert isinstance(nerf, MegaNeRF)
    from mega_nerf.rendering import render_rays
    g = load(fixture)
    s = common.SCENE
    with torch.no_grad():
        render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']),
                    *[bool(v) for v in g['flags']])
    assert nerf._last_routed_split is True                  # the split-precision launch really served it


@pytest.mark.parametrize('name', ['fg', 'bg'])
def test_split_precision_routed_launch_ragged_cells(name):
    """mnr_mlp_forward_cells_h2 against mnr_mlp_forward_cells (the fp32 routed launch) on hand-made row lists: an empty cell, a
    one-row cell, counts that are not multiples of the 128-row workgroup tile, rows listed by two cells -- same rows, same cell
    weights, outputs within the split kernel's rounding of the fp32 kernel's."""
    from mega_nerf import _native as N
    lib = N.lib()
    hp, cfg, w = mlp_variant(name)
    rng = np.random.default_rng(17)
    B, S = 1536, 8
    n_ray = B // S
    xyz = rng.uniform(-1, 1, (B, cfg.xyz_dim)).astype(f32)
    dirs = rng.standard_normal((n_ray, 3)).astype(f32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    idx = rng.integers(0, 100, n_ray).astype(f32)
    counts = [0, 1, 130, 517, 1536]
    kids = [native_nerf(cfg, {k: (v * f32(1 + 0.03 * i)).astype(f32) for k, v in w.items()}) for i in range(len(counts))]
    lists = torch.zeros(len(counts), B, dtype=torch.int32, device=DEV)
    for i, c in enumerate(counts):
        lists[i, :c] = torch.from_numpy(np.sort(rng.permutation(B)[:c]).astype(np.int32)).to(DEV)
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    t = [T(a) for a in (xyz, dirs, idx)]
    outs = {}
    for split in (False, True):
        sub_out = torch.full((len(counts), B, 4), -7.0, device=DEV)
        rows = []
        for i, child in enumerate(kids):
            _, packed = child.packed_h2() if split else child.packed()
            rows.append([packed.data_ptr(), child.embedding_a.weight.data_ptr(), lists[i].data_ptr(), cnt[i:i + 1].data_ptr(), sub_out[i].data_ptr()])
        cells = torch.tensor(rows, dtype=torch.int64).to(DEV)
        desc, _ = kids[0].packed()
        io = kids[0].mlp_io(t[0], cfg.xyz_dim, t[1], 3, t[2], 1, S, B, sub_out[0], None, None, 0)
        fn = lib.mnr_mlp_forward_cells_h2 if split else lib.mnr_mlp_forward_cells
        N.check(fn(C.byref(desc), cells.data_ptr(), len(counts), C.byref(io), None))
        torch.cuda.synchronize()
        outs[split] = sub_out.cpu().numpy()
    for i, c in enumerate(counts):
        np.testing.assert_allclose(outs[True][i, :c], outs[False][i, :c], rtol=2e-5, atol=2e-6, err_msg='cell %d' % i)
        assert (outs[True][i, c:] == -7.0).all(), 'rows past the count of cell %d were written' % i
