"""Round-2 parity additions ("close the parity holes"): stage kernels against the reference vectors that
no GPU test read before (spherical harmonics of every degree, positional encodings), the remaining MLP variants, the new
render goldens (SH degree 3, 8-cell container, 512-channel container), the benchmark-shaped 1024-ray render against the
numpy oracle, and a per-ray account of WHERE the training render may deviate from the reference."""
import ctypes as C
from argparse import Namespace

import numpy as np
import pytest
import torch

import common
import fp64_ref
from oracle import nerf_oracle as O
from test_oracle_golden import OVERFIT_KEYS, build_case, check_index_agreement, load, mlp_variant, overfit_case, rays_beyond_bound
from test_gpu_parity import DEV, T, close, native_models, native_nerf

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.mark.parametrize('deg', [0, 1, 2, 3, 4])
def test_sh_apply_every_degree(deg):
    """mnr_sh_apply (rendering.py:300-305 / spherical_harmonics.py:55-107) against eval_sh of the reference, deg 0..4."""
    from mega_nerf import _native as N
    g = load('stages')
    coef, dirs, ref = g['sh_in_%d' % deg], g['sh_dirs_%d' % deg], g['sh_out_%d' % deg]
    B, nb = coef.shape[0], (deg + 1) ** 2
    inp = np.concatenate([coef.reshape(B, 3 * nb), np.full((B, 1), 0.25, f32)], 1).astype(f32)      # [coefficients | sigma]
    out = torch.empty(B, 4, device=DEV)
    inp_d, dirs_d = T(inp), T(dirs)                     # (keep the device tensors alive across the asynchronous launch)
    N.check(N.lib().mnr_sh_apply(out.data_ptr(), 4, inp_d.data_ptr(), 3 * nb + 1, dirs_d.data_ptr(), 3, 1, deg, B, None))
    close(out[:, :3], 1.0 / (1.0 + np.exp(-ref.astype(np.float64))), 2e-5, 2e-6)
    close(out[:, 3], np.full(B, 0.25, f32), 0, 0)


def test_embed_matches_reference():
    """mnr_embed (nerf.py:20-25) against Embedding(12) / Embedding(4) of the reference, column order included."""
    from mega_nerf import _native as N
    g = load('stages')
    x = g['emb_x']
    for L, cols, key in ((12, x.shape[1], 'emb_12'), (4, 3, 'emb_4')):
        xin = T(np.ascontiguousarray(x[:, :cols]))
        width = cols * (1 + 2 * L)
        out = torch.empty(x.shape[0], width, device=DEV)
        N.check(N.lib().mnr_embed(out.data_ptr(), width, xin.data_ptr(), cols, cols, L, 1, x.shape[0], None))
        close(out, g[key], 2e-6, 2e-6)


@pytest.mark.parametrize('name', ['relu', 'plain', 'affine'])
def test_remaining_mlp_variants(name):
    """ReLU density activation (--no_shifted_softplus), the plain xyz -> rgb network (no direction, no appearance) and
    --affine_appearance (nerf.py:87-89,156-158), each against the reference's own outputs."""
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    x = T(g[name + '_x'])
    with torch.no_grad():
        close(m(x), g[name + '_out'], 1e-4, 2e-6)
        close(m(x, sigma_noise=T(g[name + '_noise'])), g[name + '_out_noise'], 1e-4, 2e-6)
        close(m(x[:, :cfg.xyz_dim].contiguous(), sigma_only=True), g[name + '_sigma_only'], 1e-4, 2e-6)


@pytest.mark.parametrize('name', ['render_sh3_eval', 'render_container8_eval', 'render_container_w512_eval', 'render_container25_eval',
                                  'render_container_2d_eval'])
def test_new_render_goldens(name):
    from mega_nerf.rendering import render_rays
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    s = common.SCENE
    idx = T(g['idx'].astype(f32))
    flags = [bool(v) for v in g['flags']]
    rnd = {'_want_inds': True}
    with torch.no_grad():
        res, present = render_rays(nerf, bg_nerf, T(g['rays']), idx, Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']), *flags,
                                   _randoms=rnd)
    ref_keys = sorted(k[4:] for k in g if k.startswith('res_'))
    assert sorted(res.keys()) == ref_keys and present == bool(g['present'])
    for k in ref_keys:
        a, b = res[k].cpu().numpy(), g['res_' + k]
        tol = dict(rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max()))) if 'variance' in k else dict(rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(a, b, err_msg=k, **tol)
    for part in ('fg', 'bg'):
        if 'inds_' + part in g and '_inds_' + part in rnd:
            check_index_agreement(name, part, rnd['_inds_' + part].cpu().numpy(), g['inds_' + part])


def test_containers_of_a_pass_share_one_launch():
    """render_container8_eval with the foreground and the background container's routed evaluations in ONE launch per pass
    (mnr_mlp_forward_cells_multi, the default) against one launch per container: the same kernel bodies over the same rows -- every
    output bit-identical; and the merged launch is what the default takes (checked through the C entry point's call count)."""
    from mega_nerf import _native as N
    from mega_nerf import rendering as R
    g = load('render_container8_eval')
    hp, nerf, bg_nerf = native_models('render_container8_eval')
    s = common.SCENE
    idx = T(g['idx'].astype(f32))
    flags = [bool(v) for v in g['flags']]
    calls = {'n': 0}
    real = N.lib().mnr_mlp_forward_cells_multi

    class Counting:
        def __call__(self, *a):
            calls['n'] += 1
            return real(*a)

    def render():
        with torch.no_grad():
            return R.render_rays(nerf, bg_nerf, T(g['rays']), idx, Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']), *flags)[0]

    lib = N.lib()
    try:
        # (the stage-by-stage sequencing: the one-call routed render issues the same merged launch from inside mnr_render_fwd, where
        # a Python-side call counter cannot see it -- tests/test_gpu_step.py compares the two paths bit for bit)
        R.FUSED_RENDER = False
        lib.mnr_mlp_forward_cells_multi = Counting()
        merged = {k: v.cpu().numpy().copy() for k, v in render().items()}
        assert calls['n'] == 2                       # coarse pass + fine pass
        R.MERGE_ROUTED = False
        single = {k: v.cpu().numpy().copy() for k, v in render().items()}
        assert calls['n'] == 2
    finally:
        R.MERGE_ROUTED = True
        R.FUSED_RENDER = True
        lib.mnr_mlp_forward_cells_multi = real
    assert merged.keys() == single.keys()
    for k in merged:
        np.testing.assert_array_equal(merged[k], single[k], err_msg=k)


def test_container_render_edge_batches():
    """Merged containers on the batches the routed launches size worst: no ray of the batch reaches the background (the background
    container's row lists are all empty: every workgroup of its segment exits on the device-side counts), one ray, and an empty batch
    -- against the oracle's routed render (mega_nerf.py:19-61 over rendering.py:33-45)."""
    from mega_nerf.rendering import render_rays
    from test_oracle_golden import build_case
    from oracle import nerf_oracle as O
    name = 'render_container8_eval'
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    hpn = Namespace(**vars(hp))
    s = common.SCENE
    ohp, onerf, obg = build_case(name)
    rays = g['rays'].copy()
    rays[:, 7] = np.minimum(rays[:, 7], 0.3)              # far well inside the ellipsoid for every ray
    idx = g['idx'].astype(f32)
    for sel in (slice(None), slice(0, 1)):
        want, present = O.render_rays(onerf, obg, rays[sel], idx[sel], ohp, s['sphere_center'], s['sphere_radius'], True, False, True)
        assert not present
        with torch.no_grad():
            res, got_present = render_rays(nerf, bg_nerf, T(rays[sel]), T(idx[sel]), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
        assert got_present is False and sorted(res) == sorted(want)
        for k in want:
            np.testing.assert_allclose(res[k].cpu().numpy(), want[k], rtol=2e-4, atol=2e-5, err_msg=k)
        assert float(res['bg_rgb_fine'].abs().max()) == 0.0
    with torch.no_grad():
        res0, p0 = render_rays(nerf, bg_nerf, T(rays[:0]), T(idx[:0]), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    assert p0 is False and res0['rgb_fine'].shape == (0, 3)


def all_ray_violations(res, ores, rnd, dbg, keys, rtol=1e-4, atol=2e-5):
    """Rays of a render that miss ``|a - b| <= atol + rtol |b|`` (the north-star 1e-4 relative bound) in any of ``keys``, with
    what is needed to explain them: the largest relative move of one of the ray's fine samples against the oracle's."""
    n = next(iter(ores.values())).shape[0]
    bad = np.zeros(n, bool)
    worst = {}
    for k in keys:
        a, b = res[k].cpu().numpy().astype(np.float64), ores[k].astype(np.float64)
        excess = (np.abs(a - b) - (atol + rtol * np.abs(b))).reshape(n, -1).max(1)
        bad |= excess > 0
        worst[k] = float((np.abs(a - b) / (atol / rtol + np.abs(b))).max())      # in units of rtol-relative error
    zmove = np.zeros(n)
    zg, zo = rnd['_fine_z_fg'].cpu().numpy(), dbg['fg']['fine_z']
    zmove = (np.abs(zg - zo) / np.maximum(np.abs(zo), 1e-9)).max(1)
    if 'bg' in dbg and 'fine_z' in dbg['bg']:
        ids = np.asarray(dbg['rays_with_bg'])
        zb, zbo = rnd['_fine_z_bg'].cpu().numpy()[:len(ids)], dbg['bg']['fine_z']
        zmove[ids] = np.maximum(zmove[ids], (np.abs(zb - zbo) / np.maximum(np.abs(zbo), 1e-9)).max(1))
    return np.flatnonzero(bad), worst, zmove


def _benchmark_shape_check(train_steps=0, max_offenders=9, same_batch=False):
    """The bench.py shape -- 1024 rays x (64 + 128) samples, fg + bg, eval flags -- against the numpy oracle on the same
    rays / weights: every workgroup, compaction and tile boundary of the stage kernels at the size that is benchmarked.
    ALL 1024 rays must meet the north-star tolerance (1e-4 relative on rgb / depth) in every output.  A fine-sample index
    may differ from the oracle's where a u value sits within GEMM rounding (~1e-6) of a cdf entry -- the last u = 1.0 against
    cdf[-1] = 1 -+ ulp does so on ~20 % of the rays -- but _sample_cdf is continuous across an entry (rendering.py:524-535:
    t -> 1 in bin k meets t -> 0 in bin k + 1), so a moved index does not move the sample.  The one genuine discontinuity is
    a run of cdf entries that are EQUAL in fp32 (zero-probability bins: searchsorted(right=True) jumps across the whole run):
    a ray may miss the bound only if one of its fine samples really sits elsewhere (relative z move > 1e-5), and there may be
    at most 9 such rays."""
    from mega_nerf import ray_utils
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    from mega_nerf.rendering import render_rays
    s = common.SCENE
    hp = O.make_hparams(coarse_samples=64, fine_samples=128)
    A = s['appearance_count']
    fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
    fw, bw = common.make_weights(fcfg, A, 1000), common.make_weights(bcfg, A, 1500)

    def native(cfg, w):
        m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim, False, A, 3,
                 cfg.xyz_dim, ShiftedSoftplus())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        return m.to(DEV).eval()

    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, torch.device(DEV))
    rays_all = ray_utils.get_rays(d, T(s['c2w']), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8).cpu().numpy()
    rays, idx = common.pick_rays(rays_all, 1024, 7)
    rnd = {'_want_inds': True}
    nf, nb = native(fcfg, fw), native(bcfg, bw)
    if train_steps:
        # the weights bench.py evaluates with: a few fused Adam steps (random targets, training-mode randomness) away from the seeded
        # initialisation -- density concentrates, runs of zero-probability coarse bins get longer
        from mega_nerf.training import FusedTrainStep
        nf.train(), nb.train()
        step = FusedTrainStep([(nf, nb)], Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']), 1024, seed=11)
        gen = torch.Generator(device='cpu').manual_seed(3)
        fixed = (T(rays), T(idx.astype(np.int32)), torch.rand(1024, 3, generator=gen).to(DEV))
        for it in range(train_steps):
            if same_batch:                       # bench.py's protocol: every step on the batch that is rendered afterwards
                step([fixed])
                continue
            r_, i_ = common.pick_rays(rays_all, 1024, 100 + it)
            step([(T(r_), T(i_.astype(np.int32)), torch.rand(1024, 3, generator=gen).to(DEV))])
        torch.cuda.synchronize()
        del step
        nf.eval(), nb.eval()
        fw = {k: v.detach().cpu().numpy().copy() for k, v in nf.state_dict().items()}
        bw = {k: v.detach().cpu().numpy().copy() for k, v in nb.state_dict().items()}
    with torch.no_grad():
        res, present = render_rays(nf, nb, T(rays), T(idx.astype(f32)), Namespace(**vars(hp)),
                                   T(s['sphere_center']), T(s['sphere_radius']), True, False, True, _randoms=rnd)
    dbg = {}
    ores, opresent = O.render_rays(O.Model(fcfg, fw), O.Model(bcfg, bw), rays, idx.astype(f32), hp, s['sphere_center'],
                                   s['sphere_radius'], True, False, True, debug=dbg)
    assert present == opresent and sorted(res.keys()) == sorted(ores.keys())
    assert np.isfinite(res['rgb_fine'].cpu().numpy()).all()
    moved = rnd['_inds_fg'].cpu().numpy() != dbg['fg']['inds']
    keys = ('rgb_fine', 'fg_rgb_fine', 'bg_rgb_fine', 'depth_fine', 'fg_depth_fine', 'bg_depth_fine', 'bg_lambda_fine')
    offenders, worst, zmove = all_ray_violations(res, ores, rnd, dbg, keys)
    per_output = {k: int(rays_beyond_bound(res[k].cpu().numpy(), ores[k], 1024).sum()) for k in keys}
    print('rays beyond the bound per output:', per_output)
    print('moved fine indices: %d of %d (%d at the last u), rays with a moved index: %d; worst error per output in units of the '
          'bound: %s; rays missing the bound: %s' % (moved.sum(), moved.size, moved[:, -1].sum(), moved.any(1).sum(),
                                                     {k: '%.3f' % v for k, v in worst.items()}, offenders.tolist()))
    if train_steps == 0:
        # "bit-exact sample indices", stated as the mechanism: an index may differ from the oracle's only (i) at the last u = 1.0, which
        # sits on cdf[-1] = 1 -+ ulp, or (ii) where u is within GEMM rounding of a cdf entry, in which case the SAMPLE does not move
        # (_sample_cdf is continuous across an entry), or (iii) across a run of zero-probability bins (equal cdf entries)
        assert moved[:, :-1].sum() <= 8 and moved[:, -1].sum() <= 1024          # measured: 4 / 221; the last u: <= one per ray
        zg_, zo_ = rnd['_fine_z_fg'].cpu().numpy(), dbg['fg']['fine_z']
        inner = np.argwhere(moved[:, :-1])
        jumps = 0
        for r_, j_ in inner:
            if abs(zg_[r_, j_] - zo_[r_, j_]) <= 1e-5 * max(abs(zo_[r_, j_]), 1e-9):
                continue                                                    # (ii): same sample to 1e-5 relative
            a_, b_ = sorted((int(rnd['_inds_fg'][r_, j_]), int(dbg['fg']['inds'][r_, j_])))
            w_ = dbg['fg']['weights_coarse'][r_, 1:-1]
            pdf = (w_ + 1e-8) / (w_ + 1e-8).sum()
            assert pdf[max(a_ - 1, 0):b_].sum() <= 1e-6, ('a moved index that is not explained', r_, j_, a_, b_)      # (iii)
            jumps += 1
        print('moved indices away from the last u: %d, of which across zero-probability runs: %d' % (len(inner), jumps))
        assert jumps <= 9
    elif moved.mean() >= 5e-3:
        r = int(np.argmax(moved.sum(1)))
        wc = dbg['fg']['weights_coarse'][r]
        cols = np.flatnonzero(moved[r])
        print('ray %d: %d moved; oracle weights_coarse nonzero at %s (values %s); moved u slots %s; got %s; oracle %s; |dz| max of the ray %.3e'
              % (r, len(cols), np.flatnonzero(wc > 1e-6).tolist(), wc[wc > 1e-6][:6].tolist(), cols[:12].tolist(),
                 rnd['_inds_fg'].cpu().numpy()[r, cols[:12]].tolist(), dbg['fg']['inds'][r, cols[:12]].tolist(), zmove[r]))
    unexplained = [int(r) for r in offenders if not zmove[r] > 1e-5]
    assert not unexplained, ('rays miss 1e-4 without a moved sample', unexplained, worst)
    assert len(offenders) <= max_offenders, (offenders.tolist(), zmove[offenders].tolist())


def test_benchmark_shape_render_against_oracle():
    _benchmark_shape_check()


def test_benchmark_shape_render_against_oracle_after_training_steps():
    """The same all-ray assertion on lightly trained weights -- 25 fused training steps (a fresh batch each) away from the
    initialisation: a ray may still miss the bound only where one of its fine samples sits elsewhere than the oracle's (measured: none)."""
    _benchmark_shape_check(train_steps=25, max_offenders=9)


@pytest.mark.parametrize('fixture', ['render_overfit_eval', 'render_overfit_hip_eval'])
@pytest.mark.parametrize('path', ['stages', 'fused', 'split'])
def test_overfit_regime_against_the_references_own_fp32_and_fp64_runs(path, fixture, monkeypatch):
    """Where the reference's OWN importance sampling is decided by rounding, pinned by the reference itself
    (tests/golden/make_golden.py::run_overfit): weights ~30 Adam steps into overfitting one 1024-ray batch -- trained by the reference
    (render_overfit_eval) or by this implementation's one-call step with bench.py's protocol (render_overfit_hip_eval: the weights the
    round-3 tests allowed 100 offenders on) -- rendered BY THE REFERENCE in fp32 and in fp64.  Its two runs draw different fine
    samples on practically every ray (14 % / 45 % of the indices) and differ beyond the north-star bound on 131 rays in depth_fine /
    bg_depth_fine (first fixture) resp. on 28 rays in rgb_fine / bg_rgb_fine / depth_fine / bg_depth_fine (second); foreground outputs
    and bg_lambda agree on every ray.  This implementation (stage-by-stage launches, the one-call render, the split-precision kernels)
    is held to exactly that: an output the reference pins with both of its runs must be met on EVERY ray against both; in an output
    where the reference's fp32 run misses its fp64 run on `own` rays, at most 1.5 own + 4 rays may miss the fp64 run and at most
    2 own + 4 the fp32 run (two rounding-noise parties)."""
    from mega_nerf import rendering
    from mega_nerf.rendering import render_rays
    g, hp, fcfg, bcfg, fw, bw = overfit_case(fixture)
    s = common.SCENE
    nerf, bg_nerf = native_nerf(fcfg, fw).to(DEV).eval(), native_nerf(bcfg, bw).to(DEV).eval()
    monkeypatch.setattr(rendering, 'FUSED_RENDER', path != 'stages')
    monkeypatch.setattr(rendering, 'SPLIT_PRECISION', path == 'split')
    rnd = {'_want_inds': True}
    with torch.no_grad():
        res, present = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), Namespace(**vars(hp)), T(s['sphere_center']),
                                   T(s['sphere_radius']), True, False, True, _randoms=rnd if path == 'stages' else None)
    n = g['rays'].shape[0]
    assert present == bool(g['present'])
    report = {}
    for k in OVERFIT_KEYS:
        self_bad = rays_beyond_bound(g['res_f32_' + k], g['res_f64_' + k], n)
        bad64 = rays_beyond_bound(res[k].cpu().numpy(), g['res_f64_' + k], n)
        bad32 = rays_beyond_bound(res[k].cpu().numpy(), g['res_f32_' + k], n)
        report[k] = (int(bad64.sum()), int(bad32.sum()), int(self_bad.sum()), int((bad64 & ~self_bad).sum()))
    print(path, '(vs fp64, vs fp32, reference fp32 vs fp64, vs fp64 outside the reference\'s own rays):', report)
    if path == 'stages':
        mine = rnd['_inds_fg'].cpu().numpy()
        print('fg fine indices differing: mine vs ref fp32 %d, mine vs ref fp64 %d, ref fp32 vs ref fp64 %d of %d' % (
            (mine != g['inds_f32_fg']).sum(), (mine != g['inds_f64_fg']).sum(), (g['inds_f32_fg'] != g['inds_f64_fg']).sum(), mine.size))
        # the index disagreement is the regime's, not this implementation's: no larger against either reference run than theirs with each other
        own = int((g['inds_f32_fg'] != g['inds_f64_fg']).sum())
        assert (mine != g['inds_f64_fg']).sum() <= 1.25 * own and (mine != g['inds_f32_fg']).sum() <= 1.6 * own
    for k, (b64, b32, own, outside) in report.items():
        if own == 0:        # the reference agrees with itself on every ray: so must this implementation, with both of its runs
            assert b64 == 0 and b32 == 0, (k, report)
        else:
            assert b64 <= 1.5 * own + 4 and b32 <= 2 * own + 4, (k, report)


def test_training_render_deviates_only_where_sample_indices_moved():
    """render_fgbg_train (reference outputs with captured randoms): per ray, rgb_fine agrees with the reference to 1e-4
    wherever this implementation drew the same fine-sample indices as the reference did; rays with a moved index are
    counted and bounded, so the loose end-to-end gradient tolerance of test_gpu_parity is attributable to them."""
    from mega_nerf.rendering import render_rays_async
    g = load('render_fgbg_train')
    hp, nerf, bg_nerf = native_models('render_fgbg_train')
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    rnd['_want_inds'] = True
    with torch.no_grad():        # same kernels as the differentiable path's forward (training-mode randomness comes from rnd)
        res = render_rays_async(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(np.int32)), Namespace(**vars(hp)), T(s['sphere_center']),
                                T(s['sphere_radius']), False, True, False, _randoms=rnd)[0]
    same = (rnd['_inds_fg'].cpu().numpy()[:g['inds_fg'].shape[0]] == g['inds_fg'].astype(np.int64)).all(axis=1)
    a, b = res['rgb_fine'].cpu().numpy(), g['res_rgb_fine']
    # a ray without a background segment depends on the fg indices only
    no_bg = res['bg_lambda_fine'].cpu().numpy() < 1e-6
    tight = same & no_bg
    assert tight.sum() >= 0.5 * len(a)
    np.testing.assert_allclose(a[tight], b[tight], rtol=1e-4, atol=2e-5)
    assert (~same).mean() < 0.25


@pytest.mark.parametrize('mode', ['slabs', 'atomic_fallback'])
def test_batched_weight_gradients(mode, monkeypatch):
    """mnr_mlp_backward_weights_multi (csrc/wgrad.hip): foreground region with one dense range + background region with two
    device-counted ranges in ONE launch, against torch fp64 autograd of the same rows; also with the slab slots switched off,
    so that every flush takes the atomic fallback."""
    from mega_nerf import _native as N
    from test_gpu_parity import _torch_nerf_forward
    if mode == 'atomic_fallback':
        monkeypatch.setenv('MNR_WGRAD_MAX_EPISODES', '0')
    lib = N.lib()
    rng = np.random.default_rng(31)
    regions, keep, refs = [], [], []
    for name, S, n_ray, counted in (('fg', 32, 40, False), ('bg', 32, 24, True)):
        hp, cfg, w = mlp_variant(name)
        m = native_nerf(cfg, w)
        B = S * n_ray
        n_used = n_ray - 5 if counted else n_ray                      # device-side count below the host bound
        xyz = rng.uniform(-1, 1, (2 * B, cfg.xyz_dim)).astype(f32)     # two passes ("coarse", "fine") of B rows each
        dirs = rng.standard_normal((n_ray, 3)).astype(f32)
        dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
        idx = rng.integers(0, 100, n_ray).astype(f32)
        d_out = rng.standard_normal((2 * B, 4)).astype(f32)
        cap = 2 * B
        fpr = m.tape_floats_per_row()
        tape, gtape = torch.zeros(cap * fpr, device=DEV), torch.zeros(cap * fpr, device=DEV)
        dheads, out = torch.zeros(cap, 4, device=DEV), torch.empty(cap, 4, device=DEV)
        xyz_t, dirs_t, idx_t, dout_t = T(xyz), T(dirs), T(idx), T(d_out)
        nun = torch.tensor([n_used], device=DEV, dtype=torch.int32) if counted else None
        grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
        desc, packed = m.packed()
        pb = m.packed_bwd()
        counter = torch.zeros(1, device=DEV, dtype=torch.int32)
        gios = []
        for p in range(2):
            io = m.mlp_io(xyz_t[p * B:], cfg.xyz_dim, dirs_t, 3, idx_t, 1, S, B, out[p * B:], None, nun, S)
            m.evaluate_train(io, tape, cap, p * B)
            g = N.MlpGradIO()
            g.tape, g.gtape, g.tape_rows, g.tape_row0 = tape.data_ptr(), gtape.data_ptr(), cap, p * B
            g.d_out, g.d_out_stride, g.out, g.out_stride = dout_t[p * B:].data_ptr(), 4, out[p * B:].data_ptr(), 4
            g.dheads, g.idx, g.idx_stride, g.idx_is_float, g.rows_per_ray = dheads.data_ptr(), idx_t.data_ptr(), 1, 1, S
            g.n_rows, g.work_counter, g.grad = B, counter.data_ptr(), m.grad_struct(grads)
            if counted:
                g.n_units_dev, g.rows_per_unit = nun.data_ptr(), S
            N.check(lib.mnr_mlp_backward_data(packed.data_ptr(), pb.data_ptr(), C.byref(desc), C.byref(g), None))
            gios.append(g)
        rg = N.WgradRegion()
        rg.desc, rg.tape, rg.gtape, rg.tape_rows, rg.grad = C.pointer(desc), tape.data_ptr(), gtape.data_ptr(), cap, gios[0].grad
        if counted:
            rg.n_ranges = 2
            for p in range(2):
                rg.row0[p], rg.n_rows[p], rg.n_units_dev[p], rg.rows_per_unit[p] = p * B, B, nun.data_ptr(), S
        else:
            rg.n_ranges, rg.row0[0], rg.n_rows[0] = 1, 0, cap
        regions.append(rg)
        keep.append((m, desc, packed, pb, tape, gtape, dheads, out, xyz_t, dirs_t, idx_t, dout_t, nun, counter, gios, grads))
        # reference gradients: fp64 autograd over the rows that count, with the ReLU masks the kernels actually used (read back
        # from the tape: tests/fp64_ref.py explains why)
        torch.cuda.synchronize()
        rows = np.concatenate([np.arange(p * B, p * B + n_used * S) for p in range(2)])
        mk = fp64_ref.tape_masks(lib, m, desc, tape, cap, 0, cap)
        mk = dict(act=[a[rows] for a in mk['act']], dact=mk['dact'][rows])
        ray = (rows % B) // S
        x_full = np.concatenate([xyz[rows], dirs[ray], idx[ray][:, None]], 1)
        refs.append(fp64_ref.autograd_grads64(w, cfg, x_full, None, d_out[rows], mk))
    ws = torch.empty(lib.mnr_wgrad_workspace_bytes(), dtype=torch.uint8, device=DEV)
    arr = (N.WgradRegion * 2)(*regions)
    N.check(lib.mnr_mlp_backward_weights_multi(arr, 2, ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    bad = {}
    for name, k_, ref in zip(('fg', 'bg'), keep, refs):
        for k, got in k_[-1].items():
            if k.split('.')[0] in ('embedding_a', 'sigma', 'rgb'):
                continue                                              # head / embedding gradients come from backward_data
            e = fp64_ref.rel_to_scale(got.cpu().numpy(), ref[k])
            if not (e < 2e-4 and np.abs(ref[k]).max() > 0):
                bad[name + '.' + k] = e
    assert not bad, bad


def _flat_backward(name, S, n_ray, row0, pad, seed, d_scale=1.0):
    """One training-mode launch of model ``name`` over n_ray x S rows written at tape row ``row0`` + its data-gradient chain /
    head gradients; returns everything the weight-gradient launch and the fp64 checker need."""
    from mega_nerf import _native as N
    lib = N.lib()
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    rng = np.random.default_rng(seed)
    B = S * n_ray
    xyz = rng.uniform(-1, 1, (B, cfg.xyz_dim)).astype(f32)
    dirs = rng.standard_normal((n_ray, 3)).astype(f32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    idx = rng.integers(0, 100, n_ray).astype(f32)
    noise = rng.uniform(0, 1, B).astype(f32)
    d_out = (rng.standard_normal((B, 4)) * d_scale).astype(f32)
    cap = row0 + B + pad
    fpr = m.tape_floats_per_row()
    tape, gtape = torch.zeros(cap * fpr, device=DEV), torch.zeros(cap * fpr, device=DEV)
    dheads, out = torch.zeros(cap, 4, device=DEV), torch.empty(B, 4, device=DEV)
    t = [T(a) for a in (xyz, dirs, idx, noise, d_out)]
    io = m.mlp_io(t[0], cfg.xyz_dim, t[1], 3, t[2], 1, S, B, out, t[3])
    m.evaluate_train(io, tape, cap, row0)
    grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    desc, packed = m.packed()
    pb = m.packed_bwd()
    counter = torch.zeros(1, device=DEV, dtype=torch.int32)
    g = N.MlpGradIO()
    g.tape, g.gtape, g.tape_rows, g.tape_row0 = tape.data_ptr(), gtape.data_ptr(), cap, row0
    g.d_out, g.d_out_stride, g.out, g.out_stride = t[4].data_ptr(), 4, out.data_ptr(), 4
    g.dheads, g.idx, g.idx_stride, g.idx_is_float, g.rows_per_ray = dheads.data_ptr(), t[2].data_ptr(), 1, 1, S
    g.n_rows, g.work_counter, g.grad = B, counter.data_ptr(), m.grad_struct(grads)
    N.check(lib.mnr_mlp_backward_data(packed.data_ptr(), pb.data_ptr(), C.byref(desc), C.byref(g), None))
    rg = N.WgradRegion()
    rg.desc, rg.tape, rg.gtape, rg.tape_rows, rg.grad = C.pointer(desc), tape.data_ptr(), gtape.data_ptr(), cap, g.grad
    rg.n_ranges, rg.row0[0], rg.n_rows[0] = 1, row0, B
    x_full = np.concatenate([xyz, np.repeat(dirs, S, 0), np.repeat(idx, S)[:, None]], 1)
    return dict(m=m, cfg=cfg, w=w, desc=desc, tape=tape, cap=cap, row0=row0, B=B, region=rg, grads=grads, x=x_full, noise=noise,
                d_out=d_out, out=out, keep=(packed, pb, gtape, dheads, t, counter, g))


@pytest.mark.parametrize('h2', [False, True], ids=['f32', 'split'])
@pytest.mark.parametrize('size', ['r592', 'benchmark'])
def test_wgrad2_against_fp64_autograd(size, h2):
    """k_wgrad2 (mnr_mlp_backward_weights_multi: the weight-gradient launch of a training step) against torch fp64 autograd of
    the reference's NeRF.forward (nerf.py:115-160): EVERY parameter gradient within 2e-4 of its tensor's scale -- at the 592
    ragged rows of test_mlp_backward_against_fp64_autograd and at the benchmark's row counts (fg 1024 x 192 = 196 608 rows, bg
    138 x 96 = 13 248), fg + bg regions in ONE launch.  The fp64 side uses the ReLU masks found on the kernel's own tape, so
    the comparison measures the kernels, not which way a pre-activation within an ulp of zero was rounded.
    ``split``: the opt-in split-precision form of the same launch (mnr_mlp_backward_weights_multi_h2: f16 hi/lo operands, plane-wise
    power-of-two scaling of dZ) at the same tolerance, with output gradients of realistic size (1e-7: far below the f16 range)."""
    from mega_nerf import _native as N
    lib = N.lib()
    d_scale = 1e-7 if h2 else 1.0
    shapes = dict(r592=(('fg', 16, 37, 24, 40), ('bg', 16, 37, 24, 40)),
                  benchmark=(('fg', 192, 1024, 0, 0), ('bg', 96, 138, 0, 0)))[size]
    runs = [_flat_backward(name, S, n_ray, row0, pad, 40 + i, d_scale) for i, (name, S, n_ray, row0, pad) in enumerate(shapes)]
    ws = torch.empty(lib.mnr_wgrad_workspace_bytes(), dtype=torch.uint8, device=DEV)
    arr = (N.WgradRegion * len(runs))(*[r['region'] for r in runs])
    fn = lib.mnr_mlp_backward_weights_multi_h2 if h2 else lib.mnr_mlp_backward_weights_multi
    N.check(fn(arr, len(runs), ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    worst, flips = {}, {}
    for (name, *_), r in zip(shapes, runs):
        mk = fp64_ref.tape_masks(lib, r['m'], r['desc'], r['tape'], r['cap'], r['row0'], r['B'])
        ref = fp64_ref.autograd_grads64(r['w'], r['cfg'], r['x'], r['noise'], r['d_out'], mk)
        for k, got in r['grads'].items():
            worst[name + '.' + k] = fp64_ref.rel_to_scale(got.cpu().numpy(), ref[k])
    print(size, {k: '%.1e' % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    assert not bad, bad


def test_default_width_model_built_under_inference_mode():
    """Reference-style callers (render_images.py, create_octree.py, the merge / convert scripts) build or load the model inside
    ``torch.inference_mode()``: its parameters then carry no version counter.  The 256-wide model goes through the packed-weight
    cache of the fused kernel (the 32-wide merge-script golden does not), which must neither raise nor go stale."""
    from test_gpu_parity import mlp_variant
    hp, cfg, w = mlp_variant('fg')
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1, 1, (300, 3)), rng.standard_normal((300, 3)), rng.integers(0, 100, (300, 1))], 1).astype(f32)
    with torch.inference_mode():
        m = native_nerf(cfg, w)
        assert all(p.is_inference() for p in m.parameters())
        a = m(T(x)).cpu().numpy()
        b = m(T(x)).cpu().numpy()                   # second call: cache hit
    with torch.no_grad():
        c = native_nerf(cfg, w)(T(x)).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize('xyz_dim', [3, 4])
def test_pair_kernel_equals_the_one_wavefront_kernel_and_fp64(xyz_dim, monkeypatch):
    """k_mlp_fwd_pair (csrc/mlp_fwd_pair.hip: the 512-wide default architectures with two wavefronts per SIMD, a wavefront pair splitting
    every layer's output features) against (i) k_mlp_fwd<MlpCfg<.., 512, ..>> on the same packed image -- a feature's K loop is the same
    fmaf chain in both, so the densities must be BIT-identical (colours: to the order of the rgb head's last addition) -- for plain launches (ragged row count, sigma noise, sigma_only) and (ii)
    an fp64 torch evaluation of the same weights (1e-5 of the output scale)."""
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, layer_dim=512, bg_layer_dim=512)
    cfg = common.model_cfg(hp, xyz_dim, 512)
    A = common.SCENE['appearance_count']
    w = common.make_weights(cfg, A, 4400 + xyz_dim)
    m = native_nerf(cfg, w).to(DEV).eval()
    assert m.is_wide_default_arch()
    rng = np.random.default_rng(17)
    B = 1000 + 37                                           # not a multiple of the 64 rows of a workgroup
    x = np.concatenate([rng.uniform(-.8, .8, (B, xyz_dim)), rng.standard_normal((B, 3)), rng.integers(0, A, (B, 1))], 1).astype(f32)
    noise = rng.uniform(0, 1, (B, 1)).astype(f32)
    outs = {}
    for mode in ('pair', 'one'):
        if mode == 'one':
            monkeypatch.setenv('MNR_NO_PAIR_KERNEL', '1')
        else:
            monkeypatch.delenv('MNR_NO_PAIR_KERNEL', raising=False)
        with torch.no_grad():
            outs[mode] = (m(T(x)).cpu().numpy(), m(T(x), sigma_noise=T(noise)).cpu().numpy(),
                          m(T(x[:, :xyz_dim].copy()), sigma_only=True).cpu().numpy())
    for a, b in zip(outs['pair'], outs['one']):
        # the trunk (hence sigma) is bit-identical; the rgb head sums its 256 inputs as two halves of 128 in the pair kernel
        np.testing.assert_array_equal(a[:, -1], b[:, -1])
        np.testing.assert_allclose(a, b, rtol=0, atol=3e-7)
    # fp64 evaluation of nerf.py:115-160 with the same weights
    W64 = {k: torch.from_numpy(v).double() for k, v in w.items()}
    xt = torch.from_numpy(x).double()

    def emb(v, L):
        return torch.cat([v] + [f(v * 2.0 ** k) for k in range(L) for f in (torch.sin, torch.cos)], -1)
    e = emb(xt[:, :xyz_dim], 12)
    h = e
    for i in range(8):
        inp = torch.cat([e, h], -1) if i == 4 else h
        h = torch.relu(inp @ W64['xyz_encodings.%d.0.weight' % i].T + W64['xyz_encodings.%d.0.bias' % i])
    sigma = torch.nn.functional.softplus(h @ W64['sigma.weight'].T + W64['sigma.bias'] - 1)
    f = h @ W64['xyz_encoding_final.weight'].T + W64['xyz_encoding_final.bias']
    app = W64['embedding_a.weight'][xt[:, -1].long()]
    d = torch.relu(torch.cat([f, emb(xt[:, xyz_dim:xyz_dim + 3], 4), app], -1) @ W64['dir_a_encoding.0.weight'].T + W64['dir_a_encoding.0.bias'])
    rgb = torch.sigmoid(d @ W64['rgb.weight'].T + W64['rgb.bias'])
    ref = torch.cat([rgb, sigma], -1).numpy()
    err = np.abs(outs['pair'][0] - ref).max(0) / np.maximum(np.abs(ref).max(0), 1e-12)
    assert err.max() < 1e-5, err


@pytest.mark.parametrize('name', ['render_container8_eval', 'render_container_sh2_eval', 'render_container_w512_eval'])
def test_one_call_routed_render_equals_the_stage_by_stage_render_at_ragged_sizes(name):
    """mnr_render_fwd with merged containers (route -> all cells in one launch -> blend inside the call) against the stage-by-stage
    sequencing of the same kernels at ray counts that leave ragged last workgroups / row blocks everywhere (1, 7, 33, 100, 257 rays of
    the benchmark camera, some with a background segment): every output bit-identical."""
    from mega_nerf import rendering as R
    from oracle import nerf_oracle as O
    hp, nerf, bg_nerf = native_models(name)
    hpn = Namespace(**vars(hp))
    s = common.SCENE
    d = O.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True)
    rays_all = O.get_rays(d, s['c2w'], s['near'], s['far'], s['ray_altitude_range']).reshape(-1, 8)
    for n in (1, 7, 33, 100, 257):
        rays, idx = common.pick_rays(rays_all, n, 1000 + n)
        args = (nerf, bg_nerf, T(rays), T(idx.astype(f32)), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
        try:
            with torch.no_grad():
                assert R._fused_render_ok(nerf, bg_nerf, hpn, args[3], args[6], False, {})
                fused, p1 = R.render_rays(*args)
                R.FUSED_RENDER = False
                stage, p2 = R.render_rays(*args)
        finally:
            R.FUSED_RENDER = True
        assert p1 == p2 and sorted(fused) == sorted(stage)
        for k in fused:
            np.testing.assert_array_equal(fused[k].cpu().numpy(), stage[k].cpu().numpy(), err_msg='%s n=%d %s' % (name, n, k))


@pytest.mark.gpu
@pytest.mark.parametrize('width', [256, 512])
def test_routed_launch_workgroup_order_does_not_change_a_bit(width, monkeypatch):
    """Gather-mode launches of a merged container hand every XCD a contiguous run of the cells' workgroups (csrc/mlp_fwd_kernels.h
    ``xcd_contiguous``; ``MNR_NO_XCD_ORDER=1``: the hardware's round-robin).  Same tiles on other CUs: MegaNeRF.forward
    (mega_nerf.py:19-61) must return the same bits either way -- 3 / 25 / 64 cells (64 = the router's maximum), row counts from one row to
    several rounds of workgroups, most cells empty at the small ones, hard and blended routing."""
    from mega_nerf.models.mega_nerf import MegaNeRF
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, layer_dim=width, bg_layer_dim=width)
    cfg = common.model_cfg(hp, 3, width)
    A = common.SCENE['appearance_count']
    rng = np.random.default_rng(91)
    base = [common.make_weights(cfg, A, 9100 + i, sharpen=False) for i in range(3)]
    for n_cells, rows in ((3, (1, 63, 65, 4097)), (25, (7, 640, 20011)), (64, (129, 9000))):
        g = int(np.ceil(np.sqrt(n_cells)))
        cent = np.array([[0.0, -.7 + 1.4 * (i // g) / max(1, g - 1), -.7 + 1.4 * (i % g) / max(1, g - 1)] for i in range(n_cells)], f32)
        subs = [native_nerf(cfg, base[i % 3]) for i in range(n_cells)]
        for margin in (1.0, 1.15):
            m = MegaNeRF(subs, torch.from_numpy(cent), margin, False, False).to(DEV).eval()
            for B in rows:
                x = np.concatenate([rng.uniform(-.8, .8, (B, 3)), rng.standard_normal((B, 3)), rng.integers(0, A, (B, 1))], 1).astype(f32)
                out = {}
                for order in ('xcd', 'round_robin'):
                    if order == 'xcd':
                        monkeypatch.delenv('MNR_NO_XCD_ORDER', raising=False)
                    else:
                        monkeypatch.setenv('MNR_NO_XCD_ORDER', '1')
                    with torch.no_grad():
                        out[order] = m(T(x)).cpu().numpy()
                assert np.isfinite(out['xcd']).all()
                np.testing.assert_array_equal(out['xcd'], out['round_robin'], err_msg='%d cells, margin %.2f, %d rows' % (n_cells, margin, B))
