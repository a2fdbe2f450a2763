"""GPU parity tests: the HIP path (through the C ABI, via the mega_nerf host package) against
 (a) golden vectors recorded from the real reference (tests/golden/*.npz) and
 (b) the numpy oracle on the same seeded inputs.

Tolerances (north star): rgb/depth within 1e-4 relative; sample indices bit-exact for identical
(bins, weights, u); pure elementwise fp32 chains (z values, sample positions) bit-exact.
"""
import ctypes as C
import os
from argparse import Namespace
from pathlib import Path

import numpy as np
import pytest
import torch

import common
from oracle import nerf_oracle as O
from test_oracle_golden import MLP_VARIANTS, RENDER_CASES, build_case, check_index_agreement, load, mlp_variant  # noqa: F401

pytestmark = pytest.mark.gpu
f32 = np.float32
DEV = 'cuda'


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def native_nerf(cfg, weights, appearance_count=100):
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    m = NeRF(cfg.pos_xyz_dim, cfg.pos_dir_dim, cfg.layers, cfg.skip_layers, cfg.layer_dim, cfg.appearance_dim,
             cfg.affine_appearance, appearance_count, cfg.rgb_dim, cfg.xyz_dim,
             ShiftedSoftplus() if cfg.shifted_softplus else torch.nn.ReLU())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    return m.to(DEV).eval()


# ---- ray generation ------------------------------------------------------------------------------
def test_ray_generation():
    from mega_nerf import ray_utils as RU
    g = load('rays')
    W, H = int(g['W']), int(g['H'])
    fx, fy, cx, cy = [float(v) for v in g['intr']]
    for cp in (1, 0):
        d = RU.get_ray_directions(W, H, fx, fy, cx, cy, bool(cp), torch.device(DEV))
        close(d, g['dirs_c%d' % cp], 2e-6, 2e-7)
    d = T(g['dirs_c1'])
    c2w = T(g['c2w'])
    close(RU.get_rays(d, c2w, 0.01, 1e5, [-0.5, 0.2]), g['rays_alt'], 5e-6, 2e-7)
    close(RU.get_rays(d, c2w, 0.05, 2.0, None), g['rays_noalt'], 5e-6, 2e-7)
    close(RU.get_rays(d, c2w, 0.3, 0.9, [-0.35, -0.1]), g['rays_alt2'], 5e-6, 2e-7)
    close(RU.get_rays_batch(T(g['batch_dirs']), T(g['batch_c2w']), 0.01, 1e5, [-0.5, 0.2]), g['rays_batch'], 5e-6, 2e-7)
    # non-contiguous view input, empty input
    full = RU.get_rays(d, c2w, 0.01, 1e5, [-0.5, 0.2])
    part = RU.get_rays(d[:, 3:17], c2w, 0.01, 1e5, [-0.5, 0.2])
    assert torch.equal(part, full[:, 3:17])
    assert RU.get_rays(d[:0], c2w, 0.01, 1e5, None).shape == (0, W, 8)


# ---- stage kernels -------------------------------------------------------------------------------
def test_ray_setup_and_bg_points():
    from mega_nerf import _native as N
    g = load('stages')
    s = common.SCENE
    rays = T(g['rays'])
    n = rays.shape[0]
    far, ld = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    lst = torch.zeros(n, device=DEV, dtype=torch.int32)
    slot = torch.empty(n, device=DEV, dtype=torch.int32)
    sc = torch.zeros(2, device=DEV, dtype=torch.int32)
    lib = N.lib()
    N.check(lib.mnr_ray_setup(rays.data_ptr(), n, N.host3(s['sphere_center']), N.host3(s['sphere_radius']),
                              far.data_ptr(), ld.data_ptr(), lst.data_ptr(), slot.data_ptr(), sc[0:1].data_ptr(),
                              sc[1:2].data_ptr(), None))
    fg_far = np.maximum(g['fg_far'], g['rays'][:, 6])
    has_bg = g['rays'][:, 7] > fg_far
    assert int(sc[1]) == 0
    assert int(sc[0]) == int(has_bg.sum())
    exp_list = np.nonzero(has_bg)[0]
    assert np.array_equal(lst.cpu().numpy()[:len(exp_list)], exp_list)
    exp_slot = -np.ones(n, np.int64)
    exp_slot[exp_list] = np.arange(len(exp_list))
    assert np.array_equal(slot.cpu().numpy(), exp_slot)
    close(far, np.minimum(g['rays'][:, 7], fg_far), 2e-6, 1e-7)
    close(ld, np.where(has_bg, fg_far, f32(1e10)), 2e-6, 1e-7)
    # camera outside the ellipsoid -> device error flag
    bad = rays.clone()
    bad[:, :3] *= 30
    sc.zero_()
    N.check(lib.mnr_ray_setup(bad.data_ptr(), n, N.host3(s['sphere_center']), N.host3(s['sphere_radius']),
                              far.data_ptr(), ld.data_ptr(), lst.data_ptr(), slot.data_ptr(), sc[0:1].data_ptr(),
                              sc[1:2].data_ptr(), None))
    assert int(sc[1]) == 1
    # _depth2pts_outside on caller-provided inverse depths, all three layouts
    depth = T(g['depth'])
    for xr, c2 in ((0, 0), (1, 0), (1, 1)):
        ncol = 7 if xr else 4
        pts = torch.empty(n, 32, ncol, device=DEV)
        dr = torch.empty(n, 32, device=DEV)
        N.check(lib.mnr_bg_samples(rays.data_ptr(), None, None, n, 32, None, 0.0, None, depth.data_ptr(),
                                   N.host3(s['sphere_center']), N.host3(s['sphere_radius']), xr, c2, None,
                                   pts.data_ptr(), dr.data_ptr(), None))
        close(pts, g['pts_%d%d' % (xr, c2)], 3e-5, 3e-6)
        close(dr, g['depth_real_%d%d' % (xr, c2)], 3e-5, 2e-6)


def test_perturbed_z_bit_exact():
    from mega_nerf import _native as N
    g = load('stages')
    rays = np.zeros((64, 8), f32)
    rays[:, 6], rays[:, 7] = 0.0, 1.0          # near=0, far=1 -> z = t exactly, then jitter (rendering.py:472-483)
    z = torch.empty(64, 32, device=DEV)
    rays_t, t_t, r_t = T(rays), T(g['linspace_32']), T(g['perturb_rand'])   # keep alive across the launch
    N.check(N.lib().mnr_fg_samples(rays_t.data_ptr(), None, 64, 32, t_t.data_ptr(), 0.7, r_t.data_ptr(), z.data_ptr(),
                                   None, None))
    assert np.array_equal(z.cpu().numpy(), g['perturbed'])


@pytest.mark.parametrize('n', [62, 30, 254])
@pytest.mark.parametrize('det', [True, False])
def test_sample_pdf_bit_exact(n, det):
    from mega_nerf import _native as N
    g = load('stages')
    nf = 128 if det else 64
    tag = '%d_%s' % (n, 'det' if det else 'rnd')
    bins, w = T(g['pdf_bins_%d' % n]), T(g['pdf_w_%d' % n])
    u = T(g['linspace_%d' % nf]) if det else T(g['pdf_u_' + tag])
    smp = torch.empty(64, nf, device=DEV)
    inds = torch.empty(64, nf, device=DEV, dtype=torch.int32)
    N.check(N.lib().mnr_sample_pdf(bins.data_ptr(), n + 1, w.data_ptr(), n, 64, None, n, nf, int(det), u.data_ptr(),
                                   smp.data_ptr(), inds.data_ptr(), None))
    assert np.array_equal(inds.cpu().numpy(), g['pdf_inds_' + tag].astype(np.int32))
    assert np.array_equal(smp.cpu().numpy(), g['pdf_samples_' + tag])


def test_merge_and_composite_against_oracle():
    from mega_nerf import _native as N
    rng = np.random.default_rng(5)
    n, Sc, Sf = 37, 64, 128
    for flip in (0, 1):
        zc = np.sort(rng.uniform(0.1, 3, (n, Sc)).astype(f32), -1)
        zf = rng.uniform(0.1, 3, (n, Sf)).astype(f32)            # unsorted (training draws)
        zf[:, 5] = zc[:, 7]                                      # exact ties: stable order matters
        if flip:
            zc = zc[:, ::-1].copy()
        rawc = rng.uniform(0, 1, (n, Sc, 4)).astype(f32)
        rawf = rng.uniform(0, 1, (n, Sf, 4)).astype(f32)
        rawc[..., 3] *= 30
        rawf[..., 3] *= 30
        drc, drf = rng.uniform(1, 9, (n, Sc)).astype(f32), rng.uniform(1, 9, (n, Sf)).astype(f32)
        last = np.where(rng.uniform(size=n) < 0.5, f32(1e10), rng.uniform(3.5, 4, n)).astype(f32)
        z_o, order = O._stable_sort(np.concatenate([zf, zc], -1), bool(flip))
        raw_o = np.take_along_axis(np.concatenate([rawf, rawc], 1), order[..., None], 1)
        dr_o = np.take_along_axis(np.concatenate([drf, drc], 1), order, 1)
        St = Sc + Sf
        z_m, raw_m, dr_m = (torch.empty(n, St, device=DEV), torch.empty(n, St, 4, device=DEV),
                            torch.empty(n, St, device=DEV))
        ordr = torch.empty(n, St, device=DEV, dtype=torch.int32)
        ins = [T(a) for a in (zf, rawf, drf, zc, rawc, drc)]                 # keep alive across the launch
        N.check(N.lib().mnr_merge_sorted(ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), Sf, ins[3].data_ptr(),
                                         ins[4].data_ptr(), ins[5].data_ptr(), Sc, n, None, flip, z_m.data_ptr(),
                                         raw_m.data_ptr(), dr_m.data_ptr(), ordr.data_ptr(), None))
        assert np.array_equal(ordr.cpu().numpy(), order.astype(np.int32))
        assert np.array_equal(z_m.cpu().numpy(), z_o)
        assert np.array_equal(raw_m.cpu().numpy(), raw_o)
        assert np.array_equal(dr_m.cpu().numpy(), dr_o)
        # compositing vs oracle.inference on the merged arrays
        res = {}
        has = last < 1e10
        diff = np.where(has, zf.max(-1), 0).astype(f32)

        class Fake:          # oracle.inference wants a model; feed the merged raw values straight through
            training = False

            def __call__(self, x, sigma_noise=None, use_coarse=None):
                return raw_o.reshape(-1, 4)
        hp = O.make_hparams(coarse_samples=Sc, fine_samples=Sf, appearance_dim=0)
        # inference() re-flips z when flip is set and no coarse z is stored (rendering.py:271-273): pre-flip it
        O.inference(res, 'fine', Fake(), np.zeros((n, 1, 3), f32), None, hp, np.zeros((n, St, 3), f32),
                    z_o[:, ::-1] if flip else z_o,
                    (last - diff)[:, None].astype(f32), True, True, True, True, True, bool(flip), dr_o)
        io = N.CompositeIO()
        io.z, io.raw, io.depth_real = z_m.data_ptr(), raw_m.data_ptr(), dr_m.data_ptr()
        lt, zfT = T(last), T(zf)
        io.last_delta, io.zmax_src, io.zmax_S = lt.data_ptr(), zfT.data_ptr(), Sf
        io.flip, io.N, io.S = flip, n, St
        outs = dict(weights=torch.empty(n, St, device=DEV), rgb=torch.empty(n, 3, device=DEV),
                    depth=torch.empty(n, device=DEV), depth_var=torch.empty(n, device=DEV),
                    bg_lambda=torch.empty(n, device=DEV))
        for k, v in outs.items():
            setattr(io, k, v.data_ptr())
        N.check(N.lib().mnr_composite(C.byref(io), None))
        close(outs['weights'], res['weights_fine'], 2e-5, 1e-7)
        close(outs['rgb'], res['rgb_fine'], 2e-5, 1e-6)
        close(outs['depth'], res['depth_fine'], 2e-5, 1e-6)
        close(outs['depth_var'], res['depth_variance_fine'], 1e-4, 1e-5)
        close(outs['bg_lambda'], res['bg_lambda_fine'], 2e-5, 1e-9)


# ---- fused MLP -----------------------------------------------------------------------------------
SUPPORTED_MLP = ['fg', 'bg', 'sh2', 'noapp', 'w64', 'w512']


@pytest.mark.parametrize('name', SUPPORTED_MLP)
def test_mlp_forward_matches_reference(name):
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    x = T(g[name + '_x'])
    with torch.no_grad():
        close(m(x), g[name + '_out'], 1e-4, 2e-6)
        close(m(x, sigma_noise=T(g[name + '_noise'])), g[name + '_out_noise'], 1e-4, 2e-6)
        close(m(x[:, :cfg.xyz_dim].contiguous(), sigma_only=True), g[name + '_sigma_only'], 1e-4, 2e-6)
        # ragged sizes: 1 row, a non-multiple of the 128-row workgroup tile, empty
        close(m(x[:1]), g[name + '_out'][:1], 1e-4, 2e-6)
        close(m(x[:131]), g[name + '_out'][:131], 1e-4, 2e-6)
        assert m(x[:0]).shape == (0, cfg.rgb_dim + 1)
        with pytest.raises(Exception, match='Unexpected input shape'):
            m(x[:, :-1])


@pytest.mark.parametrize('name', ['fg', 'bg'])
def test_mlp_tile32_variant(name):
    """32-samples-per-wave kernel (v_mfma_f32_32x32x2_f32) gives the same numbers as the default (16)."""
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    m.mfma_tile = 32
    x = T(g[name + '_x'])
    with torch.no_grad():
        close(m(x), g[name + '_out'], 1e-4, 2e-6)
        close(m(x[:77], sigma_noise=T(g[name + '_noise'][:77])), g[name + '_out_noise'][:77], 1e-4, 2e-6)


def test_mlp_repacks_after_weight_update():
    hp, cfg, w = mlp_variant('fg')
    g = load('mlp')
    m = native_nerf(cfg, w)
    x = T(g['fg_x'])
    with torch.no_grad():
        a = m(x).clone()
        m.sigma.bias.add_(0.5)              # in-place update bumps the version counter -> re-pack
        b = m(x)
    w2 = dict(w)
    w2['sigma.bias'] = w['sigma.bias'] + f32(0.5)
    close(b, O.nerf_forward(w2, cfg, g['fg_x']), 1e-4, 2e-6)
    assert not torch.allclose(a[:, 3], b[:, 3])


def test_layerwise_padded_weights_cached_until_the_parameter_changes():
    """models/layerwise.py: the zero-padded weight copies of the tiled-GEMM path (layer 0, skip layer, dir_a) are built once per
    parameter version, not once per row chunk -- and rebuilt after an in-place update or ``weights_changed()``."""
    hp, cfg, w = mlp_variant('w512')
    g = load('mlp')
    m = native_nerf(cfg, w)
    m.fused_supported = lambda: False                    # small launches of this width take the fused kernel: force the layer-by-layer path
    x = T(g['w512_x'])
    with torch.no_grad():
        close(m(x), g['w512_out'], 1e-4, 2e-6)
        cache = dict(m.__dict__.get('_padded_weights', {}))
        if not cache:
            pytest.skip('this build evaluates w512 without padded copies')
        m(x)
        assert all(m._padded_weights[k][1] is v[1] for k, v in cache.items())           # second call: the same tensors
        name = next(iter(cache))
        layer = dict(m.named_modules())[name]
        layer.weight.mul_(1.5)                                                           # in-place: version bump
        m(x)
        assert m._padded_weights[name][1] is not cache[name][1]
        w2 = {k: (v * f32(1.5) if k == name + '.weight' else v) for k, v in w.items()}
        close(m(x), O.nerf_forward(w2, cfg, g['w512_x']), 1e-4, 2e-6)
        before = m._padded_weights[name][1]
        m.weights_changed()
        m(x)
        assert m._padded_weights[name][1] is not before


def test_affine_appearance_needs_rgb():
    """nerf.py:156-158 multiplies a 3x3 colour transform with the colour: rgb_dim != 3 cannot work (the reference raises a shape error)."""
    from mega_nerf.models.nerf import NeRF, ShiftedSoftplus
    m = NeRF(12, 0, 8, [4], 64, 48, True, 10, 27, 3, ShiftedSoftplus()).to(DEV).eval()        # SH colour: no direction input (nerf.py:52-53)
    x = torch.zeros(8, 4, device=DEV)
    with torch.no_grad(), pytest.raises(Exception, match='rgb_dim == 3'):
        m(x)


def test_mlp_large_batch_against_oracle():
    """Full benchmark shape (1024 rays x 192 samples) -- every workgroup / chunk boundary exercised."""
    hp, cfg, w = mlp_variant('fg')
    rng = np.random.default_rng(3)
    B = 1024 * 192
    x = np.concatenate([rng.uniform(-1, 1, (B, 3)), rng.standard_normal((B, 3)), rng.integers(0, 100, (B, 1))], 1).astype(f32)
    m = native_nerf(cfg, w)
    with torch.no_grad():
        out = m(T(x)).cpu().numpy()
    sel = rng.permutation(B)[:4096]
    np.testing.assert_allclose(out[sel], O.nerf_forward(w, cfg, x[sel]), rtol=1e-4, atol=2e-6)
    assert np.isfinite(out).all()


# ---- end-to-end render_rays ----------------------------------------------------------------------
SUPPORTED_RENDER = ['render_fgbg_eval', 'render_fgonly_eval', 'render_q13_eval', 'render_default_samples_eval',
                    'render_sh2_eval', 'render_container_eval', 'render_cascade_eval', 'render_coarse_only_eval',
                    'render_relu_noapp_eval', 'render_w512_eval', 'render_container_sh2_eval',
                    'render_container_default_samples_eval', 'render_container_sh3_eval', 'render_cascade_bg_eval',
                    'render_container_q13_eval', 'render_container_fgonly_eval']


def native_models(name):
    from mega_nerf.models.cascade import Cascade
    from mega_nerf.models.mega_nerf import MegaNeRF
    hp, onerf, obg = build_case(name)

    def conv(om):
        if om is None:
            return None
        if om.cascade is not None:
            m = Cascade(native_nerf(om.cfg, om.cascade[0]), native_nerf(om.cfg, om.cascade[1]))
        elif om.subs is not None:
            m = MegaNeRF([native_nerf(om.cfg, s) for s in om.subs], torch.from_numpy(om.centroids), om.boundary_margin,
                         om.xyz_real, om.cluster_2d)
        else:
            m = native_nerf(om.cfg, om.params)
        m = m.to(DEV)
        m.train(om.training)
        return m
    return hp, conv(onerf), conv(obg)


@pytest.mark.parametrize('name', SUPPORTED_RENDER)
def test_render_rays_matches_reference(name):
    from mega_nerf.rendering import render_rays
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    hp = Namespace(**vars(hp))
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    rnd['_want_inds'] = True
    idx = T(g['idx'].astype(f32)) if hp.appearance_dim > 0 else None
    flags = [bool(v) for v in g['flags']]
    with torch.no_grad():
        res, present = render_rays(nerf, bg_nerf, T(g['rays']), idx, hp,
                                   T(s['sphere_center']) if bg_nerf is not None else None,
                                   T(s['sphere_radius']) if bg_nerf is not None else None, *flags, _randoms=rnd)
    ref_keys = sorted(k[4:] for k in g if k.startswith('res_'))
    assert sorted(res.keys()) == ref_keys
    assert present == bool(g['present'])
    for k in ref_keys:
        a, b = res[k].cpu().numpy(), g['res_' + k]
        assert a.shape == b.shape, k
        if 'variance' in k:
            # depth_variance = sum w (z - depth)^2 is a difference of nearly equal numbers (values of 1e-6 .. 1e-3 from z of 0.1 .. 2): its
            # relative error is the rgb / depth error amplified by depth^2 / variance, hence 1e-3 here (the north star states 1e-4 for rgb / depth)
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max())), err_msg=k)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5, err_msg=k)   # north star: 1e-4 rel on rgb/depth
    # end-to-end sample-index agreement with the reference: measured and explained per fixture (test_oracle_golden.check_index_agreement)
    for part in ('fg', 'bg'):
        if 'inds_' + part in g and '_inds_' + part in rnd:
            check_index_agreement(name, part, rnd['_inds_' + part].cpu().numpy(), g['inds_' + part])


def test_render_rays_raises_when_camera_outside_sphere():
    from mega_nerf.rendering import render_rays
    g = load('render_fgbg_eval')
    hp, nerf, bg_nerf = native_models('render_fgbg_eval')
    s = common.SCENE
    rays = T(g['rays']).clone()
    rays[:, :3] *= 40
    with pytest.raises(Exception, match='Not all your cameras are bounded by the unit sphere'):
        render_rays(nerf, bg_nerf, rays, T(g['idx'].astype(f32)), Namespace(**vars(hp)), T(s['sphere_center']),
                    T(s['sphere_radius']), True, False, True)


# ---- training ------------------------------------------------------------------------------------
def _torch_nerf_forward(w, cfg, x, noise):
    """fp64 torch restatement of nerf.py:115-160 for autograd reference gradients (test infrastructure)."""
    def emb(v, L):
        out = [v]
        for k in range(L):
            out += [torch.sin(2.0 ** k * v), torch.cos(2.0 ** k * v)]
        return torch.cat(out, -1)
    inp = emb(x[:, :cfg.xyz_dim], cfg.pos_xyz_dim)
    h = inp
    for i in range(cfg.layers):
        if i in cfg.skip_layers:
            h = torch.cat([inp, h], -1)
        h = torch.relu(h @ w['xyz_encodings.%d.0.weight' % i].T + w['xyz_encodings.%d.0.bias' % i])
    sig = h @ w['sigma.weight'].T + w['sigma.bias'] + noise.view(-1, 1)
    sig = torch.nn.functional.softplus(sig - 1, 1, 20)
    f = h @ w['xyz_encoding_final.weight'].T + w['xyz_encoding_final.bias']
    idx = x[:, -1].long()
    d_in = torch.cat([f, emb(x[:, -4:-1], cfg.pos_dir_dim), w['embedding_a.weight'][idx]], -1)
    d = torch.relu(d_in @ w['dir_a_encoding.0.weight'].T + w['dir_a_encoding.0.bias'])
    rgb = torch.sigmoid(d @ w['rgb.weight'].T + w['rgb.bias'])
    return torch.cat([rgb, sig], -1)


@pytest.mark.parametrize('name', ['fg', 'bg'])
def test_mlp_backward_against_fp64_autograd(name):
    """mnr_mlp_forward_train + mnr_mlp_backward_{data,weights} on a flat batch vs torch fp64 autograd."""
    from mega_nerf import _native as N
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    m = native_nerf(cfg, w)
    rng = np.random.default_rng(21)
    S, n_ray = 16, 37                                    # 592 rows: ragged last workgroup; 16 rows per ray
    B = S * n_ray
    xyz = rng.uniform(-1, 1, (B, cfg.xyz_dim)).astype(f32)
    dirs = rng.standard_normal((n_ray, 3)).astype(f32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    idx = rng.integers(0, 100, n_ray).astype(f32)
    noise = rng.uniform(0, 1, B).astype(f32)
    d_out = rng.standard_normal((B, 4)).astype(f32)
    # reference gradients (fp64, CPU)
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    x_full = np.concatenate([xyz, np.repeat(dirs, S, 0), np.repeat(idx, S)[:, None]], 1)
    out_ref = _torch_nerf_forward(wt, cfg, torch.tensor(x_full, dtype=torch.float64), torch.tensor(noise, dtype=torch.float64))
    (out_ref * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    # native
    lib = N.lib()
    xyz_t, dirs_t, idx_t, noise_t, dout_t = T(xyz), T(dirs), T(idx), T(noise), T(d_out)
    out = torch.empty(B, 4, device=DEV)
    cap = B + 40                                          # tape with a row offset, like the fine pass of a render
    row0 = 24
    fpr = m.tape_floats_per_row()
    tape, gtape = torch.zeros(cap * fpr, device=DEV), torch.zeros(cap * fpr, device=DEV)
    dheads = torch.zeros(cap, 4, device=DEV)
    io = m.mlp_io(xyz_t, cfg.xyz_dim, dirs_t, 3, idx_t, 1, S, B, out, noise_t)
    m.evaluate_train(io, tape, cap, row0)
    close(out, out_ref.detach().numpy(), 1e-4, 2e-6)
    grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    desc, packed = m.packed()
    pb = m.packed_bwd()
    gio = N.MlpGradIO()
    gio.tape, gio.gtape, gio.tape_rows, gio.tape_row0 = tape.data_ptr(), gtape.data_ptr(), cap, row0
    gio.d_out, gio.d_out_stride, gio.out, gio.out_stride = dout_t.data_ptr(), 4, out.data_ptr(), 4
    gio.dheads = dheads.data_ptr()
    gio.idx, gio.idx_stride, gio.idx_is_float, gio.rows_per_ray = idx_t.data_ptr(), 1, 1, S
    gio.n_rows = B
    counter = torch.zeros(1, device=DEV, dtype=torch.int32)
    gio.work_counter = counter.data_ptr()
    gio.grad = m.grad_struct(grads)
    N.check(lib.mnr_mlp_backward_data(packed.data_ptr(), pb.data_ptr(), C.byref(desc), C.byref(gio), None))
    N.check(lib.mnr_mlp_backward_weights(C.byref(desc), C.byref(gio), None))
    worst = {}
    for k, gt in grads.items():
        ref = wt[k].grad.numpy()
        got = gt.cpu().numpy()
        scale = max(float(np.abs(ref).max()), 1e-20)
        worst[k] = float(np.abs(got - ref).max()) / scale
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    assert not bad, bad


def test_composite_backward_against_autograd():
    from mega_nerf import _native as N
    rng = np.random.default_rng(8)
    n, S = 45, 192
    for flip in (0, 1):
        z = np.sort(rng.uniform(0.1, 3, (n, S)).astype(f32), -1)
        if flip:
            z = z[:, ::-1].copy()
        raw = rng.uniform(0, 1, (n, S, 4)).astype(f32)
        raw[..., 3] *= 20
        last = np.where(rng.uniform(size=n) < 0.5, f32(1e10), rng.uniform(3.5, 4, n)).astype(f32)
        zsub = rng.uniform(0.1, 3, (n, 128)).astype(f32)
        d_rgb = rng.standard_normal((n, 3)).astype(f32)
        d_lam = rng.standard_normal(n).astype(f32)
        # fp64 autograd reference of rendering.py:353-373
        rt = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
        zt = torch.tensor(z, dtype=torch.float64)
        ld = torch.tensor(last, dtype=torch.float64)
        ld = torch.where(ld < 1e10, ld - torch.tensor(zsub, dtype=torch.float64).max(-1)[0], ld)
        deltas = (zt[:, :-1] - zt[:, 1:]) if flip else (zt[:, 1:] - zt[:, :-1])
        deltas = torch.cat([deltas, ld[:, None]], -1)
        alphas = 1 - torch.exp(-deltas * rt[..., 3])
        Tt = torch.cumprod(1 - alphas + 1e-8, -1)
        lam = Tt[:, -1]
        Tt = torch.cat([torch.ones_like(Tt[:, :1]), Tt[:, :-1]], -1)
        rgb = ((alphas * Tt)[..., None] * rt[..., :3]).sum(1)
        ((rgb * torch.tensor(d_rgb, dtype=torch.float64)).sum() + (lam * torch.tensor(d_lam, dtype=torch.float64)).sum()).backward()
        io = N.CompositeGradIO()
        ts = [T(a) for a in (z, raw, last, zsub, d_rgb, d_lam)]
        d_raw = torch.empty(n, S, 4, device=DEV)
        io.z, io.raw, io.last_delta, io.zmax_src, io.zmax_S = ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), 128
        io.flip, io.N, io.S = flip, n, S
        io.d_rgb, io.d_bg_lambda, io.d_raw = ts[4].data_ptr(), ts[5].data_ptr(), d_raw.data_ptr()
        N.check(N.lib().mnr_composite_backward(C.byref(io), None))
        ref = rt.grad.numpy()
        got = d_raw.cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * float(np.abs(ref).max()))


def test_train_render_and_gradients_match_reference():
    """End to end: training-mode render_rays with the reference's captured random draws, then loss.backward().
    Outputs must match to 1e-4; gradients are compared against the reference's own fp32 and fp64 gradients
    (check_gradients_against_reference); the tight check of the backward kernels is
    test_train_gradients_against_fp64_with_the_kernels_own_relu_masks."""
    from mega_nerf.rendering import render_rays
    name = 'render_fgbg_train'
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    hp = Namespace(**vars(hp))
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    idx = T(g['idx'].astype(np.int32))
    flags = [bool(v) for v in g['flags']]
    assert nerf.training and bg_nerf.training
    res, present = render_rays(nerf, bg_nerf, T(g['rays']), idx, hp, T(s['sphere_center']), T(s['sphere_radius']), *flags,
                               _randoms=rnd)
    assert present == bool(g['present'])
    ref_keys = sorted(k[4:] for k in g if k.startswith('res_'))
    assert sorted(res.keys()) == ref_keys
    for k in ref_keys:
        a, b = res[k].detach().cpu().numpy(), g['res_' + k]
        if 'variance' in k:
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max())), err_msg=k)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5, err_msg=k)
    loss = torch.nn.functional.mse_loss(res['rgb_fine'], T(g['target']))
    np.testing.assert_allclose(float(loss.detach()), float(g['loss']), rtol=1e-4)
    loss.backward()
    check_gradients_against_reference(g, (('fg', nerf), ('bg', bg_nerf)), 'stagewise:' + name)


# How many tensors of each training fixture needed the one-in-ten allowance below when the check last ran on the MI355X (a flipped ReLU
# unit on a dominant row: discrete events, so the count is a property of fixture + kernels).  A fixture may not need MORE than recorded
# here: a regression can no longer hide in the allowance.  Keys: '<caller>:<fixture>'.  MNR_RECORD_ALLOWANCE=<file>
# appends the measured counts as JSON lines instead of asserting (to re-record after a deliberate kernel change).
ALLOWANCE_USED = {}
try:
    import json as _json
    ALLOWANCE_USED = _json.loads((Path(__file__).resolve().parent / 'golden' / 'gradient_allowance.json').read_text())
except Exception:        # (file absent: every fixture must then pass without the allowance)
    pass


def check_gradients_against_reference(g, models, key=None):
    """Parameter gradients against the golden file's two recordings of the reference's own gradients: fp32 autograd
    (``grad_*`` / ``gsub_*`` = every 37th element) and the same reference run in fp64 on the same random numbers (``g64_*``).
    The fp32 reference is itself off the fp64 one by up to 7e-2 of a tensor's scale: a ReLU unit whose pre-activation lies
    within an ulp of zero falls on one side in one fp32 implementation and on the other in the next (tests/fp64_ref.py), and a
    sharpened field's trunk gradients are sums of a few dominant rows.  Those are discrete events, so the bound has two parts
    (measured on the MI355X: the implementation's errors track the reference's tensor by tensor, mostly to the digit):
      * per tensor: error against fp64 <= 2e-4 of the tensor's scale + twice the reference's own fp32 error on that tensor;
        at most one tensor in ten may miss this (a unit that flipped here and not in the reference), and then
      * no tensor may be further from the exact gradient than the reference's own worst tensor of that model.
    Gradient norms: 2e-2.  The tight check of the backward kernels themselves (fp64 with the kernels' own masks, 2e-4) is
    test_train_gradients_against_fp64_with_the_kernels_own_relu_masks."""
    errs, norm_bad = {}, {}
    for tag, m in models:
        if m is None:
            continue
        for pn, p in m.named_parameters():
            got = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), f32)
            gn = float(g['gnorm_%s_%s' % (tag, pn)])
            nerr = abs(float(np.linalg.norm(got)) - gn) / max(gn, 1e-20) if gn > 0 else 0.0
            if 'grad_%s_%s' % (tag, pn) in g:
                r32 = g['grad_%s_%s' % (tag, pn)]
            else:
                r32, got = g['gsub_%s_%s' % (tag, pn)], got.reshape(-1)[::int(g['gstride']) if 'gstride' in g else 37]
            r64 = g['g64_%s_%s' % (tag, pn)].reshape(r32.shape)
            scale = float(np.abs(r64).max())
            if scale == 0:
                e32 = e64 = float(np.abs(got).max())
                eref = 0.0
            else:
                e32, e64 = float(np.abs(got - r32).max()) / scale, float(np.abs(got - r64).max()) / scale
                eref = float(np.abs(r32.astype(np.float64) - r64).max()) / scale
            errs[(tag, pn)] = (e64, eref, e32, nerr)
            if not nerr < 2e-2:
                norm_bad['%s.%s' % (tag, pn)] = nerr
    print({'%s.%s' % k: 'vs64 %.1e (ref32 vs64 %.1e) vs32 %.1e norm %.1e' % v for k, v in errs.items()})
    assert not norm_bad, norm_bad
    over = {k: v for k, v in errs.items() if not v[0] <= 2e-4 + 2 * v[1]}
    assert len(over) <= max(1, len(errs) // 10), over
    rec = os.environ.get('MNR_RECORD_ALLOWANCE')
    if rec and key is not None:
        with open(rec, 'a') as f:
            f.write(_json.dumps({key: len(over)}) + '\n')
    elif key is not None:
        assert len(over) <= ALLOWANCE_USED.get(key, 0), (key, 'tensors outside 2e-4 + 2 x reference error', over, 'recorded', ALLOWANCE_USED.get(key, 0))
    for (tag, pn), v in over.items():
        worst_ref = max(e[1] for (t, _), e in errs.items() if t == tag)
        assert v[0] <= worst_ref, ((tag, pn), v, worst_ref)


class _MaskedTorchNeRF:
    """fp64 NeRF.forward for oracle/torch_oracle.render_rays whose ReLU masks come from a queue (one entry per MLP pass, in the
    order torch_oracle evaluates them: bg coarse, bg fine, fg coarse, fg fine)."""

    def __init__(self, cfg, w, queue, dtype, training=True):
        self.cfg, self.w, self.queue, self.dtype, self.training = cfg, w, queue, dtype, training

    def __call__(self, x, noise=None):
        import fp64_ref
        mk = self.queue.pop(0)
        assert mk['dact'].shape[0] == x.shape[0], (mk['dact'].shape, x.shape)
        mk = dict(act=[a.to(self.dtype) for a in mk['act']], dact=mk['dact'].to(self.dtype))
        return fp64_ref.nerf_forward64(self.w, self.cfg, x, noise.view(-1) if noise is not None else None, mk)


def test_train_gradients_against_fp64_with_the_kernels_own_relu_masks():
    """The tight end-to-end gradient check: training-mode render_rays + MSE + backward on the reference's captured random draws
    (render_fgbg_train), against an fp64 restatement of the whole render (oracle/torch_oracle.py, pinned to the goldens) that is
    handed the ReLU masks found on the kernels' activation tapes.  What remains is fp32 rounding of the forward / backward
    kernels: every parameter gradient within 2e-4 of its tensor's scale (+ twice the error a plain fp32 CPU evaluation of the
    same masked function makes on that tensor, which only matters for the two background sigma-head tensors)."""
    import fp64_ref
    from mega_nerf import _native as N
    from mega_nerf.rendering import render_rays
    from oracle import torch_oracle as TO
    name = 'render_fgbg_train'
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    _, onerf, obg = build_case(name)
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    res, present = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(np.int32)), Namespace(**vars(hp)), T(s['sphere_center']),
                               T(s['sphere_radius']), False, True, False, _randoms=rnd)
    node = res['rgb_fine'].grad_fn              # training.RenderFunction's context: the branches with their tapes
    torch.cuda.synchronize()
    n_bg = int(node.bgb.part.n_units.item())
    lib = N.lib()
    queues = {}
    for tag, b, n in (('bg', node.bgb, n_bg), ('fg', node.fgb, node.fgb.n)):
        desc, _ = b.model.packed()
        queues[tag] = [fp64_ref.tape_masks(lib, b.model, desc, b.tape, b.cap, 0, n * b.Sc),
                       fp64_ref.tape_masks(lib, b.model, desc, b.tape, b.cap, b.rows_c, n * b.Sf)]
    torch.nn.functional.mse_loss(res['rgb_fine'], T(g['target'])).backward()
    # the same restatement on the CPU, in fp64 (the exact gradient of the masked function) and in fp32 (what a plain fp32
    # implementation of it gets: the yardstick for the tensors whose sum cancels to ~1e-5 of its terms, i.e. bg sigma.*)
    def restate(dtype):
        torch.set_default_dtype(dtype)
        try:
            w = {t: {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in om.params.items()} for t, om in (('fg', onerf), ('bg', obg))}
            q = {t: [dict(act=list(m_['act']), dact=m_['dact']) for m_ in queues[t]] for t in queues}
            fgm, bgm = _MaskedTorchNeRF(onerf.cfg, w['fg'], q['fg'], dtype), _MaskedTorchNeRF(obg.cfg, w['bg'], q['bg'], dtype)
            rr = {k[4:]: torch.from_numpy(v).to(dtype) for k, v in g.items() if k.startswith('rnd_')}
            out = TO.render_rays(fgm, bgm, torch.from_numpy(g['rays']).to(dtype), torch.from_numpy(g['idx']), hp,
                                 torch.from_numpy(s['sphere_center']).to(dtype), torch.from_numpy(s['sphere_radius']).to(dtype), randoms=rr)
            torch.nn.functional.mse_loss(out['rgb_fine'], torch.from_numpy(g['target']).to(dtype)).backward()
            assert not q['fg'] and not q['bg']
            return out, w
        finally:
            torch.set_default_dtype(torch.float32)
    r64, w64 = restate(torch.float64)
    r32, w32 = restate(torch.float32)
    np.testing.assert_allclose(res['rgb_fine'].detach().cpu().numpy(), r64['rgb_fine'].detach().numpy(), rtol=1e-4, atol=2e-5)
    worst = {}
    for tag, m in (('fg', nerf), ('bg', bg_nerf)):
        for pn, p in m.named_parameters():
            ref = w64[tag][pn].grad.numpy()
            worst['%s.%s' % (tag, pn)] = (fp64_ref.rel_to_scale(p.grad.cpu().numpy(), ref), fp64_ref.rel_to_scale(w32[tag][pn].grad.numpy(), ref))
    print({k: 'hip %.1e cpu-fp32 %.1e' % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v[0] <= 2e-4 + 2 * v[1]}
    assert not bad, bad
    assert sum(v[0] > 2e-4 for v in worst.values()) <= 2, worst      # (bg sigma.weight / sigma.bias: scale 3e-9, cancelling sum)


def test_train_step_reduces_loss():
    """A few Adam steps through TrainStep on a fixed batch must reduce the photometric loss."""
    from mega_nerf.training import TrainStep
    g = load('render_fgbg_train')
    hp, nerf, bg_nerf = native_models('render_fgbg_train')
    s = common.SCENE
    step = TrainStep(nerf, bg_nerf, Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']))
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    losses = [float(step(rays, idx, tgt)[0].detach()) for _ in range(8)]
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0]


def test_train_step_equals_a_plain_adam_loop():
    """TrainStep (fused optimiser launches) against the reference-style loop it stands for (runner.py:244-277: render_rays,
    mse_loss, backward, torch.optim.Adam.step on fg and bg) from the same seed: same loss trajectory.
    Guards the packed-weight caches: an optimiser that updates parameters without bumping their version counters (torch's
    fused Adam) must not leave the kernels running on the previous step's weights."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import TrainStep
    g = load('render_fgbg_train')
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    runs = []
    for use_step in (True, False):
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        hpn = Namespace(**vars(hp))
        nerf.eval(), bg_nerf.eval()          # deterministic render (no jitter / noise): TrainStep draws its random numbers from its
        torch.manual_seed(1234)              # own counter-based generator, the plain loop from torch's
        losses = []
        if use_step:
            step = TrainStep(nerf, bg_nerf, hpn, sc, sr)
            for _ in range(6):
                losses.append(float(step(rays, idx, tgt)[0].detach()))
        else:
            opts = [torch.optim.Adam(m.parameters(), lr=5e-4) for m in (nerf, bg_nerf)]
            gamma = 0.1 ** (1 / 500000)
            for it in range(6):
                for o in opts:
                    o.zero_grad(set_to_none=True)
                res, _ = render_rays(nerf, bg_nerf, rays, idx, hpn, sc, sr, False, True, False)
                loss = torch.nn.functional.mse_loss(res['rgb_fine'], tgt)
                loss.backward()
                for o in opts:
                    o.step()
                    for pg in o.param_groups:
                        pg['lr'] = 5e-4 * gamma ** (it + 1)
                losses.append(float(loss.detach()))
        runs.append(losses)
    # (with the weights frozen at their initial values the loss only jitters with the random draws: 0.08411 -> 0.08409 over four
    # steps is what smoke() printed before the fix)
    np.testing.assert_allclose(runs[0], runs[1], rtol=5e-5)
    assert runs[1][-1] < runs[1][0] and runs[0][-1] < runs[0][0], runs


@pytest.mark.parametrize('cluster_2d', [False, True])
@pytest.mark.parametrize('xyz_real', [False, True])
def test_meganerf_router_forward_matches_oracle(cluster_2d, xyz_real):
    """MegaNeRF.forward (device routing + gathered per-cell launches; mega_nerf.py:19-61) vs the numpy oracle: hard and blended routing,
    3-D and `cluster_2d` distances, foreground rows and background rows that carry their routing point in front (`xyz_real`, Q15).
    Every row must agree to 1e-4 unless its ROUTING sits on a rounding edge -- counted and explained, not budgeted: blended, a cell
    whose distance ratio is within 1e-5 of the margin (it is in or out of the blend); hard, two nearest centroids within 1e-6."""
    from mega_nerf.models.mega_nerf import MegaNeRF
    hp, cfg, _ = mlp_variant('bg' if xyz_real else 'fg')
    rng = np.random.default_rng(4)
    cent = np.array([[0, -.4, -.4], [0, -.4, .4], [0, .4, -.4], [0, .4, .4]], f32)
    subs_w = [common.make_weights(cfg, 100, 300 + i, sharpen=False) for i in range(4)]
    B = 1000
    pos = rng.uniform(-.8, .8, (B, cfg.xyz_dim))
    x = np.concatenate([pos, rng.standard_normal((B, 3)), rng.integers(0, 100, (B, 1))], 1).astype(f32)
    if xyz_real:                              # [xyz_real(3) | p_sphere(3) inv_depth(1) | dir(3) | idx(1)]: routed on the first three columns
        x = np.concatenate([rng.uniform(-.8, .8, (B, 3)).astype(f32), x], 1)
    c0 = 1 if cluster_2d else 0
    d = np.sqrt(((x[:, None, c0:3].astype(np.float64) - cent[None, :, c0:].astype(np.float64)) ** 2).sum(-1))
    ds = np.sort(d, 1)
    for margin in (1.0, 1.15):
        m = MegaNeRF([native_nerf(cfg, w) for w in subs_w], torch.from_numpy(cent), margin, xyz_real, cluster_2d).to(DEV).eval()
        with torch.no_grad():
            got = m(T(x)).cpu().numpy()
            got_s = m(T(np.ascontiguousarray(x[:, :(3 if xyz_real else 0) + cfg.xyz_dim])), sigma_only=True).cpu().numpy()
        exp = O.mega_nerf_forward(subs_w, cfg, cent, margin, xyz_real, cluster_2d, x)
        exp_s = O.mega_nerf_forward(subs_w, cfg, cent, margin, xyz_real, cluster_2d, x[:, :(3 if xyz_real else 0) + cfg.xyz_dim], sigma_only=True)
        if margin > 1:
            edge = (np.abs(d / ds[:, :1] - margin) < 1e-5).any(1)
        else:
            edge = (ds[:, 1] - ds[:, 0]) < 1e-6 * ds[:, 1]
        for a, b in ((got, exp), (got_s, exp_s)):
            bad = np.abs(a - b).max(-1) > 1e-4 * (1 + np.abs(b).max(-1))
            print('margin %.2f cluster_2d %s xyz_real %s: rows beyond 1e-4: %d of %d, of which on a routing edge: %d'
                  % (margin, cluster_2d, xyz_real, bad.sum(), B, (bad & edge).sum()))
            assert not (bad & ~edge).any(), (margin, np.flatnonzero(bad & ~edge)[:8])
            assert bad.sum() <= 2                  # (measured: 0)


def test_psnr_within_0p05_db_of_reference():
    """North-star PSNR criterion on identical rays/weights: PSNR of our render and of the reference render against the
    same target image differ by far less than 0.05 dB (fg+bg eval fixture, 96 rays)."""
    from mega_nerf.metrics import psnr
    from mega_nerf.rendering import render_rays
    g = load('render_fgbg_eval')
    hp, nerf, bg_nerf = native_models('render_fgbg_eval')
    s = common.SCENE
    with torch.no_grad():
        res, _ = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), Namespace(**vars(hp)),
                             T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    rng = np.random.default_rng(0)
    ref = T(g['res_rgb_fine'])
    for target in (T(rng.uniform(0, 1, tuple(ref.shape)).astype(f32)), (ref + 0.01 * torch.randn_like(ref)).clamp(0, 1)):
        ours, theirs = psnr(res['rgb_fine'], target), psnr(ref, target)            # device-side metric (mnr_image_metrics)
        assert abs(ours - theirs) < 0.05
        assert abs(theirs - O.psnr(ref.cpu().numpy(), target.cpu().numpy())) < 1e-4
    assert psnr(res['rgb_fine'], ref) > 80.0                 # image-level agreement with the reference itself


@pytest.mark.parametrize('kw', [dict(layer_dim=96), dict(layer_dim=2048, appearance_dim=0), dict(layer_dim=96, xyz_dim=4),
                                dict(layer_dim=160, pos_dir_dim=0, appearance_dim=0), dict(layer_dim=96, layers=6, skip_layers=[3])])
def test_generic_width_fallback_matches_oracle(kw):
    """Architectures without a fused instantiation (configs/nerf: layer_dim 2048; odd widths / depths) run through the
    layer-by-layer exact-fp32 MFMA path and still match the oracle."""
    kw = dict(kw)
    xyz_dim = kw.pop('xyz_dim', 3)
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, **kw)
    cfg = common.model_cfg(hp, xyz_dim, hp.layer_dim)
    w = common.make_weights(cfg, 100, 77, sharpen=False)
    m = native_nerf(cfg, w)
    assert not m.fused_supported()
    rng = np.random.default_rng(9)
    B = 300
    cols = [rng.uniform(-1, 1, (B, xyz_dim))]
    if cfg.pos_dir_dim > 0:
        cols.append(rng.standard_normal((B, 3)))
    if cfg.appearance_dim > 0:
        cols.append(rng.integers(0, 100, (B, 1)).astype(np.float64))
    x = np.concatenate(cols, 1).astype(f32)
    noise = rng.uniform(0, 1, (B, 1)).astype(f32)
    with torch.no_grad():
        close(m(T(x)), O.nerf_forward(w, cfg, x), 1e-4, 2e-6)
        close(m(T(x), sigma_noise=T(noise)), O.nerf_forward(w, cfg, x, sigma_noise=noise), 1e-4, 2e-6)
        close(m(T(x[:, :xyz_dim]).contiguous(), sigma_only=True), O.nerf_forward(w, cfg, x[:, :xyz_dim], sigma_only=True), 1e-4, 2e-6)


def test_cascade_render_with_unfused_width():
    """configs/nerf-shaped render (cascade, no bg, no appearance -> quirk Q8) at a width that has no fused kernel."""
    from mega_nerf.models.cascade import Cascade
    from mega_nerf.rendering import render_rays
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, use_cascade=True, appearance_dim=0, layer_dim=96)
    cfg = common.model_cfg(hp, 3, 96)
    wc, wf = common.make_weights(cfg, 100, 5), common.make_weights(cfg, 100, 6)
    g = load('render_cascade_eval')
    nerf = Cascade(native_nerf(cfg, wc), native_nerf(cfg, wf)).to(DEV).eval()
    with torch.no_grad():
        res, _ = render_rays(nerf, None, T(g['rays']), None, Namespace(**vars(hp)), None, None, True, False, True)
    ores, _ = O.render_rays(O.Model(cfg, cascade=(wc, wf)), None, g['rays'], None, hp, None, None, True, False, True)
    assert sorted(res) == sorted(ores)
    for k in ores:
        np.testing.assert_allclose(res[k].cpu().numpy(), ores[k], rtol=2e-4, atol=2e-5, err_msg=k)


def test_render_edge_batches_no_background_rays_and_empty():
    """A bg model is present but no ray of the batch leaves the ellipsoid before `far` (the reference returns
    bg_nerf_rays_present == False and zero bg terms, rendering.py:33-45,102-139); and an empty batch."""
    from mega_nerf.rendering import render_rays
    g = load('render_fgbg_eval')
    hp, nerf, bg_nerf = native_models('render_fgbg_eval')
    hp = Namespace(**vars(hp))
    s = common.SCENE
    rays = g['rays'].copy()
    rays[:, 7] = np.minimum(rays[:, 7], 0.3)              # far well inside the sphere for every ray
    idx = g['idx'].astype(f32)
    ohp, onerf, obg = build_case('render_fgbg_eval')
    want, present = O.render_rays(onerf, obg, rays, idx, ohp, s['sphere_center'], s['sphere_radius'], True, False, True)
    assert not present
    with torch.no_grad():
        res, got_present = render_rays(nerf, bg_nerf, T(rays), T(idx), hp, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    assert got_present is False
    assert sorted(res) == sorted(want)
    for k in want:
        np.testing.assert_allclose(res[k].cpu().numpy(), want[k], rtol=2e-4, atol=2e-5, err_msg=k)
    assert float(res['bg_rgb_fine'].abs().max()) == 0.0
    # empty batch: shapes only
    with torch.no_grad():
        res0, p0 = render_rays(nerf, bg_nerf, T(rays[:0]), T(idx[:0]), hp, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    assert p0 is False and res0['rgb_fine'].shape == (0, 3)


def test_training_step_without_background_rays():
    """Train-mode render + backward when no ray of the batch has a background segment: the bg branch runs over zero rows,
    bg gradients are exactly zero, fg gradients equal those of the same batch rendered without a bg model."""
    from mega_nerf.rendering import render_rays
    g = load('render_fgbg_train')
    s = common.SCENE
    rays = g['rays'].copy()
    rays[:, 7] = np.minimum(rays[:, 7], 0.3)
    idx, tgt = T(g['idx'].astype(np.int32)), T(g['target'])
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_fg')}
    grads = []
    for with_bg in (True, False):
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        hp = Namespace(**vars(hp))
        res, present = render_rays(nerf, bg_nerf if with_bg else None, T(rays), idx, hp, T(s['sphere_center']) if with_bg else None,
                                   T(s['sphere_radius']) if with_bg else None, False, True, False, _randoms=dict(rnd))
        assert present is False
        torch.nn.functional.mse_loss(res['rgb_fine'], tgt).backward()
        if with_bg:
            assert all(float(p.grad.abs().max()) == 0.0 for p in bg_nerf.parameters())
            assert 'bg_lambda_fine' in res
        grads.append({k: p.grad.clone() for k, p in nerf.named_parameters()})
        assert all(torch.isfinite(v).all() for v in grads[-1].values())
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-20), k


def test_render_accepts_views_and_int64_indices_and_no_altitude_range():
    """Callers hand in slices of larger tensors and int64 index vectors (runner.py:570, dataset_utils.py:39);
    get_rays without an altitude range keeps the constant bounds (ray_utils.py:44-62)."""
    from mega_nerf import ray_utils as RU
    from mega_nerf.rendering import render_rays
    g = load('render_fgbg_eval')
    hp, nerf, bg_nerf = native_models('render_fgbg_eval')
    hp = Namespace(**vars(hp))
    s = common.SCENE
    big = torch.zeros(g['rays'].shape[0] + 7, 11, device=DEV)
    big[3:-4, 2:10] = T(g['rays'])
    view = big[3:-4, 2:10]                                   # non-contiguous, storage offset
    with torch.no_grad():
        a, _ = render_rays(nerf, bg_nerf, view, T(g['idx'].astype(np.int64)), hp, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
        b, _ = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), hp, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    for k in b:
        assert torch.equal(a[k], b[k]), k
    d = RU.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, torch.device(DEV))
    r = RU.get_rays(d, T(s['c2w']), 0.05, 7.0, None)
    want = O.get_rays(O.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True), s['c2w'], 0.05, 7.0, None)
    np.testing.assert_allclose(r.cpu().numpy(), want, rtol=2e-6, atol=2e-7)
    assert float(r[..., 6].min()) == float(r[..., 6].max()) == np.float32(0.05) and float(r[..., 7].max()) == 7.0
