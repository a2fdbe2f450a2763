"""Pins the numpy oracle (oracle/nerf_oracle.py) to golden vectors produced by the REAL reference
(tests/golden/make_golden.py, run in the build container).  CPU only.

Tolerances: integer/index outputs bit-exact; z/sample values that are pure elementwise fp32 chains
bit-exact; anything downstream of a GEMM / libm call within 2e-5 relative (BLAS and libm differ
between numpy and torch builds)."""
from pathlib import Path

import numpy as np
import pytest

import common
from oracle import nerf_oracle as O

G = Path(__file__).resolve().parent / 'golden'
f32 = np.float32


def load(name):
    return dict(np.load(G / (name + '.npz'), allow_pickle=False))


def close(a, b, rtol=2e-5, atol=2e-6):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_ray_generation():
    g = load('rays')
    W, H = int(g['W']), int(g['H'])
    fx, fy, cx, cy = [float(v) for v in g['intr']]
    for cp in (1, 0):
        d = O.get_ray_directions(W, H, fx, fy, cx, cy, bool(cp))
        close(d, g['dirs_c%d' % cp], 1e-6, 1e-7)
    d = g['dirs_c1']
    close(O.get_rays(d, g['c2w'], 0.01, 1e5, [-0.5, 0.2]), g['rays_alt'], 2e-6, 1e-7)
    close(O.get_rays(d, g['c2w'], 0.05, 2.0, None), g['rays_noalt'], 2e-6, 1e-7)
    close(O.get_rays(d, g['c2w'], 0.3, 0.9, [-0.35, -0.1]), g['rays_alt2'], 2e-6, 1e-7)
    close(O.get_rays_batch(g['batch_dirs'], g['batch_c2w'], 0.01, 1e5, [-0.5, 0.2]), g['rays_batch'], 2e-6, 1e-7)


def test_sphere_and_bg_points():
    g = load('stages')
    s = common.SCENE
    rays = g['rays']
    o, d = rays[:, :3], rays[:, 3:6]
    close(O.intersect_sphere(o, d, s['sphere_center'], s['sphere_radius']), g['fg_far'], 2e-6, 1e-7)
    close(O.intersect_sphere((o * f32(0.3)), d, None, None), g['fg_far_nosphere'], 2e-6, 1e-7)
    with pytest.raises(O.CameraOutsideSphere):
        O.intersect_sphere(o * f32(30), d, s['sphere_center'], s['sphere_radius'])
    for xr, c2 in ((0, 0), (1, 0), (1, 1)):
        pts, dr = O.depth2pts_outside(o[:, None], d[:, None], g['depth'], s['sphere_center'], s['sphere_radius'],
                                      bool(xr), bool(c2))
        close(pts, g['pts_%d%d' % (xr, c2)], 2e-5, 2e-6)
        close(dr, g['depth_real_%d%d' % (xr, c2)], 2e-5, 1e-6)


def test_perturb_bit_exact():
    g = load('stages')
    z = O.expand_and_perturb_z_vals(g['linspace_32'], 32, 0.7, 64, g['perturb_rand'])
    assert np.array_equal(z, g['perturbed'])


@pytest.mark.parametrize('n', [62, 30, 254])
@pytest.mark.parametrize('det', [True, False])
def test_sample_pdf_indices_bit_exact(n, det):
    g = load('stages')
    nf = 128 if det else 64
    tag = '%d_%s' % (n, 'det' if det else 'rnd')
    w, bins = g['pdf_w_%d' % n], g['pdf_bins_%d' % n]
    # the normaliser / cdf must reproduce torch's association order exactly
    ww = (w + f32(1e-8)).astype(f32)
    cdf = O.torch_cpu_cumsum((ww / O.torch_cpu_row_sum(ww)[:, None]).astype(f32))
    assert np.array_equal(np.concatenate([np.zeros((64, 1), f32), cdf], 1), g['pdf_cdf_' + tag])
    smp, inds = O.sample_pdf(bins, w, nf, det, u_rand=g['pdf_u_' + tag], t_fine=g['linspace_%d' % nf] if det else None,
                             return_inds=True)
    assert np.array_equal(inds, g['pdf_inds_' + tag].astype(np.int64))
    assert np.array_equal(smp, g['pdf_samples_' + tag])


def test_embedding_and_sh():
    g = load('stages')
    close(O.embedding(g['emb_x'], 12), g['emb_12'], 1e-6, 1e-6)
    close(O.embedding(g['emb_x'][:, :3], 4), g['emb_4'], 1e-6, 1e-6)
    for deg in range(5):
        close(O.eval_sh(deg, g['sh_in_%d' % deg], g['sh_dirs_%d' % deg]), g['sh_out_%d' % deg], 1e-5, 1e-6)


MLP_VARIANTS = dict(
    fg=dict(xyz_dim=3), bg=dict(xyz_dim=4), w512=dict(xyz_dim=3, layer_dim=512),
    sh2=dict(xyz_dim=3, sh_deg=2, pos_dir_dim=0), noapp=dict(xyz_dim=3, appearance_dim=0),
    relu=dict(xyz_dim=3, shifted_softplus=False), w64=dict(xyz_dim=4, layer_dim=64),
    plain=dict(xyz_dim=3, appearance_dim=0, pos_dir_dim=0), affine=dict(xyz_dim=3, affine_appearance=True))


def mlp_variant(name):
    v = dict(MLP_VARIANTS[name])
    xyz_dim = v.pop('xyz_dim')
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, **v)
    cfg = common.model_cfg(hp, xyz_dim, hp.layer_dim)
    return hp, cfg, common.make_weights(cfg, 100, 100 + len(name), sharpen=False)


@pytest.mark.parametrize('name', list(MLP_VARIANTS))
def test_mlp_forward(name):
    g = load('mlp')
    hp, cfg, w = mlp_variant(name)
    x = g[name + '_x']
    close(O.nerf_forward(w, cfg, x), g[name + '_out'])
    close(O.nerf_forward(w, cfg, x, sigma_noise=g[name + '_noise']), g[name + '_out_noise'])
    close(O.nerf_forward(w, cfg, x[:, :cfg.xyz_dim], sigma_only=True), g[name + '_sigma_only'])
    with pytest.raises(Exception, match='Unexpected input shape'):
        O.nerf_forward(w, cfg, x[:, :-1] if x.shape[1] > cfg.xyz_dim else np.zeros((3, cfg.xyz_dim + 5), f32))



# ---- "bit-exact sample indices", end to end ------------------------------------------------------------------------------------
# Stage level (identical inputs) the indices ARE bit-exact (test_sample_pdf_*).  End to end the inputs are a GEMM's outputs, and two
# fp32 GEMM implementations round differently; what that can move, measured over every fixture (numpy oracle AND the HIP path):
#  * the LAST deterministic sample of a ray: u = linspace(0, 1, Nf) ends in exactly 1.0 while cdf[-1] is 1 - ulp or 1 (+ ulp) depending
#    on the last bits of the pdf -> searchsorted(right=True) returns Nc - 1 or Nc - 2 (`inds` differs by one) -- and _sample_cdf
#    (rendering.py:521-534) yields the same z either way: below / above clamp to the last bin, t -> 1.  18 of 96 rays in
#    render_fgbg_eval, 3 of 13 background rays; never in training mode (random u).  Counted, not bounded: at most one per ray.
#  * anything else is a u that straddles a cdf entry by an ulp: measured 0 in every 64 + 128 fixture, 1 of 4 096 at 256 + 512 samples
#    (cdf steps of 1e-3 instead of 1e-2).  Bound: INDEX_OTHER_MAX (<= 2 x measured, 0 where 0 was measured).
INDEX_OTHER_MAX = {'render_default_samples_eval': 2, 'render_default_samples_train': 2, 'render_container_default_samples_eval': 2, 'render_container_fgonly_eval': 2}
INDEX_LOG = []


def check_index_agreement(name, part, got, ref, other_max=None):
    ref = np.asarray(ref).astype(np.int64)
    got = np.asarray(got).astype(np.int64)[:ref.shape[0]]
    diff = got != ref
    last, other = int(diff[..., -1].sum()), int(diff[..., :-1].sum())
    INDEX_LOG.append(dict(fixture=name, part=part, indices=int(ref.size), rays=int(ref.shape[0]), last_u=last, other=other))
    bound = INDEX_OTHER_MAX.get(name, 0) if other_max is None else other_max
    assert other <= bound, (name, part, 'indices off the last column that differ', other, 'of', ref.size, 'bound', bound)
    assert np.abs(got - ref)[diff].max(initial=0) <= 1, (name, part, 'an index moved by more than one bin')
    return last, other


# ---- end-to-end render_rays ---------------------------------------------------------------------
RENDER_CASES = {
    'render_fgbg_eval': dict(hp=dict(), seed=1),
    'render_fgbg_train': dict(hp=dict(), seed=2, fg_train=True, bg_train=True),
    'render_fgonly_eval': dict(hp=dict(), seed=3, bg=False),
    'render_sh2_eval': dict(hp=dict(sh_deg=2, pos_dir_dim=0), seed=4),
    'render_cascade_eval': dict(hp=dict(use_cascade=True, appearance_dim=0, layer_dim=64), seed=5, bg=False,
                                cascade=True),
    'render_cascade_bg_train': dict(hp=dict(use_cascade=True, layer_dim=64, bg_layer_dim=64), seed=6, cascade=True,
                                    fg_train=True, bg_train=True),
    'render_q13_eval': dict(hp=dict(), seed=7, bg_train=True),
    'render_container_eval': dict(hp=dict(container_path='dummy'), seed=8, container=4),
    'render_default_samples_eval': dict(hp=dict(coarse_samples=256, fine_samples=512), seed=9),
    'render_w512_eval': dict(hp=dict(layer_dim=512, bg_layer_dim=512), seed=10),
    'render_coarse_only_eval': dict(hp=dict(fine_samples=0), seed=11, bg=False),
    'render_relu_noapp_eval': dict(hp=dict(shifted_softplus=False, appearance_dim=0), seed=12),
    'render_sh2_train': dict(hp=dict(sh_deg=2, pos_dir_dim=0, layer_dim=128, bg_layer_dim=128), seed=13, fg_train=True, bg_train=True),
    'render_noapp_train': dict(hp=dict(appearance_dim=0, shifted_softplus=False, layer_dim=128, bg_layer_dim=128), seed=14,
                               fg_train=True, bg_train=True),
    'render_sh2_256_train': dict(hp=dict(sh_deg=2, pos_dir_dim=0), seed=18, fg_train=True, bg_train=True),
    'render_noapp256_train': dict(hp=dict(appearance_dim=0), seed=16, fg_train=True, bg_train=True),
    'render_sh3_eval': dict(hp=dict(sh_deg=3, pos_dir_dim=0), seed=21),
    'render_sh3_256_train': dict(hp=dict(sh_deg=3, pos_dir_dim=0), seed=26, fg_train=True, bg_train=True),
    'render_default_samples_train': dict(hp=dict(coarse_samples=256, fine_samples=512), seed=27, fg_train=True, bg_train=True),
    'render_container8_eval': dict(hp=dict(container_path='dummy'), seed=22, container=8),
    'render_container_w512_eval': dict(hp=dict(container_path='dummy', layer_dim=512, bg_layer_dim=512), seed=23, container=4),
    'render_joint_train': dict(hp=dict(train_mega_nerf='dummy', layer_dim=64, bg_layer_dim=64), seed=17, container=4, joint=True,
                               fg_train=True, bg_train=True),
    'render_w512_train': dict(hp=dict(layer_dim=512, bg_layer_dim=256), seed=24, fg_train=True, bg_train=True),
    'render_container25_eval': dict(hp=dict(container_path='dummy', layer_dim=512, bg_layer_dim=512), seed=25, container=25),
    'render_nerf_cfg_train': dict(hp=dict(coarse_samples=48, fine_samples=0, use_cascade=True, appearance_dim=0, layer_dim=160),
                                  seed=15, bg=False, cascade=True, fg_train=True),
    # BASELINE configs[0] at its real width: configs/nerf/*.yaml (use_cascade, layer_dim 2048, appearance_dim 0, no_bg_nerf), 8 rays
    'render_nerf_w2048_train': dict(hp=dict(use_cascade=True, appearance_dim=0, layer_dim=2048), seed=30, bg=False, cascade=True, fg_train=True),
    'render_container_default_samples_eval': dict(hp=dict(container_path='dummy', coarse_samples=256, fine_samples=512), seed=32, container=4),
    'render_container_sh3_eval': dict(hp=dict(container_path='dummy', sh_deg=3, pos_dir_dim=0), seed=33, container=4),
    'render_cascade_bg_eval': dict(hp=dict(use_cascade=True, layer_dim=64, bg_layer_dim=64), seed=34, cascade=True),
    'render_joint_sh2_train': dict(hp=dict(train_mega_nerf='dummy', sh_deg=2, pos_dir_dim=0, layer_dim=64, bg_layer_dim=64), seed=36, container=4,
                                   joint=True, fg_train=True, bg_train=True),
    'render_affine_train': dict(hp=dict(affine_appearance=True, layer_dim=64, bg_layer_dim=64), seed=37, fg_train=True, bg_train=True),
    'render_container_q13_eval': dict(hp=dict(container_path='dummy'), seed=38, container=4, bg_train=True),
    'render_container_fgonly_eval': dict(hp=dict(container_path='dummy'), seed=39, container=4, bg=False),
    'render_fgonly_train': dict(hp=dict(), seed=40, bg=False, fg_train=True),
    'render_container_sh2_eval': dict(hp=dict(container_path='dummy', sh_deg=2, pos_dir_dim=0), seed=31, container=4),
    # cluster_2d (Quad configs): distances over dims 1:3, background routed per sample on the true far-away point
    'render_container_2d_eval': dict(hp=dict(container_path='dummy'), seed=28, container=4, cluster_2d=True),
    'render_joint_2d_train': dict(hp=dict(train_mega_nerf='dummy', layer_dim=64, bg_layer_dim=64), seed=29, container=4, joint=True,
                                  cluster_2d=True, fg_train=True, bg_train=True),
}


def build_case(name):
    """(hp, nerf, bg_nerf) as oracle Models, from the same seeds the golden generator used."""
    c = RENDER_CASES[name]
    kw = dict(coarse_samples=64, fine_samples=128)
    kw.update(c['hp'])
    hp = O.make_hparams(**kw)
    seed = c['seed']
    A = common.SCENE['appearance_count']
    fcfg = common.model_cfg(hp, 3, hp.layer_dim)
    bcfg = common.model_cfg(hp, 4, hp.bg_layer_dim)
    bg = c.get('bg', True)
    ft, bt = c.get('fg_train', False), c.get('bg_train', False)
    if c.get('container'):
        n = c['container']
        g = load(name)
        cent = g['centroids']
        margin = 1.0 if c.get('joint') else hp.boundary_margin          # --train_mega_nerf routes hard (model_utils.py:37-42)
        nerf = O.Model(fcfg, subs=[common.make_weights(fcfg, A, seed * 1000 + i) for i in range(n)], centroids=cent,
                       boundary_margin=margin, xyz_real=False, cluster_2d=c.get('cluster_2d', False), training=ft)
        bg_nerf = O.Model(bcfg, subs=[common.make_weights(bcfg, A, seed * 1000 + 500 + i) for i in range(n)],
                          centroids=cent, boundary_margin=margin, xyz_real=True, cluster_2d=c.get('cluster_2d', False), training=bt) if bg else None
    elif c.get('cascade'):
        nerf = O.Model(fcfg, cascade=(common.make_weights(fcfg, A, seed * 1000),
                                      common.make_weights(fcfg, A, seed * 1000 + 1)), training=ft)
        bg_nerf = O.Model(bcfg, cascade=(common.make_weights(bcfg, A, seed * 1000 + 500),
                                         common.make_weights(bcfg, A, seed * 1000 + 501)), training=bt) if bg else None
    else:
        nerf = O.Model(fcfg, common.make_weights(fcfg, A, seed * 1000), training=ft)
        bg_nerf = O.Model(bcfg, common.make_weights(bcfg, A, seed * 1000 + 500), training=bt) if bg else None
    return hp, nerf, bg_nerf


def oracle_render(name, debug=None):
    g = load(name)
    hp, nerf, bg_nerf = build_case(name)
    s = common.SCENE
    rnd = {k[4:]: v for k, v in g.items() if k.startswith('rnd_')}
    idx = g['idx'].astype(f32) if hp.appearance_dim > 0 else None
    flags = [bool(v) for v in g['flags']]
    res, present = O.render_rays(nerf, bg_nerf, g['rays'], idx, hp,
                                 s['sphere_center'] if bg_nerf is not None else None,
                                 s['sphere_radius'] if bg_nerf is not None else None, *flags, rnd=rnd, debug=debug)
    return g, res, present


@pytest.mark.parametrize('name', list(RENDER_CASES))
def test_render_rays_matches_reference(name):
    dbg = {}
    g, res, present = oracle_render(name, dbg)
    ref_keys = sorted(k[4:] for k in g if k.startswith('res_'))
    assert sorted(res.keys()) == ref_keys
    assert present == bool(g['present'])
    for k in ref_keys:
        a, b = res[k], g['res_' + k]
        if 'depth' in k and 'variance' not in k:
            # bg depths are ~1e7 (quirk Q2): compare relatively
            close(a, b, 2e-4, 1e-5)
        elif 'variance' in k:
            close(a, b, 1e-3, 1e-4 * max(1.0, float(np.abs(b).max())))
        else:
            close(a, b, 2e-4, 2e-5)
    for part in ('fg', 'bg'):
        if 'inds_' + part in g and part in dbg and 'inds' in dbg[part]:
            check_index_agreement(name, part, dbg[part]['inds'], g['inds_' + part])


# ---- torch-CPU baseline oracle (oracle/torch_oracle.py) -------------------------------------------
@pytest.mark.parametrize('name', ['render_fgbg_eval', 'render_default_samples_eval'])
def test_torch_oracle_matches_reference(name):
    import torch
    from oracle import torch_oracle as TO
    g = load(name)
    hp, nerf, bg_nerf = build_case(name)
    s = common.SCENE
    fg, bg = TO.make_models(hp, nerf.cfg, nerf.params, bg_nerf.cfg, bg_nerf.params, s['appearance_count'])
    fg.eval(), bg.eval()
    with torch.no_grad():
        res = TO.render_rays(fg, bg, torch.from_numpy(g['rays']), torch.from_numpy(g['idx']), hp,
                             torch.from_numpy(s['sphere_center']), torch.from_numpy(s['sphere_radius']))
    for k in ('rgb_fine', 'fg_rgb_fine', 'bg_rgb_fine', 'depth_fine', 'bg_lambda_fine', 'fg_depth_fine'):
        np.testing.assert_allclose(res[k].numpy(), g['res_' + k], rtol=2e-4, atol=2e-5, err_msg=k)


# ---- the regime in which the reference's own sampling is decided by rounding (round 4) ------------------------------------------------
OVERFIT_KEYS = ('rgb_fine', 'fg_rgb_fine', 'bg_rgb_fine', 'depth_fine', 'fg_depth_fine', 'bg_depth_fine', 'bg_lambda_fine')


def overfit_case(name='render_overfit_eval'):
    """render_overfit_eval / render_overfit_hip_eval: (fixture, hparams, fg cfg, bg cfg, fg weights, bg weights).  The weights are the seeded initialisation plus
    the int8-quantised displacement of 30 reference Adam steps on the rendered batch -- decoded with exactly the expression of
    make_golden.run_overfit, so both reference renders in the fixture (fp32 and fp64) are renders of THESE weights."""
    g = load(name)
    hp = O.make_hparams(coarse_samples=64, fine_samples=128)
    A = common.SCENE['appearance_count']
    fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
    ws = []
    for tag, cfg, sd in (('fg', fcfg, int(g['seed_fg'])), ('bg', bcfg, int(g['seed_bg']))):
        init = common.make_weights(cfg, A, sd)
        ws.append({k: (init[k] + g['dq_%s_%s' % (tag, k)].astype(np.float32) * np.float32(g['ds_%s_%s' % (tag, k)])).astype(np.float32) for k in init})
    return g, hp, fcfg, bcfg, ws[0], ws[1]


def rays_beyond_bound(got, ref, n, rtol=1e-4, atol=2e-5):
    a, b = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return (np.abs(a - b) > atol + rtol * np.abs(b)).reshape(n, -1).any(1)


@pytest.mark.parametrize('name', ['render_overfit_eval', 'render_overfit_hip_eval'])
def test_overfit_fixture_is_the_reference_disagreeing_with_itself(name):
    """The fixtures' own content: the reference in fp32 and in fp64 on identical weights and rays draws different fine samples (a
    seventh / nearly half of the indices, on practically every ray) and differs beyond 1e-4 relative only in outputs that carry the
    background branch (depth through `depth_real`, quirk Q2: entries of 1e8; in the second fixture also the background colour);
    foreground colour / depth and bg_lambda agree on every ray."""
    g = load(name)
    n = g['rays'].shape[0]
    moved = g['inds_f32_fg'] != g['inds_f64_fg']
    assert moved.mean() > 0.05 and moved.any(1).mean() > 0.9
    total = 0
    for k in OVERFIT_KEYS:
        bad = rays_beyond_bound(g['res_f32_' + k], g['res_f64_' + k], n)
        assert int(bad.sum()) == int(g['selfdiff_' + k])
        if k in ('fg_rgb_fine', 'fg_depth_fine', 'bg_lambda_fine'):
            assert bad.sum() == 0, k
        total += int(bad.sum())
    assert total > 0 and int(g['selfdiff_bg_depth_fine']) > 0


@pytest.mark.parametrize('name', ['render_overfit_eval', 'render_overfit_hip_eval'])
def test_oracle_in_the_overfit_regime_meets_the_references_own_yardstick(name):
    """The numpy oracle on the first 128 rays of the overfit fixtures: an output the reference pins with both of its runs (fp32 and
    fp64 agree on all of these rays) is met on every ray; elsewhere no more rays may miss the fp64 run than 1.5 x the reference's own
    fp32 run does, + 4."""
    g, hp, fcfg, bcfg, fw, bw = overfit_case(name)
    s = common.SCENE
    n = 128
    res, present = O.render_rays(O.Model(fcfg, fw), O.Model(bcfg, bw), g['rays'][:n], g['idx'][:n].astype(np.float32), hp, s['sphere_center'],
                                 s['sphere_radius'], True, False, True)
    assert present == bool(g['present'])
    for k in OVERFIT_KEYS:
        own = int(rays_beyond_bound(g['res_f32_' + k][:n], g['res_f64_' + k][:n], n).sum())
        bad = int(rays_beyond_bound(res[k], g['res_f64_' + k][:n], n).sum())
        assert bad <= (1.5 * own + 4 if own else 0), (k, bad, own)
