"""General training path (cascade / generic widths / no appearance / spherical harmonics): layer-by-layer adjoint vs
fp64 autograd, and end-to-end training renders + gradients vs the reference's (tests/golden/render_*_train.npz)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import common
from oracle import nerf_oracle as O
from test_gpu_parity import DEV, T, check_gradients_against_reference, close, native_models, native_nerf
from test_oracle_golden import load

pytestmark = pytest.mark.gpu
f32 = np.float32

_SH_C = [0.28209479177387814, 0.4886025119029199,
         [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]]


def _eval_sh2(sh, d):
    """spherical_harmonics.py:55-107 for deg 2 in torch (any dtype): sh (B, 3, 9), d (B, 3)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C2 = _SH_C[2]
    r = _SH_C[0] * sh[..., 0] - _SH_C[1] * y * sh[..., 1] + _SH_C[1] * z * sh[..., 2] - _SH_C[1] * x * sh[..., 3]
    r = r + C2[0] * x * y * sh[..., 4] + C2[1] * y * z * sh[..., 5] + C2[2] * (2 * z * z - x * x - y * y) * sh[..., 6] \
        + C2[3] * x * z * sh[..., 7] + C2[4] * (x * x - y * y) * sh[..., 8]
    return r


def torch_forward(w, cfg, xyz, dirq, idx, noise, sh_dirs=None):
    """fp64 restatement of nerf.py:115-160 (+ the SH colour of rendering.py:300-305) for reference gradients."""
    def emb(v, L):
        out = [v]
        for k in range(L):
            out += [torch.sin(2.0 ** k * v), torch.cos(2.0 ** k * v)]
        return torch.cat(out, -1)
    inp = emb(xyz, cfg.pos_xyz_dim)
    h = inp
    for i in range(cfg.layers):
        if i in cfg.skip_layers:
            h = torch.cat([inp, h], -1)
        h = torch.relu(h @ w['xyz_encodings.%d.0.weight' % i].T + w['xyz_encodings.%d.0.bias' % i])
    sig = h @ w['sigma.weight'].T + w['sigma.bias'] + noise.view(-1, 1)
    sig = torch.nn.functional.softplus(sig - 1, 1, 20) if cfg.shifted_softplus else torch.relu(sig)
    if 'xyz_encoding_final.weight' in w:
        parts = [h @ w['xyz_encoding_final.weight'].T + w['xyz_encoding_final.bias']]
        if cfg.pos_dir_dim > 0:
            parts.append(emb(dirq, cfg.pos_dir_dim))
        if cfg.appearance_dim > 0 and 'affine.weight' not in w:
            parts.append(w['embedding_a.weight'][idx])
        h = torch.relu(torch.cat(parts, -1) @ w['dir_a_encoding.0.weight'].T + w['dir_a_encoding.0.bias'])
    rgb = h @ w['rgb.weight'].T + w['rgb.bias']
    if 'affine.weight' in w:                       # nerf.py:156-158
        t = (w['embedding_a.weight'][idx] @ w['affine.weight'].T + w['affine.bias']).view(-1, 3, 4)
        rgb = (t[:, :, :3] @ rgb.unsqueeze(-1) + t[:, :, 3:]).squeeze(-1)
    if cfg.rgb_dim > 3:
        rgb = torch.sigmoid(_eval_sh2(rgb.view(rgb.shape[0], 3, -1), sh_dirs))
    else:
        rgb = torch.sigmoid(rgb)
    return torch.cat([rgb, sig], -1)


LW_VARIANTS = dict(
    w96=dict(xyz_dim=3, layer_dim=96),
    w96_bg=dict(xyz_dim=4, layer_dim=96),
    noapp=dict(xyz_dim=3, layer_dim=128, appearance_dim=0),
    plain_relu=dict(xyz_dim=3, layer_dim=64, appearance_dim=0, pos_dir_dim=0, shifted_softplus=False),
    sh2=dict(xyz_dim=3, layer_dim=128, sh_deg=2, pos_dir_dim=0),
    w320_skip2=dict(xyz_dim=3, layer_dim=320, layers=5, skip_layers=[2]),
    affine=dict(xyz_dim=3, layer_dim=256, affine_appearance=True),
    affine_bg_w128=dict(xyz_dim=4, layer_dim=128, affine_appearance=True),
)
# architectures whose training runs on the fused register-chained kernels (tape + hand-written chain)
FUSED_VARIANTS = dict(
    fused_sh2=dict(xyz_dim=3, layer_dim=256, sh_deg=2, pos_dir_dim=0),
    fused_sh2_bg=dict(xyz_dim=4, layer_dim=256, sh_deg=2, pos_dir_dim=0),
    fused_noapp=dict(xyz_dim=4, layer_dim=256, appearance_dim=0),
)


@pytest.mark.parametrize('name', list(LW_VARIANTS) + list(FUSED_VARIANTS))
def test_layerwise_backward_against_fp64_autograd(name):
    v = dict(LW_VARIANTS[name] if name in LW_VARIANTS else FUSED_VARIANTS[name])
    xyz_dim = v.pop('xyz_dim')
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, **v)
    cfg = common.model_cfg(hp, xyz_dim, hp.layer_dim)
    w = common.make_weights(cfg, 100, 900 + len(name), sharpen=False)
    m = native_nerf(cfg, w)
    assert m.fused_train_supported() == (name in FUSED_VARIANTS)
    rng = np.random.default_rng(5)
    S, n_ray = 12, 29
    B = S * n_ray
    xyz = rng.uniform(-1, 1, (B, xyz_dim)).astype(f32)
    dirs = rng.standard_normal((n_ray, 3)).astype(f32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    idx = rng.integers(0, 100, n_ray).astype(f32)
    noise = rng.uniform(0, 1, B).astype(f32)
    d_out = rng.standard_normal((B, 4)).astype(f32)
    sh_deg = 2 if cfg.rgb_dim > 3 else -1
    q8 = cfg.pos_dir_dim > 0 and cfg.appearance_dim == 0
    dir_rows = np.repeat(dirs, S, 0)
    dirq = np.concatenate([xyz[:, -1:], dir_rows[:, :2]], 1) if q8 else dir_rows       # quirk Q8 (nerf.py:146)
    wt = {k: torch.tensor(x, dtype=torch.float64, requires_grad=True) for k, x in w.items()}
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)   # noqa: E731
    ref = torch_forward(wt, cfg, t64(xyz), t64(dirq), torch.tensor(np.repeat(idx, S).astype(np.int64)), t64(noise), t64(dir_rows))
    (ref * t64(d_out)).sum().backward()
    out = torch.empty(B, 4, device=DEV)
    xyz_t, dirs_t, idx_t, noise_t = T(xyz), T(dirs), T(idx), T(noise)
    if q8:
        dq = T(dirq)
        tape = m.train_eval(xyz_t, xyz_dim, dq, 3, 1, None, 0, 1, B, out, noise_t, sh_deg, None, 0)
    else:
        tape = m.train_eval(xyz_t, xyz_dim, dirs_t, 3, S, idx_t if cfg.appearance_dim > 0 else None, 1, S, B, out, noise_t, sh_deg,
                            None, 0, dirs_t if sh_deg >= 0 else None, 3)
    close(out, ref.detach().numpy(), 1e-4, 3e-6)
    grads = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    tape.backward(T(d_out), 4, grads)
    worst = {}
    for k, gt in grads.items():
        r = wt[k].grad.numpy()
        worst[k] = float(np.abs(gt.cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-20)
    bad = {k: e for k, e in worst.items() if not e < 2e-4}
    assert not bad, bad


def test_nerf_forward_autograd_matches_fp64():
    """models.NeRF(x) with grad enabled (the reference module API) differentiates w.r.t. its parameters."""
    hp = O.make_hparams(coarse_samples=64, fine_samples=128)
    for xyz_dim, width in ((3, 256), (3, 96)):          # fused tape / layer-by-layer
        hp.layer_dim = width
        cfg = common.model_cfg(hp, xyz_dim, width)
        w = common.make_weights(cfg, 100, 77, sharpen=False)
        m = native_nerf(cfg, w).train()
        rng = np.random.default_rng(9)
        B = 333
        x = np.concatenate([rng.uniform(-1, 1, (B, 3)), rng.standard_normal((B, 3)), rng.integers(0, 100, (B, 1))], 1).astype(f32)
        d_out = rng.standard_normal((B, 4)).astype(f32)
        out = m(T(x))
        (out * T(d_out)).sum().backward()
        wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
        x64 = torch.tensor(x, dtype=torch.float64)
        ref = torch_forward(wt, cfg, x64[:, :3], x64[:, 3:6], x64[:, 6].long(), torch.zeros(B, dtype=torch.float64))
        (ref * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
        close(out, ref.detach().numpy(), 1e-4, 3e-6)
        for k, p in m.named_parameters():
            r = wt[k].grad.numpy()
            err = float(np.abs(p.grad.cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-20)
            assert err < 3e-4, (width, k, err)


TRAIN_CASES = ['render_fgbg_train', 'render_w512_train', 'render_cascade_bg_train', 'render_sh2_train', 'render_sh2_256_train', 'render_sh3_256_train', 'render_default_samples_train', 'render_noapp_train',
               'render_noapp256_train',
               'render_nerf_cfg_train', 'render_nerf_w2048_train', 'render_joint_train', 'render_joint_2d_train', 'render_joint_sh2_train', 'render_affine_train', 'render_fgonly_train']


@pytest.mark.parametrize('name', TRAIN_CASES)
def test_general_training_render_and_gradients_match_reference(name):
    """Training-mode render_rays through GeneralRenderFunction with the reference's captured random draws, loss as in
    runner.py:370-379, backward.  Outputs to 1e-4; gradients against the reference's fp32 and fp64 gradients
    (test_gpu_parity.check_gradients_against_reference)."""
    import mega_nerf.training as TRN
    from mega_nerf.rendering import render_rays
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    hp = Namespace(**vars(hp))
    s = common.SCENE
    rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
    idx = T(g['idx'].astype(np.int32)) if hp.appearance_dim > 0 else None
    flags = [bool(v) for v in g['flags']]
    sc = T(s['sphere_center']) if bg_nerf is not None else None
    sr = T(s['sphere_radius']) if bg_nerf is not None else None
    if name == 'render_noapp256_train':          # fused training kernels without the appearance input (Q8 directions)
        assert nerf.fused_train_supported() and bg_nerf.fused_train_supported() and not TRN._fast_path_ok(nerf, bg_nerf, hp)
    TRN.FORCE_GENERAL = True
    try:
        res, present = render_rays(nerf, bg_nerf, T(g['rays']), idx, hp, sc, sr, *flags, _randoms=rnd)
    finally:
        TRN.FORCE_GENERAL = False
    assert present == bool(g['present'])
    ref_keys = sorted(k[4:] for k in g if k.startswith('res_'))
    assert sorted(res.keys()) == ref_keys
    for k in ref_keys:
        a, b = res[k].detach().cpu().numpy(), g['res_' + k]
        if 'variance' in k:
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(b).max())), err_msg=k)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5, err_msg=k)
    typ = 'fine' if 'rgb_fine' in res else 'coarse'
    loss = torch.nn.functional.mse_loss(res['rgb_' + typ], T(g['target']))
    if hp.use_cascade and typ == 'fine':
        loss = (loss + torch.nn.functional.mse_loss(res['rgb_coarse'], T(g['target']))) / 2
    np.testing.assert_allclose(float(loss.detach()), float(g['loss']), rtol=1e-4)
    loss.backward()
    check_gradients_against_reference(g, (('fg', nerf), ('bg', bg_nerf)), 'general:' + name)


def test_general_path_agrees_with_tuned_path():
    """The default configuration through both implementations: same outputs, same gradients (fp32 summation order aside)."""
    import mega_nerf.training as TRN
    from mega_nerf.rendering import render_rays
    name = 'render_fgbg_train'
    g = load(name)
    s = common.SCENE
    grads = []
    for force in (False, True):
        hp, nerf, bg_nerf = native_models(name)
        rnd = {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}
        TRN.FORCE_GENERAL = force
        try:
            res, _ = render_rays(nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(np.int32)), Namespace(**vars(hp)), T(s['sphere_center']),
                                 T(s['sphere_radius']), False, True, False, _randoms=rnd)
        finally:
            TRN.FORCE_GENERAL = False
        torch.nn.functional.mse_loss(res['rgb_fine'], T(g['target'])).backward()
        grads.append((res['rgb_fine'].detach().cpu().numpy(),
                      {k: p.grad.cpu().numpy() for m_, t in ((nerf, 'fg.'), (bg_nerf, 'bg.')) for k, p in
                       ((t + kk, pp) for kk, pp in m_.named_parameters())}))
    np.testing.assert_array_equal(grads[0][0], grads[1][0])
    for k in grads[0][1]:
        a, b = grads[0][1][k], grads[1][1][k]
        assert float(np.abs(a - b).max()) <= 2e-4 * max(float(np.abs(a).max()), 1e-20), k


@pytest.mark.parametrize('width', [192, 2048])
def test_train_step_with_cascade_wide_model_reduces_loss(width):
    """configs/nerf-shaped training (cascade, no appearance, no bg; layer_dim 2048 in the reference's yaml) through the
    Runner's loss."""
    from mega_nerf.models.cascade import Cascade
    from mega_nerf.rendering import render_rays
    hp = O.make_hparams(coarse_samples=32, fine_samples=32, use_cascade=True, appearance_dim=0, layer_dim=width)
    cfg = common.model_cfg(hp, 3, width)
    nerf = Cascade(native_nerf(cfg, common.make_weights(cfg, 1, 5, sharpen=False)),
                   native_nerf(cfg, common.make_weights(cfg, 1, 6, sharpen=False))).train()
    g = load('render_fgbg_train')
    rays, tgt = T(g['rays']), T(g['target'])
    opt = torch.optim.Adam(nerf.parameters(), lr=5e-4)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        res, _ = render_rays(nerf, None, rays, None, Namespace(**vars(hp)), None, None, False, True, False)
        loss = (torch.nn.functional.mse_loss(res['rgb_fine'], tgt) + torch.nn.functional.mse_loss(res['rgb_coarse'], tgt)) / 2
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
