"""mnr_train_step (csrc/step.hip): the whole training iteration of one or several cells as one C call -- against the
stage-by-stage path (the autograd node of mega_nerf/training.py, itself pinned to the reference's outputs and gradients), against
the reference's own recorded gradients, and for the independence of the cells that share its launches."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import common
from test_gpu_parity import DEV, T, check_gradients_against_reference, native_models
from test_oracle_golden import load

pytestmark = pytest.mark.gpu
f32 = np.float32


def _randoms_of(g):
    return {k[4:]: T(v).reshape(-1) if 'noise' in k else T(v) for k, v in g.items() if k.startswith('rnd_')}


def _grads(models):
    return {'%s.%s' % (t, k): p.grad.detach().cpu().numpy().copy() for t, m in models for k, p in m.named_parameters()}


@pytest.mark.parametrize('name', ['render_fgbg_train', 'render_sh2_256_train', 'render_sh3_256_train', 'render_default_samples_train', 'render_w512_train'])
def test_fused_step_equals_the_stagewise_path_and_the_reference(name):
    """render_w512_train: the Building shape (512-wide foreground, README "Larger models") -- forward on the wavefront-pair kernel, backward as
    tiled GEMMs + weight-gradient jobs sequenced inside the step, no host read;
    render_fgbg_train / render_sh2_256_train / render_sh3_256_train (configs/mega-nerf-sh-3: sh_deg 2, pos_dir_dim 0, and the
    degree-3 head BASELINE.json words -- the colour head's adjoint runs in
    k_sh_head_bwd inside the step; render_default_samples_train: the default models at the reference's 256 + 512 samples per ray, the
    other instantiation of the step's ray-stage kernels) on the reference's captured random draws: loss, rgb_fine, depth variance, bg_lambda and every parameter
    gradient of ONE mnr_train_step call against (i) the stage-by-stage path on the same random numbers -- same kernels for the
    MLP, restated kernels for the ray stages: equal up to the summation order of the atomically accumulated head / embedding
    gradients -- and (ii) the reference's own outputs and gradients."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import FusedTrainStep, fused_step_supported
    g = load(name)
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    # (i) stage by stage
    hp, nerf, bg_nerf = native_models(name)
    hpn = Namespace(**vars(hp))
    res, _ = render_rays(nerf, bg_nerf, rays, idx, hpn, sc, sr, False, True, False, _randoms=_randoms_of(g))
    loss = torch.nn.functional.mse_loss(res['rgb_fine'], tgt)
    loss.backward()
    ref = _grads((('fg', nerf), ('bg', bg_nerf)))
    ref_out = {k: v.detach().cpu().numpy() for k, v in res.items()}
    # (ii) one call
    hp, nerf2, bg2 = native_models(name)
    assert fused_step_supported(nerf2, bg2, hpn, rays.shape[0])
    step = FusedTrainStep([(nerf2, bg2)], hpn, sc, sr, rays.shape[0])
    l2, n_bg, err = step([(rays, idx, tgt)], _randoms=[_randoms_of(g)], optimize=False)
    torch.cuda.synchronize()
    assert int(err[0]) == 0 and int(n_bg[0]) > 0
    np.testing.assert_allclose(float(l2[0]), float(loss.detach()), rtol=2e-6)
    np.testing.assert_allclose(float(l2[0]), float(g['loss']), rtol=1e-4)
    np.testing.assert_array_equal(step.rgb[0].cpu().numpy(), ref_out['rgb_fine'])
    np.testing.assert_array_equal(step.depth_variance[0].cpu().numpy(), ref_out['depth_variance_fine'])
    np.testing.assert_array_equal(step.bg_lambda[0].cpu().numpy(), ref_out['bg_lambda_fine'])
    np.testing.assert_allclose(step.rgb[0].cpu().numpy(), g['res_rgb_fine'], rtol=1e-4, atol=2e-5)
    got = _grads((('fg', nerf2), ('bg', bg2)))
    worst = {k: float(np.abs(got[k] - ref[k]).max()) / max(float(np.abs(ref[k]).max()), 1e-30) for k in ref}
    print({k: '%.1e' % v for k, v in worst.items()})
    assert max(worst.values()) < 2e-5, {k: v for k, v in worst.items() if v >= 2e-5}
    check_gradients_against_reference(g, (('fg', nerf2), ('bg', bg2)), 'fused:' + name)


def _cell(seed, n_rays, sh=False, fg_width=256):
    """A cell of the benchmark's kind: default fg + bg models (``sh``: their spherical-harmonics form of that degree; True = 2;
    ``fg_width`` 512: the Building shape) with their own weights, their own batch."""
    from oracle import nerf_oracle as O
    from test_gpu_parity import native_nerf
    s = common.SCENE
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, **(dict(sh_deg=2 if sh is True else int(sh), pos_dir_dim=0) if sh else {}))
    fcfg, bcfg = common.model_cfg(hp, 3, fg_width), common.model_cfg(hp, 4, 256)
    fg = native_nerf(fcfg, common.make_weights(fcfg, s['appearance_count'], seed)).train()
    bg = native_nerf(bcfg, common.make_weights(bcfg, s['appearance_count'], seed + 500)).train()
    d = O.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True)
    rays_all = O.get_rays(d, s['c2w'], s['near'], s['far'], s['ray_altitude_range']).reshape(-1, 8)
    rays, idx = common.pick_rays(rays_all, n_rays, seed)
    tgt = np.random.default_rng(seed).uniform(0, 1, (n_rays, 3)).astype(f32)
    return hp, fg, bg, (T(rays), T(idx.astype(np.int32)), T(tgt))


@pytest.mark.parametrize('split,sh', [(False, False), (True, False), (False, True), (False, 3), (False, 'wide')], ids=['f32', 'split', 'f32-sh2', 'f32-sh3', 'f32-w512'])
def test_cells_sharing_a_step_are_independent(split, sh):
    """Three cells (own weights, own batches, own optimiser moments) stepped by ONE plan -- their rows side by side in every MLP
    launch -- against the same cells stepped one plan each (cell c of a plan draws its random numbers with key seed + c, so a
    lone plan seeded seed + c sees the same numbers): per-cell loss, rendered colours and gradients of the first step agree --
    colours exactly, gradients to the summation order of the atomics and of the weight-gradient partials -- and so do the weights
    after three Adam steps (parscripts/run_8.txt: independent trainers).  ``split``: the same through the split-precision kernels
    (per-cell exponent words of the weight-gradient scaling, device tables of the h2 images; the lone plans run their background
    branch on a side stream, the shared plan does not).  ``f32-w512``: 512-wide foreground cells (configs[3]: a rank's Building cells in
    one plan) -- the per-cell offsets of the tiled backward (csrc/step.hip wide_fg_backward) and of the padded weight copies."""
    from mega_nerf.training import FusedTrainStep
    s = common.SCENE
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    n_rays, seeds = 128, (11, 12, 13)
    wide = sh == 'wide'
    if wide:
        sh = False

    def run(groups):
        cells = [_cell(sd, n_rays, sh, 512 if wide else 256) for sd in seeds]
        hpn = Namespace(**vars(cells[0][0]))
        first = {}
        for grp in groups:
            step = FusedTrainStep([(cells[i][1], cells[i][2]) for i in grp], hpn, sc, sr, n_rays, seed=77 + grp[0], split_precision=split)
            for it in range(3):
                loss, n_bg, err = step([cells[i][3] for i in grp])
                if it == 0:
                    torch.cuda.synchronize()
                    assert int(err.max()) == 0 and int(n_bg.min()) > 0
                    for j, i in enumerate(grp):
                        first[i] = (float(loss[j]), step.rgb[j].cpu().numpy().copy(),
                                    {'%d.%s' % (k, n): v.cpu().numpy().copy() for k in range(2) for n, v in step.grad_views[2 * j + k].items()})
        torch.cuda.synchronize()
        return first, [{'%d.%s' % (q, k): p.detach().cpu().numpy().copy() for q, m in enumerate((c[1], c[2])) for k, p in m.named_parameters()}
                       for c in cells]

    (a, wa), (b, wb) = run([(0, 1, 2)]), run([(0,), (1,), (2,)])
    for i in range(len(seeds)):
        np.testing.assert_allclose(a[i][0], b[i][0], rtol=2e-6)
        np.testing.assert_array_equal(a[i][1], b[i][1])
        for k in a[i][2]:
            sc_ = max(float(np.abs(b[i][2][k]).max()), 1e-30)
            assert float(np.abs(a[i][2][k] - b[i][2][k]).max()) / sc_ < 2e-5, (i, k)
        for k in wa[i]:
            # three Adam steps move a weight by <= 3 lr; two summation orders of a noise-level gradient may disagree on its sign
            assert float(np.abs(wa[i][k] - wb[i][k]).max()) <= 3 * 5e-4 * 2 + 1e-6, (i, k)
            assert float(np.abs(wa[i][k] - wb[i][k]).mean()) <= 2e-5, (i, k)


def test_fused_adam_equals_torch_adam():
    """Six optimisation steps of the fused step (its own Adam kernel + ExponentialLR, re-pack) against the reference-style loop
    (render_rays, mse_loss, backward, torch.optim.Adam.step, runner.py:244-277) on eval-mode models (deterministic render):
    same loss trajectory, same final weights."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import FusedTrainStep
    g = load('render_fgbg_train')
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    hp, nerf, bg_nerf = native_models('render_fgbg_train')
    hpn = Namespace(**vars(hp))
    nerf.eval(), bg_nerf.eval()
    step = FusedTrainStep([(nerf, bg_nerf)], hpn, sc, sr, rays.shape[0])
    fused = [float(step([(rays, idx, tgt)])[0][0]) for _ in range(6)]
    hp, n2, b2 = native_models('render_fgbg_train')
    n2.eval(), b2.eval()
    opts = [torch.optim.Adam(m.parameters(), lr=5e-4) for m in (n2, b2)]
    gamma = 0.1 ** (1 / 500000)
    plain = []
    for it in range(6):
        for o in opts:
            o.zero_grad(set_to_none=True)
        res, _ = render_rays(n2, b2, rays, idx, hpn, sc, sr, False, True, False)
        loss = torch.nn.functional.mse_loss(res['rgb_fine'], tgt)
        loss.backward()
        for o in opts:
            o.step()
            for pg in o.param_groups:
                pg['lr'] = 5e-4 * gamma ** (it + 1)
        plain.append(float(loss.detach()))
    np.testing.assert_allclose(fused, plain, rtol=5e-5)
    assert fused[-1] < fused[0]
    # the images the kernels run on are the current weights (the bug class of the fused torch optimiser: tests/test_gpu_parity.py)
    with torch.no_grad():
        a = render_rays(nerf, bg_nerf, rays, idx, hpn, sc, sr, False, True, False)[0]['rgb_fine'].cpu().numpy()
        b = render_rays(n2, b2, rays, idx, hpn, sc, sr, False, True, False)[0]['rgb_fine'].cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)


def test_generated_random_numbers_are_uniform_and_keyed():
    """The step's own random numbers (training mode, nothing injected): two steps differ, the same (seed, step) repeats, and the
    loss stays in the range of the injected-randoms run."""
    from mega_nerf.training import FusedTrainStep
    s = common.SCENE
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    hp, fg, bg, batch = _cell(31, 128)
    hpn = Namespace(**vars(hp))
    st = FusedTrainStep([(fg, bg)], hpn, sc, sr, 128, seed=5)
    l1 = float(st([batch], optimize=False)[0][0])
    rgb1 = st.rgb[0].cpu().numpy().copy()
    l2 = float(st([batch], optimize=False)[0][0])
    rgb2 = st.rgb[0].cpu().numpy().copy()
    assert np.isfinite([l1, l2]).all() and not np.array_equal(rgb1, rgb2)
    hp, fg, bg, batch = _cell(31, 128)
    st2 = FusedTrainStep([(fg, bg)], hpn, sc, sr, 128, seed=5)
    np.testing.assert_allclose(float(st2([batch], optimize=False)[0][0]), l1, rtol=2e-6)      # (the loss is an atomic sum over the rays)
    np.testing.assert_array_equal(st2.rgb[0].cpu().numpy(), rgb1)
    assert abs(l1 - l2) < 0.05 * max(l1, l2)


@pytest.mark.parametrize('name,split', [('render_fgbg_eval', False), ('render_fgbg_eval', True), ('render_default_samples_eval', False),
                                        ('render_default_samples_eval', True), ('render_sh2_eval', False), ('render_sh3_eval', False),
                                        ('render_w512_eval', False),
                                        # merged containers: route -> all cells in one launch -> blend inside the same call (mega_nerf.py:19-61)
                                        ('render_container_eval', False), ('render_container8_eval', False), ('render_container25_eval', False),
                                        ('render_container_w512_eval', False), ('render_container_sh2_eval', False),
                                        ('render_container_default_samples_eval', False), ('render_container_sh3_eval', False),
                                        # cluster_2d: distances over y, z; the background routed per sample on o + d * depth_real (rendering.py:458-461)
                                        ('render_container_2d_eval', False)])
def test_fused_render_equals_the_stagewise_path_and_the_reference(name, split):
    """mnr_render_fwd (six launches; routed containers: thirteen) against the stage-by-stage render -- identical outputs, bit for bit, for
    the fp32 kernels -- and against the reference's outputs at the north-star tolerance; also on the split-precision MLP kernel."""
    from mega_nerf import rendering as R
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    s = common.SCENE
    hpn = Namespace(**vars(hp))
    args = (nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    R.SPLIT_PRECISION = split
    try:
        with torch.no_grad():
            assert R._fused_render_ok(nerf, bg_nerf, hpn, args[3], args[6], False, {})
            fused, present = R.render_rays(*args)
            R.FUSED_RENDER = False
            stage, present2 = R.render_rays(*args)
    finally:
        R.FUSED_RENDER, R.SPLIT_PRECISION = True, False
    assert present == present2 == bool(g['present']) and sorted(fused) == sorted(stage) == sorted(k[4:] for k in g if k.startswith('res_'))
    for k in fused:
        a, b = fused[k].cpu().numpy(), g['res_' + k]
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5, err_msg=k)
        np.testing.assert_array_equal(a, stage[k].cpu().numpy(), err_msg=k)


def test_cluster_2d_containers_can_be_kept_on_the_stage_by_stage_path(monkeypatch):
    """`cluster_2d` containers go through the one-call render like the others (the case above); MNR_NO_FUSED_2D_ROUTED_RENDER=1 keeps them
    on the stage-by-stage path (comparison runs)."""
    from mega_nerf import rendering as R
    name = 'render_container_2d_eval'
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    hpn = Namespace(**vars(hp))
    assert nerf.cluster_dim_start == 1
    assert R._fused_render_ok(nerf, bg_nerf, hpn, T(g['idx'].astype(f32)), T(common.SCENE['sphere_radius']), False, {})
    monkeypatch.setenv('MNR_NO_FUSED_2D_ROUTED_RENDER', '1')
    assert not R._fused_render_ok(nerf, bg_nerf, hpn, T(g['idx'].astype(f32)), T(common.SCENE['sphere_radius']), False, {})


def test_fused_render_benchmark_shape_all_rays():
    """The benchmark's 1024 x (64 + 128) render through mnr_render_fwd: every output of every ray within 1e-4 of the numpy oracle,
    the error flag raised for cameras outside the ellipsoid, an empty background handled."""
    from mega_nerf import ray_utils
    from mega_nerf.rendering import render_rays
    from oracle import nerf_oracle as O
    from test_gpu_parity import native_nerf
    s = common.SCENE
    hp = O.make_hparams(coarse_samples=64, fine_samples=128)
    A = s['appearance_count']
    fcfg, bcfg = common.model_cfg(hp, 3, 256), common.model_cfg(hp, 4, 256)
    fw, bw = common.make_weights(fcfg, A, 1000), common.make_weights(bcfg, A, 1500)
    d = ray_utils.get_ray_directions(s['W'], s['H'], s['fx'], s['fy'], s['cx'], s['cy'], True, torch.device(DEV))
    rays_all = ray_utils.get_rays(d, T(s['c2w']), s['near'], s['far'], s['ray_altitude_range']).view(-1, 8).cpu().numpy()
    rays, idx = common.pick_rays(rays_all, 1024, 7)
    fg, bg = native_nerf(fcfg, fw), native_nerf(bcfg, bw)
    hpn = Namespace(**vars(hp))
    with torch.no_grad():
        res, present = render_rays(fg, bg, T(rays), T(idx.astype(f32)), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    ores, opresent = O.render_rays(O.Model(fcfg, fw), O.Model(bcfg, bw), rays, idx.astype(f32), hp, s['sphere_center'], s['sphere_radius'], True, False, True)
    assert present == opresent and sorted(res) == sorted(ores)
    for k in ores:
        np.testing.assert_allclose(res[k].cpu().numpy(), ores[k], rtol=1e-4, atol=2e-5, err_msg=k)
    bad = T(rays).clone()
    bad[:, :3] *= 40
    with pytest.raises(Exception, match='Not all your cameras are bounded by the unit sphere'):
        render_rays(fg, bg, bad, T(idx.astype(f32)), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    # rays that all end inside the sphere: no background segment
    inside = T(rays).clone()
    inside[:, 7] = 0.3
    with torch.no_grad():
        res2, present2 = render_rays(fg, bg, inside, T(idx.astype(f32)), hpn, T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    assert not present2 and float(res2['bg_rgb_fine'].abs().max()) == 0.0 and np.isfinite(res2['rgb_fine'].cpu().numpy()).all()


@pytest.mark.parametrize('split', [False, True])
def test_fused_step_gradients_against_fp64_with_the_kernels_own_relu_masks(split):
    """The tight gradient check of tests/test_gpu_parity.py on the fused step -- fp32 kernels and the opt-in split-precision
    forward / data-gradient chain: mnr_train_step on the reference's captured random draws against an fp64 restatement of the
    whole render that is handed the ReLU masks found on the step's own activation tapes; every parameter gradient within 2e-4 of
    its tensor's scale (+ twice a plain fp32 CPU evaluation's error: the two background sigma-head tensors)."""
    import fp64_ref
    from mega_nerf import _native as N
    from mega_nerf.training import FusedTrainStep
    from oracle import torch_oracle as TO
    from test_gpu_parity import _MaskedTorchNeRF
    from test_oracle_golden import build_case
    name = 'render_fgbg_train'
    g = load(name)
    hp, nerf, bg_nerf = native_models(name)
    _, onerf, obg = build_case(name)
    s = common.SCENE
    hpn = Namespace(**vars(hp))
    n = g['rays'].shape[0]
    step = FusedTrainStep([(nerf, bg_nerf)], hpn, T(s['sphere_center']), T(s['sphere_radius']), n, split_precision=split)
    loss, n_bg, err = step([(T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target']))], _randoms=[_randoms_of(g)], optimize=False)
    torch.cuda.synchronize()
    nb = int(n_bg[0])
    lay, lib = step.layout, N.lib()
    wsf = step.workspace.view(torch.float32)
    Nc, Nf = hp.coarse_samples, hp.fine_samples
    queues = {}
    for tag, m, off, rows, units, Sc, Sf in (('bg', bg_nerf, lay.tape_bg_offset, lay.tape_bg_rows, nb, Nc // 2, Nf // 2),
                                             ('fg', nerf, lay.tape_fg_offset, lay.tape_fg_rows, n, Nc, Nf)):
        desc = m.model_desc()
        tape = wsf[off // 4:off // 4 + rows * m.tape_floats_per_row()]
        queues[tag] = [fp64_ref.tape_masks(lib, m, desc, tape, rows, 0, units * Sc), fp64_ref.tape_masks(lib, m, desc, tape, rows, n * Sc, units * Sf)]

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
            return out, w
        finally:
            torch.set_default_dtype(torch.float32)
    r64, w64 = restate(torch.float64)
    r32, w32 = restate(torch.float32)
    np.testing.assert_allclose(step.rgb[0].cpu().numpy(), r64['rgb_fine'].detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(step.rgb[0].cpu().numpy(), g['res_rgb_fine'], rtol=1e-4, atol=2e-5)
    worst = {}
    for k, (tag, m) in enumerate((('fg', nerf), ('bg', bg_nerf))):
        for pn, gv in step.grad_views[k].items():
            ref = w64[tag][pn].grad.numpy()
            worst['%s.%s' % (tag, pn)] = (fp64_ref.rel_to_scale(gv.cpu().numpy(), ref), fp64_ref.rel_to_scale(w32[tag][pn].grad.numpy(), ref))
    print('split' if split else 'fp32', {k: 'hip %.1e cpu-fp32 %.1e' % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if not v[0] <= 2e-4 + 2 * v[1]}
    assert not bad, bad
    assert sum(v[0] > 2e-4 for v in worst.values()) <= 2, worst
    check_gradients_against_reference(g, (('fg', nerf), ('bg', bg_nerf)), 'fused-%s:render_fgbg_train' % ('split' if split else 'f32'))


def test_split_precision_step_trains_like_the_fp32_step():
    """Six optimisation steps (eval-mode models: deterministic render) of the split-precision step against the fp32 step: same
    loss trajectory to 2e-5, and two cells in one split-precision plan behave like lone cells."""
    from mega_nerf.training import FusedTrainStep
    g = load('render_fgbg_train')
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    traj = []
    for split in (False, True):
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        nerf.eval(), bg_nerf.eval()
        step = FusedTrainStep([(nerf, bg_nerf)], Namespace(**vars(hp)), sc, sr, rays.shape[0], split_precision=split)
        traj.append([float(step([(rays, idx, tgt)])[0][0]) for _ in range(6)])
    np.testing.assert_allclose(traj[1], traj[0], rtol=2e-5)
    assert traj[1][-1] < traj[1][0]
    cells = [_cell(sd, 128) for sd in (41, 42)]
    hpn = Namespace(**vars(cells[0][0]))
    joint = FusedTrainStep([(c[1], c[2]) for c in cells], hpn, sc, sr, 128, seed=9, split_precision=True)
    lj, nbj, _ = joint([c[3] for c in cells], optimize=False)
    torch.cuda.synchronize()
    for i, c in enumerate([_cell(sd, 128) for sd in (41, 42)]):
        lone = FusedTrainStep([(c[1], c[2])], hpn, sc, sr, 128, seed=9 + i, split_precision=True)
        l1, nb1, _ = lone([c[3]], optimize=False)
        torch.cuda.synchronize()
        assert int(nb1[0]) == int(nbj[i])
        np.testing.assert_allclose(float(lj[i]), float(l1[0]), rtol=2e-6)
        np.testing.assert_array_equal(joint.rgb[i].cpu().numpy(), lone.rgb[0].cpu().numpy())


@pytest.mark.parametrize('split', [False, True])
def test_fused_step_at_the_reference_default_sample_counts(split):
    """256 + 512 samples per ray (opts.py:32-35; the other instantiation of the ray-stage kernels: 12 / 6 merged samples per lane):
    one fused step on eval-mode models (deterministic render) against the stage-by-stage path -- loss, colours, gradients."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import FusedTrainStep
    name = 'render_default_samples_eval'
    g = load(name)
    s = common.SCENE
    rays, idx = T(g['rays']), T(g['idx'].astype(np.int32))
    tgt = T(np.random.default_rng(3).uniform(0, 1, (rays.shape[0], 3)).astype(f32))
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    hp, nerf, bg_nerf = native_models(name)
    hpn = Namespace(**vars(hp))
    assert (hp.coarse_samples, hp.fine_samples) == (256, 512)
    res, _ = render_rays(nerf, bg_nerf, rays, idx, hpn, sc, sr, False, True, False)
    loss = torch.nn.functional.mse_loss(res['rgb_fine'], tgt)
    loss.backward()
    ref = _grads((('fg', nerf), ('bg', bg_nerf)))
    hp, n2, b2 = native_models(name)
    step = FusedTrainStep([(n2, b2)], hpn, sc, sr, rays.shape[0], split_precision=split)
    l2, n_bg, err = step([(rays, idx, tgt)], optimize=False)
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    np.testing.assert_allclose(float(l2[0]), float(loss.detach()), rtol=2e-5 if split else 2e-6)
    a, b = step.rgb[0].cpu().numpy(), res['rgb_fine'].detach().cpu().numpy()
    if split:
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-5)
    else:
        np.testing.assert_array_equal(a, b)
        got = _grads((('fg', n2), ('bg', b2)))
        worst = {k: float(np.abs(got[k] - ref[k]).max()) / max(float(np.abs(ref[k]).max()), 1e-30) for k in ref}
        assert max(worst.values()) < 2e-5, {k: v for k, v in worst.items() if v >= 2e-5}


@pytest.mark.parametrize('split', [False, True], ids=['f32', 'split'])
def test_background_branch_on_the_side_stream_changes_nothing(split, monkeypatch):
    """Single-cell plans may run the background branch of the forward on a plan-owned side stream (default for the split-precision step,
    MNR_STEP_TWO_STREAMS for the fp32 step; MNR_STEP_ONE_STREAM turns it off): same kernels, same inputs -- colours bit-identical,
    loss and gradients equal up to the order of the atomically accumulated sums, over several steps (fork / join every step)."""
    from mega_nerf.training import FusedTrainStep
    g = load('render_fgbg_train')
    s = common.SCENE
    batch = (T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target']))
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    out = {}
    for mode in ('one', 'two'):
        monkeypatch.delenv('MNR_STEP_ONE_STREAM', raising=False)
        monkeypatch.delenv('MNR_STEP_TWO_STREAMS', raising=False)
        monkeypatch.setenv('MNR_STEP_ONE_STREAM' if mode == 'one' else 'MNR_STEP_TWO_STREAMS', '1')
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        step = FusedTrainStep([(nerf, bg_nerf)], Namespace(**vars(hp)), sc, sr, batch[0].shape[0], seed=5, split_precision=split)
        loss, n_bg, err = step([batch], optimize=False)           # first: gradients on identical weights
        torch.cuda.synchronize()
        assert int(err.max()) == 0 and int(n_bg.min()) > 0
        first = (float(loss[0]), step.rgb[0].cpu().numpy().copy(),
                 {'%d.%s' % (k, n): v.cpu().numpy().copy() for k in range(2) for n, v in step.grad_views[k].items()})
        losses = [float(step([batch])[0][0]) for _ in range(4)]   # then four optimisation steps: fork / join every step
        torch.cuda.synchronize()
        out[mode] = (first, losses)
        del step
    (l1, rgb1, g1), (l2, rgb2, g2) = out['one'][0], out['two'][0]
    np.testing.assert_allclose(l1, l2, rtol=2e-6)
    np.testing.assert_array_equal(rgb1, rgb2)
    for k in g1:
        sc_ = max(float(np.abs(g1[k]).max()), 1e-30)
        assert float(np.abs(g1[k] - g2[k]).max()) / sc_ < 2e-5, k
    np.testing.assert_allclose(out['one'][1], out['two'][1], rtol=1e-3)


def test_fused_render_on_two_streams_changes_nothing(monkeypatch):
    """mnr_render_fwd with a lent side stream (mnr_render_io::side; MNR_RENDER_TWO_STREAMS in the Python mirror): bit-identical outputs."""
    from mega_nerf import rendering as R
    g = load('render_fgbg_eval')
    hp, nerf, bg_nerf = native_models('render_fgbg_eval')
    s = common.SCENE
    args = (nerf, bg_nerf, T(g['rays']), T(g['idx'].astype(f32)), Namespace(**vars(hp)), T(s['sphere_center']), T(s['sphere_radius']), True, False, True)
    outs = []
    for two in (False, True, True):
        if two:
            monkeypatch.setenv('MNR_RENDER_TWO_STREAMS', '1')
        else:
            monkeypatch.delenv('MNR_RENDER_TWO_STREAMS', raising=False)
        with torch.no_grad():
            res = R.render_rays(*args)[0]
        outs.append({k: v.cpu().numpy().copy() for k, v in res.items()})
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
        np.testing.assert_array_equal(outs[0][k], outs[2][k], err_msg=k)


def test_background_optimiser_is_gated_on_the_device_and_state_dict_round_trips():
    """runner.py:268-272: the background optimiser steps only on batches that had background rays.  The fused step decides that on the
    device (AdamTensor::gate = the cell's background-ray count) and keeps torch.optim.Adam's per-optimiser step count there too
    (mnr_step_model::adam_steps_dev).  Protocol: batch A (far bound inside the ellipsoid: no background ray), batch B (the golden batch),
    A again -- against the reference-style loop with the same rule, eval-mode models; then the optimiser state exported in the
    reference's checkpoint layout, loaded into a second plan and into plain torch optimisers, continues identically."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import FusedTrainStep
    g = load('render_fgbg_train')
    s = common.SCENE
    rays_b, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    rays_a = rays_b.clone()
    rays_a[:, 7] = 0.3                    # far = 0.3 from a camera well inside the ellipsoid: every ray ends before the sphere
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    hp, nerf, bg_nerf = native_models('render_fgbg_train')
    hpn = Namespace(**vars(hp))
    nerf.eval(), bg_nerf.eval()
    bg0 = {k: v.detach().clone() for k, v in bg_nerf.state_dict().items()}
    step = FusedTrainStep([(nerf, bg_nerf)], hpn, sc, sr, rays_b.shape[0])
    seq = [rays_a, rays_b, rays_a, rays_b]
    fused, nbg = [], []
    for i, r in enumerate(seq):
        l, nb, err = step([(r, idx, tgt)])
        fused.append(float(l[0]))
        nbg.append(int(nb[0]))
        assert int(err[0]) == 0
        if i == 0:       # no background ray: the background model has not moved, its optimiser has not stepped
            assert nbg[0] == 0
            for k, v in bg_nerf.state_dict().items():
                np.testing.assert_array_equal(v.cpu().numpy(), bg0[k].cpu().numpy())
            assert step.adam_t.tolist() == [[1, 0]]
            assert float(step.adam_m[0, sum((p.numel() + 3) // 4 * 4 for p in nerf.parameters()):].abs().max()) == 0.0
    assert nbg[1] > 0 and step.adam_t.tolist() == [[4, 2]]
    assert int(step.sticky.max()) == 0
    # reference-style loop
    hp, n2, b2 = native_models('render_fgbg_train')
    n2.eval(), b2.eval()
    opts = {'nerf': torch.optim.Adam(n2.parameters(), lr=5e-4), 'bg_nerf': torch.optim.Adam(b2.parameters(), lr=5e-4)}
    scheds = [torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.1 ** (1 / 500000)) for o in opts.values()]
    plain = []
    for r in seq:
        for o in opts.values():
            o.zero_grad(set_to_none=True)
        res, present = render_rays(n2, b2, r, idx, hpn, sc, sr, False, True, False)
        loss = torch.nn.functional.mse_loss(res['rgb_fine'], tgt)
        loss.backward()
        for key, o in opts.items():
            if key == 'bg_nerf' and not present:
                continue
            o.step()
        for sch in scheds:
            sch.step()
        plain.append(float(loss.detach()))
    np.testing.assert_allclose(fused, plain, rtol=5e-5)
    for (k, a), (_, b) in zip(bg_nerf.state_dict().items(), b2.state_dict().items()):
        assert float((a - b).abs().max()) <= 2 * 5e-4 * 2 + 1e-6, k        # two Adam steps: |dw| <= 2 lr each side
        assert float((a - b).abs().mean()) <= 2e-5, k
    # ---- checkpoint layout ----
    sd = step.state_dict()
    ref_sd = {k: o.state_dict() for k, o in opts.items()}
    for key in ('nerf', 'bg_nerf'):
        assert sd[key]['param_groups'][0].keys() == ref_sd[key]['param_groups'][0].keys()
        assert sd[key]['state'].keys() == ref_sd[key]['state'].keys()
        assert abs(sd[key]['param_groups'][0]['lr'] - ref_sd[key]['param_groups'][0]['lr']) < 1e-15
        for i in sd[key]['state']:
            assert float(sd[key]['state'][i]['step']) == float(ref_sd[key]['state'][i]['step']) == (4.0 if key == 'nerf' else 2.0)
            a, b = sd[key]['state'][i]['exp_avg_sq'], ref_sd[key]['state'][i]['exp_avg_sq']
            # (second moments of noise-level gradients: two summation orders of g, squared)
            assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-20
    # a second plan over copies of the models, fed the exported state, takes the same next step
    hp, n3, b3 = native_models('render_fgbg_train')
    n3.load_state_dict(nerf.state_dict()), b3.load_state_dict(bg_nerf.state_dict())
    n3.eval(), b3.eval()
    step3 = FusedTrainStep([(n3, b3)], hpn, sc, sr, rays_b.shape[0])
    step3.load_state_dict(sd)
    assert step3.adam_t.tolist() == [[4, 2]] and abs(step3.lr - step.lr) < 1e-18
    step3.repack()
    l1 = float(step([(rays_b, idx, tgt)])[0][0])
    l3 = float(step3([(rays_b, idx, tgt)])[0][0])
    np.testing.assert_allclose(l3, l1, rtol=2e-6)
    for (k, a), (_, b) in zip(nerf.state_dict().items(), n3.state_dict().items()):
        assert float((a - b).abs().max()) <= 2 * 5e-4 + 1e-6 and float((a - b).abs().mean()) <= 1e-6, k


def test_cell_trainer_mixes_fused_and_autograd_steps_on_one_state():
    """training.CellTrainer: batches of the planned size run mnr_train_step, any other batch the stage-by-stage autograd path -- on
    the same torch.optim.Adam objects, whose moment tensors are views of the plan's buffers (one source of truth).
    Against the reference-style loop over the same batch sequence (eval-mode models), incl. the ragged 592 -> 200-ray batch."""
    from mega_nerf.rendering import render_rays
    from mega_nerf.training import CellTrainer
    g = load('render_fgbg_train')
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    n = rays.shape[0]
    small = slice(0, 200)
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    hp, nerf, bg_nerf = native_models('render_fgbg_train')
    hpn = Namespace(**vars(hp))
    nerf.eval(), bg_nerf.eval()
    tr = CellTrainer(nerf, bg_nerf, hpn, sc, sr)
    seq = ['full', 'full', 'small', 'full', 'small', 'full']
    got = []
    for what in seq:
        b = (rays, idx, tgt) if what == 'full' else (rays[small], idx[small], tgt[small])
        got.append(float(tr.step(*b)[0]))
    assert tr.fused is not None and tr.fused.n_rays == n
    tr.sync()
    for key, o in tr.optimizers.items():
        for st in o.state.values():
            assert float(st['step']) == 6.0
    assert tr.fused.adam_t.tolist() == [[6, 6]]
    hp, n2, b2 = native_models('render_fgbg_train')
    n2.eval(), b2.eval()
    opts = [torch.optim.Adam(m.parameters(), lr=5e-4) for m in (n2, b2)]
    scheds = [torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.1 ** (1 / 500000)) for o in opts]
    plain = []
    for what in seq:
        b = (rays, idx, tgt) if what == 'full' else (rays[small], idx[small], tgt[small])
        for o in opts:
            o.zero_grad(set_to_none=True)
        res, _ = render_rays(n2, b2, b[0], b[1], hpn, sc, sr, False, True, False)
        loss = torch.nn.functional.mse_loss(res['rgb_fine'], b[2])
        loss.backward()
        for o in opts:
            o.step()
        for sch in scheds:
            sch.step()
        plain.append(float(loss.detach()))
    np.testing.assert_allclose(got, plain, rtol=1e-4)
    assert abs(tr.optimizers['nerf'].param_groups[0]['lr'] - opts[0].param_groups[0]['lr']) < 1e-15


def test_gathered_batch_equals_the_materialised_batch():
    """mnr_step_batch::select (training.GatheredBatch): the step's first kernel gathers rows of a device-resident training set -- rays,
    image indices, uint8 colours through the CPU's i / 255. table (dataset_utils.py:30) -- exactly as MemoryDataset.__getitem__ +
    collation would: colours bit-identical, loss and gradients equal up to the order of atomically accumulated sums, against the same batch
    materialised with torch indexing."""
    from mega_nerf.datasets.memory_dataset import unit_rgb, unit_table
    from mega_nerf.training import FusedTrainStep, GatheredBatch
    g = load('render_fgbg_train')
    s = common.SCENE
    n = g['rays'].shape[0]
    rng = np.random.default_rng(5)
    P = 5 * n
    # a "training set" of P rows that contains the golden batch's rays at shuffled positions, byte colours, int32 indices
    pos = rng.permutation(P)[:n]
    rays_all = np.tile(g['rays'], (5, 1)).astype(f32)
    rays_all[:, :3] += rng.normal(0, 1e-3, (P, 3)).astype(f32)
    rays_all[pos] = g['rays']
    idx_all = rng.integers(0, s['appearance_count'], P).astype(np.int32)
    idx_all[pos] = g['idx'].astype(np.int32)
    rgb_all = rng.integers(0, 256, (P, 3), dtype=np.uint8)
    src = (T(rays_all), T(idx_all), T(rgb_all), unit_table(DEV))
    sel = T(pos.astype(np.int64))
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    out = []
    for gathered in (True, False):
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        step = FusedTrainStep([(nerf, bg_nerf)], Namespace(**vars(hp)), sc, sr, n, seed=3)
        batch = GatheredBatch(src[0], src[1], src[2], sel, src[3]) if gathered else (src[0][sel], src[1][sel], unit_rgb(src[2][sel]))
        loss, n_bg, err = step([batch], optimize=False)
        torch.cuda.synchronize()
        assert int(err[0]) == 0 and int(n_bg[0]) > 0
        out.append((float(loss[0]), step.rgb[0].cpu().numpy().copy(), {k: v.cpu().numpy().copy() for q in range(2) for k, v in
                                                                         (('%d.%s' % (q, a), b) for a, b in step.grad_views[q].items())}))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)            # (the loss is an atomically accumulated sum over rays)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    # (gradients: head / embedding sums are accumulated with atomics -- equal up to their order)
    for k in out[0][2]:
        sc_ = max(float(np.abs(out[1][2][k]).max()), 1e-30)
        assert float(np.abs(out[0][2][k] - out[1][2][k]).max()) / sc_ < 2e-5, k


def test_sticky_health_bits_raise_what_the_reference_raises():
    """The trainer checks its steps' health every k iterations instead of synchronising every iteration (runner.py:260-261 checks every
    metric every step; rendering.py:412-414 raises inside render_rays): mnr_train_step ORs 'loss not finite' / 'camera outside the unit
    ellipsoid' into words that survive the per-step memset, CellTrainer.health() turns them into the reference's exceptions -- also when
    the offending step is followed by healthy ones."""
    from mega_nerf import _native as N
    from mega_nerf.training import CellTrainer
    g = load('render_fgbg_train')
    s = common.SCENE
    rays, idx, tgt = T(g['rays']), T(g['idx'].astype(np.int32)), T(g['target'])
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])

    def trainer():
        hp, nerf, bg_nerf = native_models('render_fgbg_train')
        return CellTrainer(nerf, bg_nerf, Namespace(**vars(hp)), sc, sr, seed=1)
    tr = trainer()
    tr.step(rays, idx, tgt)
    tr.health()                                                   # nothing to report
    assert int(tr.fused.sticky.max()) == 0
    bad = rays.clone()
    bad[3, :3] *= 40                                              # one camera far outside the ellipsoid
    tr.step(bad, idx, tgt)
    tr.step(rays, idx, tgt)                                       # a healthy step afterwards must not clear the flag
    assert int(tr.fused.sticky[0]) & N.MNR_STEP_STICKY_OUTSIDE
    with pytest.raises(Exception, match='Not all your cameras are bounded by the unit sphere'):
        tr.health()
    assert int(tr.fused.sticky.max()) == 0                        # reported once, then cleared
    tr2 = trainer()
    nan_t = tgt.clone()
    nan_t[5, 1] = float('nan')
    tr2.step(rays, idx, nan_t)
    tr2.step(rays, idx, tgt)
    assert int(tr2.fused.sticky[0]) & N.MNR_STEP_STICKY_NONFINITE
    with pytest.raises(Exception, match='Train metrics not finite'):
        tr2.health()


def test_gathered_batches_in_a_multi_cell_plan():
    """Two cells in one plan, each fed by row selections of its own resident training set (one GatheredBatch per cell, one cell's
    colours as bytes and the other's batch materialised): per-cell loss and colours equal the all-materialised call."""
    from mega_nerf.datasets.memory_dataset import unit_rgb, unit_table
    from mega_nerf.training import FusedTrainStep, GatheredBatch
    s = common.SCENE
    sc, sr = T(s['sphere_center']), T(s['sphere_radius'])
    n = 128
    rng = np.random.default_rng(9)

    def source(cell):
        rays, idx, _ = cell[3]
        P = 3 * n
        pos = rng.permutation(P)[:n]
        ra = rays.repeat(3, 1).clone()
        ra[:, :3] += 1e-3 * torch.randn(P, 3, device=ra.device)
        ra[T(pos.astype(np.int64))] = rays
        ia = torch.randint(0, s['appearance_count'], (P,), device=ra.device, dtype=torch.int32)
        ia[T(pos.astype(np.int64))] = idx
        ca = torch.randint(0, 256, (P, 3), device=ra.device, dtype=torch.uint8)
        return ra.contiguous(), ia.contiguous(), ca.contiguous(), T(pos.astype(np.int64))
    out = []
    for gathered in (True, False):
        rng = np.random.default_rng(9)
        torch.manual_seed(4)
        cells = [_cell(sd, n) for sd in (21, 22)]
        srcs = [source(c) for c in cells]
        hpn = Namespace(**vars(cells[0][0]))
        step = FusedTrainStep([(c[1], c[2]) for c in cells], hpn, sc, sr, n, seed=5)
        mat = [(sr_[0][sr_[3]], sr_[1][sr_[3]], unit_rgb(sr_[2][sr_[3]])) for sr_ in srcs]
        batches = [GatheredBatch(srcs[0][0], srcs[0][1], srcs[0][2], srcs[0][3], unit_table(DEV)), mat[1]] if gathered else mat
        loss, n_bg, err = step(batches, optimize=False)
        torch.cuda.synchronize()
        assert int(err.max()) == 0
        out.append((loss.cpu().numpy().copy(), step.rgb.cpu().numpy().copy()))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)
    np.testing.assert_array_equal(out[0][1], out[1][1])
