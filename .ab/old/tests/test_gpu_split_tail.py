"""The feature-split tail body (csrc/mlp_fwd_split.h): the background segment of a multi-segment launch run as 32-row workgroups whose
wavefront pairs split every layer's output features.  Its contract is BIT-IDENTITY with the 64-row body (`MNR_NO_SPLIT_TAIL=1` takes that
one): same K order per output feature, same heads, same tape -- checked here on outputs and on every tape float, for ragged row counts
either side of the switch (the kernel takes the split body when the device-side row count needs at most one workgroup per CU)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import common
from test_gpu_parity import DEV, T, native_nerf
from test_oracle_golden import O

pytestmark = pytest.mark.gpu
f32 = np.float32


def _models(sh_deg):
    kw = dict(sh_deg=sh_deg, pos_dir_dim=0) if sh_deg else {}
    hp = O.make_hparams(coarse_samples=64, fine_samples=128, **kw)
    out = []
    for xyz_dim, seed in ((3, 11), (4, 12)):
        cfg = common.model_cfg(hp, xyz_dim, hp.layer_dim)
        out.append((cfg, native_nerf(cfg, common.make_weights(cfg, 100, seed, sharpen=False))))
    return out


def _inputs(cfg, n, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    x = np.concatenate([rng.uniform(-1, 1, (n, cfg.xyz_dim)), d / np.linalg.norm(d, axis=-1, keepdims=True),
                        rng.integers(0, 100, (n, 1))], 1).astype(f32)
    return T(x), T(rng.standard_normal(n).astype(f32))


def _launch(models, xs, counts, sh_deg, train, split, device_count):
    """fg + bg rows through ONE mnr_mlp_forward_multi launch; returns (outputs, tapes) as numpy."""
    from mega_nerf import _native as N
    segs = (N.MlpLaunch * 2)()
    keep, outs, tapes = [], [], []
    for sg, (cfg, m), (x, noise), n in zip(segs, models, xs, counts):
        ncol = x.shape[1]
        out = torch.full((x.shape[0], 4), -7.0, device=DEV)
        units = torch.tensor([n], dtype=torch.int32, device=DEV) if device_count else None
        # device-side count: the launch covers the whole buffer (capacity), the kernel stops at `units` rows
        io = m.mlp_io(x, ncol, x[:, ncol - 4:], ncol, x[:, ncol - 1:], ncol, 1, x.shape[0] if device_count else n, out, noise, units, 1)
        if sh_deg:
            io.apply_sh_deg = sh_deg
        desc, packed = m.packed()
        sg.packed_dev, sg.desc, sg.io = packed.data_ptr(), C.pointer(desc), C.pointer(io)
        if train:
            tape = torch.full((x.shape[0] * m.tape_floats_per_row(),), -3.0, device=DEV)
            sg.tape_dev, sg.tape_rows, sg.tape_row0 = tape.data_ptr(), x.shape[0], 0
            tapes.append(tape)
        keep.append((io, desc, packed, units))
        outs.append(out)
    if split:
        os.environ.pop('MNR_NO_SPLIT_TAIL', None)
    else:
        os.environ['MNR_NO_SPLIT_TAIL'] = '1'
    try:
        N.check(N.lib().mnr_mlp_forward_multi(segs, 2, N.stream_ptr()))
        torch.cuda.synchronize()
    finally:
        os.environ.pop('MNR_NO_SPLIT_TAIL', None)
    return [o.cpu().numpy() for o in outs], [t.cpu().numpy() for t in tapes]


@pytest.mark.parametrize('sh_deg', [0, 2, 3])
@pytest.mark.parametrize('train', [False, True])
def test_split_tail_is_bit_identical_to_the_full_body(sh_deg, train):
    """Background row counts 1 .. 8192 take the split body on a 256-CU device (<= 256 workgroups of 32 rows), 8193+ the 64-row body on
    the finer grid; host-side counts and device-side counts (`n_units_dev`, what the training step passes) both."""
    models = _models(sh_deg)
    cap = 8704
    xs = [_inputs(models[0][0], 700, 1), _inputs(models[1][0], cap, 2)]
    for n_bg, device_count in ((1, False), (31, True), (33, False), (1000, True), (4417, True), (8192, False), (8193, True), (8700, False)):
        a = _launch(models, xs, (700, n_bg), sh_deg, train, True, device_count)
        b = _launch(models, xs, (700, n_bg), sh_deg, train, False, device_count)
        for got, ref in zip(a[0] + a[1], b[0] + b[1]):
            assert got.tobytes() == ref.tobytes(), (n_bg, device_count, int((got != ref).sum()))
        assert (a[0][1][:n_bg] != -7.0).all() and (a[0][1][n_bg:] == -7.0).all()      # every row written, none past the count


def test_split_tail_matches_the_single_model_forward():
    """... and the split body against the one-segment kernel (k_mlp_fwd, the golden-fixture-tested path) on the same rows."""
    models = _models(0)
    xs = [_inputs(models[0][0], 128, 3), _inputs(models[1][0], 2000, 4)]
    (o_fg, o_bg), _ = _launch(models, xs, (128, 2000), 0, False, True, False)
    with torch.no_grad():
        for (cfg, m), (x, noise), got in zip(models, xs, (o_fg, o_bg)):
            ref = m(x, sigma_noise=noise).cpu().numpy()
            assert got.tobytes() == ref.tobytes()
