"""Seeded synthetic scene + deterministic model weights shared by bench.py, the golden generator (which feeds them to
the *reference*), the oracle tests and the GPU parity tests.  Pure numpy; no oracle and no reference code.

Weights are produced by numpy's PCG64 (stable across numpy versions), NOT by torch
initialisers, so fixtures only need to store the seed.  Shapes and key names follow the
reference ``NeRF`` state_dict (mega_nerf/models/nerf.py:45-113).
"""
from types import SimpleNamespace

import numpy as np

f32 = np.float32


def model_cfg(hp, xyz_dim: int, layer_dim: int) -> SimpleNamespace:
    rgb_dim = 3 * ((hp.sh_deg + 1) ** 2) if hp.sh_deg is not None else 3
    return SimpleNamespace(xyz_dim=xyz_dim, pos_xyz_dim=hp.pos_xyz_dim, pos_dir_dim=hp.pos_dir_dim,
                           layers=hp.layers, skip_layers=list(hp.skip_layers), layer_dim=layer_dim,
                           appearance_dim=hp.appearance_dim, affine_appearance=hp.affine_appearance,
                           rgb_dim=rgb_dim, shifted_softplus=hp.shifted_softplus)


def make_weights(cfg, appearance_count: int, seed: int, sharpen: bool = True) -> dict:
    """state_dict (numpy fp32) for one NeRF with nn.Linear-like U(-1/sqrt(fan_in), 1/sqrt(fan_in)) init.
    ``sharpen`` scales the sigma head so the density field is peaky like a trained model."""
    rng = np.random.default_rng(seed)
    p = {}

    def lin(name, fin, fout):
        b = 1.0 / np.sqrt(fin)
        p[name + '.weight'] = rng.uniform(-b, b, (fout, fin)).astype(f32)
        p[name + '.bias'] = rng.uniform(-b, b, (fout,)).astype(f32)

    W = cfg.layer_dim
    in_xyz = cfg.xyz_dim + cfg.xyz_dim * cfg.pos_xyz_dim * 2
    for i in range(cfg.layers):
        fin = in_xyz if i == 0 else (W + in_xyz if i in cfg.skip_layers else W)
        lin('xyz_encodings.%d.0' % i, fin, W)
    in_dir = 3 + 3 * cfg.pos_dir_dim * 2 if cfg.pos_dir_dim > 0 else 0
    if cfg.appearance_dim > 0:
        p['embedding_a.weight'] = rng.standard_normal((appearance_count, cfg.appearance_dim)).astype(f32)
    if cfg.affine_appearance:
        lin('affine', cfg.appearance_dim, 12)
    uses_final = cfg.pos_dir_dim > 0 or (cfg.appearance_dim > 0 and not cfg.affine_appearance)
    if uses_final:
        lin('xyz_encoding_final', W, W)
        lin('dir_a_encoding.0', W + in_dir + (cfg.appearance_dim if not cfg.affine_appearance else 0), W // 2)
    lin('sigma', W, 1)
    lin('rgb', W // 2 if uses_final else W, cfg.rgb_dim)
    if sharpen:
        p['sigma.weight'] = (p['sigma.weight'] * f32(40)).astype(f32)
        p['sigma.bias'] = (p['sigma.bias'] + f32(2)).astype(f32)
    return p


# --- the synthetic camera / scene of SURVEY.md section 8(d) -----------------------------------------
SCENE = dict(W=400, H=400, fx=300.0, fy=300.0, cx=200.0, cy=200.0,
             c2w=np.array([[.6, 0, -.8, -.3], [0, 1, 0, .1], [.8, 0, .6, .05]], f32),
             near=0.01, far=1e5, ray_altitude_range=[-0.5, 0.2],
             sphere_center=np.array([-.15, 0, 0], f32), sphere_radius=np.array([.6, 1.2, 1.2], f32),
             appearance_count=100)


def pick_rays(all_rays: np.ndarray, n: int, seed: int):
    """n rays of the (H*W, 8) image + image indices ~ U{0..99}, by seeded permutation."""
    rng = np.random.default_rng(seed)
    sel = rng.permutation(all_rays.shape[0])[:n]
    idx = rng.integers(0, SCENE['appearance_count'], n)
    return np.ascontiguousarray(all_rays[sel]), idx
