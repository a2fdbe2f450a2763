"""Torch-CPU restatement of the reference hot path -- the CPU *baseline* leg of bench.py.

TEST INFRASTRUCTURE ONLY (same rules as nerf_oracle.py: imported by tests/ and bench.py's cpu_baseline leg,
never by the product path).  The numpy oracle (nerf_oracle.py) is the parity checker; this file exists because the
north star asks for "the reference PyTorch CPU path timed on the same box's host cores" and /root/reference is not
present on the GPU box: it restates the same algorithm (render_rays, non-cascade, single NeRF per branch:
rendering.py:15-536, nerf.py:8-160) with the same torch CPU ops the reference issues (MKL addmm, cat, sort,
searchsorted, cumprod, gather), so its speed is representative of the reference on that host, for both the
forward render and a full training step (autograd backward + 2x Adam, runner.py:246-277).
It is pinned to the same golden vectors as the numpy oracle (tests/test_oracle_golden.py::test_torch_oracle_*).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def embedding(x: torch.Tensor, L: int) -> torch.Tensor:
    out = [x]
    for k in range(L):
        out += [torch.sin((2.0 ** k) * x), torch.cos((2.0 ** k) * x)]
    return torch.cat(out, -1)


class TorchNeRF(torch.nn.Module):
    """Parameter names identical to the reference state_dict, so golden weights load directly."""

    def __init__(self, cfg, appearance_count: int):
        super().__init__()
        self.cfg = cfg
        W, D = cfg.layer_dim, cfg.xyz_dim
        in_xyz = D * (1 + 2 * cfg.pos_xyz_dim)
        in_dir = 3 * (1 + 2 * cfg.pos_dir_dim) if cfg.pos_dir_dim > 0 else 0
        self.xyz_encodings = torch.nn.ModuleList(
            torch.nn.Sequential(torch.nn.Linear(in_xyz if i == 0 else W + (in_xyz if i in cfg.skip_layers else 0), W))
            for i in range(cfg.layers))
        self.embedding_a = torch.nn.Embedding(appearance_count, cfg.appearance_dim)
        self.xyz_encoding_final = torch.nn.Linear(W, W)
        self.dir_a_encoding = torch.nn.Sequential(torch.nn.Linear(W + in_dir + cfg.appearance_dim, W // 2))
        self.sigma = torch.nn.Linear(W, 1)
        self.rgb = torch.nn.Linear(W // 2, 3)

    def forward(self, x: torch.Tensor, sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        cfg = self.cfg
        inp = embedding(x[:, :cfg.xyz_dim], cfg.pos_xyz_dim)
        h = inp
        for i, enc in enumerate(self.xyz_encodings):
            if i in cfg.skip_layers:
                h = torch.cat([inp, h], -1)
            h = torch.relu_(enc(h))            # (in place, as the reference's nn.ReLU(True): nerf.py:70)
        sigma = self.sigma(h)
        if sigma_noise is not None:
            sigma = sigma + sigma_noise
        sigma = F.softplus(sigma - 1, 1, 20) if cfg.shifted_softplus else torch.relu(sigma)
        f = self.xyz_encoding_final(h)
        d_in = torch.cat([f, embedding(x[:, -4:-1], cfg.pos_dir_dim), self.embedding_a(x[:, -1].long())], -1)
        rgb = torch.sigmoid(self.rgb(torch.relu_(self.dir_a_encoding(d_in))))
        return torch.cat([rgb, sigma], -1)


def intersect_sphere(o, d, c, r):
    o, d = (o - c) / r, d / r
    d1 = -(d * o).sum(-1) / (d * d).sum(-1)
    p = o + d1.unsqueeze(-1) * d
    return d1 + torch.sqrt(1. - (p * p).sum(-1)) / torch.norm(d, dim=-1)


def depth2pts_outside(o, d, depth, c, r):
    o, d = (o - c) / r, d / r
    d1 = -(d * o).sum(-1) / (d * d).sum(-1)
    p_mid = o + d1.unsqueeze(-1) * d
    pn = torch.norm(p_mid, dim=-1)
    d2 = torch.sqrt(1. - pn * pn) / d.norm(dim=-1)
    ps = o + (d1 + d2).unsqueeze(-1) * d
    ax = torch.cross(o.expand_as(ps), ps, dim=-1)
    ax = ax / (torch.norm(ax, dim=-1, keepdim=True) + 1e-8)
    theta = torch.asin(pn * depth)
    ang = (torch.asin(pn) - theta).unsqueeze(-1)
    pnew = ps * torch.cos(ang) + torch.cross(ax.expand(*ang.shape[:-1], 3), ps.expand(*ang.shape[:-1], 3), dim=-1) * torch.sin(ang) \
        + ax * (ax * ps).sum(-1, keepdim=True) * (1. - torch.cos(ang))
    pnew = pnew / torch.norm(pnew, dim=-1, keepdim=True)
    depth_real = 1. / (depth + 1e-8) * torch.cos(theta) + d1
    return torch.cat([pnew, depth.unsqueeze(-1)], -1), depth_real


def _draw(rnd, key, shape):
    """torch.rand(shape), or the first rows of the caller-supplied tensor rnd[key] (shared with the product path's
    ``_randoms`` so that both implementations can be driven with identical random numbers)."""
    if rnd is not None and key in rnd:
        n = 1
        for s_ in shape:
            n *= s_
        return rnd[key].reshape(-1)[:n].reshape(shape)
    return torch.rand(shape)


def perturb_z(z, perturb, n, rnd=None, key=None):
    z = z.expand(n, z.shape[-1])
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * _draw(rnd, key, tuple(z.shape)))
    return z


def sample_pdf(bins, weights, n, det, rnd=None, key=None):
    w = weights + 1e-8
    cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = torch.linspace(0, 1, n).expand(cdf.shape[0], n).contiguous() if det else _draw(rnd, key, (cdf.shape[0], n)).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = (inds - 1).clamp_min(0), inds.clamp_max(cdf.shape[1] - 1)
    cb, ca = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bb, ba = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = ca - cb
    denom = torch.where(denom < 1e-8, torch.ones_like(denom), denom)
    return bb + (u - cb) / denom * (ba - bb)


def _eval(model, xyz, dirs, idx, chunk=32768, rnd=None, key=None):
    n, S = xyz.shape[:2]
    x = torch.cat([xyz.reshape(n * S, -1), dirs.repeat(1, S, 1).view(-1, 3), idx.repeat(1, S, 1).view(-1, 1)], 1)
    noise = _draw(rnd, key, (x.shape[0], 1)) if (model.training and rnd is not None and key in rnd) else None
    outs = []
    for i in range(0, x.shape[0], chunk):
        xc = x[i:i + chunk]
        nz = noise[i:i + chunk] if noise is not None else (torch.rand(len(xc), 1) if model.training else None)
        outs.append(model(xc, nz))
    return torch.cat(outs).view(n, S, 4)


def _composite(z, raw, last_delta, flip, depth_src=None):
    deltas = (z[:, :-1] - z[:, 1:]) if flip else (z[:, 1:] - z[:, :-1])
    deltas = torch.cat([deltas, last_delta], -1)
    alphas = 1 - torch.exp(-deltas * raw[..., 3])
    T = torch.cumprod(1 - alphas + 1e-8, -1)
    lam = T[:, -1]
    T = torch.cat([torch.ones_like(T[:, :1]), T[:, :-1]], -1)
    w = alphas * T
    rgb = (w.unsqueeze(-1) * raw[..., :3]).sum(1)
    with torch.no_grad():
        depth = (w * (depth_src if depth_src is not None else z)).sum(1)
        var = (w * (z - depth.unsqueeze(1)).square()).sum(-1)
    return w, rgb, depth, var, lam


def _branch(model, hp, o, d, idx, z, xyz, last_delta, flip, depth_real, points_fn, rnd=None, tag=''):
    perturb = hp.perturb if model.training else 0
    has = last_delta[:, 0] < 1e10
    diff = torch.zeros_like(last_delta)
    if has.any():
        diff[has, 0] = z[has].max(-1)[0]
    xc, zc = (xyz.flip(1), z.flip(1)) if flip else (xyz, z)
    raw_c = _eval(model, xc, d, idx, rnd=rnd, key=tag + '_noise_coarse')
    w, *_ = _composite(zc, raw_c, last_delta - diff, flip)
    nf = hp.fine_samples // 2 if flip else hp.fine_samples
    zf = sample_pdf(0.5 * (z[:, :-1] + z[:, 1:]), w[:, 1:-1].detach(), nf, perturb == 0, rnd, tag + '_u')
    xf, dr_f = points_fn(zf)
    diff = torch.zeros_like(last_delta)
    if has.any():
        diff[has, 0] = zf[has].max(-1)[0]
    raw_f = _eval(model, xf, d, idx, rnd=rnd, key=tag + '_noise_fine')
    zm, order = torch.sort(torch.cat([zf, zc], -1), -1, descending=flip)
    raw_m = torch.gather(torch.cat([raw_f, raw_c], 1), 1, order.unsqueeze(-1).expand(-1, -1, 4))
    dr_m = torch.gather(torch.cat([dr_f, depth_real], 1), 1, order) if depth_real is not None else None
    _, rgb, depth, var, lam = _composite(zm, raw_m, last_delta - diff, flip, dr_m)
    return rgb, depth, var, lam


def render_rays(nerf, bg_nerf, rays, image_indices, hp, sphere_center, sphere_radius, randoms=None) -> Dict[str, torch.Tensor]:
    """fg + bg render (rendering.py:15-139) returning rgb/depth/depth_variance/bg_lambda + fg/bg splits.
    ``randoms``: optional dict of pre-drawn uniforms replacing the torch.rand draws of the training mode (keys
    ``{fg,bg}_{perturb,noise_coarse,u,noise_fine}``: the first rows are used; background rows are the compacted rays)."""
    N = rays.shape[0]
    o, d = rays[:, None, 0:3], rays[:, None, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    idx = image_indices.float().view(N, 1, 1)
    perturb = hp.perturb if nerf.training else 0
    last_delta = 1e10 * torch.ones(N, 1)
    fg_far = torch.maximum(intersect_sphere(o[:, 0], d[:, 0], sphere_center, sphere_radius), near[:, 0])
    bgi = torch.nonzero(far[:, 0] > fg_far)[:, 0]
    bg = None
    if bgi.numel() > 0:
        last_delta[bgi, 0] = fg_far[bgi]
        far = torch.minimum(far[:, 0], fg_far).unsqueeze(-1)
        bz = perturb_z(torch.linspace(0, 1, hp.coarse_samples // 2), perturb, bgi.numel(), randoms, 'bg_perturb')
        ob, db = o[bgi], d[bgi]
        pts, dr = depth2pts_outside(ob, db, bz, sphere_center, sphere_radius)
        bg = _branch(bg_nerf, hp, ob, db, idx[bgi], bz, pts, 1e10 * torch.ones(bgi.numel(), 1), True, dr,
                     lambda zf: depth2pts_outside(ob, db, zf, sphere_center, sphere_radius), randoms, 'bg')
    t = torch.linspace(0, 1, hp.coarse_samples)
    z = perturb_z(near * (1 - t) + far * t, perturb, N, randoms, 'fg_perturb')
    rgb, depth, var, lam = _branch(nerf, hp, o, d, idx, z, o + d * z.unsqueeze(-1), last_delta, False, None,
                                   lambda zf: (o + d * zf.unsqueeze(-1), None), randoms, 'fg')
    res = {'fg_rgb_fine': rgb, 'fg_depth_fine': depth, 'depth_variance_fine': var, 'bg_lambda_fine': lam}
    bg_rgb, bg_depth = torch.zeros_like(rgb), torch.zeros_like(depth)
    if bg is not None:
        bg_rgb = bg_rgb.index_put((bgi,), bg[0] * lam[bgi].unsqueeze(-1))
        bg_depth = bg_depth.index_put((bgi,), bg[1] * lam[bgi])
    res.update({'bg_rgb_fine': bg_rgb, 'bg_depth_fine': bg_depth, 'rgb_fine': rgb + bg_rgb, 'depth_fine': depth + bg_depth})
    return res


def make_models(hp, fcfg, fw, bcfg, bw, appearance_count) -> Tuple[TorchNeRF, TorchNeRF]:
    out = []
    for cfg, w in ((fcfg, fw), (bcfg, bw)):
        m = TorchNeRF(cfg, appearance_count)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        out.append(m)
    return out[0], out[1]
