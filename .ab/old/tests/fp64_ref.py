"""fp64 torch restatements used as *checkers* by the GPU gradient tests (test infrastructure, never imported by the product).

Why ReLU masks are an input here: a ReLU unit whose pre-activation lies within an fp32 ulp of zero is "on" in one fp32
implementation and "off" in another (and in fp64).  The reference's own fp32 autograd gradients differ from an fp64 run of
the same reference by up to 7e-2 of a tensor's scale for exactly this reason (tests/golden/make_golden.py records both:
``grad_* / gsub_*`` and ``g64_*``).  To check the backward kernels -- rather than which side of zero a borderline unit fell --
the fp64 restatement can be told the masks the kernel under test actually used (read back from its activation tape).
Semantics: mega_nerf/models/nerf.py:115-160 of the reference.
"""
import numpy as np
import torch


def embedding64(v, L):
    out = [v]
    for k in range(L):
        out += [torch.sin(2.0 ** k * v), torch.cos(2.0 ** k * v)]
    return torch.cat(out, -1)


def nerf_forward64(w, cfg, x, noise=None, masks=None):
    """NeRF.forward in fp64.  ``masks``: None (plain ReLU) or dict(act=[bool [B, W]] * layers, dact=bool [B, W/2])."""
    inp = embedding64(x[:, :cfg.xyz_dim], cfg.pos_xyz_dim)
    h = inp
    for i in range(cfg.layers):
        if i in cfg.skip_layers:
            h = torch.cat([inp, h], -1)
        pre = h @ w['xyz_encodings.%d.0.weight' % i].T + w['xyz_encodings.%d.0.bias' % i]
        h = torch.relu(pre) if masks is None else pre * masks['act'][i]
    sig = h @ w['sigma.weight'].T + w['sigma.bias']
    if noise is not None:
        sig = sig + noise.view(-1, 1)
    sig = torch.nn.functional.softplus(sig - 1, 1, 20) if cfg.shifted_softplus else torch.relu(sig)
    f = h @ w['xyz_encoding_final.weight'].T + w['xyz_encoding_final.bias']
    idx = x[:, -1].long()
    d_in = torch.cat([f, embedding64(x[:, -4:-1], cfg.pos_dir_dim), w['embedding_a.weight'][idx]], -1)
    pre = d_in @ w['dir_a_encoding.0.weight'].T + w['dir_a_encoding.0.bias']
    d = torch.relu(pre) if masks is None else pre * masks['dact']
    rgb = torch.sigmoid(d @ w['rgb.weight'].T + w['rgb.bias'])
    return torch.cat([rgb, sig], -1)


def tape_masks(lib, model, desc, tape, cap, row0, n_rows):
    """ReLU masks of tape rows [row0, row0 + n_rows) of a fused-kernel activation tape (csrc/mlp_layout.h TapeLayout:
    plane l = post-ReLU output of trunk layer l at float offset l * W * cap; dir_a plane via mnr_tape_plane_offset)."""
    import ctypes as C
    W, L = model.layer_dim, model.layers
    assert int(lib.mnr_tape_plane_offset(C.byref(desc), 2)) == (L - 1) * W
    act = [(tape[l * W * cap:(l + 1) * W * cap].view(cap, W)[row0:row0 + n_rows] > 0).cpu() for l in range(L)]
    off = int(lib.mnr_tape_plane_offset(C.byref(desc), 0)) * cap
    dact = (tape[off:off + (W // 2) * cap].view(cap, W // 2)[row0:row0 + n_rows] > 0).cpu()
    return dict(act=act, dact=dact)


def autograd_grads64(w_np, cfg, x, noise, d_out, masks=None, chunk=16384):
    """Parameter gradients of sum(out * d_out) in fp64, rows processed in chunks (gradients accumulate)."""
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w_np.items()}
    B = x.shape[0]
    for r0 in range(0, B, chunk):
        sl = slice(r0, min(B, r0 + chunk))
        mk = None
        if masks is not None:
            mk = dict(act=[m[sl].double() for m in masks['act']], dact=masks['dact'][sl].double())
        nz = torch.tensor(noise[sl], dtype=torch.float64) if noise is not None else None
        out = nerf_forward64(wt, cfg, torch.tensor(x[sl], dtype=torch.float64), nz, mk)
        (out * torch.tensor(d_out[sl], dtype=torch.float64)).sum().backward()
    return {k: v.grad.numpy() for k, v in wt.items()}


def rel_to_scale(got, ref):
    sc = max(float(np.abs(ref).max()), 1e-30)
    return float(np.abs(np.asarray(got, np.float64) - ref).max()) / sc
