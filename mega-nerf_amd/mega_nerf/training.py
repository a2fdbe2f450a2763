"""Training path of render_rays on the MI355X: forward with an activation tape + hand-written backward.

The reference trains through torch autograd over ``rendering.render_rays`` (runner.py:347-381, 263-277).
Here the whole differentiable part of the render -- fg/bg blend, compositing, coarse/fine merge and the NeRF
MLP -- has explicit HIP backward kernels; :class:`RenderFunction` exposes them to autograd as ONE node whose
inputs are the model parameters and whose output is ``rgb_fine``, so ``loss.backward()`` / ``optimizer.step()``
in a reference-style training loop work unchanged.  :class:`TrainStep` is that loop body for the benchmark.

Two implementations share the stage kernels:
  * :class:`RenderFunction` -- the tuned path for the default configuration (single 8x256 NeRF per branch, no
    cascade, fine_samples > 0): one activation tape per branch, coarse + fine rows in one weight-gradient launch;
  * :class:`GeneralRenderFunction` -- every other configuration of the reference (``use_cascade``, any
    ``layer_dim``, ``appearance_dim 0``, spherical harmonics): one tape per MLP evaluation
    (``NeRF.train_eval``: fused kernels where they exist, else the layer-by-layer GEMM path), outputs
    ``rgb_fine`` and, with cascade, ``rgb_coarse``.
Importance-sampling weights are detached exactly as in the reference (rendering.py:215), so gradients reach the
MLP only through the compositing of ``rgb_*``.  ``--train_mega_nerf`` (all cells of a MegaNeRF trained in one process,
rows routed hard to the nearest centroid) goes through the general path with per-cell tapes
(``MegaNeRF.train_eval_routed``).
"""
from __future__ import annotations

import math

import ctypes as C
from argparse import Namespace
from typing import NamedTuple, Dict, Optional

import time

import torch
from torch import nn

from mega_nerf import _native as N
from mega_nerf import rendering as R


def _f(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)


def _timed(tag, fn):
    """Run ``fn`` between two HIP events on the launch stream when bench.py asked for kernel timings."""
    if R.KERNEL_EVENTS is None:
        return fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    R.KERNEL_EVENTS.append((tag, a, b))
    return out


class _Branch:
    """Everything one branch (fg or bg) keeps between forward and backward."""
    pass


def _multi_ok(*models) -> bool:
    """True if every model is one of the two default architectures the multi-segment launches cover."""
    import os
    if os.environ.get('MNR_NO_MULTI'):
        return False
    return all(m is None or getattr(m, 'is_default_arch', lambda: False)() for m in models)


def _launch_forward(branches, which: str) -> None:
    """One training-mode MLP pass (``which`` = 'c' coarse / 'f' fine) of every branch: one launch for all of them when the
    architectures allow it (the compacted background rows ride along with the foreground's), else one launch each."""
    lib = N.lib()
    if len(branches) > 1 and _multi_ok(*[b.model for b in branches]):
        segs = (N.MlpLaunch * len(branches))()
        keep = []
        for sg, b in zip(segs, branches):
            io, row0 = (b.io_c, 0) if which == 'c' else (b.io_f, b.rows_c)
            desc, packed = b.model.packed()
            keep.append((desc, packed, io))
            sg.packed_dev, sg.desc, sg.io = packed.data_ptr(), C.pointer(desc), C.pointer(io)
            sg.tape_dev, sg.tape_rows, sg.tape_row0 = b.tape.data_ptr(), b.cap, row0
        _timed('fwd_' + which, lambda: N.check(lib.mnr_mlp_forward_multi(segs, len(branches), N.stream_ptr())))
        return
    for b in branches:
        io, row0 = (b.io_c, 0) if which == 'c' else (b.io_f, b.rows_c)
        _timed('%s_%s' % (b.part.tag, 'coarse' if which == 'c' else 'fine'), lambda: b.model.evaluate_train(io, b.tape, b.cap, row0))


def _branch_begin(model, hparams: Namespace, part, flip: bool, rnd: dict, tag: str) -> _Branch:
    """Training-mode _get_results (rendering.py:176-248, non-cascade, Nf > 0), stage 1: buffers + the coarse MLP launch
    description (``b.io_c``).  The activation tape holds the coarse rows, then the fine rows."""
    b = _Branch()
    dev = part.z.device
    n, Sc = part.z.shape
    nf = hparams.fine_samples // 2 if flip else hparams.fine_samples
    b.n, b.Sc, b.Sf, b.flip, b.part, b.model, b.tag = n, Sc, nf, flip, part, model, tag
    b.rows_c, b.rows_f = n * Sc, n * nf
    b.cap = b.rows_c + b.rows_f
    b.tape = _f(b.cap * model.tape_floats_per_row(), device=dev)
    b.raw_all = _f(b.cap, 4, device=dev)                       # coarse rows, then fine rows
    xyz_c, z_c = part.xyz, part.z
    if flip:                                                   # rendering.py:271-273
        xyz_c, z_c = xyz_c.flip(1).contiguous(), z_c.flip(1).contiguous()
    b.z_c, b.xyz_c = z_c, xyz_c
    noise_c = rnd.get(tag + '_noise_coarse') if model.training else None
    if model.training and noise_c is None:
        noise_c = torch.rand(b.rows_c, device=dev)
    b.noise_c = noise_c
    b.io_c = model.mlp_io(xyz_c, xyz_c.shape[-1], part.dirs, part.dirs.stride(0), part.idx, 1, Sc, b.rows_c, b.raw_all[:b.rows_c],
                          noise_c, part.n_units, Sc)
    return b


def _branch_mid(b: _Branch, hparams: Namespace, rnd: dict) -> None:
    """Stage 2 (after the coarse MLP pass): importance sampling from the coarse weights, fine points, fine launch
    description (``b.io_f``)."""
    lib = N.lib()
    part, model, tag = b.part, b.model, b.tag
    dev = part.z.device
    n, Sc, nf = b.n, b.Sc, b.Sf
    nunits = part.n_units.data_ptr() if part.n_units is not None else None
    raw_c = b.raw_all[:b.rows_c].view(n, Sc, 4)
    comp = R._composite(b.z_c, raw_c, n, Sc, part, part.last_delta, part.z, b.flip, part.depth_real, {'weights'}, dev)
    det = (hparams.perturb if model.training else 0) == 0
    if det:
        u = R.linspace01(nf, dev)
    else:
        u = rnd.get(tag + '_u')
        if u is None:
            u = torch.rand(n, nf, device=dev)
    z_f = _f(n, nf, device=dev)
    N.check(lib.mnr_sample_fine(part.z.data_ptr(), comp['weights'].data_ptr(), n, nunits, Sc, nf, int(det), u.data_ptr(),
                                z_f.data_ptr(), None, N.stream_ptr()))
    b.z_f = z_f
    b.xyz_f, b.dr_f = part.points(z_f)
    noise_f = rnd.get(tag + '_noise_fine') if model.training else None
    if model.training and noise_f is None:
        noise_f = torch.rand(b.rows_f, device=dev)
    b.noise_f = noise_f
    b.io_f = model.mlp_io(b.xyz_f, b.xyz_f.shape[-1], part.dirs, part.dirs.stride(0), part.idx, 1, nf, b.rows_f,
                          b.raw_all[b.rows_c:], noise_f, part.n_units, nf)


def _branch_end(b: _Branch, get_bg_lambda: bool) -> None:
    """Stage 3 (after the fine MLP pass): coarse/fine merge + compositing of the merged samples -> ``b.out``."""
    lib = N.lib()
    part = b.part
    dev = part.z.device
    n, Sc, nf = b.n, b.Sc, b.Sf
    nunits = part.n_units.data_ptr() if part.n_units is not None else None
    raw_c, raw_f = b.raw_all[:b.rows_c].view(n, Sc, 4), b.raw_all[b.rows_c:].view(n, nf, 4)
    Sm = nf + Sc
    b.Sm = Sm
    b.z_m, b.raw_m = _f(n, Sm, device=dev), _f(n, Sm, 4, device=dev)
    dr_m = _f(n, Sm, device=dev) if b.dr_f is not None else None
    b.order = torch.empty(n, Sm, device=dev, dtype=torch.int32)
    N.check(lib.mnr_merge_sorted(b.z_f.data_ptr(), raw_f.data_ptr(), N.ptr(b.dr_f), nf, b.z_c.data_ptr(), raw_c.data_ptr(),
                                 N.ptr(part.depth_real), Sc, n, nunits, int(b.flip), b.z_m.data_ptr(), b.raw_m.data_ptr(),
                                 N.ptr(dr_m), b.order.data_ptr(), N.stream_ptr()))
    want = {'rgb', 'depth', 'depth_var'}
    if get_bg_lambda:
        want.add('bg_lambda')
    b.out = R._composite(b.z_m, b.raw_m, n, Sm, part, part.last_delta, b.z_f, b.flip, dr_m, want, dev)


def _branch_backward_begin(b: _Branch, d_rgb: torch.Tensor, d_lambda: Optional[torch.Tensor], grads: Dict[str, torch.Tensor]):
    """Adjoint of stage 3 (compositing + merge) and the launch descriptions of the branch's two data-gradient segments
    (coarse rows, fine rows) and of its weight-gradient region."""
    lib = N.lib()
    part, model = b.part, b.model
    dev = b.z_m.device
    n, Sc, Sf, Sm = b.n, b.Sc, b.Sf, b.Sm
    nunits = part.n_units.data_ptr() if part.n_units is not None else None
    rows_c, rows_f = b.rows_c, b.rows_f
    # compositing backward -> gradient of the merged raw outputs
    d_raw_m = _f(n, Sm, 4, device=dev)
    io = N.CompositeGradIO()
    io.z, io.raw = b.z_m.data_ptr(), b.raw_m.data_ptr()
    io.last_delta = part.last_delta.data_ptr() if part.last_delta is not None else None
    if part.last_delta is not None:
        io.zmax_src, io.zmax_S = b.z_f.data_ptr(), Sf
    io.flip, io.N, io.S = int(b.flip), n, Sm
    io.n_units_dev = nunits
    io.d_rgb = d_rgb.data_ptr()
    io.d_bg_lambda = d_lambda.data_ptr() if d_lambda is not None else None
    io.d_raw = d_raw_m.data_ptr()
    N.check(lib.mnr_composite_backward(C.byref(io), N.stream_ptr()))
    # un-merge: fine rows / coarse rows of one [cap][4] gradient array (same row space as the tape)
    d_raw_all = _f(b.cap, 4, device=dev)
    N.check(lib.mnr_merge_backward(d_raw_m.data_ptr(), b.order.data_ptr(), Sf, Sc, n, nunits,
                                   d_raw_all[rows_c:].data_ptr(), d_raw_all[:rows_c].data_ptr(), N.stream_ptr()))
    desc, packed = model.packed()
    packed_bwd = model.packed_bwd()
    gtape = _f(b.tape.numel(), device=dev)
    dheads = _f(b.cap, 4, device=dev)
    gs = model.grad_struct(grads)
    counter = torch.zeros(1, device=dev, dtype=torch.int32)

    def gio(row0, rows, rows_per_ray):
        g = N.MlpGradIO()
        g.tape, g.gtape, g.tape_rows, g.tape_row0 = b.tape.data_ptr(), gtape.data_ptr(), b.cap, row0
        g.d_out, g.d_out_stride = d_raw_all[row0:].data_ptr(), 4
        g.out, g.out_stride = b.raw_all[row0:].data_ptr(), 4
        g.dheads = dheads.data_ptr()
        if part.idx is not None:
            g.idx, g.idx_stride = part.idx.data_ptr(), 1
            g.idx_is_float = 1 if part.idx.dtype == torch.float32 else 0
        g.rows_per_ray = rows_per_ray
        g.n_rows = rows
        g.n_units_dev = nunits
        g.rows_per_unit = rows_per_ray
        g.work_counter = counter.data_ptr()
        g.grad = gs
        return g

    b.g_c, b.g_f = gio(0, rows_c, Sc), gio(rows_c, rows_f, Sf)
    b.g_all = gio(0, b.cap, Sc) if part.n_units is None else None
    b.bwd_keep = (desc, packed, packed_bwd, gtape, dheads, d_raw_all, d_raw_m, counter)
    b.aligned = not (Sc % 32 or Sf % 32)
    rg = N.WgradRegion()
    rg.desc = C.pointer(desc)
    rg.tape, rg.gtape, rg.tape_rows = b.tape.data_ptr(), gtape.data_ptr(), b.cap
    if part.n_units is None:
        rg.n_ranges = 1                                  # foreground: coarse + fine rows are one dense region of the tape
        rg.row0[0], rg.n_rows[0] = 0, b.cap
    else:
        rg.n_ranges = 2                                  # compacted background: device-side row counts per pass
        rg.row0[0], rg.n_rows[0], rg.n_units_dev[0], rg.rows_per_unit[0] = 0, rows_c, nunits, Sc
        rg.row0[1], rg.n_rows[1], rg.n_units_dev[1], rg.rows_per_unit[1] = rows_c, rows_f, nunits, Sf
    rg.grad = gs
    b.wgrad_region = rg


def _launch_backward(branches, dev: torch.device) -> None:
    """Data-gradient chains of every (branch, pass) segment in one launch, then the weight gradients of every branch in one
    launch (+ its reduction); per-segment / per-branch launches where the batched kernels do not apply."""
    lib = N.lib()
    if _multi_ok(*[b.model for b in branches]):
        segs = (N.MlpGradLaunch * (2 * len(branches)))()
        i = 0
        for b in branches:
            desc, packed, packed_bwd = b.bwd_keep[:3]
            for g in (b.g_c, b.g_f):
                segs[i].packed_fwd_dev, segs[i].packed_bwd_dev = packed.data_ptr(), packed_bwd.data_ptr()
                segs[i].desc, segs[i].io = C.pointer(desc), C.pointer(g)
                i += 1
        _timed('bwd', lambda: N.check(lib.mnr_mlp_backward_chain_multi(segs, i, N.stream_ptr())))
        _timed('head_grads', lambda: N.check(lib.mnr_mlp_head_grads_multi(segs, i, N.stream_ptr())))
    else:
        for b in branches:
            desc, packed, packed_bwd = b.bwd_keep[:3]
            for g, t in ((b.g_c, 'coarse'), (b.g_f, 'fine')):
                _timed('%s_bwd_%s' % (b.part.tag, t), lambda: N.check(lib.mnr_mlp_backward_data(
                    packed.data_ptr(), packed_bwd.data_ptr(), C.byref(desc), C.byref(g), N.stream_ptr())))
    regions = []
    batched = _multi_ok(*[b.model for b in branches])      # the batched weight-gradient launch covers the same two architectures
    for b in branches:
        if b.aligned and batched:
            regions.append(b.wgrad_region)
            continue
        # sample counts that are not multiples of the 32-row tile: per-branch launches with ragged-tile handling
        desc = b.bwd_keep[0]
        for g in ([b.g_all] if b.g_all is not None else [b.g_c, b.g_f]):
            N.check(lib.mnr_mlp_backward_weights(C.byref(desc), C.byref(g), N.stream_ptr()))
    _weight_gradients(regions, dev, 'wgrad')




def _wgrad_workspace(dev: torch.device) -> torch.Tensor:
    """Scratch of the batched weight-gradient launch (partial-sum slabs), allocated once per device."""
    return N.wgrad_workspace(dev)


def _weight_gradients(regions, dev: torch.device, tag: str = 'wgrad') -> None:
    """One launch (+ one reduction launch) for the weight gradients of every region (fg + bg of a training step)."""
    regions = [r for r in regions if r is not None]
    if not regions:
        return
    arr = (N.WgradRegion * len(regions))(*regions)
    ws = _wgrad_workspace(dev)
    _timed(tag, lambda: N.check(N.lib().mnr_mlp_backward_weights_multi(arr, len(regions), ws.data_ptr(), ws.numel(), N.stream_ptr())))


def _zero_grads(names, params) -> Dict[str, torch.Tensor]:
    """Gradient tensors of all parameters of one model as views of ONE zero-filled buffer (a single memset)."""
    if not params:
        return {}
    sizes = [(p.numel() + 3) // 4 * 4 for p in params]          # keep every view 16-byte aligned
    flat = torch.zeros(sum(sizes), device=params[0].device, dtype=torch.float32)
    out, o = {}, 0
    for k, p, n in zip(names, params, sizes):
        out[k] = flat[o:o + p.numel()].view(p.shape)
        o += n
    return out


def _param_list(m: Optional[nn.Module]):
    return [] if m is None else [(k, p) for k, p in m.named_parameters()]


def _release(ctx, *branches) -> None:
    """Drop every buffer a render kept for its backward pass the moment that pass has been enqueued.  The branch objects sit
    in reference cycles (closures of the sampling stages), so without this their tapes -- 2-3 GB per training step -- would
    live until Python's cyclic collector happens to run: measured 60-70 GB of reserved HBM and ~2 hipMalloc calls per step
    inside the benchmark's timed region.  (Backward through the same render twice is not supported, as in the reference's
    training loop, which never retains the graph.)"""
    for b in branches:
        if b is not None:
            b.__dict__.clear()
    ctx.fgb = ctx.bgb = ctx.bg_slot = ctx.params = None


class RenderFunction(torch.autograd.Function):
    """rgb_fine = render(params...) with hand-written HIP backward.  Non-differentiable by-products (depth
    variance, bg_lambda, device flags) are returned through ``ctx_out``."""

    @staticmethod
    def forward(ctx, nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, rnd, ctx_out, *params):
        N.require_device(rays, 'rays')
        lib = N.lib()
        dev = rays.device
        rays = rays.contiguous().float()
        n_rays = rays.shape[0]
        Nc = hparams.coarse_samples
        if image_indices is not None:
            if image_indices.dtype not in (torch.float32, torch.int32):
                image_indices = image_indices.float()
            image_indices = image_indices.contiguous()
        perturb = float(hparams.perturb) if nerf.training else 0.0
        far = last_delta = bg_slot = n_bg = err = None
        bgb = None
        if bg_nerf is not None:
            c, r = R._host_vec(sphere_center), R._host_vec(sphere_radius)
            far, last_delta = _f(n_rays, device=dev), _f(n_rays, device=dev)
            bg_list = torch.zeros(max(n_rays, 1), device=dev, dtype=torch.int32)
            bg_slot = torch.empty(max(n_rays, 1), device=dev, dtype=torch.int32)
            scal = torch.zeros(2, device=dev, dtype=torch.int32)
            n_bg, err = scal[0:1], scal[1:2]
            N.check(lib.mnr_ray_setup(rays.data_ptr(), n_rays, N.host3(c), N.host3(r), far.data_ptr(),
                                      last_delta.data_ptr(), bg_list.data_ptr(), bg_slot.data_ptr(), n_bg.data_ptr(),
                                      err.data_ptr(), N.stream_ptr()))
            bg_part = R._background_part(bg_nerf, nerf, hparams, rays, image_indices, bg_list, n_bg, n_rays, perturb,
                                         c, r, rnd, dev)
            bgb = _branch_begin(bg_nerf, hparams, bg_part, True, rnd, 'bg')
        t_c = R.linspace01(Nc, dev)
        prnd = None
        if perturb > 0:
            prnd = rnd.get('fg_perturb')
            if prnd is None:
                prnd = torch.rand(n_rays, Nc, device=dev)
        z, xyz = _f(n_rays, Nc, device=dev), _f(n_rays, Nc, 3, device=dev)
        N.check(lib.mnr_fg_samples(rays.data_ptr(), N.ptr(far), n_rays, Nc, t_c.data_ptr(), perturb, N.ptr(prnd),
                                   z.data_ptr(), xyz.data_ptr(), N.stream_ptr()))

        def fg_points(zf):
            p = _f(n_rays, zf.shape[1], 3, device=dev)
            N.check(lib.mnr_fg_points(rays.data_ptr(), n_rays, zf.shape[1], zf.data_ptr(), p.data_ptr(), N.stream_ptr()))
            return p, None

        fg_part = R._Part(z=z, xyz=xyz, depth_real=None, last_delta=last_delta, n_units=None, dirs=rays[:, 3:6],
                          idx=image_indices, points=fg_points, rays=rays, tag='fg')
        fgb = _branch_begin(nerf, hparams, fg_part, False, rnd, 'fg')
        # both branches advance pass by pass on ONE stream; each MLP pass is one launch over the rows of both
        # (a side stream for the background made its small launches compete with the foreground's for CUs: round 1)
        branches = [fgb] + ([bgb] if bgb is not None else [])
        _launch_forward(branches, 'c')
        for b in branches:
            _branch_mid(b, hparams, rnd)
        _launch_forward(branches, 'f')
        _branch_end(fgb, bg_nerf is not None)
        if bgb is not None:
            _branch_end(bgb, False)
        rgb = fgb.out['rgb']
        ctx.fg_rgb_unblended = None
        if bgb is not None:
            # blend in place (rendering.py:102-131); depth is not part of the training outputs
            N.check(lib.mnr_bg_blend(rgb.data_ptr(), None, fgb.out['bg_lambda'].data_ptr(), bg_slot.data_ptr(),
                                     bgb.out['rgb'].data_ptr(), None, n_rays, None, None, None, None, N.stream_ptr()))
        ctx.fgb, ctx.bgb, ctx.bg_slot, ctx.n_rays = fgb, bgb, bg_slot, n_rays
        ctx.names_fg = [k for k, _ in _param_list(nerf)]
        ctx.names_bg = [k for k, _ in _param_list(bg_nerf)]
        ctx.params = params
        ctx_out['depth_variance_fine'] = fgb.out['depth_var']
        if bg_nerf is not None:
            ctx_out['bg_lambda_fine'] = fgb.out['bg_lambda']
        ctx_out['n_bg'], ctx_out['err'] = n_bg, err
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        lib = N.lib()
        fgb, bgb = ctx.fgb, ctx.bgb
        if fgb is None:
            raise RuntimeError('backward through the same render_rays call twice: its tapes were released after the first pass')
        d_rgb = d_rgb.contiguous().float()
        dev = d_rgb.device
        n_fg, n_bgp = len(ctx.names_fg), len(ctx.names_bg)
        grads_fg = _zero_grads(ctx.names_fg, ctx.params[:n_fg])
        grads_bg = _zero_grads(ctx.names_bg, ctx.params[n_fg:n_fg + n_bgp])
        d_lambda = None
        if bgb is not None:
            d_lambda = _f(ctx.n_rays, device=dev)
            d_bg_rgb = torch.zeros(bgb.n, 3, device=dev)
            N.check(lib.mnr_bg_blend_backward(d_rgb.data_ptr(), fgb.out['bg_lambda'].data_ptr(), ctx.bg_slot.data_ptr(),
                                              bgb.out['rgb'].data_ptr(), ctx.n_rays, d_lambda.data_ptr(),
                                              d_bg_rgb.data_ptr(), N.stream_ptr()))
            _branch_backward_begin(bgb, d_bg_rgb, None, grads_bg)
        _branch_backward_begin(fgb, d_rgb, d_lambda, grads_fg)
        _launch_backward([fgb] + ([bgb] if bgb is not None else []), dev)
        out = [grads_fg[k] for k in ctx.names_fg] + [grads_bg[k] for k in ctx.names_bg]
        _release(ctx, fgb, bgb)
        return (None,) * 9 + tuple(out)


# =====================================================================================================================
# general path: cascade / generic widths / no appearance / spherical harmonics
# =====================================================================================================================

def _train_model_eval(nerf, typ: str, hparams: Namespace, xyz: torch.Tensor, part, S: int, noise):
    """Training twin of rendering._model_eval_inner: (raw [n, S, 4], tape, parameter-name prefix of the sub-model)."""
    from mega_nerf.models.cascade import Cascade
    from mega_nerf.models.mega_nerf import MegaNeRF
    model, prefix = nerf, ''
    if isinstance(model, Cascade):
        model, prefix = (model.coarse, 'coarse.') if typ == 'coarse' else (model.fine, 'fine.')
    n = xyz.shape[0]
    out = _f(n, S, 4, device=xyz.device)
    sh_deg = hparams.sh_deg if (hparams.pos_dir_dim == 0 and hparams.sh_deg is not None) else -1
    if isinstance(model, MegaNeRF):
        # --train_mega_nerf: every cell trained in this process, rows routed hard to their nearest centroid
        return out, model.train_eval_routed(xyz, part, S, out, noise, sh_deg), prefix
    if model.has_dir and model.embedding_a is None:
        # quirk Q8 (nerf.py:146): the encoded "direction" is [last xyz coordinate, d_x, d_y] of every sample
        d = part.dirs.view(n, 1, 3).expand(n, S, 3)
        q = torch.cat([xyz[..., -1:], d[..., :2]], -1).contiguous()
        tape = model.train_eval(xyz, xyz.shape[-1], q, 3, 1, None, 0, 1, n * S, out.view(-1, 4), noise, sh_deg, part.n_units, S)
        tape.keepalive = q
        return out, tape, prefix
    need_dir = model.has_dir or sh_deg >= 0
    dirs = part.dirs if need_dir else None
    dstride = part.dirs.stride(0) if need_dir else 0
    tape = model.train_eval(xyz, xyz.shape[-1], dirs, dstride, S, part.idx, 1, S, n * S, out.view(-1, 4), noise, sh_deg,
                            part.n_units, S, dirs if sh_deg >= 0 else None, dstride)
    return out, tape, prefix


def _general_branch_forward(nerf, hparams: Namespace, part, flip: bool, get_bg_lambda: bool, get_depth_variance: bool,
                            rnd: dict, tag: str) -> _Branch:
    """rendering._get_results (reference rendering.py:176-248) in training mode, recording what the adjoint needs.
    ``b.results`` holds rgb_/bg_lambda_/depth_variance_ per pass; ``b.stages[typ]`` the composite inputs + tapes."""
    lib = N.lib()
    b = _Branch()
    dev = part.z.device
    n, Sc = part.z.shape
    Nf, cascade = hparams.fine_samples, hparams.use_cascade
    b.n, b.flip, b.part, b.results, b.stages = n, flip, part, {}, {}
    nunits = part.n_units.data_ptr() if part.n_units is not None else None
    if Nf == 0 and not cascade:
        raise NotImplementedError('fine_samples == 0 without --use_cascade produces no rgb to train on (rendering.py:204)')

    xyz_c, z_c = part.xyz, part.z
    if flip:
        xyz_c, z_c = xyz_c.flip(1).contiguous(), z_c.flip(1).contiguous()
    noise_c = rnd.get(tag + '_noise_coarse') if nerf.training else None
    if nerf.training and noise_c is None:
        noise_c = torch.rand(n * Sc, device=dev)
    raw_c, tape_c, prefix_c = _train_model_eval(nerf, 'coarse', hparams, xyz_c, part, Sc, noise_c)
    want = set()
    if Nf > 0:
        want.add('weights')
    if cascade:
        want.add('rgb')
        if get_bg_lambda:
            want.add('bg_lambda')
    if Nf == 0 and get_depth_variance:
        want.update(('depth', 'depth_var'))
    comp = R._composite(z_c, raw_c, n, Sc, part, part.last_delta, part.z, flip, part.depth_real, want, dev)
    if cascade:
        b.results['rgb_coarse'] = comp['rgb']
        if get_bg_lambda:
            b.results['bg_lambda_coarse'] = comp['bg_lambda']
        b.stages['coarse'] = dict(z=z_c, raw=raw_c, S=Sc, zmax_src=part.z, merged=False, tapes=[(tape_c, prefix_c)])
    if Nf == 0:
        if get_depth_variance:
            b.results['depth_variance_coarse'] = comp['depth_var']
        return b

    nf = Nf // 2 if flip else Nf
    det = (hparams.perturb if nerf.training else 0) == 0
    if det:
        u = R.linspace01(nf, dev)
    else:
        u = rnd.get(tag + '_u')
        if u is None:
            u = torch.rand(n, nf, device=dev)
    z_f = _f(n, nf, device=dev)
    N.check(lib.mnr_sample_fine(part.z.data_ptr(), comp['weights'].data_ptr(), n, nunits, Sc, nf, int(det), u.data_ptr(),
                                z_f.data_ptr(), None, N.stream_ptr()))
    zmax_src = z_f
    if cascade:
        z_all = _f(n, Sc + nf, device=dev)
        N.check(lib.mnr_sort_rows(part.z.data_ptr(), Sc, z_f.data_ptr(), nf, n, nunits, z_all.data_ptr(), N.stream_ptr()))
        z_f, nf = z_all, Sc + nf
        zmax_src = z_f
    xyz_f, dr_f = part.points(z_f)
    if flip and cascade:
        xyz_f, z_f = xyz_f.flip(1).contiguous(), z_f.flip(1).contiguous()
    noise_f = rnd.get(tag + '_noise_fine') if nerf.training else None
    if nerf.training and noise_f is None:
        noise_f = torch.rand(n * nf, device=dev)
    raw_f, tape_f, prefix_f = _timed(tag + '_fine', lambda: _train_model_eval(nerf, 'fine', hparams, xyz_f, part, nf, noise_f))
    if cascade:
        z_m, raw_m, dr_m, Sm = z_f, raw_f, dr_f, nf
        stage = dict(z=z_m, raw=raw_m, S=Sm, zmax_src=zmax_src, merged=False, tapes=[(tape_f, prefix_f)])
    else:
        Sm = nf + Sc
        z_m, raw_m = _f(n, Sm, device=dev), _f(n, Sm, 4, device=dev)
        dr_m = _f(n, Sm, device=dev) if dr_f is not None else None
        order = torch.empty(n, Sm, device=dev, dtype=torch.int32)
        N.check(lib.mnr_merge_sorted(z_f.data_ptr(), raw_f.data_ptr(), N.ptr(dr_f), nf, z_c.data_ptr(), raw_c.data_ptr(),
                                     N.ptr(part.depth_real), Sc, n, nunits, int(flip), z_m.data_ptr(), raw_m.data_ptr(),
                                     N.ptr(dr_m), order.data_ptr(), N.stream_ptr()))
        stage = dict(z=z_m, raw=raw_m, S=Sm, zmax_src=zmax_src, merged=True, order=order, Sf=nf, Sc=Sc,
                     tapes=[(tape_f, prefix_f), (tape_c, prefix_c)])
    want = {'rgb'}
    if get_bg_lambda:
        want.add('bg_lambda')
    if get_depth_variance:
        want.update(('depth', 'depth_var'))
    comp = R._composite(z_m, raw_m, n, Sm, part, part.last_delta, zmax_src, flip, dr_m, want, dev)
    b.results['rgb_fine'] = comp['rgb']
    if get_bg_lambda:
        b.results['bg_lambda_fine'] = comp['bg_lambda']
    if get_depth_variance:
        b.results['depth_variance_fine'] = comp['depth_var']
    b.stages['fine'] = stage
    return b


def _sub_grads(grads: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return grads if not prefix else {k[len(prefix):]: v for k, v in grads.items() if k.startswith(prefix)}


def _general_branch_backward(b: _Branch, typ: str, d_rgb: torch.Tensor, d_lambda: Optional[torch.Tensor],
                             grads: Dict[str, torch.Tensor]) -> None:
    """Adjoint of one pass ('coarse' / 'fine') of a branch: d rgb_typ (+ d bg_lambda_typ) -> parameter gradients."""
    lib = N.lib()
    st, part = b.stages[typ], b.part
    dev = d_rgb.device
    n, S = b.n, st['S']
    nunits = part.n_units.data_ptr() if part.n_units is not None else None
    d_raw = _f(n, S, 4, device=dev)
    io = N.CompositeGradIO()
    io.z, io.raw = st['z'].data_ptr(), st['raw'].data_ptr()
    io.last_delta = part.last_delta.data_ptr() if part.last_delta is not None else None
    if part.last_delta is not None:
        io.zmax_src, io.zmax_S = st['zmax_src'].data_ptr(), st['zmax_src'].shape[1]
    io.flip, io.N, io.S = int(b.flip), n, S
    io.n_units_dev = nunits
    io.d_rgb = d_rgb.data_ptr()
    io.d_bg_lambda = d_lambda.data_ptr() if d_lambda is not None else None
    io.d_raw = d_raw.data_ptr()
    N.check(lib.mnr_composite_backward(C.byref(io), N.stream_ptr()))
    if st['merged']:
        Sf, Sc = st['Sf'], st['Sc']
        d_f, d_c = _f(n * Sf, 4, device=dev), _f(n * Sc, 4, device=dev)
        N.check(lib.mnr_merge_backward(d_raw.data_ptr(), st['order'].data_ptr(), Sf, Sc, n, nunits, d_f.data_ptr(), d_c.data_ptr(),
                                       N.stream_ptr()))
        pieces = [d_f, d_c]
    else:
        pieces = [d_raw.view(-1, 4)]
    for (tape, prefix), d in zip(st['tapes'], pieces):
        _timed('%s_bwd_%s' % (part.tag, typ), lambda: tape.backward(d, 4, _sub_grads(grads, prefix)))


class GeneralRenderFunction(torch.autograd.Function):
    """(rgb_fine?, rgb_coarse?) = render(params...) for every configuration outside the tuned default path."""

    @staticmethod
    def forward(ctx, nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, rnd, ctx_out, *params):
        N.require_device(rays, 'rays')
        lib = N.lib()
        dev = rays.device
        rays = rays.contiguous().float()
        n_rays = rays.shape[0]
        Nc = hparams.coarse_samples
        if image_indices is not None:
            if image_indices.dtype not in (torch.float32, torch.int32):
                image_indices = image_indices.float()
            image_indices = image_indices.contiguous()
        perturb = float(hparams.perturb) if nerf.training else 0.0
        far = last_delta = bg_slot = n_bg = err = None
        bgb = None
        get_var = ctx_out.pop('_get_depth_variance', True)
        if bg_nerf is not None:
            c, r = R._host_vec(sphere_center), R._host_vec(sphere_radius)
            far, last_delta = _f(n_rays, device=dev), _f(n_rays, device=dev)
            bg_list = torch.zeros(max(n_rays, 1), device=dev, dtype=torch.int32)
            bg_slot = torch.empty(max(n_rays, 1), device=dev, dtype=torch.int32)
            scal = torch.zeros(2, device=dev, dtype=torch.int32)
            n_bg, err = scal[0:1], scal[1:2]
            N.check(lib.mnr_ray_setup(rays.data_ptr(), n_rays, N.host3(c), N.host3(r), far.data_ptr(),
                                      last_delta.data_ptr(), bg_list.data_ptr(), bg_slot.data_ptr(), n_bg.data_ptr(),
                                      err.data_ptr(), N.stream_ptr()))
            bg_part = R._background_part(bg_nerf, nerf, hparams, rays, image_indices, bg_list, n_bg, n_rays, perturb, c, r, rnd, dev)
            bgb = _general_branch_forward(bg_nerf, hparams, bg_part, True, False, False, rnd, 'bg')
        t_c = R.linspace01(Nc, dev)
        prnd = None
        if perturb > 0:
            prnd = rnd.get('fg_perturb')
            if prnd is None:
                prnd = torch.rand(n_rays, Nc, device=dev)
        z, xyz = _f(n_rays, Nc, device=dev), _f(n_rays, Nc, 3, device=dev)
        N.check(lib.mnr_fg_samples(rays.data_ptr(), N.ptr(far), n_rays, Nc, t_c.data_ptr(), perturb, N.ptr(prnd),
                                   z.data_ptr(), xyz.data_ptr(), N.stream_ptr()))

        def fg_points(zf):
            p = _f(n_rays, zf.shape[1], 3, device=dev)
            N.check(lib.mnr_fg_points(rays.data_ptr(), n_rays, zf.shape[1], zf.data_ptr(), p.data_ptr(), N.stream_ptr()))
            return p, None

        fg_part = R._Part(z=z, xyz=xyz, depth_real=None, last_delta=last_delta, n_units=None, dirs=rays[:, 3:6],
                          idx=image_indices, points=fg_points, rays=rays, tag='fg')
        fgb = _general_branch_forward(nerf, hparams, fg_part, False, bg_nerf is not None, get_var, rnd, 'fg')
        types = [t for t in ('fine', 'coarse') if 'rgb_' + t in fgb.results]
        if bgb is not None:
            for typ in types:
                # blend in place (rendering.py:102-131); the composite adjoint recomputes what it needs from raw
                N.check(lib.mnr_bg_blend(fgb.results['rgb_' + typ].data_ptr(), None, fgb.results['bg_lambda_' + typ].data_ptr(),
                                         bg_slot.data_ptr(), bgb.results['rgb_' + typ].data_ptr(), None, n_rays, None, None, None,
                                         None, N.stream_ptr()))
        ctx.fgb, ctx.bgb, ctx.bg_slot, ctx.n_rays, ctx.types = fgb, bgb, bg_slot, n_rays, types
        ctx.names_fg = [k for k, _ in _param_list(nerf)]
        ctx.names_bg = [k for k, _ in _param_list(bg_nerf)]
        ctx.params = params
        for k, v in fgb.results.items():
            if not k.startswith('rgb_'):
                ctx_out[k] = v
        ctx_out['n_bg'], ctx_out['err'], ctx_out['types'] = n_bg, err, types
        return tuple(fgb.results['rgb_' + t] for t in types)

    @staticmethod
    def backward(ctx, *d_rgbs):
        lib = N.lib()
        fgb, bgb = ctx.fgb, ctx.bgb
        if fgb is None:
            raise RuntimeError('backward through the same render_rays call twice: its tapes were released after the first pass')
        n_fg, n_bgp = len(ctx.names_fg), len(ctx.names_bg)
        grads_fg = _zero_grads(ctx.names_fg, ctx.params[:n_fg])
        grads_bg = _zero_grads(ctx.names_bg, ctx.params[n_fg:n_fg + n_bgp])
        for typ, d_rgb in zip(ctx.types, d_rgbs):
            if d_rgb is None:
                continue
            d_rgb = d_rgb.contiguous().float()
            dev = d_rgb.device
            d_lambda = None
            if bgb is not None:
                d_lambda = _f(ctx.n_rays, device=dev)
                d_bg_rgb = torch.zeros(bgb.n, 3, device=dev)
                N.check(lib.mnr_bg_blend_backward(d_rgb.data_ptr(), fgb.results['bg_lambda_' + typ].data_ptr(), ctx.bg_slot.data_ptr(),
                                                  bgb.results['rgb_' + typ].data_ptr(), ctx.n_rays, d_lambda.data_ptr(),
                                                  d_bg_rgb.data_ptr(), N.stream_ptr()))
                _general_branch_backward(bgb, typ, d_bg_rgb, None, grads_bg)
            _general_branch_backward(fgb, typ, d_rgb, d_lambda, grads_fg)
        out = [grads_fg[k] for k in ctx.names_fg] + [grads_bg[k] for k in ctx.names_bg]
        _release(ctx, fgb, bgb)
        return (None,) * 9 + tuple(out)


def _fast_path_ok(nerf, bg_nerf, hparams) -> bool:
    from mega_nerf.models.nerf import NeRF
    if hparams.use_cascade or hparams.fine_samples == 0 or (hparams.sh_deg is not None and hparams.pos_dir_dim == 0):
        return False
    for m in (nerf, bg_nerf):
        if m is not None and not (isinstance(m, NeRF) and m.fused_train_supported()):
            return False
        if m is not None and m.has_dir and m.embedding_a is None:
            return False         # quirk Q8 needs per-sample "directions": handled by the general path
    return True


FORCE_GENERAL = False       # tests: run the default configuration through GeneralRenderFunction as well


def render_rays_train(nerf: nn.Module, bg_nerf: Optional[nn.Module], rays: torch.Tensor,
                      image_indices: Optional[torch.Tensor], hparams: Namespace, sphere_center, sphere_radius,
                      get_depth: bool, get_depth_variance: bool, get_bg_fg_rgb: bool, _randoms: Optional[dict] = None):
    """Differentiable render (training flags of runner.py:349-358).  Returns (results, n_bg_dev, err_dev)."""
    if hparams.container_path is not None:
        raise NotImplementedError('merged containers are inference-only (rendering.py never trains one either: '
                                  'model_utils.py:22-29 loads them as frozen TorchScript)')
    if get_depth or get_bg_fg_rgb:
        raise NotImplementedError('training render returns rgb / depth_variance / bg_lambda only')
    if rays.shape[0] == 0:
        # an empty batch renders to empty results (the inference path does the same); the rgb tensor hangs off the
        # parameters so that loss.backward() still works and leaves zero gradients
        res = R._empty_results(hparams, bg_nerf is not None, False, get_depth_variance, False, rays.device)
        anchor = sum(p.sum() for p in list(nerf.parameters())[:1]) * 0
        for k in list(res):
            if k.startswith('rgb_'):
                res[k] = res[k] + anchor
        return res, None, None
    rnd = _randoms if _randoms is not None else {}
    params = [p for _, p in _param_list(nerf)] + [p for _, p in _param_list(bg_nerf)]
    aux: Dict[str, torch.Tensor] = {}
    if _fast_path_ok(nerf, bg_nerf, hparams) and not FORCE_GENERAL:
        rgb = RenderFunction.apply(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, rnd, aux, *params)
        results = {'rgb_fine': rgb}
        if get_depth_variance:
            results['depth_variance_fine'] = aux['depth_variance_fine']
        if bg_nerf is not None:
            results['bg_lambda_fine'] = aux['bg_lambda_fine']
        return results, aux.get('n_bg'), aux.get('err')
    aux['_get_depth_variance'] = get_depth_variance
    rgbs = GeneralRenderFunction.apply(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, rnd, aux, *params)
    results = {'rgb_' + t: v for t, v in zip(aux['types'], rgbs)}
    for k, v in aux.items():
        if k.startswith(('depth_variance_', 'bg_lambda_')):
            results[k] = v
    return results, aux.get('n_bg'), aux.get('err')


def fused_step_supported(nerf, bg_nerf, hparams, n_rays: int, split_precision: bool = False) -> bool:
    """True if mnr_train_step (csrc/step.hip) covers this configuration: the default foreground / background architectures or their
    spherical-harmonics form (configs/mega-nerf-sh-3: sh_deg 2, pos_dir_dim 0; also sh_deg 3; fp32 kernels only), no cascade, 64 + 128 or 256 + 512
    samples per ray, background rows of a batch filling whole 64-row tiles."""
    import os
    from mega_nerf.models.nerf import NeRF
    if os.environ.get('MNR_NO_FUSED_STEP') or bg_nerf is None or hparams.use_cascade or hparams.fine_samples == 0:
        return False
    if not (isinstance(nerf, NeRF) and isinstance(bg_nerf, NeRF)):
        return False
    sh = hparams.sh_deg is not None and hparams.pos_dir_dim == 0
    if sh:
        if hparams.sh_deg not in (2, 3) or split_precision or not (nerf.is_sh_arch(hparams.sh_deg) and bg_nerf.is_sh_arch(hparams.sh_deg)):
            return False
    elif hparams.sh_deg is not None:
        return False
    elif nerf.is_wide_default_arch():
        # the Building shape (README "Larger models": 512-wide foreground, 256-wide background): forward on the wavefront-pair kernel, backward
        # as tiled GEMMs + weight-gradient jobs sequenced inside the step (csrc/step.hip wide_fg_backward); fp32 kernels only
        if split_precision or os.environ.get('MNR_NO_PAIR_KERNEL') or os.environ.get('MNR_NO_WIDE_FUSED_STEP'):
            return False
        if not (bg_nerf.is_default_arch() and bg_nerf.fused_train_supported() and nerf.affine is None and nerf.embedding_a is not None):
            return False
    elif not (_fast_path_ok(nerf, bg_nerf, hparams) and nerf.is_default_arch() and bg_nerf.is_default_arch()):
        return False
    if nerf.xyz_dim != 3 or bg_nerf.xyz_dim != 4 or hparams.container_path is not None or hparams.train_mega_nerf is not None:
        return False
    if (hparams.coarse_samples, hparams.fine_samples) not in ((64, 128), (256, 512)):
        return False
    return (n_rays * (hparams.coarse_samples // 2)) % 64 == 0 and (n_rays * (hparams.fine_samples // 2)) % 64 == 0


class GatheredBatch(NamedTuple):
    """A batch given as rows ``select`` of a device-resident training set (``mnr_step_batch::select``): the step's first kernel does
    the gather that MemoryDataset.__getitem__ / a DataLoader's collation would do.  ``rays`` [P, 8] fp32, ``img_indices`` [P] int32 or
    fp32, ``rgbs_u8`` [P, 3] uint8, ``select`` [n_rays] int64, ``u8_table`` [256] fp32 (the CPU's ``i / 255.`` values)."""
    rays: torch.Tensor
    img_indices: torch.Tensor
    rgbs_u8: torch.Tensor
    select: torch.Tensor
    u8_table: torch.Tensor


class FusedTrainStep:
    """The reference trainer's iteration (runner.py:244-277: render_rays with the training flags, mse_loss, backward, Adam on the
    foreground and the background model, ExponentialLR) for ONE OR SEVERAL independent cells a rank owns, as one call of
    ``mnr_train_step``: 12 kernel launches + one memset (+ 2 per further cell), no torch kernels, no host synchronisation.  Every cell keeps its own models,
    optimiser moments and batch (parscripts/run_8.txt: one trainer per cell).  After a call ``param.grad`` of every model
    parameter is that step's gradient (a view into the step's workspace).

    Optimiser state (all device resident, per cell): ``adam_m`` / ``adam_v`` = torch.optim.Adam's exp_avg / exp_avg_sq of every
    parameter in ``named_parameters()`` order (fg, then bg; each tensor padded to 16 bytes), ``adam_t[cell, 0 | 1]`` = the number
    of updates the cell's fg | bg optimiser has applied (torch's per-parameter ``step``).  A background model is updated only by
    batches that had rays with a background segment -- the rule of runner.py:268-272 -- decided on the device.
    ``state_dict()`` / ``load_state_dict()`` speak the reference's checkpoint layout (runner.py:519-538 ``optimizers``)."""

    def __init__(self, cells, hparams: Namespace, sphere_center, sphere_radius, n_rays: int, lr: float = 5e-4,
                 lr_decay_factor: float = 0.1, train_iterations: int = 500000, seed: Optional[int] = None, split_precision: bool = False,
                 state=None, rng_cells: Optional[list] = None):
        lib = N.lib()
        self.cells = [(f, b) for f, b in cells]
        # random streams of cell i: keyed seed + rng_cells[i] instead of seed + i (mnr_step_batch::rng_cell_plus1) -- JointCells passes
        # zeros, so that every cell of a shared plan draws what it would draw in a plan of its own
        self.rng_cells = None if rng_cells is None else [int(v) for v in rng_cells]
        assert self.rng_cells is None or len(self.rng_cells) == len(self.cells)
        assert 1 <= len(self.cells) <= N.MNR_STEP_MAX_CELLS
        f0, b0 = self.cells[0]
        dev = next(f0.parameters()).device
        self.dev, self.hparams, self.n_rays = dev, hparams, n_rays
        # ExponentialLR in its chained form (torch/optim/lr_scheduler.py: lr <- lr * gamma after every step), in Python doubles
        self.lr0, self.gamma = lr, lr_decay_factor ** (1 / train_iterations)
        self.lr = float(lr)
        self.step_count = 0
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xffffffffffffffff
        c, r = R._host_vec(sphere_center), R._host_vec(sphere_radius)
        Nc, Nf = hparams.coarse_samples, hparams.fine_samples
        self._tables = [R.linspace01(n, torch.device('cpu')).numpy().astype('float32').copy() for n in (Nc, Nc // 2, Nf, Nf // 2)]
        cfg = N.StepCfg()
        cfg.n_cells, cfg.n_rays, cfg.coarse_samples, cfg.fine_samples = len(self.cells), n_rays, Nc, Nf
        cfg.perturb = float(hparams.perturb) if f0.training else 0.0
        cfg.sigma_noise = 1 if f0.training else 0
        for i in range(3):
            cfg.sphere_center[i], cfg.sphere_radius[i] = c[i], r[i]
        cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps = 0.9, 0.999, 1e-8
        cfg.t_coarse, cfg.t_bg_coarse, cfg.t_fine, cfg.t_bg_fine = [t.ctypes.data_as(N.c_float_p) for t in self._tables]
        # opt-in: tape-writing forward + data-gradient chain on the 16-bit matrix pipe with split-precision operands
        cfg.split_precision = 1 if split_precision else 0
        self.split_precision = bool(split_precision)
        # gradient area of a cell: every parameter of its fg model, then of its bg model, each padded to 16 bytes
        def sizes(m):
            return [(k, p, (p.numel() + 3) // 4 * 4) for k, p in m.named_parameters()]
        per_cell = sum(n for m in (f0, b0) for _, _, n in sizes(m))
        cfg.grad_floats_per_cell = per_cell
        lay = N.StepLayout()
        N.check(lib.mnr_step_query(C.byref(cfg), C.byref(f0.model_desc()), C.byref(b0.model_desc()), C.byref(lay)))
        self.workspace = torch.empty(lay.workspace_bytes, dtype=torch.uint8, device=dev)
        self.layout = lay
        wsf = self.workspace.view(torch.float32)
        if state is None:
            self.adam_m = torch.zeros(len(self.cells), per_cell, device=dev)
            self.adam_v = torch.zeros(len(self.cells), per_cell, device=dev)
            self.adam_t = torch.zeros(len(self.cells), 2, device=dev, dtype=torch.int32)
        else:       # a second plan (another batch size) of the same cells shares the first one's optimiser state
            self.adam_m, self.adam_v, self.adam_t = state
            assert self.adam_m.shape == (len(self.cells), per_cell) and self.adam_t.shape == (len(self.cells), 2)
        self._packed = []
        models = (N.StepModel * (2 * len(self.cells)))()
        self.grad_views, self.m_views, self.v_views = [], [], []
        for ci, (f, b) in enumerate(self.cells):
            g0 = (lay.grad_offset + ci * lay.grad_stride) // 4
            o = 0
            for k, m in enumerate((f, b)):
                N.require_device(next(m.parameters()), 'NeRF parameter')
                sm = models[2 * ci + k]
                sm.desc = m.model_desc()
                views = [{}, {}, {}]
                for name, p, n in sizes(m):
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise N.NativeError('NeRF parameters must be contiguous float32')
                    views[0][name] = wsf[g0 + o:g0 + o + p.numel()].view(p.shape)
                    views[1][name] = self.adam_m[ci, o:o + p.numel()].view(p.shape)
                    views[2][name] = self.adam_v[ci, o:o + p.numel()].view(p.shape)
                    o += n
                sm.grad, sm.adam_m, sm.adam_v = m.grad_struct(views[0]), m.grad_struct(views[1]), m.grad_struct(views[2])
                sm.adam_steps_dev = self.adam_t[ci, k:k + 1].data_ptr()
                if split_precision:
                    pk = torch.empty(lib.mnr_packed_model_h2_bytes(C.byref(sm.desc)), dtype=torch.uint8, device=dev)
                    pb = torch.empty(lib.mnr_packed_bwd_h2_bytes(C.byref(sm.desc)), dtype=torch.uint8, device=dev)
                    sm.packed_h2_dev, sm.packed_bwd_h2_dev = pk.data_ptr(), pb.data_ptr()
                elif k == 0 and m.layer_dim == 512:
                    # 512-wide foreground: no transposed image (its data gradients are tiled GEMMs over the nn.Linear weights themselves)
                    pk, pb = torch.empty(lib.mnr_packed_model_bytes(C.byref(sm.desc)), dtype=torch.uint8, device=dev), None
                    sm.packed_dev = pk.data_ptr()
                else:
                    pk = torch.empty(lib.mnr_packed_model_bytes(C.byref(sm.desc)), dtype=torch.uint8, device=dev)
                    pb = torch.empty(lib.mnr_packed_bwd_bytes(C.byref(sm.desc)), dtype=torch.uint8, device=dev)
                    sm.packed_dev, sm.packed_bwd_dev = pk.data_ptr(), pb.data_ptr()
                self._packed.append((pk, pb))
                self.grad_views.append(views[0])
                self.m_views.append(views[1])
                self.v_views.append(views[2])
        self._models = models
        plan = C.c_void_p()
        N.check(lib.mnr_step_create(C.byref(plan), C.byref(cfg), models, self.workspace.data_ptr(), self.workspace.numel(), N.stream_ptr()))
        self._plan = plan
        nc = len(self.cells)
        self.loss = wsf[lay.loss_offset // 4:lay.loss_offset // 4 + nc]
        self.rgb = wsf[lay.rgb_offset // 4:lay.rgb_offset // 4 + nc * n_rays * 3].view(nc, n_rays, 3)
        self.depth_variance = wsf[lay.depth_var_offset // 4:lay.depth_var_offset // 4 + nc * n_rays].view(nc, n_rays)
        self.bg_lambda = wsf[lay.bg_lambda_offset // 4:lay.bg_lambda_offset // 4 + nc * n_rays].view(nc, n_rays)
        wsi = self.workspace.view(torch.int32)
        self.n_bg = wsi[lay.n_bg_offset // 4:lay.n_bg_offset // 4 + nc]
        self.err = wsi[lay.err_offset // 4:lay.err_offset // 4 + nc]
        # MNR_STEP_STICKY_* bits OR-ed over every optimising step since the plan was made (or since health() last cleared them)
        self.sticky = wsi[lay.sticky_offset // 4:lay.sticky_offset // 4 + nc]

    def __del__(self):
        plan = getattr(self, '_plan', None)
        if plan is not None and plan.value:
            try:
                N.lib().mnr_step_destroy(plan)
            except Exception:          # interpreter shutdown: the module globals may be gone already
                pass
            self._plan = None

    def profile(self, n_slots: int) -> None:
        """Record HIP events around every kernel group of the next steps (slot = step index mod n_slots); 0 = off."""
        N.check(N.lib().mnr_step_profile(self._plan, n_slots))

    def kernel_times(self, slot: int) -> dict:
        """{span name: ms} of a finished profiled step (synchronise first)."""
        ms = (C.c_float * N.MNR_STEP_SPANS)()
        N.check(N.lib().mnr_step_kernel_times(self._plan, slot, ms))
        return dict(zip(N.STEP_SPAN_NAMES, [float(v) for v in ms]))

    def repack(self) -> None:
        """After the parameters were changed from outside (checkpoint load): refresh the step's weight images."""
        N.check(N.lib().mnr_step_repack(self._plan, N.stream_ptr()))

    def health(self, clear: bool = True) -> None:
        """The checks the reference makes on every iteration (runner.py:260-261 'Train metrics not finite', rendering.py:412-414
        cameras outside the sphere), made for all optimising steps since the last call: ONE host synchronisation per call."""
        bits = self.sticky.cpu()
        if clear:
            self.sticky.zero_()
        if int((bits & N.MNR_STEP_STICKY_OUTSIDE).max()) != 0:
            raise Exception(R._ERR_TEXT)
        if int((bits & N.MNR_STEP_STICKY_NONFINITE).max()) != 0:
            raise Exception('Train metrics not finite: {}'.format({'loss': self.loss.tolist()}))

    # ---- optimiser state in torch.optim.Adam's terms ---------------------------------------------------------------------------
    def _torch_adam(self, ci: int, k: int, clone: bool) -> torch.optim.Adam:
        """A torch.optim.Adam over the parameters of model k (0 = fg, 1 = bg) of cell ci whose state IS this plan's state
        (views of the moment buffers, or copies when ``clone``)."""
        m = self.cells[ci][k]
        opt = torch.optim.Adam(m.parameters(), lr=self.lr0)
        group = opt.param_groups[0]
        group['initial_lr'], group['lr'] = self.lr0, self.lr
        t = float(self.adam_t[ci, k].item())
        mv, vv = self.m_views[2 * ci + k], self.v_views[2 * ci + k]
        for name, p in m.named_parameters():
            a, b = mv[name], vv[name]
            opt.state[p] = {'step': torch.tensor(t, dtype=torch.float32), 'exp_avg': a.clone() if clone else a,
                            'exp_avg_sq': b.clone() if clone else b}
        return opt

    def state_dict(self, cell: int = 0) -> Dict[str, dict]:
        """{'nerf': ..., 'bg_nerf': ...}: what ``{k: v.state_dict() for k, v in optimizers.items()}`` of the reference's trainer
        holds for this cell (runner.py:523) -- loadable by ``torch.optim.Adam.load_state_dict`` there and here."""
        return {key: self._torch_adam(cell, k, True).state_dict() for k, key in enumerate(('nerf', 'bg_nerf'))}

    def load_state_dict(self, sd: Dict[str, dict], cell: int = 0) -> None:
        """Inverse of :meth:`state_dict` (also accepts the ``optimizers`` entry of a checkpoint written by the reference)."""
        for k, key in enumerate(('nerf', 'bg_nerf')):
            opt = self._torch_adam(cell, k, True)
            merged = opt.state_dict()
            merged.update(sd[key])                            # the reference's update-then-load (runner.py:181-184)
            opt.load_state_dict(merged)
            self.adopt(opt, cell, k)

    def adopt(self, opt: torch.optim.Optimizer, cell: int, k: int) -> None:
        """Copy a torch.optim.Adam's state (moments, step count, current lr) into the plan's buffers."""
        m = self.cells[cell][k]
        mv, vv = self.m_views[2 * cell + k], self.v_views[2 * cell + k]
        steps = set()
        for name, p in m.named_parameters():
            st = opt.state.get(p)
            if not st:                                        # never stepped
                mv[name].zero_(), vv[name].zero_()
                steps.add(0)
                continue
            if st['exp_avg'].data_ptr() != mv[name].data_ptr():
                mv[name].copy_(st['exp_avg'])
                vv[name].copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) != 1:
            raise N.NativeError('torch.optim.Adam state with different step counts per parameter: {}'.format(sorted(steps)))
        self.adam_t[cell, k] = steps.pop()
        if k == 0:
            self.lr = float(opt.param_groups[0]['lr'])

    def __call__(self, batches, _randoms=None, optimize: bool = True, lr: Optional[float] = None):
        """batches: one (rays [n_rays, 8], image_indices [n_rays], rgbs [n_rays, 3]) per cell.  Returns (loss [cells], n_bg
        [cells], err [cells]) as device tensors (views of the workspace: valid until the next call).  ``lr``: this step's learning
        rate when the caller runs the schedule (Runner: torch's ExponentialLR object); default: the plan's own chained decay."""
        nc = len(self.cells)
        assert len(batches) == nc
        arr = (N.StepBatch * nc)()
        keep = []
        for i, batch in enumerate(batches):
            if self.rng_cells is not None:
                arr[i].rng_cell_plus1 = self.rng_cells[i] + 1
            if isinstance(batch, GatheredBatch):
                N.require_device(batch.rays, 'rays')
                ok = (batch.rays.dtype == torch.float32 and batch.rays.is_contiguous() and batch.rays.shape[1:] == (8,) and
                      batch.rgbs_u8.dtype == torch.uint8 and batch.rgbs_u8.is_contiguous() and batch.rgbs_u8.shape == (batch.rays.shape[0], 3) and
                      batch.img_indices.dtype in (torch.float32, torch.int32) and batch.img_indices.is_contiguous() and
                      batch.img_indices.numel() == batch.rays.shape[0] and batch.select.dtype == torch.int64 and batch.select.is_contiguous() and
                      batch.u8_table.dtype == torch.float32 and batch.u8_table.numel() == 256)
                if not ok or batch.select.numel() != self.n_rays:
                    raise N.NativeError('GatheredBatch: contiguous fp32 rays [P, 8], uint8 colours [P, 3], int32 / fp32 indices [P], int64 select '
                                        '[{}] expected'.format(self.n_rays))
                keep.append(batch)
                arr[i].rays, arr[i].idx, arr[i].target_u8 = batch.rays.data_ptr(), batch.img_indices.data_ptr(), batch.rgbs_u8.data_ptr()
                arr[i].select, arr[i].u8_table = batch.select.data_ptr(), batch.u8_table.data_ptr()
                arr[i].idx_is_float = 1 if batch.img_indices.dtype == torch.float32 else 0
                continue
            rays, idx, rgbs = batch
            N.require_device(rays, 'rays')
            rays, rgbs = rays.contiguous().float(), rgbs.contiguous().float()
            if idx.dtype not in (torch.float32, torch.int32):
                idx = idx.float()
            idx = idx.contiguous()
            if rays.shape != (self.n_rays, 8) or rgbs.shape != (self.n_rays, 3) or idx.numel() != self.n_rays:
                raise N.NativeError('FusedTrainStep was planned for batches of {} rays'.format(self.n_rays))
            keep.append((rays, idx, rgbs))
            arr[i].rays, arr[i].idx, arr[i].target = rays.data_ptr(), idx.data_ptr(), rgbs.data_ptr()
            arr[i].idx_is_float = 1 if idx.dtype == torch.float32 else 0
        inj = None
        if _randoms is not None:
            inj = (N.StepRandoms * nc)()
            for i, rd in enumerate(_randoms):
                for k in ('fg_perturb', 'bg_perturb', 'fg_noise_coarse', 'fg_noise_fine', 'bg_noise_coarse', 'bg_noise_fine', 'fg_u', 'bg_u'):
                    if rd is not None and rd.get(k) is not None:
                        t = rd[k].contiguous().float()
                        keep.append(t)
                        setattr(inj[i], k, t.data_ptr())
        self.step_count += 1
        step_lr = self.lr if lr is None else float(lr)
        N.check(N.lib().mnr_train_step(self._plan, arr, inj, step_lr, self.step_count, self.seed, 0 if optimize else N.MNR_STEP_NO_OPTIMIZER,
                                       N.stream_ptr()))
        if optimize and lr is None:
            self.lr *= self.gamma                # ExponentialLR.step()
        self._keep = keep                    # inputs stay alive until the enqueued step has read them (next call at the latest)
        for ci, (f, b) in enumerate(self.cells):
            for k, m in enumerate((f, b)):
                gv = self.grad_views[2 * ci + k]
                for name, p in m.named_parameters():
                    p.grad = gv[name]
                if optimize:
                    m.weights_changed()     # the parameters moved without a version bump: NeRF.packed()'s own cache is stale
        return self.loss, self.n_bg, self.err


class CellTrainer:
    """Optimiser side of the reference trainer for one cell (runner.py:169-194 Adam + ExponentialLR per model, :244-277 the
    iteration) with ONE source of truth for its state: the torch.optim.Adam objects in ``optimizers`` (the reference's dict:
    'nerf', 'bg_nerf') -- whose exp_avg / exp_avg_sq tensors are VIEWS of the fused plan's moment buffers once a plan exists.

    ``step(rays, image_indices, rgbs)`` runs ``mnr_train_step`` (csrc/step.hip: the whole iteration as one call, no host
    synchronisation) when the configuration and the batch size allow it, and otherwise the stage-by-stage autograd path with
    the same optimisers (a ragged last batch of an epoch, an architecture outside the fused step): both paths read and write
    the same moments, step counts and learning rate, and the fused plan's weight images are refreshed after a torch step.
    ``optimizers[k].state_dict()`` after :meth:`sync` is the checkpoint entry of runner.py:523."""

    def __init__(self, nerf: nn.Module, bg_nerf: Optional[nn.Module], hparams: Namespace, sphere_center, sphere_radius,
                 optimizers: Optional[Dict[str, torch.optim.Optimizer]] = None, schedulers: Optional[dict] = None,
                 lr: float = 5e-4, lr_decay_factor: float = 0.1, train_iterations: int = 500000, seed: Optional[int] = None,
                 split_precision: bool = False, iteration: int = 0, plan_rays: Optional[int] = None):
        self.nerf, self.bg_nerf, self.hparams = nerf, bg_nerf, hparams
        # the batch size the one-call plan is built for (the trainer's nominal --batch_size): a ragged batch that happens to come first --
        # right after a resume, say -- must not pin the plan to ITS size and send every full batch down the autograd path
        self.plan_rays = plan_rays
        self.sc, self.sr = sphere_center, sphere_radius
        self._seed, self._split = seed, split_precision
        if optimizers is None:
            optimizers = {'nerf': torch.optim.Adam(nerf.parameters(), lr=lr)}
            if bg_nerf is not None:
                optimizers['bg_nerf'] = torch.optim.Adam(bg_nerf.parameters(), lr=lr)
        if schedulers is None:
            gamma = lr_decay_factor ** (1 / train_iterations)
            schedulers = {k: torch.optim.lr_scheduler.ExponentialLR(o, gamma=gamma) for k, o in optimizers.items()}
        self.optimizers, self.schedulers = optimizers, schedulers
        for o in optimizers.values():
            o._opt_called = True             # (ExponentialLR warns when its first step() precedes optimizer.step(): the fused step IS that step)
        self.fused: Optional[FusedTrainStep] = None
        self.cell = 0                        # this trainer's cell inside `fused` (a rank's cells may share ONE plan: JointCells)
        self.iteration = iteration           # keys the fused step's random streams: continues across a resume
        self._steps_stale = False            # torch's per-parameter `step` tensors lag behind the device counters

    # ---- state plumbing --------------------------------------------------------------------------------------------------------
    def _models(self):
        return [('nerf', self.nerf, 0)] + ([('bg_nerf', self.bg_nerf, 1)] if self.bg_nerf is not None else [])

    def _make_plan(self, n_rays: int) -> None:
        lr = float(self.optimizers['nerf'].param_groups[0]['lr'])
        self.fused = FusedTrainStep([(self.nerf, self.bg_nerf)], self.hparams, self.sc, self.sr, n_rays, lr, seed=self._seed,
                                    split_precision=self._split)
        self.fused.step_count = self.iteration
        self._adopt_into_plan()

    def _adopt_into_plan(self) -> None:
        """This cell's torch.optim.Adam state goes into cell ``self.cell`` of ``self.fused``; from here on torch's state tensors ARE the
        plan's buffers."""
        ci = self.cell
        for key, m, k in self._models():
            opt = self.optimizers[key]
            self.fused.adopt(opt, ci, k)
            mv, vv = self.fused.m_views[2 * ci + k], self.fused.v_views[2 * ci + k]
            t = float(self.fused.adam_t[ci, k].item())
            for name, p in m.named_parameters():
                opt.state[p] = {'step': torch.tensor(t, dtype=torch.float32), 'exp_avg': mv[name], 'exp_avg_sq': vv[name]}

    def sync(self) -> None:
        """Bring torch's per-parameter ``step`` counters up to date with the device (one host synchronisation): call before
        ``optimizers[k].state_dict()`` (checkpoints) -- the moments and the learning rate are shared and always current."""
        if self.fused is None or not self._steps_stale:
            return
        t = self.fused.adam_t[self.cell].tolist()
        for key, m, k in self._models():
            for p in m.parameters():
                self.optimizers[key].state[p]['step'] = torch.tensor(float(t[k]), dtype=torch.float32)
        self._steps_stale = False

    def _after_torch_step(self) -> None:
        if self.fused is None:
            return
        for key, m, k in self._models():
            st = self.optimizers[key].state
            self.fused.adam_t[self.cell, k] = int(float(st[next(iter(m.parameters()))]['step']))
        self.fused.repack()                   # the plan's weight images must follow the parameters

    def health(self) -> None:
        if self.fused is not None:
            self.fused.health()

    # ---- the iteration ---------------------------------------------------------------------------------------------------------
    def step_gathered(self, batch: 'GatheredBatch'):
        """:meth:`step` for a batch given as rows of a device-resident training set: the fused step gathers them itself (no torch
        kernels at all in the iteration); anything the fused step does not take is gathered here and goes through :meth:`step`."""
        n = batch.select.numel()
        if fused_step_supported(self.nerf, self.bg_nerf, self.hparams, n) and self._plan_takes(n):
            return self._fused_call(batch, n)
        sel = batch.select
        return self.step(batch.rays[sel], batch.img_indices[sel], batch.u8_table[batch.rgbs_u8[sel].long()])

    def _plan_takes(self, n: int) -> bool:
        if self.fused is not None:
            return n == self.fused.n_rays
        return self.plan_rays is None or n == self.plan_rays

    def _fused_call(self, batch, n: int):
        if self.fused is None:
            self._make_plan(n)
        self.iteration += 1
        lrs = [float(o.param_groups[0]['lr']) for o in self.optimizers.values()]
        assert all(v == lrs[0] for v in lrs), 'foreground and background optimisers on different learning rates'
        self.fused.step_count = self.iteration - 1
        loss, n_bg, err = self.fused([batch], lr=lrs[0])
        self.last_mse = loss[0]
        self._steps_stale = True
        for s in self.schedulers.values():
            s.step()
        return loss[0], n_bg[0:1], err[0:1]

    def step(self, rays: torch.Tensor, image_indices: Optional[torch.Tensor], rgbs: torch.Tensor):
        """One optimisation step.  Returns (loss, n_bg, err): device scalars / 1-element device tensors, no host sync on the
        fused path."""
        n = rays.shape[0]
        fusable = image_indices is not None and fused_step_supported(self.nerf, self.bg_nerf, self.hparams, n)
        if fusable and self._plan_takes(n):
            return self._fused_call((rays, image_indices, rgbs), n)
        self.iteration += 1
        # stage-by-stage path under autograd, same optimisers (their state tensors are the plan's buffers when one exists)
        self.sync()
        for o in self.optimizers.values():
            o.zero_grad(set_to_none=True)
        results, n_bg, err = render_rays_train(self.nerf, self.bg_nerf, rays, image_indices, self.hparams, self.sc, self.sr,
                                               False, True, False)
        typ = 'fine' if 'rgb_fine' in results else 'coarse'
        loss = torch.nn.functional.mse_loss(results['rgb_' + typ], rgbs, reduction='mean')
        self.last_mse = loss.detach()          # MSE of rgb_{fine|coarse} alone: what the reference's logged PSNR is made of (runner.py:252-256)
        if self.hparams.use_cascade and typ != 'coarse':
            loss = (loss + torch.nn.functional.mse_loss(results['rgb_coarse'], rgbs, reduction='mean')) / 2
        # What the reference checks every iteration (camera inside the sphere: rendering.py:412-414; finite loss: runner.py:260-261) and
        # what gates the background optimiser (runner.py:268-272) are three scalars the FORWARD pass has produced.  They go to pinned host
        # memory behind the forward; the backward pass is enqueued meanwhile, and the host waits for that copy only -- the GPU is then
        # busy with the backward, so neither the check nor the optimisers' enqueue leaves it idle (two full synchronisations per step
        # before: one mid-step, one in front of the optimisers).  Still BEFORE Adam can write a non-finite update into the weights.
        dev = loss.device
        zero = torch.zeros((), device=dev)
        stats = torch.stack([err.max().float() if err is not None else zero, torch.isfinite(loss.detach()).float(),
                             n_bg.reshape(-1)[0].float() if n_bg is not None else zero])
        if getattr(self, '_host_stats', None) is None:
            self._host_stats = torch.empty(3, dtype=torch.float32).pin_memory()
            self._stats_ready = torch.cuda.Event()
        self._host_stats.copy_(stats, non_blocking=True)
        self._stats_ready.record()
        loss.backward()
        t_wait = time.perf_counter()
        self._stats_ready.synchronize()
        self.host_wait_s = getattr(self, 'host_wait_s', 0.0) + (time.perf_counter() - t_wait)       # (blocked, not enqueuing: bench.py's host figure excludes it)
        h_err, h_finite, h_nbg = (float(v) for v in self._host_stats)
        if h_err != 0:
            from mega_nerf.rendering import _ERR_TEXT
            raise Exception(_ERR_TEXT)
        if h_finite == 0:
            raise Exception('Train metrics not finite: {}'.format({'loss': float(loss.detach())}))
        bg_present = n_bg is not None and h_nbg > 0
        for key, o in self.optimizers.items():
            if key == 'bg_nerf' and not bg_present:
                continue
            o.step()
        for m in (self.nerf, self.bg_nerf):
            for sub in (m.modules() if m is not None else ()):
                if hasattr(sub, 'weights_changed'):
                    sub.weights_changed()
        self._after_torch_step()
        for s in self.schedulers.values():
            s.step()
        return loss.detach(), n_bg, err


class TrainStep(CellTrainer):
    """One optimisation step of the reference trainer (runner.py:244-277, fp32): render -> MSE -> backward ->
    Adam on fg and bg -> LR decay, as a callable (bench.py, smoke, tests).  See :class:`CellTrainer`."""

    def __init__(self, nerf: nn.Module, bg_nerf: Optional[nn.Module], hparams: Namespace, sphere_center, sphere_radius,
                 lr: float = 5e-4, lr_decay_factor: float = 0.1, train_iterations: int = 500000):
        super().__init__(nerf, bg_nerf, hparams, sphere_center, sphere_radius, lr=lr, lr_decay_factor=lr_decay_factor,
                         train_iterations=train_iterations)

    @property
    def opts(self):
        return list(self.optimizers.values())

    @property
    def _fused(self):
        return self.fused

    def __call__(self, rays: torch.Tensor, image_indices: Optional[torch.Tensor], rgbs: torch.Tensor):
        return self.step(rays, image_indices, rgbs)


class JointCells:
    """Several cells of ONE rank trained side by side through one plan (``mnr_train_step`` with ``n_cells`` > 1: the cells' rows share
    the MLP launches and fill each other's launch tails), each cell still driven by its own, unchanged ``Runner.train()`` loop -- own
    cluster-masked dataset, own epochs, own checkpoints and logs (parscripts/run_8.txt: one trainer per cell; Building has 25 cells on
    8 GPUs: 4,3,3,...).

    Every cell's loop runs on a host thread of its own; ``member(i)`` is the trainer factory handed to cell i's Runner
    (``Runner.trainer_factory``).  A loop calls ``step_gathered(batch)`` once per iteration: the call parks the thread until every cell
    of the rank has brought its batch, the last one to arrive enqueues ONE joint step, and all resume with their own loss.  Exactly one
    thread runs at any time (the ``baton``: held while a loop executes Python, handed over while it waits), so nothing of the loops --
    validation renders, checkpoints, dataset chunk loads -- interleaves.  An iteration in which some cell brings a batch the plan does
    not take (the ragged last batch of an epoch) is stepped cell by cell on the stage-by-stage path, on the same optimiser state.

    Every cell draws its random numbers as it would alone (``rng_cells`` = 0): a joint job and a one-cell-after-the-other job see the
    same batches and the same random numbers; their weights then differ only by the summation order of atomically accumulated and
    dynamically scheduled partial sums (tests/test_gpu_runner.py)."""

    def __init__(self, n_cells: int):
        import threading
        assert 1 <= n_cells <= N.MNR_STEP_MAX_CELLS
        self.n = n_cells
        self.baton = threading.Condition(threading.Lock())
        self.members: list = [None] * n_cells
        self.pending: dict = {}
        self.results: dict = {}
        self.generation = 0
        self.active = n_cells
        self.plan: Optional[FusedTrainStep] = None
        self.error: Optional[BaseException] = None
        self.joint_steps = self.separate_steps = 0

    def member(self, index: int):
        """The ``trainer_factory`` of cell ``index``: called by its Runner with CellTrainer's arguments."""
        def make(*args, **kwargs):
            m = _JointMember(self, index, *args, **kwargs)
            self.members[index] = m
            return m
        return make

    def run(self, loops) -> None:
        """Run one loop (a callable, e.g. ``runner.train``) per cell to the end, each on its own thread under the baton; re-raises the
        first exception any of them raised."""
        import threading
        assert len(loops) == self.n
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def body(fn):
            with self.baton:
                try:
                    if dev is not None:
                        torch.cuda.set_device(dev)          # (a new thread starts on device 0)
                    fn()
                except BaseException as e:          # noqa: BLE001  (re-raised on the caller's thread)
                    if self.error is None:
                        self.error = e
                finally:
                    self.active -= 1
                    self.baton.notify_all()
        threads = [threading.Thread(target=body, args=(fn,), daemon=True) for fn in loops]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if self.error is not None:
            raise self.error

    # ---- called by a member, baton held --------------------------------------------------------------------------------------------
    def submit(self, index: int, batch: 'GatheredBatch'):
        if self.error is not None:
            raise RuntimeError('another cell of this rank failed') from self.error
        self.pending[index] = batch
        gen = self.generation
        if len(self.pending) == self.n:
            try:
                self._step_all()
            except BaseException as e:
                self.error = e
                self.baton.notify_all()
                raise
            self.pending = {}
            self.generation += 1
            self.baton.notify_all()
        else:
            while self.generation == gen and self.error is None:
                if self.active < self.n:
                    self.error = RuntimeError('a cell of this rank ended its loop while the others still train: joint training needs '
                                              'every cell to run the same number of iterations')
                    self.baton.notify_all()
                    break
                self.baton.wait()
            if self.error is not None:
                raise RuntimeError('another cell of this rank failed') from self.error
        return self.results.pop(index)

    def _make_plan(self) -> None:
        ms = self.members
        n_rays = ms[0].plan_rays
        hp = ms[0].hparams
        lr = float(ms[0].optimizers['nerf'].param_groups[0]['lr'])
        self.plan = FusedTrainStep([(m.nerf, m.bg_nerf) for m in ms], hp, ms[0].sc, ms[0].sr, n_rays, lr, seed=ms[0]._seed,
                                   split_precision=ms[0]._split, rng_cells=[0] * self.n)
        for i, m in enumerate(ms):
            m.fused, m.cell = self.plan, i
            m._adopt_into_plan()

    def _joinable(self) -> bool:
        ms = self.members
        if any(m is None for m in ms) or ms[0].plan_rays is None:
            return False
        if self.plan is not None:
            n = self.plan.n_rays
        else:
            n = ms[0].plan_rays
            if any(m.plan_rays != n or m.iteration != ms[0].iteration for m in ms):
                return False
            if not all(fused_step_supported(m.nerf, m.bg_nerf, m.hparams, n) for m in ms):
                return False
        return all(self.pending[i].select.numel() == n for i in range(self.n)) and len({m.iteration for m in ms}) == 1

    def _step_all(self) -> None:
        ms = self.members
        if self._joinable():
            if self.plan is None:
                self._make_plan()
            lrs = {float(o.param_groups[0]['lr']) for m in ms for o in m.optimizers.values()}
            assert len(lrs) == 1, 'the cells of a rank must be on the same learning-rate schedule'
            for m in ms:
                m.iteration += 1
            self.plan.step_count = ms[0].iteration - 1
            loss, n_bg, err = self.plan([self.pending[i] for i in range(self.n)], lr=lrs.pop())
            for i, m in enumerate(ms):
                m.last_mse = loss[i]
                m._steps_stale = True
                for s_ in m.schedulers.values():
                    s_.step()
                self.results[i] = (loss[i], n_bg[i:i + 1], err[i:i + 1])
            self.joint_steps += 1
            return
        for i, m in enumerate(ms):          # this iteration cell by cell, stage by stage, on the same optimiser state
            b = self.pending[i]
            sel = b.select
            self.results[i] = CellTrainer.step(m, b.rays[sel], b.img_indices[sel], b.u8_table[b.rgbs_u8[sel].long()])
        self.separate_steps += 1


class _JointMember(CellTrainer):
    """What the Runner of one cell of a :class:`JointCells` group holds as its trainer."""

    def __init__(self, joint: JointCells, index: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.joint, self.cell = joint, index

    def step_gathered(self, batch: 'GatheredBatch'):
        return self.joint.submit(self.cell, batch)

    def _plan_takes(self, n: int) -> bool:      # (CellTrainer.step from JointCells._step_all: never the one-cell fused call on the shared plan)
        return False

    def _make_plan(self, n_rays: int) -> None:
        raise AssertionError('the shared plan belongs to JointCells')
