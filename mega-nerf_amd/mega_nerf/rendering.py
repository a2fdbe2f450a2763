"""render_rays -- drop-in for the reference ``mega_nerf.rendering.render_rays`` (rendering.py:15-173)
running entirely on MI355X kernels (csrc/render.hip, csrc/mlp_fwd.hip) through the C ABI.

Host code here only sequences kernel launches on the current HIP stream and owns the (torch-allocated)
device buffers; there is no data-dependent host synchronisation until the very end of
:func:`render_rays`, where the reference API forces one (it returns a Python bool and raises when a
camera lies outside the bounding ellipsoid).  :func:`render_rays_async` is the sync-free form used by
the trainer / benchmark.

Random numbers (training) are drawn with torch on the device in the reference's draw order and can be
injected (``_randoms``) so that train-mode parity is testable.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from argparse import Namespace
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from mega_nerf import _native as N

_ERR_TEXT = ('Not all your cameras are bounded by the unit sphere; please make sure the cameras are normalized '
             'properly!')

# bench.py sets this to a list to get (tag, start_event, end_event) around every MLP launch, recorded on
# the launch stream (kernel-level timing without a profiler); None = no events.
KERNEL_EVENTS = None

# Opt-in: inference MLP passes of the default architectures on the 16-bit matrix pipe with split-precision operands
# (csrc/mlp_fwd_h2.hip: fp32-class accuracy -- 4.6e-7 per layer against fp64 -- at ~2.5x the speed).  The fp32 kernels are the
# default and what every headline number is measured with.
SPLIT_PRECISION = False
# Merged containers: the foreground and the background container's routed evaluations of a pass share ONE launch
# (mnr_mlp_forward_cells_multi); False = one launch per container (tests compare the two)
MERGE_ROUTED = True

_tables: Dict[Tuple[int, str], torch.Tensor] = {}
_host_cache: Dict[int, tuple] = {}          # id(tensor) -> (weakref to it, version, host list)


def linspace01(n: int, device: torch.device) -> torch.Tensor:
    """torch.linspace(0, 1, n) evaluated by the CPU kernel (the parity target: rendering.py:47,82,511),
    cached on the device.  The device-side linspace kernel differs by 1 ulp in places."""
    key = (n, str(device))
    t = _tables.get(key)
    if t is None:
        t = torch.linspace(0, 1, n, device='cpu').to(device)
        _tables[key] = t
    return t


def _host_vec(v) -> Optional[list]:
    """Host copy of a small device tensor (sphere centre/radius).  Cached per tensor *object* (weakly) and version, so a
    trainer that keeps its sphere tensors alive never synchronises in steady state; a fresh tensor costs one copy."""
    if v is None:
        return None
    if not isinstance(v, torch.Tensor):
        return [float(x) for x in v]
    hit = _host_cache.get(id(v))
    ver = -1 if v.is_inference() else v._version          # inference tensors track no version (and cannot change in place)
    if hit is not None and hit[0]() is v and hit[1] == ver:
        return hit[2]
    if len(_host_cache) > 64:
        for k in [k for k, e in _host_cache.items() if e[0]() is None]:
            del _host_cache[k]
    h = v.detach().float().cpu().tolist()
    _host_cache[id(v)] = (weakref.ref(v), ver, h)
    return h


def _f(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)


class _Part:
    """Per-branch (foreground / background) geometry handed to :func:`_get_results`."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _model_eval(nerf, typ, hparams, xyz, part, S, noise):
    if KERNEL_EVENTS is None:
        return _model_eval_inner(nerf, typ, hparams, xyz, part, S, noise)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = _model_eval_inner(nerf, typ, hparams, xyz, part, S, noise)
    b.record()
    KERNEL_EVENTS.append(('%s_%s' % (part.tag, typ), a, b))
    return out


def _model_eval_inner(nerf: nn.Module, typ: str, hparams: Namespace, xyz: torch.Tensor, part: _Part, S: int,
                      noise: Optional[torch.Tensor]) -> torch.Tensor:
    """The MLP pass of _inference (rendering.py:275-331) for n x S samples -> raw [n, S, 4]."""
    from mega_nerf.models.cascade import Cascade
    from mega_nerf.models.mega_nerf import MegaNeRF
    n = xyz.shape[0]
    out = _f(n, S, 4, device=xyz.device)
    model = nerf
    if isinstance(model, Cascade):
        model = model.coarse if typ == 'coarse' else model.fine
    sh_deg = hparams.sh_deg if (hparams.pos_dir_dim == 0 and hparams.sh_deg is not None) else -1
    if isinstance(model, MegaNeRF):
        model.evaluate_routed(xyz, part, S, out, noise, sh_deg)
        return out
    need_dir = model.has_dir or sh_deg >= 0
    if model.has_dir and model.embedding_a is None:
        # quirk Q8 (nerf.py:146): without an appearance column the encoded "direction" is
        # [last xyz coordinate, d_x, d_y]; materialise that 3-vector per sample.
        d = part.dirs.view(n, 1, 3).expand(n, S, 3)
        q = torch.cat([xyz[..., -1:], d[..., :2]], -1).contiguous()
        model.evaluate(xyz, xyz.shape[-1], q, 3, None, 0, 1, n * S, out.view(-1, 4), noise, False, sh_deg,
                       part.n_units, S)
        return out
    model.evaluate(xyz, xyz.shape[-1], part.dirs if need_dir else None, part.dirs.stride(0) if need_dir else 0,
                   part.idx, 1, S, n * S, out.view(-1, 4), noise, False, sh_deg, part.n_units, S)
    return out


def _composite(z, raw, n, S, part: _Part, last_delta, zmax_src, flip, depth_real, want, device):
    io = N.CompositeIO()
    io.z, io.raw = z.data_ptr(), raw.data_ptr()
    io.depth_real = depth_real.data_ptr() if depth_real is not None else None
    io.last_delta = last_delta.data_ptr() if last_delta is not None else None
    if zmax_src is not None and last_delta is not None:
        io.zmax_src, io.zmax_S = zmax_src.data_ptr(), zmax_src.shape[1]
    io.flip, io.N, io.S = int(flip), n, S
    io.n_units_dev = part.n_units.data_ptr() if part.n_units is not None else None
    out = {}
    for k, shape in (('weights', (n, S)), ('rgb', (n, 3)), ('depth', (n,)), ('depth_var', (n,)), ('bg_lambda', (n,))):
        if k in want:
            out[k] = _f(*shape, device=device)
            setattr(io, k, out[k].data_ptr())
    N.check(N.lib().mnr_composite(C.byref(io), N.stream_ptr()))
    return out


class _EvalReq:
    """One MLP pass of a branch, as yielded by :func:`_get_results_gen`: the driver answers with raw [n, S, 4]."""

    def __init__(self, nerf, typ, hparams, xyz, part, S, noise):
        self.nerf, self.typ, self.hparams, self.xyz, self.part, self.S, self.noise = nerf, typ, hparams, xyz, part, S, noise


def _serve(reqs) -> list:
    """Run the MLP passes the branches are waiting for.  Two default-architecture NeRFs (the foreground and the background
    model of a render) go out as ONE launch (mnr_mlp_forward_multi): the compacted background rows alone fill half the
    chip at best, side by side with the foreground's they only lengthen its tail.  Everything else: one launch each."""
    from mega_nerf.models.nerf import NeRF
    split = SPLIT_PRECISION and not torch.is_grad_enabled()
    if (len(reqs) > 1 or split) and all(isinstance(q.nerf, NeRF) and q.nerf.is_default_arch() and q.part.idx is not None for q in reqs):
        segs = (N.MlpLaunch * len(reqs))()
        outs, keep = [], []
        for sg, q in zip(segs, reqs):
            n = q.xyz.shape[0]
            out = _f(n, q.S, 4, device=q.xyz.device)
            io = q.nerf.mlp_io(q.xyz, q.xyz.shape[-1], q.part.dirs, q.part.dirs.stride(0), q.part.idx, 1, q.S, n * q.S,
                               out.view(-1, 4), q.noise, q.part.n_units, q.S)
            desc, packed = q.nerf.packed_h2() if split else q.nerf.packed()
            keep.append((io, desc, packed))
            sg.packed_dev, sg.desc, sg.io = packed.data_ptr(), C.pointer(desc), C.pointer(io)
            outs.append(out)

        def launch():
            if split:
                N.check(N.lib().mnr_mlp_forward_multi_h2(segs, len(reqs), N.stream_ptr()))
            else:
                N.check(N.lib().mnr_mlp_forward_multi(segs, len(reqs), N.stream_ptr()))
        if KERNEL_EVENTS is None:
            launch()
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            launch()
            b.record()
            KERNEL_EVENTS.append(('fwd_%s' % reqs[0].typ, a, b))
        return outs
    if len(reqs) > 1 and not split and MERGE_ROUTED:
        # merged containers (foreground and background): both routed evaluations of the pass in ONE gather-mode launch
        from mega_nerf.models.mega_nerf import MegaNeRF, evaluate_routed_together
        if all(isinstance(q.nerf, MegaNeRF) for q in reqs):
            outs = [_f(q.xyz.shape[0], q.S, 4, device=q.xyz.device) for q in reqs]
            a = b = None
            if KERNEL_EVENTS is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            hp = reqs[0].hparams
            sh_deg = hp.sh_deg if (hp.pos_dir_dim == 0 and hp.sh_deg is not None) else -1
            if evaluate_routed_together([(q.nerf, q.xyz, q.part, q.S, o, q.noise, sh_deg) for q, o in zip(reqs, outs)]):
                if a is not None:
                    b.record()
                    KERNEL_EVENTS.append(('fwd_%s' % reqs[0].typ, a, b))
                return outs
    return [_model_eval(q.nerf, q.typ, q.hparams, q.xyz, q.part, q.S, q.noise) for q in reqs]


def _run_branches(gens) -> list:
    """Advance the branch generators in lockstep, serving their MLP passes together; returns their result dicts."""
    results = [None] * len(gens)
    pending = {}
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as e:
            results[i] = e.value
    while pending:
        order = sorted(pending)
        outs = _serve([pending[i] for i in order])
        nxt = {}
        for i, out in zip(order, outs):
            try:
                nxt[i] = gens[i].send(out)
            except StopIteration as e:
                results[i] = e.value
        pending = nxt
    return results


def _get_results(nerf: nn.Module, hparams: Namespace, part: _Part, get_depth: bool, get_depth_variance: bool,
                 get_bg_lambda: bool, flip: bool, rnd: dict, tag: str) -> Dict[str, torch.Tensor]:
    """rendering.py:176-248 for one branch on its own (every MLP pass is its own launch)."""
    return _run_branches([_get_results_gen(nerf, hparams, part, get_depth, get_depth_variance, get_bg_lambda, flip, rnd, tag)])[0]


def _get_results_gen(nerf: nn.Module, hparams: Namespace, part: _Part, get_depth: bool, get_depth_variance: bool,
                     get_bg_lambda: bool, flip: bool, rnd: dict, tag: str):
    """rendering.py:176-248 for one branch as a generator: it yields an :class:`_EvalReq` wherever the reference calls the
    model and is resumed with the raw output, so that a driver can serve the passes of several branches with one launch.
    ``part`` carries z_coarse [n,Sc], xyz_coarse, depth_real, last_delta.  Returns (StopIteration.value) the result dict."""
    lib = N.lib()
    dev = part.z.device
    n, Sc = part.z.shape
    Nf = hparams.fine_samples
    cascade = hparams.use_cascade
    results: Dict[str, torch.Tensor] = {}
    nunits = part.n_units.data_ptr() if part.n_units is not None else None

    # ---- coarse pass (rendering.py:195-210) ----
    xyz_c, z_c = part.xyz, part.z
    if flip:                                   # :271-273 (depth_real is *not* flipped: quirk Q2)
        xyz_c, z_c = xyz_c.flip(1).contiguous(), z_c.flip(1).contiguous()
    noise_c = rnd.get(tag + '_noise_coarse') if nerf.training else None
    if nerf.training and noise_c is None:
        noise_c = torch.rand(n * Sc, device=dev)
    raw_c = yield _EvalReq(nerf, 'coarse', hparams, xyz_c, part, Sc, noise_c)
    want = set()
    if Nf > 0:
        want.add('weights')
    if cascade:
        want.add('rgb')
        if get_bg_lambda:
            want.add('bg_lambda')
    if Nf == 0 and (get_depth or get_depth_variance):
        want.add('depth')
        if get_depth_variance:
            want.add('depth_var')
    comp = _composite(z_c, raw_c, n, Sc, part, part.last_delta, part.z, flip, part.depth_real, want, dev)
    if cascade:
        results['rgb_coarse'] = comp['rgb']
        if get_bg_lambda:
            results['bg_lambda_coarse'] = comp['bg_lambda']
    else:
        results['zvals_coarse'] = z_c
        results['raw_rgb_coarse'] = raw_c[..., :3]
        results['raw_sigma_coarse'] = raw_c[..., 3]
        if part.depth_real is not None:
            results['depth_real_coarse'] = part.depth_real
    if Nf == 0:
        if get_depth:
            results['depth_coarse'] = comp['depth']
        if get_depth_variance:
            results['depth_variance_coarse'] = comp['depth_var']
        return results

    # ---- importance sampling (rendering.py:212-219) ----
    nf = Nf // 2 if flip else Nf
    det = (hparams.perturb if nerf.training else 0) == 0
    if det:
        u = linspace01(nf, dev)
    else:
        u = rnd.get(tag + '_u')
        if u is None:
            u = torch.rand(n, nf, device=dev)
    z_f = _f(n, nf, device=dev)
    inds = torch.empty(n, nf, device=dev, dtype=torch.int32) if rnd.get('_want_inds') else None
    N.check(lib.mnr_sample_fine(part.z.data_ptr(), comp['weights'].data_ptr(), n, nunits, Sc, nf, int(det),
                                u.data_ptr(), z_f.data_ptr(), N.ptr(inds), N.stream_ptr()))
    if inds is not None:
        rnd['_inds_' + tag] = inds
        rnd['_fine_z_' + tag] = z_f
    zmax_src = z_f                                           # last_delta uses the fine-only max (quirk Q4)
    if cascade:
        z_all = _f(n, Sc + nf, device=dev)
        N.check(lib.mnr_sort_rows(part.z.data_ptr(), Sc, z_f.data_ptr(), nf, n, nunits, z_all.data_ptr(), N.stream_ptr()))
        z_f, nf = z_all, Sc + nf
        zmax_src = z_f
    xyz_f, depth_real_f = part.points(z_f)

    # ---- fine pass (rendering.py:227-242) ----
    if flip and cascade:                                     # 'zvals_coarse' absent -> flip again (:271-273)
        xyz_f, z_f = xyz_f.flip(1).contiguous(), z_f.flip(1).contiguous()
    noise_f = rnd.get(tag + '_noise_fine') if nerf.training else None
    if nerf.training and noise_f is None:
        noise_f = torch.rand(n * nf, device=dev)
    raw_f = yield _EvalReq(nerf, 'fine', hparams, xyz_f, part, nf, noise_f)
    if cascade:
        z_m, raw_m, dr_m, Sm = z_f, raw_f, depth_real_f, nf
    else:
        Sm = nf + Sc
        z_m, raw_m = _f(n, Sm, device=dev), _f(n, Sm, 4, device=dev)
        dr_m = _f(n, Sm, device=dev) if depth_real_f is not None else None
        N.check(lib.mnr_merge_sorted(z_f.data_ptr(), raw_f.data_ptr(), N.ptr(depth_real_f), nf, z_c.data_ptr(),
                                     raw_c.data_ptr(), N.ptr(part.depth_real), Sc, n, nunits, int(flip),
                                     z_m.data_ptr(), raw_m.data_ptr(), N.ptr(dr_m), None, N.stream_ptr()))
    want = {'rgb'}
    if get_bg_lambda:
        want.add('bg_lambda')
    if get_depth or get_depth_variance:
        want.add('depth')
    if get_depth_variance:
        want.add('depth_var')
    comp = _composite(z_m, raw_m, n, Sm, part, part.last_delta, zmax_src, flip, dr_m, want, dev)
    results['rgb_fine'] = comp['rgb']
    if get_bg_lambda:
        results['bg_lambda_fine'] = comp['bg_lambda']
    if get_depth:
        results['depth_fine'] = comp['depth']
    if get_depth_variance:
        results['depth_variance_fine'] = comp['depth_var']
    for k in ('zvals_coarse', 'raw_rgb_coarse', 'raw_sigma_coarse', 'depth_real_coarse'):
        results.pop(k, None)
    return results


def _background_part(bg_nerf, nerf, hparams, rays, image_indices, bg_list, n_bg, n_rays, perturb, c, r, rnd, dev) -> _Part:
    """Coarse background samples (rendering.py:47-56) for the compacted background rays, on the current stream."""
    lib = N.lib()
    Sb = hparams.coarse_samples // 2
    include_xyz_real = hparams.container_path is not None or hparams.train_mega_nerf is not None
    cluster_2d = bool(include_xyz_real and getattr(nerf, 'cluster_dim_start', 0) == 1)
    ncol = 7 if include_xyz_real else 4
    rays_bg = rays.index_select(0, bg_list.long())          # compacted rays (rows >= n_bg are padding)
    idx_bg = image_indices.index_select(0, bg_list.long()) if image_indices is not None else None
    t_bg = linspace01(Sb, dev)
    prnd = None
    if perturb > 0:
        prnd = rnd.get('bg_perturb')
        if prnd is None:
            prnd = torch.rand(n_rays, Sb, device=dev)
    bg_z = _f(n_rays, Sb, device=dev)
    bg_pts, bg_dr = _f(n_rays, Sb, ncol, device=dev), _f(n_rays, Sb, device=dev)
    N.check(lib.mnr_bg_samples(rays_bg.data_ptr(), None, n_bg.data_ptr(), n_rays, Sb, t_bg.data_ptr(), perturb,
                               N.ptr(prnd), None, N.host3(c), N.host3(r), int(include_xyz_real),
                               int(cluster_2d), bg_z.data_ptr(), bg_pts.data_ptr(), bg_dr.data_ptr(), N.stream_ptr()))

    def bg_points(zf):
        s = zf.shape[1]
        p, d = _f(n_rays, s, ncol, device=dev), _f(n_rays, s, device=dev)
        N.check(lib.mnr_bg_samples(rays_bg.data_ptr(), None, n_bg.data_ptr(), n_rays, s, None, 0.0, None,
                                   zf.data_ptr(), N.host3(c), N.host3(r), int(include_xyz_real),
                                   int(cluster_2d), None, p.data_ptr(), d.data_ptr(), N.stream_ptr()))
        return p, d

    return _Part(z=bg_z, xyz=bg_pts, depth_real=bg_dr, last_delta=None, n_units=n_bg, dirs=rays_bg[:, 3:6], idx=idx_bg,
                 points=bg_points, rays=rays_bg, tag='bg')


def _empty_results(hparams: Namespace, has_bg: bool, get_depth: bool, get_depth_variance: bool, get_bg_fg_rgb: bool,
                   dev: torch.device) -> Dict[str, torch.Tensor]:
    """The result dict of a zero-ray batch: the keys render_rays produces for these flags, with empty tensors."""
    Nf = hparams.fine_samples
    types = ['fine' if Nf > 0 else 'coarse'] + (['coarse'] if (hparams.use_cascade and Nf > 0) else [])
    out: Dict[str, torch.Tensor] = {}
    for typ in types:
        out['rgb_' + typ] = _f(0, 3, device=dev)
        main_pass = typ == types[0]
        if get_depth and main_pass:
            out['depth_' + typ] = _f(0, device=dev)
        if get_depth_variance and main_pass:
            out['depth_variance_' + typ] = _f(0, device=dev)
        if has_bg:
            out['bg_lambda_' + typ] = _f(0, device=dev)
            if get_bg_fg_rgb:
                out['fg_rgb_' + typ], out['bg_rgb_' + typ] = _f(0, 3, device=dev), _f(0, 3, device=dev)
                if get_depth and main_pass:
                    out['fg_depth_' + typ], out['bg_depth_' + typ] = _f(0, device=dev), _f(0, device=dev)
    return out


FUSED_RENDER = True          # inference renders of the default configuration go through mnr_render_fwd (six launches)
_render_ws: Dict[tuple, torch.Tensor] = {}
_render_side: Dict[tuple, C.c_void_p] = {}     # (device, stream) -> mnr_side handle (host object: stream + two events), created on first use


def release_render_workspaces(*models) -> None:
    """Drop the cached scratch of the one-call render (~2.7 GB after 65 536-ray batches at 256 + 512 samples) and, for merged containers
    passed in, their routing buffers (``MegaNeRF.release_buffers``); the next render re-allocates."""
    _render_ws.clear()
    _route_ws.clear()
    for m in models:
        for sub in (m.modules() if isinstance(m, torch.nn.Module) else ()):       # (a Cascade may hold containers)
            if hasattr(sub, 'release_buffers'):
                sub.release_buffers()



def _routed_pair(nerf, bg_nerf) -> bool:
    """Both models are merged containers (MegaNeRF routers) the one-call render covers: the same centroids, margin and clustering (3-D, or
    `cluster_2d`), the foreground routed on its points and the background on its rays' sphere-exit points -- under `cluster_2d` on every
    sample's own far-away position (mega_nerf.py:19-61, rendering.py:458-464)."""
    from mega_nerf.models.mega_nerf import MegaNeRF
    if not (isinstance(nerf, MegaNeRF) and isinstance(bg_nerf, MegaNeRF)):
        return False
    if nerf.cluster_dim_start != bg_nerf.cluster_dim_start or nerf.xyz_real or not bg_nerf.xyz_real or nerf.joint_training or bg_nerf.joint_training:
        return False
    if nerf.cluster_dim_start and os.environ.get('MNR_NO_FUSED_2D_ROUTED_RENDER'):
        return False
    if len(nerf.sub_modules) != len(bg_nerf.sub_modules) or not 1 <= len(nerf.sub_modules) <= 64:
        return False
    # (host copies, cached on the modules: no device read per render)
    if float(nerf.boundary_margin) != float(bg_nerf.boundary_margin) or list(nerf._centroids_host()) != list(bg_nerf._centroids_host()):
        return False
    from mega_nerf.models.mega_nerf import _arch_key
    for m in (nerf, bg_nerf):
        kids = list(m.sub_modules)
        if not all(c.fused_supported() and _arch_key(c) == _arch_key(kids[0]) and c.embedding_a is not None for c in kids):
            return False
    return True


def _fused_render_ok(nerf, bg_nerf, hparams, image_indices, sphere_radius, get_depth_variance, rnd) -> bool:
    import os
    from mega_nerf.models.nerf import NeRF
    if not FUSED_RENDER or os.environ.get('MNR_NO_FUSED_RENDER') or bg_nerf is None or image_indices is None or sphere_radius is None:
        return False
    if get_depth_variance or rnd or hparams.use_cascade or hparams.train_mega_nerf is not None:
        return False
    if (hparams.coarse_samples, hparams.fine_samples) not in ((64, 128), (256, 512)):
        return False
    # the default models, or their spherical-harmonics form (configs/mega-nerf-sh-3: sh_deg 2, pos_dir_dim 0; also sh_deg 3; fp32 kernels only)
    sh = hparams.sh_deg is not None and hparams.pos_dir_dim == 0
    if sh and (hparams.sh_deg not in (2, 3) or SPLIT_PRECISION):
        return False
    routed = hparams.container_path is not None
    if routed:
        # merged containers: route -> all cells of both containers in one launch -> blend, inside the same call (fp32 kernels)
        if SPLIT_PRECISION or os.environ.get('MNR_NO_FUSED_ROUTED_RENDER') or nerf.training or bg_nerf.training or not _routed_pair(nerf, bg_nerf):
            return False
        if getattr(nerf, 'routed_rows', None) is not None or getattr(bg_nerf, 'routed_rows', None) is not None:
            return False               # (bench.py's device-side tally of routed rows lives in the stage-by-stage path)
        models = (nerf.sub_modules[0], bg_nerf.sub_modules[0])
    else:
        models = (nerf, bg_nerf)
    for m in models:
        if not isinstance(m, NeRF) or (not routed and m.training):
            return False
        # default 8 x 256, its sh_deg 2 form, or (fp32 kernels only) the 512-wide Building shape on the wavefront-pair kernel
        wide = m.is_wide_default_arch() and not sh and not SPLIT_PRECISION and os.environ.get('MNR_NO_PAIR_KERNEL') is None
        if not ((m.is_sh_arch(hparams.sh_deg) if sh else m.is_default_arch()) or wide):
            return False
    return models[0].xyz_dim == 3 and models[1].xyz_dim == 4


_route_ws: Dict[tuple, torch.Tensor] = {}


def _fused_render(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, get_depth, get_bg_fg_rgb):
    """render_rays (evaluation flags) as ONE call of mnr_render_fwd: csrc/step.hip."""
    lib = N.lib()
    dev = rays.device
    n = rays.shape[0]
    Nc, Nf = hparams.coarse_samples, hparams.fine_samples
    # scratch and side handle per (device, stream): two renders enqueued on different streams must not share intermediates
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    need = lib.mnr_render_workspace_bytes(n, Nc, Nf)
    ws = _render_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _render_ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    out = _f(13 * n, device=dev)
    scal = torch.empty(2, device=dev, dtype=torch.int32)
    io = N.RenderIO()
    split = 1 if SPLIT_PRECISION else 0
    keep = []
    if hparams.container_path is not None:
        # merged containers: the cells' weight images / appearance tables as host pointer arrays, the routing buffers as one workspace
        nc = len(nerf.sub_modules)
        arrs = []
        for m in (nerf, bg_nerf):
            packs = [c.packed() for c in m.sub_modules]
            keep.append(packs)
            arrs.append((C.c_void_p * nc)(*[pk.data_ptr() for _, pk in packs]))
            arrs.append((C.c_void_p * nc)(*[c.embedding_a.weight.data_ptr() for c in m.sub_modules]))
        fd, bd = keep[0][0][0], keep[1][0][0]
        io.fg, io.bg = C.pointer(fd), C.pointer(bd)
        io.n_cells, io.fg_cell_packed, io.fg_cell_emb, io.bg_cell_packed, io.bg_cell_emb = nc, arrs[0], arrs[1], arrs[2], arrs[3]
        cent = nerf._centroids_host()
        io.centroids_host, io.boundary_margin, io.cluster_2d = cent, float(nerf.boundary_margin), int(nerf.cluster_dim_start == 1)
        sh_blend = hparams.sh_deg is not None and hparams.pos_dir_dim == 0 and float(nerf.boundary_margin) > 1
        rneed = lib.mnr_render_route_workspace_bytes(n, Nc, Nf, nc, nerf.sub_modules[0].rgb_dim + 1 if sh_blend else 4)
        rws = _route_ws.get(key)
        if rws is None or rws.numel() < rneed:
            rws = _route_ws[key] = torch.empty(rneed, dtype=torch.uint8, device=dev)
        io.route_workspace, io.route_workspace_bytes = rws.data_ptr(), rws.numel()
        keep += [arrs, cent]
    else:
        fd, fp = nerf.packed_h2() if split else nerf.packed()
        bd, bp = bg_nerf.packed_h2() if split else bg_nerf.packed()
        io.fg, io.bg, io.fg_packed, io.bg_packed = C.pointer(fd), C.pointer(bd), fp.data_ptr(), bp.data_ptr()
    io.rays, io.idx, io.idx_is_float, io.n_rays = rays.data_ptr(), image_indices.data_ptr(), 1 if image_indices.dtype == torch.float32 else 0, n
    io.coarse_samples, io.fine_samples, io.split_precision = Nc, Nf, split
    c, r = _host_vec(sphere_center), _host_vec(sphere_radius)
    for i in range(3):
        io.sphere_center[i], io.sphere_radius[i] = c[i], r[i]
    tabs = [linspace01(k, dev) for k in (Nc, Nc // 2, Nf, Nf // 2)]
    io.t_coarse_dev, io.t_bg_coarse_dev, io.t_fine_dev, io.t_bg_fine_dev = [t.data_ptr() for t in tabs]
    v = {'rgb': out[0:3 * n].view(n, 3), 'fg_rgb': out[3 * n:6 * n].view(n, 3), 'bg_rgb': out[6 * n:9 * n].view(n, 3), 'depth': out[9 * n:10 * n],
         'fg_depth': out[10 * n:11 * n], 'bg_depth': out[11 * n:12 * n], 'bg_lambda': out[12 * n:13 * n]}
    io.rgb, io.bg_lambda = v['rgb'].data_ptr(), v['bg_lambda'].data_ptr()
    if get_depth:
        io.depth = v['depth'].data_ptr()
    if get_bg_fg_rgb:
        io.fg_rgb, io.bg_rgb = v['fg_rgb'].data_ptr(), v['bg_rgb'].data_ptr()
        if get_depth:
            io.fg_depth, io.bg_depth = v['fg_depth'].data_ptr(), v['bg_depth'].data_ptr()
    io.n_bg, io.err = scal[0:1].data_ptr(), scal[1:2].data_ptr()
    io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
    if os.environ.get('MNR_RENDER_TWO_STREAMS') and hparams.container_path is None:
        # opt-in: the background branch beside the foreground's passes on a side stream (the foreground's passes are whole rounds of
        # workgroups, the background's partial rounds run inside them).  Measured at 1024 rays: fp32 render 488 K -> 508 K rays/s, the
        # split-precision render unchanged (1.27 M).  Off by default: with it every per-launch duration of the MLP kernel is an
        # overlapped one (the evidence under profiles/ is per launch).
        side = _render_side.get(key)
        if side is None:
            h = C.c_void_p()
            with torch.cuda.device(dev):
                N.check(lib.mnr_side_create(C.byref(h)))
            side = _render_side[key] = h
        io.side = side
    N.check(lib.mnr_render_fwd(C.byref(io), N.stream_ptr()))
    results = {'rgb_fine': v['rgb'], 'bg_lambda_fine': v['bg_lambda']}
    if get_depth:
        results['depth_fine'] = v['depth']
    if get_bg_fg_rgb:
        results['fg_rgb_fine'], results['bg_rgb_fine'] = v['fg_rgb'], v['bg_rgb']
        if get_depth:
            results['fg_depth_fine'], results['bg_depth_fine'] = v['fg_depth'], v['bg_depth']
    return results, scal[0:1], scal[1:2]


def render_rays_async(nerf: nn.Module, bg_nerf: Optional[nn.Module], rays: torch.Tensor,
                      image_indices: Optional[torch.Tensor], hparams: Namespace, sphere_center, sphere_radius,
                      get_depth: bool, get_depth_variance: bool, get_bg_fg_rgb: bool, _randoms: Optional[dict] = None):
    """Enqueue the whole render on the current stream.  Returns (results, n_bg_dev, err_flag_dev); the two
    device scalars are None without a background model.  No host synchronisation."""
    N.require_device(rays, 'rays')
    if (torch.is_grad_enabled() and not (get_depth or get_bg_fg_rgb)
            and any(p.requires_grad for m in (nerf, bg_nerf) if m is not None for p in m.parameters())):
        # differentiable path (hand-written backward) for the trainer's flag set (runner.py:349-358); renders that
        # ask for depth / fg-bg splits are evaluation renders and take the inference path below (no graph)
        from mega_nerf.training import render_rays_train
        return render_rays_train(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, get_depth,
                                 get_depth_variance, get_bg_fg_rgb, _randoms)
    lib = N.lib()
    dev = rays.device
    rnd = _randoms if _randoms is not None else {}
    rays = rays.contiguous().float()
    n_rays = rays.shape[0]
    Nc, Nf = hparams.coarse_samples, hparams.fine_samples
    if image_indices is not None:
        N.require_device(image_indices, 'image_indices')
        if image_indices.dtype not in (torch.float32, torch.int32):
            image_indices = image_indices.float()
        image_indices = image_indices.contiguous()
    perturb = float(hparams.perturb) if nerf.training else 0.0
    dirs = rays[:, 3:6]
    if n_rays == 0:
        return _empty_results(hparams, bg_nerf is not None, get_depth, get_depth_variance, get_bg_fg_rgb, dev), None, None
    if _fused_render_ok(nerf, bg_nerf, hparams, image_indices, sphere_radius, get_depth_variance, rnd):
        return _fused_render(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius, get_depth, get_bg_fg_rgb)

    n_bg = err = bg_slot = None
    far = None
    last_delta = None
    gens = []
    if bg_nerf is not None:
        c, r = _host_vec(sphere_center), _host_vec(sphere_radius)
        far, last_delta = _f(n_rays, device=dev), _f(n_rays, device=dev)
        bg_list = torch.zeros(max(n_rays, 1), device=dev, dtype=torch.int32)
        bg_slot = torch.empty(max(n_rays, 1), device=dev, dtype=torch.int32)
        scal = torch.zeros(2, device=dev, dtype=torch.int32)
        n_bg, err = scal[0:1], scal[1:2]
        N.check(lib.mnr_ray_setup(rays.data_ptr(), n_rays, N.host3(c), N.host3(r), far.data_ptr(),
                                  last_delta.data_ptr(), bg_list.data_ptr(), bg_slot.data_ptr(), n_bg.data_ptr(),
                                  err.data_ptr(), N.stream_ptr()))
        # The background branch (rendering.py:47-75) is independent of the foreground until the blend.  Both advance pass
        # by pass on the same stream and every MLP pass is ONE launch over the rows of both (see _serve).
        bg_part = _background_part(bg_nerf, nerf, hparams, rays, image_indices, bg_list, n_bg, n_rays, perturb, c, r, rnd, dev)
        gens.append(_get_results_gen(bg_nerf, hparams, bg_part, get_depth, get_depth_variance, False, True, rnd, 'bg'))

    # ---- foreground (rendering.py:81-100) ----
    t_c = linspace01(Nc, dev)
    prnd = None
    if perturb > 0:
        prnd = rnd.get('fg_perturb')
        if prnd is None:
            prnd = torch.rand(n_rays, Nc, device=dev)
    z = _f(n_rays, Nc, device=dev)
    xyz = _f(n_rays, Nc, 3, device=dev)
    N.check(lib.mnr_fg_samples(rays.data_ptr(), N.ptr(far), n_rays, Nc, t_c.data_ptr(), perturb, N.ptr(prnd),
                               z.data_ptr(), xyz.data_ptr(), N.stream_ptr()))

    def fg_points(zf):
        p = _f(n_rays, zf.shape[1], 3, device=dev)
        N.check(lib.mnr_fg_points(rays.data_ptr(), n_rays, zf.shape[1], zf.data_ptr(), p.data_ptr(), N.stream_ptr()))
        return p, None

    fg_part = _Part(z=z, xyz=xyz, depth_real=None, last_delta=last_delta, n_units=None, dirs=dirs,
                    idx=image_indices, points=fg_points, rays=rays, tag='fg')
    gens.insert(0, _get_results_gen(nerf, hparams, fg_part, get_depth, get_depth_variance, bg_nerf is not None, False, rnd, 'fg'))
    done = _run_branches(gens)
    results = done[0]
    bg_results = done[1] if bg_nerf is not None else None

    # ---- fg/bg blend (rendering.py:102-139) ----
    if bg_nerf is not None and n_rays > 0:
        types = ['fine' if Nf > 0 else 'coarse']
        if hparams.use_cascade and Nf > 0:
            types.append('coarse')
        for typ in types:
            lam = results['bg_lambda_%s' % typ]          # KeyError for Nf == 0 without cascade, like the reference
            rgb, depth = results.get('rgb_%s' % typ), results.get('depth_%s' % typ)
            outs = {}
            if get_bg_fg_rgb:
                for key, val in (('rgb', rgb), ('depth', depth)):
                    if val is not None:
                        outs['fg_' + key] = torch.empty_like(val)
                        outs['bg_' + key] = torch.empty_like(val)
            N.check(lib.mnr_bg_blend(N.ptr(rgb), N.ptr(depth), lam.data_ptr(), bg_slot.data_ptr(),
                                     N.ptr(bg_results.get('rgb_%s' % typ)), N.ptr(bg_results.get('depth_%s' % typ)),
                                     n_rays, N.ptr(outs.get('fg_rgb')), N.ptr(outs.get('bg_rgb')),
                                     N.ptr(outs.get('fg_depth')), N.ptr(outs.get('bg_depth')), N.stream_ptr()))
            for k, v in outs.items():
                results['%s_%s' % (k, typ)] = v
    return results, n_bg, err


def render_rays(nerf: nn.Module,
                bg_nerf: Optional[nn.Module],
                rays: torch.Tensor,
                image_indices: Optional[torch.Tensor],
                hparams: Namespace,
                sphere_center: Optional[torch.Tensor],
                sphere_radius: Optional[torch.Tensor],
                get_depth: bool,
                get_depth_variance: bool,
                get_bg_fg_rgb: bool,
                _randoms: Optional[dict] = None) -> Tuple[Dict[str, torch.Tensor], bool]:
    """Same contract as the reference (rendering.py:15-24): returns ``(results, bg_nerf_rays_present)``."""
    results, n_bg, err = render_rays_async(nerf, bg_nerf, rays, image_indices, hparams, sphere_center, sphere_radius,
                                           get_depth, get_depth_variance, get_bg_fg_rgb, _randoms)
    present = False
    if n_bg is not None:
        host = torch.stack([n_bg[0], err[0]]).cpu()      # the one sync the reference API requires
        if int(host[1]) != 0:
            raise Exception(_ERR_TEXT)
        present = int(host[0]) > 0
    return results, present
