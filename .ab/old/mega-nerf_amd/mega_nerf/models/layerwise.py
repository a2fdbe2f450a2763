"""Layer-by-layer evaluation and adjoint of the NeRF MLP (reference nerf.py:115-160) for architectures the
register-chained kernel does not cover (layer_dim 2048 of configs/nerf, spherical-harmonics heads, no appearance
embedding, ...).  One exact-fp32 MFMA GEMM launch per nn.Linear (csrc/layerwise.hip) with layer outputs in HBM;
when ``keep`` is set they stay alive as the tape of :meth:`LayerwiseTape.backward`.

Inputs are described like ``mnr_mlp_io``: row r reads xyz[r], dir[r // dir_rows], idx[r // rows_per_ray].

Layers whose width is a multiple of 256 (layer_dim 256 without the fused instantiation, 512, 1024, 2048) run on the tiled
GEMM of csrc/tgemm.hip -- forward with bias + ReLU fused, data gradients with the ReLU adjoint and the sigma head's rank-1
term fused -- and their weight gradients on the job form of csrc/wgrad.hip; odd input widths (63 embedding columns, 27 + 48
direction / appearance columns) are zero-padded to whole 32-column K tiles.  Everything else (the 1- and 3-wide heads,
widths that are not multiples of 256) stays on the 128 x 128 kernels of csrc/layerwise.hip.  ``MNR_NO_TGEMM=1`` forces
the latter everywhere (A/B measurements).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from mega_nerf import _native as N

_F4 = 4


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def _al16(*ptrs: int) -> bool:
    return all(p % 16 == 0 for p in ptrs)


def _padded_weight(w: torch.Tensor, splits: List[Tuple[int, int, int]]) -> torch.Tensor:
    """Column blocks (col0, cols, padded cols) of a weight matrix side by side, each zero-padded to its padded width, so
    that every block starts 16-byte aligned and spans whole K tiles of the tiled GEMM (csrc/tgemm.hip)."""
    w = w.detach()
    if all(n == npad for _, n, npad in splits) and len(splits) == 1:
        return w
    parts = []
    for c0, n, npad in splits:
        parts.append(w[:, c0:c0 + n])
        if npad > n:
            parts.append(w.new_zeros(w.shape[0], npad - n))
    return torch.cat(parts, 1)


def _cached_padded_weight(model, name: str, w: torch.Tensor, splits: List[Tuple[int, int, int]]) -> torch.Tensor:
    """:func:`_padded_weight`, kept on the model until the parameter changes: every row chunk of an evaluation (and every call
    between two optimiser steps) would otherwise rebuild the same copies with a torch.cat.  The key follows NeRF.packed(): storage,
    version counter (absent under inference_mode: then the model's own ``weights_changed()`` epoch), splits."""
    cache = model.__dict__.setdefault('_padded_weights', {})
    ver = None if w.is_inference() else w._version
    key = (w.data_ptr(), ver, getattr(model, '_weights_epoch', 0), tuple(splits))
    hit = cache.get(name)
    if hit is not None and hit[0] == key:
        return hit[1]
    wp = _padded_weight(w, splits)
    if wp.data_ptr() != w.data_ptr():
        cache[name] = (key, wp)
    return wp


def _act_sigma(model) -> int:
    from mega_nerf.models.nerf import ShiftedSoftplus
    return 3 if isinstance(model.sigma_activation, ShiftedSoftplus) else 1


class LayerwiseTape:
    """Forward pass over ``B`` rows; with ``keep`` every layer output is retained for :meth:`backward`."""

    def __init__(self, model, xyz: torch.Tensor, xyz_stride: int, dirs: Optional[torch.Tensor], dir_stride: int,
                 dir_rows: int, idx: Optional[torch.Tensor], idx_stride: int, rows_per_ray: int, B: int, out: torch.Tensor,
                 out_stride: int, sigma_noise: Optional[torch.Tensor], sigma_only: bool, sh_deg: int, keep: bool,
                 sh_dirs: Optional[torch.Tensor] = None, sh_dir_stride: int = 0):
        lib, st = N.lib(), N.stream_ptr
        m = self.model = model
        dev = out.device
        W, D = m.layer_dim, m.xyz_dim
        E = self.E = D * (1 + 2 * m.pos_xyz_dim)
        ED = self.ED = 3 * (1 + 2 * m.pos_dir_dim) if m.has_dir else 0
        A = self.A = m.appearance_dim if (m.embedding_a is not None and m.affine is None) else 0
        self.affine = m.affine is not None and not sigma_only     # nerf.py:156-158: 3x4 colour transform per appearance index
        if self.affine and m.rgb_dim != 3:
            # the reference fails here too: nerf.py:157-158 multiplies a [B, 3, 3] transform with the [B, rgb_dim] colour
            raise N.NativeError('affine_appearance needs rgb_dim == 3 (got %d): the 3x4 colour transform of nerf.py:156-158 acts on RGB' % m.rgb_dim)
        self.B, self.out, self.out_stride, self.sigma_only = B, out, out_stride, sigma_only
        self.idx, self.idx_stride, self.rows_per_ray = idx, idx_stride, rows_per_ray
        self.sh = sh_deg >= 0 and m.rgb_dim > 3 and not sigma_only
        self.sh_deg, self.sh_dirs, self.sh_dir_stride = sh_deg, sh_dirs, sh_dir_stride
        if self.sh and sh_dirs is None:
            raise N.NativeError('spherical-harmonics colour needs the ray directions')

        def lin(Y, ldy, X1, ld1, K1, X2, ld2, K2, layer, act, row_add=None):
            N.check(lib.mnr_linear(Y, ldy, X1, ld1, K1, X2, ld2, K2, layer.weight.data_ptr(), layer.weight.shape[1],
                                   layer.bias.data_ptr(), row_add, B, layer.weight.shape[0], act, st()))

        tiled = self.tiled = W % 256 == 0 and os.environ.get('MNR_NO_TGEMM') is None
        self.Sp = 0
        self.wp: Dict[str, torch.Tensor] = {}           # zero-padded weight copies of this pass (tiled layers with odd inputs)

        def tlin(Y: torch.Tensor, phases, name: str, layer, splits, relu: int) -> bool:
            """Y = act([phases] . W^T + b) on the tiled GEMM; False when its alignment rules do not hold."""
            n = layer.weight.shape[0]
            wp = _cached_padded_weight(m, name, layer.weight, splits)
            ldw = wp.shape[1]
            if n % 256 or not _al16(Y.data_ptr(), wp.data_ptr(), layer.bias.data_ptr(), *[x.data_ptr() for x, _, _ in phases]):
                return False
            g = N.TGemm()
            off = 0
            for p, (X, ldx, K) in enumerate(phases):
                g.a[p], g.lda[p], g.b[p], g.ldb[p], g.k[p] = X.data_ptr(), ldx, wp.data_ptr() + off * _F4, ldw, K
                off += K
            g.n_phases, g.b_kslow, g.relu = len(phases), 0, relu
            g.c, g.ldc, g.m, g.n, g.bias = Y.data_ptr(), Y.stride(0), B, n, layer.bias.data_ptr()
            N.check(lib.mnr_tgemm_run(C.byref(g), st()))
            if wp.data_ptr() != layer.weight.data_ptr():
                self.wp[name] = wp
            return True

        Ep = self.Ep = _pad32(E) if tiled else E          # embedding row pitch (zero-padded to whole K tiles)
        emb = torch.empty(B, Ep, device=dev)
        if Ep > E:
            emb[:, E:].zero_()
        N.check(lib.mnr_embed(emb.data_ptr(), Ep, xyz.data_ptr(), xyz_stride, D, m.pos_xyz_dim, 1, B, st()))
        # 512-wide default architectures (Building) in training: the whole forward is ONE launch of the wavefront-pair kernel
        # (csrc/mlp_fwd_pair.hip, TRAIN) that also writes every layer's output as a dense [rows][width] plane -- exactly the tensors the
        # tiled data-gradient / weight-gradient launches of backward() read; only the two zero-padded side inputs are built here
        if (keep and tiled and not sigma_only and not self.sh and not self.affine and dir_rows == rows_per_ray and idx is not None and
                getattr(m, 'is_wide_default_arch', lambda: False)() and os.environ.get('MNR_NO_PAIR_KERNEL') is None):
            self._fused_forward(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, out, out_stride, sigma_noise, emb)
            return
        hs = []
        ping = [torch.empty(B, W, device=dev), torch.empty(B, W, device=dev)] if not keep else None
        cur = None
        for i, enc in enumerate(m.xyz_encodings):
            h = torch.empty(B, W, device=dev) if keep else ping[i & 1]
            name = 'xyz_encodings.%d.0' % i
            if i == 0:
                if not (tiled and tlin(h, [(emb, Ep, Ep)], name, enc[0], [(0, E, Ep)], 1)):
                    lin(h.data_ptr(), W, emb.data_ptr(), Ep, E, None, 0, 0, enc[0], 1)
            elif i in m.skip_layers:
                if not (tiled and tlin(h, [(emb, Ep, Ep), (cur, W, W)], name, enc[0], [(0, E, Ep), (E, W, W)], 1)):
                    lin(h.data_ptr(), W, emb.data_ptr(), Ep, E, cur.data_ptr(), W, W, enc[0], 1)
            else:
                if not (tiled and tlin(h, [(cur, W, W)], name, enc[0], [(0, W, W)], 1)):
                    lin(h.data_ptr(), W, cur.data_ptr(), W, W, None, 0, 0, enc[0], 1)
            hs.append(h)
            cur = h
        h = cur
        # raw head outputs: straight into ``out`` unless an SH epilogue follows (then [B, rgb_dim + 1] coefficients)
        if self.sh:
            head = torch.empty(B, m.rgb_dim + 1, device=dev)
            hp, hs_ = head.data_ptr(), m.rgb_dim + 1
        else:
            head, hp, hs_ = None, out.data_ptr(), out_stride
            if not sigma_only and m.rgb_dim + 1 > out_stride:
                raise N.NativeError('output rows are too narrow for rgb_dim {} (use the SH epilogue)'.format(m.rgb_dim))
        sig_col = 0 if sigma_only else m.rgb_dim
        lin(hp + sig_col * _F4, hs_, h.data_ptr(), W, W, None, 0, 0, m.sigma, _act_sigma(m),
            sigma_noise.data_ptr() if sigma_noise is not None else None)
        f = side = dact = None
        if not sigma_only:
            rgb_act = 2 if (m.rgb_dim == 3 and not self.affine) else 0
            raw = table = None
            if self.affine:
                # A = affine(embedding_a.weight) for every appearance index; the rgb layer writes its raw output beside
                raw = torch.empty(B, 3, device=dev)
                table = torch.empty(m.appearance_count, 12, device=dev)
                N.check(lib.mnr_linear(table.data_ptr(), 12, m.embedding_a.weight.data_ptr(), m.appearance_dim, m.appearance_dim, None, 0, 0,
                                       m.affine.weight.data_ptr(), m.appearance_dim, m.affine.bias.data_ptr(), None,
                                       m.appearance_count, 12, 0, st()))
                hp, hs_ = raw.data_ptr(), 3
            if m.has_final:
                f = torch.empty(B, W, device=dev)
                if not (tiled and tlin(f, [(h, W, W)], 'xyz_encoding_final', m.xyz_encoding_final, [(0, W, W)], 0)):
                    lin(f.data_ptr(), W, h.data_ptr(), W, W, None, 0, 0, m.xyz_encoding_final, 0)
                tiled_dir = tiled and (W // 2) % 256 == 0
                Sp = self.Sp = _pad32(ED + A) if tiled_dir else ED + A      # pitch of the [direction | appearance] rows
                side = torch.empty(B, max(Sp, 1), device=dev)
                if Sp > ED + A:
                    side[:, ED + A:].zero_()
                if ED:
                    N.check(lib.mnr_embed(side.data_ptr(), Sp, dirs.data_ptr(), dir_stride, 3, m.pos_dir_dim, dir_rows, B, st()))
                if A:
                    N.check(lib.mnr_gather_rows(side.data_ptr() + ED * _F4, Sp, m.embedding_a.weight.data_ptr(), A,
                                                m.appearance_count, idx.data_ptr(), idx_stride,
                                                1 if idx.dtype == torch.float32 else 0, rows_per_ray, B, st()))
                dact = torch.empty(B, W // 2, device=dev)
                dl = m.dir_a_encoding[0]
                ph = [(f, W, W)] + ([(side, Sp, Sp)] if ED + A else [])
                sp = [(0, W, W)] + ([(W, ED + A, Sp)] if ED + A else [])
                if not (tiled_dir and tlin(dact, ph, 'dir_a_encoding.0', dl, sp, 1)):
                    lin(dact.data_ptr(), W // 2, f.data_ptr(), W, W, side.data_ptr() if ED + A else None, Sp, ED + A, dl, 1)
                lin(hp, hs_, dact.data_ptr(), W // 2, W // 2, None, 0, 0, m.rgb, rgb_act)
            else:
                lin(hp, hs_, h.data_ptr(), W, W, None, 0, 0, m.rgb, rgb_act)
            if self.sh:
                N.check(lib.mnr_sh_apply(out.data_ptr(), out_stride, head.data_ptr(), hs_, sh_dirs.data_ptr(), sh_dir_stride,
                                         rows_per_ray, sh_deg, B, st()))
            if self.affine:
                N.check(lib.mnr_affine_apply(out.data_ptr(), out_stride, raw.data_ptr(), 3, table.data_ptr(), m.appearance_count,
                                             idx.data_ptr(), idx_stride, 1 if idx.dtype == torch.float32 else 0, rows_per_ray, B, st()))
            self.raw, self.table = raw, table
        if keep:
            self.emb, self.hs, self.f, self.side, self.dact, self.head = emb, hs, f, side, dact, head

    def _fused_forward(self, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, out, out_stride, sigma_noise, emb) -> None:
        lib, st = N.lib(), N.stream_ptr
        m, dev = self.model, out.device
        W, L, E, ED, A, Ep = m.layer_dim, m.layers, self.E, self.ED, self.A, self.Ep
        fpr = m.tape_floats_per_row()
        tape = torch.empty(B * fpr, device=dev)
        io = m.mlp_io(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, B, out, sigma_noise)
        io.out_stride = out_stride
        m.evaluate_train(io, tape, B, 0)
        # TapeLayout (csrc/mlp_layout.h): act[0 .. L - 1] (W columns each), fin (W), dact (W / 2) -- plane p starts at float offset off_p * rows
        plane = lambda off, width: tape[off * B:(off + width) * B].view(B, width)      # noqa: E731
        hs = [plane(l * W, W) for l in range(L)]
        f, dact = plane(L * W, W), plane(L * W + W, W // 2)
        Sp = self.Sp = _pad32(ED + A)
        side = torch.empty(B, Sp, device=dev)
        if Sp > ED + A:
            side[:, ED + A:].zero_()
        N.check(lib.mnr_embed(side.data_ptr(), Sp, dirs.data_ptr(), dir_stride, 3, m.pos_dir_dim, rows_per_ray, B, st()))
        N.check(lib.mnr_gather_rows(side.data_ptr() + ED * _F4, Sp, m.embedding_a.weight.data_ptr(), A, m.appearance_count, idx.data_ptr(),
                                    idx_stride, 1 if idx.dtype == torch.float32 else 0, rows_per_ray, B, st()))
        # the zero-padded weight copies backward()'s tiled data gradients address (what tlin() would have left behind)
        for i in m.skip_layers:
            name = 'xyz_encodings.%d.0' % i
            self.wp[name] = _cached_padded_weight(m, name, m.xyz_encodings[i][0].weight, [(0, E, Ep), (E, W, W)])
        self.wp['dir_a_encoding.0'] = _cached_padded_weight(m, 'dir_a_encoding.0', m.dir_a_encoding[0].weight, [(0, W, W), (W, ED + A, Sp)])
        self.raw = self.table = None
        self.emb, self.hs, self.f, self.side, self.dact, self.head, self._tape = emb, hs, f, side, dact, None, tape

    # ------------------------------------------------------------------------------------------------------------
    def backward(self, d_out: torch.Tensor, d_out_stride: int, grads: Dict[str, torch.Tensor]) -> None:
        """Accumulate the parameter gradients for d(loss)/d(out) = ``d_out`` [B, out columns] into ``grads``
        (zero-initialised tensors shaped like the parameters, keyed by parameter name)."""
        lib, st = N.lib(), N.stream_ptr
        m, B = self.model, self.B
        if self.sigma_only:
            raise NotImplementedError('sigma_only evaluations are inference-only')
        dev = d_out.device
        W, E, ED, A = m.layer_dim, self.E, self.ED, self.A

        Ep, Sp = self.Ep, self.Sp

        def wgrad(name, col0, G, ldg, n_out, X, ldx, k_in):
            g = grads[name]
            N.check(lib.mnr_gemm(g.data_ptr() + col0 * _F4, g.shape[1], G, 1, ldg, X, 1, ldx, n_out, k_in, B, 1, 0, st()))

        def bgrad(name, G, ldg, n_out):
            N.check(lib.mnr_col_sum(grads[name].data_ptr(), G, ldg, B, n_out, st()))

        def dgrad(dX, ldx, G, ldg, n_out, layer, col0, k_in, accumulate=0):
            N.check(lib.mnr_gemm(dX, ldx, G, ldg, 1, layer.weight.data_ptr() + col0 * _F4, 1, layer.weight.shape[1], B, k_in,
                                 n_out, accumulate, 1, st()))

        # ---- tiled forms (csrc/tgemm.hip, csrc/wgrad.hip job form) ----
        jobs: list = []
        use_jobs = self.tiled and B % 32 == 0
        ws = N.wgrad_workspace(dev) if use_jobs else None

        held: list = []                        # gradient buffers that queued jobs still read

        def flush():
            if jobs:
                arr = (N.WgradJob * len(jobs))(*jobs)
                N.check(lib.mnr_wgrad_jobs(arr, len(jobs), B, ws.data_ptr(), ws.numel(), st()))
                del jobs[:]

        def wgrad_t(wname, bname, G: torch.Tensor, n_out: int, parts) -> bool:
            """dW (+ db) of one layer as jobs: ``parts`` = (input tensor, valid columns, first gradient column); a part is a
            [B, multiple of 256] activation or a dense zero-padded [B, 32 / 64 / 96 / 128] block."""
            if not use_jobs or n_out % 256 or not _al16(G.data_ptr()) or G.stride(0) % 4:
                return False
            for X, cols, _ in parts:
                wide = X.stride(0) > 128
                if not _al16(X.data_ptr()) or X.stride(0) % 4 or (wide and cols % 256) or (not wide and X.stride(0) % 32):
                    return False
            g, gb = grads[wname], grads[bname]
            ldw = g.shape[1]
            for mh in range(n_out // 256):
                db = gb.data_ptr() + 256 * mh * _F4
                for X, cols, col0 in parts:
                    ldx = X.stride(0)
                    for nh in range(cols // 256 if ldx > 128 else 1):
                        j = N.WgradJob()
                        j.dz, j.ldz = G.data_ptr() + 256 * mh * _F4, G.stride(0)
                        j.in_, j.ldin = X.data_ptr() + 256 * nh * _F4, ldx
                        j.in_cols, j.in_block = (256, 256) if ldx > 128 else (cols, ldx)
                        j.dw, j.ldw = g.data_ptr() + (256 * mh * ldw + col0 + 256 * nh) * _F4, ldw
                        j.db, db = db, None
                        if len(jobs) == N.WGRAD_MAX_JOBS:
                            flush()
                        jobs.append(j)
            return True

        def dgrad_t(dX: torch.Tensor, G: torch.Tensor, n_out: int, wt: torch.Tensor, col0: int, k_in: int,
                    gate: Optional[torch.Tensor] = None, r1: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> bool:
            """dX = gate( G . wt[:, col0:col0 + k_in] (+ r1_row r1_col^T) ) on the tiled GEMM; False if its rules do not hold."""
            ldw = wt.stride(0)
            ptrs = [dX.data_ptr(), G.data_ptr(), wt.data_ptr() + col0 * _F4] + ([gate.data_ptr()] if gate is not None else []) \
                + ([r1[1].data_ptr()] if r1 is not None else [])
            if not self.tiled or k_in % 256 or n_out % 32 or ldw % 4 or not _al16(*ptrs):
                return False
            g = N.TGemm()
            g.a[0], g.lda[0], g.b[0], g.ldb[0], g.k[0] = G.data_ptr(), G.stride(0), wt.data_ptr() + col0 * _F4, ldw, n_out
            g.n_phases, g.b_kslow = 1, 1
            g.c, g.ldc, g.m, g.n = dX.data_ptr(), dX.stride(0), B, k_in
            if gate is not None:
                g.gate, g.ldgate = gate.data_ptr(), gate.stride(0)
            if r1 is not None:
                g.r1_row, g.r1_stride, g.r1_col = r1[0].data_ptr(), r1[0].stride(0), r1[1].data_ptr()
            N.check(lib.mnr_tgemm_run(C.byref(g), st()))
            return True

        if self.sh:
            C1 = m.rgb_dim + 1
            d_head = torch.empty(B, C1, device=dev)
            N.check(lib.mnr_sh_backward(d_head.data_ptr(), C1, d_out.data_ptr(), d_out_stride, self.out.data_ptr(), self.out_stride,
                                        self.sh_dirs.data_ptr(), self.sh_dir_stride, self.rows_per_ray, self.sh_deg, B, st()))
            dh_p, dh_s, y_p, y_s = d_head.data_ptr(), C1, self.head.data_ptr(), C1
        else:
            dh_p, dh_s, y_p, y_s = d_out.data_ptr(), d_out_stride, self.out.data_ptr(), self.out_stride
        h_last = self.hs[-1]
        # rgb head
        C_ = m.rgb_dim
        g_rgb = torch.empty(B, C_, device=dev)
        if self.affine:
            # adjoint of the colour transform + sigmoid: d(raw rgb), and the per-row derivative with respect to its 3x4 matrix
            cnt, AD = m.appearance_count, m.appearance_dim
            d_rows = torch.empty(B, 12, device=dev)
            isf = 1 if self.idx.dtype == torch.float32 else 0
            N.check(lib.mnr_affine_backward(g_rgb.data_ptr(), 3, d_rows.data_ptr(), dh_p, dh_s, y_p, y_s, self.raw.data_ptr(), 3,
                                            self.table.data_ptr(), cnt, self.idx.data_ptr(), self.idx_stride, isf, self.rows_per_ray, B, st()))
            d_table = torch.zeros(cnt, 12, device=dev)
            N.check(lib.mnr_scatter_rows(d_table.data_ptr(), 12, cnt, self.idx.data_ptr(), self.idx_stride, isf, self.rows_per_ray,
                                         d_rows.data_ptr(), 12, B, st()))
            ew, aw = m.embedding_a.weight, m.affine.weight
            gw = grads['affine.weight']                   # d W_aff [12][AD] += d_table^T . embedding_a.weight
            N.check(lib.mnr_gemm(gw.data_ptr(), AD, d_table.data_ptr(), 1, 12, ew.data_ptr(), 1, AD, 12, AD, cnt, 1, 0, st()))
            N.check(lib.mnr_col_sum(grads['affine.bias'].data_ptr(), d_table.data_ptr(), 12, cnt, 12, st()))
            ge = grads['embedding_a.weight']              # d embedding_a [cnt][AD] += d_table . W_aff
            N.check(lib.mnr_gemm(ge.data_ptr(), AD, d_table.data_ptr(), 12, 1, aw.data_ptr(), 1, AD, cnt, AD, 12, 1, 1, st()))
        else:
            N.check(lib.mnr_act_grad(g_rgb.data_ptr(), C_, dh_p, dh_s, y_p, y_s, B, C_, 2 if C_ == 3 else 0, st()))
        src, k_src = (self.dact, W // 2) if m.has_final else (h_last, W)
        wgrad('rgb.weight', 0, g_rgb.data_ptr(), C_, C_, src.data_ptr(), k_src, k_src)
        bgrad('rgb.bias', g_rgb.data_ptr(), C_, C_)
        d_src = torch.empty(B, k_src, device=dev)
        dgrad(d_src.data_ptr(), k_src, g_rgb.data_ptr(), C_, C_, m.rgb, 0, k_src)
        gated = False                          # d_h already carries the ReLU adjoint of the last trunk layer
        if m.has_final:
            H2 = W // 2
            dl = m.dir_a_encoding[0]
            N.check(lib.mnr_act_grad(d_src.data_ptr(), H2, d_src.data_ptr(), H2, self.dact.data_ptr(), H2, B, H2, 1, st()))
            parts = [(self.f, W, 0)] + ([(self.side, ED + A, W)] if ED + A else [])
            if not (Sp % 32 == 0 and wgrad_t('dir_a_encoding.0.weight', 'dir_a_encoding.0.bias', d_src, H2, parts)):
                wgrad('dir_a_encoding.0.weight', 0, d_src.data_ptr(), H2, H2, self.f.data_ptr(), W, W)
                if ED + A:
                    wgrad('dir_a_encoding.0.weight', W, d_src.data_ptr(), H2, H2, self.side.data_ptr(), Sp, ED + A)
                bgrad('dir_a_encoding.0.bias', d_src.data_ptr(), H2, H2)
            if A:
                d_app = torch.empty(B, A, device=dev)
                dgrad(d_app.data_ptr(), A, d_src.data_ptr(), H2, H2, dl, W + ED, A)
                N.check(lib.mnr_scatter_rows(grads['embedding_a.weight'].data_ptr(), A, m.appearance_count, self.idx.data_ptr(),
                                             self.idx_stride, 1 if self.idx.dtype == torch.float32 else 0, self.rows_per_ray,
                                             d_app.data_ptr(), A, B, st()))
            d_f = torch.empty(B, W, device=dev)
            if not dgrad_t(d_f, d_src, H2, self.wp.get('dir_a_encoding.0', dl.weight.detach()), 0, W):
                dgrad(d_f.data_ptr(), W, d_src.data_ptr(), H2, H2, dl, 0, W)
            if not wgrad_t('xyz_encoding_final.weight', 'xyz_encoding_final.bias', d_f, W, [(h_last, W, 0)]):
                wgrad('xyz_encoding_final.weight', 0, d_f.data_ptr(), W, W, h_last.data_ptr(), W, W)
                bgrad('xyz_encoding_final.bias', d_f.data_ptr(), W, W)
            # sigma head (its data gradient is rank 1: d(sigma pre-activation) x sigma.weight)
            g_sig = torch.empty(B, 1, device=dev)
            N.check(lib.mnr_act_grad(g_sig.data_ptr(), 1, dh_p + C_ * _F4, dh_s, y_p + C_ * _F4, y_s, B, 1, _act_sigma(m), st()))
            wgrad('sigma.weight', 0, g_sig.data_ptr(), 1, 1, h_last.data_ptr(), W, W)
            bgrad('sigma.bias', g_sig.data_ptr(), 1, 1)
            d_h = torch.empty(B, W, device=dev)
            if dgrad_t(d_h, d_f, W, m.xyz_encoding_final.weight.detach(), 0, W, gate=h_last, r1=(g_sig, m.sigma.weight.detach())):
                gated = True
            else:
                dgrad(d_h.data_ptr(), W, d_f.data_ptr(), W, W, m.xyz_encoding_final, 0, W)
                dgrad(d_h.data_ptr(), W, g_sig.data_ptr(), 1, 1, m.sigma, 0, W, accumulate=1)
            held.append(d_f)                   # pending jobs read d_f: kept alive until they are flushed
        else:
            d_h = d_src
            g_sig = torch.empty(B, 1, device=dev)
            N.check(lib.mnr_act_grad(g_sig.data_ptr(), 1, dh_p + C_ * _F4, dh_s, y_p + C_ * _F4, y_s, B, 1, _act_sigma(m), st()))
            wgrad('sigma.weight', 0, g_sig.data_ptr(), 1, 1, h_last.data_ptr(), W, W)
            bgrad('sigma.bias', g_sig.data_ptr(), 1, 1)
            dgrad(d_h.data_ptr(), W, g_sig.data_ptr(), 1, 1, m.sigma, 0, W, accumulate=1)
        # trunk
        for i in range(m.layers - 1, -1, -1):
            name = 'xyz_encodings.%d.0' % i
            enc = m.xyz_encodings[i][0]
            if not gated:
                N.check(lib.mnr_act_grad(d_h.data_ptr(), W, d_h.data_ptr(), W, self.hs[i].data_ptr(), W, B, W, 1, st()))
            has_emb = i == 0 or i in m.skip_layers
            parts = ([(self.emb, E, 0)] if has_emb else []) + ([(self.hs[i - 1], W, E if has_emb else 0)] if i > 0 else [])
            if not (Ep % 32 == 0 and wgrad_t(name + '.weight', name + '.bias', d_h, W, parts)):
                if has_emb:
                    wgrad(name + '.weight', 0, d_h.data_ptr(), W, W, self.emb.data_ptr(), Ep, E)
                if i > 0:
                    wgrad(name + '.weight', E if has_emb else 0, d_h.data_ptr(), W, W, self.hs[i - 1].data_ptr(), W, W)
                bgrad(name + '.bias', d_h.data_ptr(), W, W)
            if i > 0:
                # every layer's dZ gets its own buffer (288 GB of HBM: ten of them are 4 GB at the benchmark's 196 608 rows), so the
                # weight-gradient jobs of several layers go out as ONE launch (24 jobs per table) instead of one launch + reduction per
                # layer -- recycling two buffers meant flushing the job table before each reuse
                held.append(d_h)
                nxt = torch.empty(B, W, device=dev)
                wt, col0 = (self.wp[name], Ep) if name in self.wp else (enc.weight.detach(), E if has_emb else 0)
                gated = dgrad_t(nxt, d_h, W, wt, col0, W, gate=self.hs[i - 1])
                if not gated:
                    dgrad(nxt.data_ptr(), W, d_h.data_ptr(), W, W, enc, E if has_emb else 0, W)
                d_h = nxt
        flush()
        del held[:]
