"""NeRF MLP module -- API/state_dict-compatible with the reference (mega_nerf/models/nerf.py:45-160)
but evaluated by the fused gfx950 kernel (csrc/mlp_fwd.hip) through the C ABI.

Parameter names (``xyz_encodings.{i}.0.*``, ``embedding_a.weight``, ``xyz_encoding_final.*``,
``dir_a_encoding.0.*``, ``sigma.*``, ``rgb.*``) are those of the reference checkpoints
(runner.py:521-536), so ``load_state_dict`` of a reference checkpoint works unchanged.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
from torch import nn

from mega_nerf import _native as N


def _off(t: Optional[torch.Tensor], elements: int) -> Optional[torch.Tensor]:
    """Flat view of ``t``'s storage starting ``elements`` items after its first element (raw-buffer addressing)."""
    if t is None:
        return None
    return t.as_strided((max(t.untyped_storage().nbytes() // t.element_size() - t.storage_offset() - elements, 0),), (1,),
                        t.storage_offset() + elements)


class ShiftedSoftplus(nn.Module):
    """softplus(x - 1) (reference nerf.py:28-39); selects sigma_activation = 1 in the kernel."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # only used by host-side tooling
        return torch.nn.functional.softplus(x - 1, 1, 20)


def _linear_act(fin: int, fout: int) -> nn.Sequential:
    # the Sequential wrapper only exists to reproduce the checkpoint key "<name>.0.weight"
    return nn.Sequential(nn.Linear(fin, fout), nn.ReLU(True))


class NullTape:
    """Tape of an evaluation over zero rows."""

    def backward(self, d_out: torch.Tensor, d_out_stride: int, grads: dict) -> None:
        return None


def _version_of(t: torch.Tensor) -> int:
    """Version counter of a tensor for cache keys; tensors created under ``torch.inference_mode()`` do not track one
    (reading it raises) and cannot be updated in place either, so a constant is exact for them."""
    return -1 if t.is_inference() else t._version


class FusedTape:
    """One fused training-mode MLP launch (activation tape in HBM) and its hand-written adjoint
    (csrc/mlp_fwd.hip TRAIN variants, csrc/mlp_bwd.hip).  Spherical-harmonics models (rgb_dim > 3, ``sh_deg`` >= 0): the
    colour epilogue and the rgb layer are differentiated here with the small stand-alone kernels (mnr_sh_backward,
    mnr_gemm, mnr_col_sum) and the fused chain picks up at the output of dir_a_encoding (``dd_in``)."""

    def __init__(self, model: 'NeRF', xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out,
                 sigma_noise, n_units_dev, rows_per_unit, sh_deg: int = -1):
        self.model, self.n_rows, self.out = model, n_rows, out
        self.idx, self.idx_stride, self.rows_per_ray = idx, idx_stride, rows_per_ray
        self.n_units_dev, self.rows_per_unit = n_units_dev, rows_per_unit
        self.sh_deg, self.dirs, self.dir_stride = sh_deg, dirs, dir_stride
        self.tape_rows = max(n_rows, 1)
        self.tape = torch.empty(self.tape_rows * model.tape_floats_per_row(), device=out.device, dtype=torch.float32)
        io = model.mlp_io(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out, sigma_noise,
                          n_units_dev, rows_per_unit)
        io.apply_sh_deg = sh_deg
        model.evaluate_train(io, self.tape, self.tape_rows, 0)

    def _colour_head_backward(self, d_out: torch.Tensor, d_out_stride: int, grads: dict) -> torch.Tensor:
        """SH models: d_out -> (rgb.* gradients, dL/d(dir_a output) [n_rows][W/2])."""
        m, dev, lib, st = self.model, d_out.device, N.lib(), N.stream_ptr
        rows = self.n_rows if self.n_units_dev is None else min(self.n_rows, int(self.n_units_dev.item()) * self.rows_per_unit)
        half, n_coef, C1 = m.layer_dim // 2, m.rgb_dim, m.rgb_dim + 1
        dd = torch.zeros(self.tape_rows, half, device=dev, dtype=torch.float32)
        if rows == 0:
            return dd
        d_coef = torch.empty(rows, C1, device=dev, dtype=torch.float32)
        N.check(lib.mnr_sh_backward(d_coef.data_ptr(), C1, d_out.data_ptr(), d_out_stride, self.out.data_ptr(), self.out.stride(0),
                                    self.dirs.data_ptr(), self.dir_stride, self.rows_per_ray, self.sh_deg, rows, st()))
        w = m.rgb.weight
        N.check(lib.mnr_gemm(dd.data_ptr(), half, d_coef.data_ptr(), C1, 1, w.data_ptr(), 1, w.shape[1], rows, half, n_coef, 0, 1, st()))
        desc = m.model_desc()
        dact = self.tape.data_ptr() + int(lib.mnr_tape_plane_offset(C.byref(desc), 0)) * self.tape_rows * 4
        gw = grads['rgb.weight']
        N.check(lib.mnr_gemm(gw.data_ptr(), gw.shape[1], d_coef.data_ptr(), 1, C1, dact, 1, half, n_coef, half, rows, 1, 0, st()))
        N.check(lib.mnr_col_sum(grads['rgb.bias'].data_ptr(), d_coef.data_ptr(), C1, rows, n_coef, st()))
        self._keep = d_coef
        return dd

    def backward(self, d_out: torch.Tensor, d_out_stride: int, grads: dict) -> None:
        m, dev = self.model, d_out.device
        if self.n_rows == 0:
            return
        desc, packed = m.packed()
        packed_bwd = m.packed_bwd()
        gtape = torch.empty(self.tape.numel(), device=dev, dtype=torch.float32)
        dheads = torch.empty(self.n_rows, 4, device=dev, dtype=torch.float32)
        counter = torch.zeros(1, device=dev, dtype=torch.int32)
        g = N.MlpGradIO()
        g.tape, g.gtape, g.tape_rows, g.tape_row0 = self.tape.data_ptr(), gtape.data_ptr(), self.tape_rows, 0
        g.d_out, g.d_out_stride = d_out.data_ptr(), d_out_stride
        g.out, g.out_stride = self.out.data_ptr(), self.out.stride(0)
        g.dheads = dheads.data_ptr()
        if self.idx is not None:
            g.idx, g.idx_stride = self.idx.data_ptr(), self.idx_stride
            g.idx_is_float = 1 if self.idx.dtype == torch.float32 else 0
        g.rows_per_ray = self.rows_per_ray
        g.n_rows = self.n_rows
        g.n_units_dev = self.n_units_dev.data_ptr() if self.n_units_dev is not None else None
        g.rows_per_unit = self.rows_per_unit
        g.work_counter = counter.data_ptr()
        g.grad = m.grad_struct(grads)
        dd = None
        if m.rgb_dim > 3:
            dd = self._colour_head_backward(d_out, d_out_stride, grads)
            g.dd_in = dd.data_ptr()
        N.check(N.lib().mnr_mlp_backward_data(packed.data_ptr(), packed_bwd.data_ptr(), C.byref(desc), C.byref(g), N.stream_ptr()))
        N.check(N.lib().mnr_mlp_backward_weights(C.byref(desc), C.byref(g), N.stream_ptr()))


class NeRF(nn.Module):
    prefer_wide_layerwise = True      # evaluate(): layer_dim >= 512 takes the tiled per-layer GEMMs for large launches (diagnostics flip it)

    def __init__(self, pos_xyz_dim: int, pos_dir_dim: int, layers: int, skip_layers: List[int], layer_dim: int,
                 appearance_dim: int, affine_appearance: bool, appearance_count: int, rgb_dim: int, xyz_dim: int,
                 sigma_activation: nn.Module):
        super().__init__()
        if rgb_dim > 3:
            assert pos_dir_dim == 0
        self.xyz_dim, self.pos_xyz_dim, self.pos_dir_dim = xyz_dim, pos_xyz_dim, pos_dir_dim
        self.layers, self.skip_layers, self.layer_dim = layers, list(skip_layers), layer_dim
        self.appearance_dim, self.appearance_count, self.rgb_dim = appearance_dim, appearance_count, rgb_dim
        in_xyz = xyz_dim * (1 + 2 * pos_xyz_dim)
        in_dir = 3 * (1 + 2 * pos_dir_dim) if pos_dir_dim > 0 else 0
        self.xyz_encodings = nn.ModuleList(
            _linear_act(in_xyz if i == 0 else layer_dim + (in_xyz if i in self.skip_layers else 0), layer_dim)
            for i in range(layers))
        self.embedding_a = nn.Embedding(appearance_count, appearance_dim) if appearance_dim > 0 else None
        if affine_appearance:
            assert appearance_dim > 0
            self.affine = nn.Linear(appearance_dim, 12)
        else:
            self.affine = None
        self.has_dir = pos_dir_dim > 0
        self.has_final = self.has_dir or (appearance_dim > 0 and not affine_appearance)
        if self.has_final:
            self.xyz_encoding_final = nn.Linear(layer_dim, layer_dim)
            self.dir_a_encoding = _linear_act(
                layer_dim + in_dir + (appearance_dim if not affine_appearance else 0), layer_dim // 2)
        else:
            self.xyz_encoding_final = None
        self.sigma = nn.Linear(layer_dim, 1)
        self.sigma_activation = sigma_activation
        self.rgb = nn.Linear(layer_dim // 2 if self.has_final else layer_dim, rgb_dim)
        self.mfma_tile = 0          # 0 = auto; 16 / 32 samples per wavefront (see include/mnr_api.h)
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None

    # ---- native plumbing ---------------------------------------------------------------------
    def _all_params(self):
        return [p for p in self.parameters()]

    def model_desc(self) -> N.ModelDesc:
        d = N.ModelDesc()
        d.xyz_dim, d.pos_xyz_dim, d.pos_dir_dim, d.layers = self.xyz_dim, self.pos_xyz_dim, self.pos_dir_dim, self.layers
        d.skip_mask = sum(1 << i for i in self.skip_layers)
        d.layer_dim, d.appearance_dim = self.layer_dim, self.appearance_dim
        d.appearance_count, d.rgb_dim = self.appearance_count, self.rgb_dim
        d.sigma_activation = 1 if isinstance(self.sigma_activation, ShiftedSoftplus) else 0
        d.mfma_tile = self.mfma_tile
        for i, enc in enumerate(self.xyz_encodings):
            d.layer_w[i], d.layer_b[i] = enc[0].weight.data_ptr(), enc[0].bias.data_ptr()
        if self.has_final:
            d.final_w, d.final_b = self.xyz_encoding_final.weight.data_ptr(), self.xyz_encoding_final.bias.data_ptr()
            d.dir_a_w, d.dir_a_b = self.dir_a_encoding[0].weight.data_ptr(), self.dir_a_encoding[0].bias.data_ptr()
        d.sigma_w, d.sigma_b = self.sigma.weight.data_ptr(), self.sigma.bias.data_ptr()
        d.rgb_w, d.rgb_b = self.rgb.weight.data_ptr(), self.rgb.bias.data_ptr()
        if self.embedding_a is not None:
            d.embedding_a = self.embedding_a.weight.data_ptr()
        return d

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() replace the parameter tensors: drop everything derived from the old storage
        self._param_cache = None
        self._packed_key = self._packed_bwd_key = None
        self._fused_ok = self._fused_train_ok = None
        return super()._apply(fn, *args, **kwargs)

    def weights_changed(self) -> None:
        """Tell the packed-weight caches that the parameters were updated by something that does not bump their version
        counters (``torch.optim.Adam(fused=True)`` does not)."""
        self._packed_key = self._packed_bwd_key = self._packed_h2_key = None
        self._weights_epoch = getattr(self, '_weights_epoch', 0) + 1          # models/layerwise.py: padded weight copies

    def packed(self):
        """(desc, packed device buffer); re-packs when any parameter changed (in-place updates bump ``_version``; storage
        replacement goes through ``_apply`` / ``load_state_dict`` and is caught by the pointer check).  The steady-state
        cost is one tuple of 25 version counters -- this runs once per MLP launch, eight cells x four passes per routed
        render, so it is kept off the per-parameter slow path."""
        cache = getattr(self, '_param_cache', None)
        if cache is not None:
            params = cache[0]
            ptrs = tuple([p.data_ptr() for p in params])
            if ptrs != cache[1] or cache[3] != self.mfma_tile:      # storage swapped / tile changed: rebuild the descriptor
                cache = None
        if cache is None:
            params = self._all_params()
            for p in params:
                N.require_device(p, 'NeRF parameter')
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise N.NativeError('NeRF parameters must be contiguous float32')
            ptrs = tuple([p.data_ptr() for p in params])
            cache = self._param_cache = (params, ptrs, self.model_desc(), self.mfma_tile)
            self._packed_key = self._packed_bwd_key = None
        desc = cache[2]
        key = tuple([_version_of(p) for p in params])
        if self._packed is None or key != self._packed_key:
            nbytes = N.lib().mnr_packed_model_bytes(C.byref(desc))
            if nbytes == 0:
                raise N.NativeError(N.lib().mnr_last_error().decode())
            if self._packed is None or self._packed.numel() != nbytes or self._packed.device != params[0].device:
                self._packed = torch.empty(nbytes, dtype=torch.uint8, device=params[0].device)
            N.check(N.lib().mnr_pack_model(self._packed.data_ptr(), nbytes, C.byref(desc), N.stream_ptr()))
            self._packed_key = key
        return desc, self._packed

    def packed_h2(self):
        """(desc, weight image of the split-precision forward, csrc/mlp_fwd_h2.hip); cached like :meth:`packed`."""
        desc, _ = self.packed()
        key = self._packed_key
        if getattr(self, '_packed_h2', None) is None or getattr(self, '_packed_h2_key', None) != key:
            nbytes = N.lib().mnr_packed_model_h2_bytes(C.byref(desc))
            if nbytes == 0:
                raise N.NativeError(N.lib().mnr_last_error().decode())
            if getattr(self, '_packed_h2', None) is None or self._packed_h2.numel() != nbytes:
                self._packed_h2 = torch.empty(nbytes, dtype=torch.uint8, device=self._packed.device)
            N.check(N.lib().mnr_pack_model_h2(self._packed_h2.data_ptr(), nbytes, C.byref(desc), N.stream_ptr()))
            self._packed_h2_key = key
        return desc, self._packed_h2

    def evaluate(self, xyz: torch.Tensor, xyz_stride: int, dirs: Optional[torch.Tensor], dir_stride: int,
                 idx: Optional[torch.Tensor], idx_stride: int, rows_per_ray: int, n_rows: int, out: torch.Tensor,
                 sigma_noise: Optional[torch.Tensor] = None, sigma_only: bool = False, apply_sh_deg: int = -1,
                 n_units_dev: Optional[torch.Tensor] = None, rows_per_unit: int = 0) -> torch.Tensor:
        """Enqueue one fused MLP launch on the current stream (no host sync).  All tensors are raw device
        buffers; see ``mnr_mlp_io`` in include/mnr_api.h for the row/ray addressing."""
        # layer_dim >= 512: once a launch fills the chip the tiled per-layer GEMMs (csrc/tgemm.hip, 118 TFLOP/s at 196 608
        # rows of the 8 x 512 model) beat the one-wavefront-per-SIMD register-chained kernel (100).  The 512-wide DEFAULT architectures
        # (Building) have the wavefront-pair kernel (csrc/mlp_fwd_pair.hip: two wavefronts per SIMD, whole render at 0.82 of the
        # fp32-MFMA peak against 0.70 through the tiled GEMMs) and stay on the fused path at every launch size.
        wide = (self.prefer_wide_layerwise and self.layer_dim >= 512 and self.layer_dim % 256 == 0 and n_units_dev is None and
                n_rows >= 65536 and os.environ.get('MNR_NO_TGEMM') is None and
                not (self.is_wide_default_arch() and os.environ.get('MNR_NO_PAIR_KERNEL') is None))
        if wide or not self.fused_supported():
            return self._evaluate_layerwise(xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out,
                                            sigma_noise, sigma_only, apply_sh_deg, n_units_dev, rows_per_unit)
        desc, packed = self.packed()
        io = N.MlpIO()
        io.xyz, io.xyz_stride = xyz.data_ptr(), xyz_stride
        io.dir, io.dir_stride = (dirs.data_ptr() if dirs is not None else None), dir_stride
        if idx is not None:
            if idx.dtype == torch.float32:
                io.idx_is_float = 1
            elif idx.dtype == torch.int32:
                io.idx_is_float = 0
            else:
                raise N.NativeError('image indices must be float32 or int32 (got {})'.format(idx.dtype))
            io.idx, io.idx_stride = idx.data_ptr(), idx_stride
        io.rows_per_ray = rows_per_ray
        io.sigma_noise = sigma_noise.data_ptr() if sigma_noise is not None else None
        io.out, io.out_stride = out.data_ptr(), out.stride(0) if out.dim() > 1 else 1
        io.n_rows = n_rows
        io.n_units_dev = n_units_dev.data_ptr() if n_units_dev is not None else None
        io.rows_per_unit = rows_per_unit
        io.sigma_only = 1 if sigma_only else 0
        io.apply_sh_deg = apply_sh_deg
        N.check(N.lib().mnr_mlp_forward(packed.data_ptr(), C.byref(desc), C.byref(io), N.stream_ptr()))
        return out

    # ---- generic-width fallback ----------------------------------------------------------------
    def fused_supported(self) -> bool:
        """True if the register-chained kernel has an instantiation for this architecture (queried once)."""
        if getattr(self, '_fused_ok', None) is None:
            # affine_appearance (nerf.py:156-158) changes the colour epilogue: evaluated layer by layer (no config uses it)
            self._fused_ok = self.affine is None and bool(N.lib().mnr_fused_supported(C.byref(self.model_desc())))
        return self._fused_ok

    def _evaluate_layerwise(self, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out,
                            sigma_noise, sigma_only, apply_sh_deg, n_units_dev, rows_per_unit, dir_rows=None):
        """nerf.py:115-160 as one exact-fp32 MFMA GEMM launch per layer (models/layerwise.py, csrc/layerwise.hip); used
        for widths / architectures without a fused kernel (e.g. configs/nerf: layer_dim 2048)."""
        from mega_nerf.models.layerwise import LayerwiseTape
        if n_units_dev is not None:
            n_rows = min(n_rows, int(n_units_dev.item()) * rows_per_unit)      # the fallback sizes launches on the host
        if n_rows == 0:
            return out
        ostride = out.stride(0) if out.dim() > 1 else 1
        # rows per pass: the 128 x 128 kernels like chunks whose activations stay in the 256 MB Infinity Cache between layers;
        # the tiled GEMM (widths that are multiples of 256) wants >= 256 output tiles of 256 rows per launch -- one per CU --
        # and is not HBM-bound, so it takes up to 1 GB of activations per layer (288 GB of HBM: no reason to go small)
        tiled = self.layer_dim % 256 == 0 and os.environ.get('MNR_NO_TGEMM') is None
        target = max(32768, (1 << 28) // self.layer_dim) if tiled else 32768
        chunk = max(rows_per_ray, (target // rows_per_ray) * rows_per_ray)
        dir_rows = rows_per_ray if dir_rows is None else dir_rows
        sh = apply_sh_deg >= 0 and self.rgb_dim > 3 and not sigma_only
        mlp_dirs = dirs if self.has_dir else None
        for r0 in range(0, n_rows, chunk):
            B = min(chunk, n_rows - r0)
            ray0 = r0 // rows_per_ray
            LayerwiseTape(self, _off(xyz, r0 * xyz_stride), xyz_stride,
                          _off(mlp_dirs, (r0 // dir_rows) * dir_stride), dir_stride, dir_rows,
                          _off(idx, ray0 * idx_stride), idx_stride, rows_per_ray, B, _off(out, r0 * ostride), ostride,
                          _off(sigma_noise, r0), sigma_only, apply_sh_deg, False,
                          _off(dirs, ray0 * dir_stride) if sh else None, dir_stride)
        return out

    def is_default_arch(self) -> bool:
        """True for the reference's default foreground / background architectures (configs/mega-nerf/*.yaml: 8 x 256, 12 / 4
        frequency bands, 48-d appearance, skip at 4, rgb head) -- the pair the multi-segment launches are instantiated for."""
        return (self.xyz_dim in (3, 4) and self.pos_xyz_dim == 12 and self.pos_dir_dim == 4 and self.layers == 8 and
                list(self.skip_layers) == [4] and self.layer_dim == 256 and self.appearance_dim == 48 and self.rgb_dim == 3 and
                self.embedding_a is not None and self.affine is None and self.mfma_tile in (0, 16))

    def is_wide_default_arch(self) -> bool:
        """True for the default architectures at 512 channels (README "Larger models", configs/mega-nerf Building): the shapes
        k_mlp_fwd_pair is instantiated for."""
        return (self.xyz_dim in (3, 4) and self.pos_xyz_dim == 12 and self.pos_dir_dim == 4 and self.layers == 8 and
                list(self.skip_layers) == [4] and self.layer_dim == 512 and self.appearance_dim == 48 and self.rgb_dim == 3 and
                self.embedding_a is not None and self.affine is None and self.mfma_tile in (0, 16))

    def is_sh_arch(self, sh_deg: int = 2) -> bool:
        """True for the default architectures in their spherical-harmonics form (configs/mega-nerf-sh-3/*.yaml: sh_deg 2, pos_dir_dim 0
        -- 27 colour coefficients, no direction encoding; sh_deg 3 -- 48 coefficients -- is the degree BASELINE.json words): the further
        pairs the multi-segment launches are instantiated for."""
        return (sh_deg in (2, 3) and self.xyz_dim in (3, 4) and self.pos_xyz_dim == 12 and self.pos_dir_dim == 0 and self.layers == 8 and
                list(self.skip_layers) == [4] and self.layer_dim == 256 and self.appearance_dim == 48 and self.rgb_dim == 3 * (sh_deg + 1) ** 2 and
                self.embedding_a is not None and self.affine is None and self.mfma_tile in (0, 16))

    def is_sh2_arch(self) -> bool:
        return self.is_sh_arch(2)

    def fused_train_supported(self) -> bool:
        """True if the fused training kernels (activation tape + hand-written backward) cover this architecture."""
        if getattr(self, '_fused_train_ok', None) is None:
            self._fused_train_ok = self.affine is None and bool(N.lib().mnr_fused_train_supported(C.byref(self.model_desc())))
        return self._fused_train_ok

    def train_eval(self, xyz, xyz_stride, dirs, dir_stride, dir_rows, idx, idx_stride, rows_per_ray, n_rows, out,
                   sigma_noise, sh_deg, n_units_dev, rows_per_unit, sh_dirs=None, sh_dir_stride=0):
        """Training-mode evaluation of ``n_rows`` rows into ``out`` [n_rows, 4]; returns a tape object whose
        ``backward(d_out, grads)`` accumulates the parameter gradients (``grads``: zero-initialised tensors keyed by
        this module's parameter names).  Fused kernels when they cover the architecture, else layer by layer."""
        sh = sh_deg >= 0 and self.rgb_dim > 3
        if self.fused_train_supported() and dir_rows == rows_per_ray and (sh or (self.rgb_dim == 3 and sh_deg < 0)):
            if sh:       # the kernel's colour epilogue reads the ray directions through the direction input
                return FusedTape(self, xyz, xyz_stride, sh_dirs, sh_dir_stride, idx, idx_stride, rows_per_ray, n_rows, out,
                                 sigma_noise, n_units_dev, rows_per_unit, sh_deg)
            return FusedTape(self, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out, sigma_noise,
                             n_units_dev, rows_per_unit)
        from mega_nerf.models.layerwise import LayerwiseTape
        B = n_rows if n_units_dev is None else min(n_rows, int(n_units_dev.item()) * rows_per_unit)
        if B == 0:
            return NullTape()
        return LayerwiseTape(self, xyz, xyz_stride, dirs if self.has_dir else None, dir_stride, dir_rows, idx, idx_stride,
                             rows_per_ray, B, out, out.stride(0), sigma_noise, False, sh_deg, True, sh_dirs, sh_dir_stride)

    def launch(self, io: 'N.MlpIO') -> None:
        """Enqueue one inference launch described by a caller-built ``mnr_mlp_io``."""
        desc, packed = self.packed()
        N.check(N.lib().mnr_mlp_forward(packed.data_ptr(), C.byref(desc), C.byref(io), N.stream_ptr()))

    # ---- training plumbing --------------------------------------------------------------------
    def packed_bwd(self):
        """Transposed weight image for the data-gradient chain (same cache key as :meth:`packed`)."""
        desc, _ = self.packed()
        key = self._packed_key
        if getattr(self, '_packed_bwd', None) is None or self._packed_bwd_key != key:
            nbytes = N.lib().mnr_packed_bwd_bytes(C.byref(desc))
            if nbytes == 0:
                raise N.NativeError(N.lib().mnr_last_error().decode())
            if getattr(self, '_packed_bwd', None) is None or self._packed_bwd.numel() != nbytes:
                self._packed_bwd = torch.empty(nbytes, dtype=torch.uint8, device=self._packed.device)
            N.check(N.lib().mnr_pack_model_bwd(self._packed_bwd.data_ptr(), nbytes, C.byref(desc), N.stream_ptr()))
            self._packed_bwd_key = key
        return self._packed_bwd

    def tape_floats_per_row(self) -> int:
        n = N.lib().mnr_tape_floats_per_row(C.byref(self.model_desc()))
        if n <= 0:
            raise N.NativeError(N.lib().mnr_last_error().decode())
        return int(n)

    def mlp_io(self, xyz, xyz_stride, dirs, dir_stride, idx, idx_stride, rows_per_ray, n_rows, out, sigma_noise=None,
               n_units_dev=None, rows_per_unit=0) -> 'N.MlpIO':
        io = N.MlpIO()
        io.xyz, io.xyz_stride = xyz.data_ptr(), xyz_stride
        io.dir, io.dir_stride = (dirs.data_ptr() if dirs is not None else None), dir_stride
        if idx is not None:
            io.idx_is_float = 1 if idx.dtype == torch.float32 else 0
            io.idx, io.idx_stride = idx.data_ptr(), idx_stride
        io.rows_per_ray = rows_per_ray
        io.sigma_noise = sigma_noise.data_ptr() if sigma_noise is not None else None
        io.out, io.out_stride = out.data_ptr(), out.stride(0)
        io.n_rows = n_rows
        io.n_units_dev = n_units_dev.data_ptr() if n_units_dev is not None else None
        io.rows_per_unit = rows_per_unit
        io.apply_sh_deg = -1
        return io

    def evaluate_train(self, io: 'N.MlpIO', tape: torch.Tensor, tape_rows: int, tape_row0: int) -> None:
        desc, packed = self.packed()
        N.check(N.lib().mnr_mlp_forward_train(packed.data_ptr(), C.byref(desc), C.byref(io), tape.data_ptr(), tape_rows,
                                              tape_row0, N.stream_ptr()))

    def grad_struct(self, grads: dict) -> 'N.ModelGrads':
        """mnr_model_grads pointing at ``grads[param_name]`` tensors (same shapes as the parameters)."""
        g = N.ModelGrads()
        for i in range(self.layers):
            g.layer_w[i] = grads['xyz_encodings.%d.0.weight' % i].data_ptr()
            g.layer_b[i] = grads['xyz_encodings.%d.0.bias' % i].data_ptr()
        g.final_w, g.final_b = grads['xyz_encoding_final.weight'].data_ptr(), grads['xyz_encoding_final.bias'].data_ptr()
        g.dir_a_w, g.dir_a_b = grads['dir_a_encoding.0.weight'].data_ptr(), grads['dir_a_encoding.0.bias'].data_ptr()
        g.sigma_w, g.sigma_b = grads['sigma.weight'].data_ptr(), grads['sigma.bias'].data_ptr()
        g.rgb_w, g.rgb_b = grads['rgb.weight'].data_ptr(), grads['rgb.bias'].data_ptr()
        if self.embedding_a is not None:
            g.embedding_a = grads['embedding_a.weight'].data_ptr()
        return g

    # ---- reference API -----------------------------------------------------------------------
    def forward(self, x: torch.Tensor, sigma_only: bool = False,
                sigma_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        expected = self.xyz_dim + (0 if (sigma_only or not self.has_dir) else 3) \
            + (0 if (sigma_only or self.embedding_a is None) else 1)
        if x.shape[1] != expected:
            raise Exception(
                'Unexpected input shape: {} (expected: {}, xyz_dim: {})'.format(x.shape, expected, self.xyz_dim))
        N.require_device(x, 'x')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from mega_nerf.autograd import mlp_forward_with_grad
            return mlp_forward_with_grad(self, x, sigma_only, sigma_noise)
        x = x.contiguous().float()
        B, ncol = x.shape
        out_cols = 1 if sigma_only else self.rgb_dim + 1
        out = torch.empty(B, out_cols, device=x.device, dtype=torch.float32)
        if B == 0:
            return out
        dirs = idx = None
        if not sigma_only:
            if self.has_dir:
                dirs = x[:, ncol - 4:] if ncol >= 4 else None    # x[:, -4:-1] (nerf.py:146, quirk Q8)
            if self.embedding_a is not None:
                idx = x[:, ncol - 1:]
        noise = sigma_noise.contiguous().float().view(-1) if sigma_noise is not None else None
        self.evaluate(x, ncol, dirs, ncol, idx, ncol, 1, B, out, noise, sigma_only)
        return out
