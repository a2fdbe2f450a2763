"""ctypes binding of libmeganerf_hip.so (C ABI declared in include/mnr_api.h).

The product path has NO fallback: if the library is missing or no HIP device is present, every
kernel entry raises.  Only raw pointers, sizes and the current HIP stream cross this boundary --
torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get('MNR_LIB', _HERE.parent / 'lib' / 'libmeganerf_hip.so'))

MNR_MAX_LAYERS = 16
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    """struct mnr_model_desc"""
    _fields_ = [
        ('xyz_dim', C.c_int32), ('pos_xyz_dim', C.c_int32), ('pos_dir_dim', C.c_int32), ('layers', C.c_int32),
        ('skip_mask', C.c_int32), ('layer_dim', C.c_int32), ('appearance_dim', C.c_int32),
        ('appearance_count', C.c_int32), ('rgb_dim', C.c_int32), ('sigma_activation', C.c_int32),
        ('mfma_tile', C.c_int32),
        ('layer_w', C.c_void_p * MNR_MAX_LAYERS), ('layer_b', C.c_void_p * MNR_MAX_LAYERS),
        ('final_w', C.c_void_p), ('final_b', C.c_void_p), ('dir_a_w', C.c_void_p), ('dir_a_b', C.c_void_p),
        ('sigma_w', C.c_void_p), ('sigma_b', C.c_void_p), ('rgb_w', C.c_void_p), ('rgb_b', C.c_void_p),
        ('embedding_a', C.c_void_p),
    ]


class MlpIO(C.Structure):
    """struct mnr_mlp_io"""
    _fields_ = [
        ('xyz', C.c_void_p), ('xyz_stride', C.c_int64),
        ('dir', C.c_void_p), ('dir_stride', C.c_int64),
        ('idx', C.c_void_p), ('idx_stride', C.c_int64),
        ('idx_is_float', C.c_int32), ('rows_per_ray', C.c_int32),
        ('sigma_noise', C.c_void_p),
        ('out', C.c_void_p), ('out_stride', C.c_int64),
        ('n_rows', C.c_int64),
        ('n_units_dev', C.c_void_p), ('rows_per_unit', C.c_int32),
        ('sigma_only', C.c_int32), ('apply_sh_deg', C.c_int32),
        ('row_index', C.c_void_p),
    ]


class CompositeIO(C.Structure):
    """struct mnr_composite_io"""
    _fields_ = [
        ('z', C.c_void_p), ('raw', C.c_void_p), ('depth_real', C.c_void_p), ('last_delta', C.c_void_p),
        ('zmax_src', C.c_void_p), ('zmax_S', C.c_int32), ('flip', C.c_int32), ('N', C.c_int64),
        ('n_units_dev', C.c_void_p), ('S', C.c_int32),
        ('weights', C.c_void_p), ('rgb', C.c_void_p), ('depth', C.c_void_p), ('depth_var', C.c_void_p),
        ('bg_lambda', C.c_void_p),
    ]


class ModelGrads(C.Structure):
    """struct mnr_model_grads"""
    _fields_ = [('layer_w', C.c_void_p * MNR_MAX_LAYERS), ('layer_b', C.c_void_p * MNR_MAX_LAYERS),
                ('final_w', C.c_void_p), ('final_b', C.c_void_p), ('dir_a_w', C.c_void_p), ('dir_a_b', C.c_void_p),
                ('sigma_w', C.c_void_p), ('sigma_b', C.c_void_p), ('rgb_w', C.c_void_p), ('rgb_b', C.c_void_p),
                ('embedding_a', C.c_void_p)]


class MlpGradIO(C.Structure):
    """struct mnr_mlp_grad_io"""
    _fields_ = [('tape', C.c_void_p), ('gtape', C.c_void_p), ('tape_rows', C.c_int64), ('tape_row0', C.c_int64),
                ('d_out', C.c_void_p), ('d_out_stride', C.c_int64), ('out', C.c_void_p), ('out_stride', C.c_int64),
                ('dheads', C.c_void_p), ('idx', C.c_void_p), ('idx_stride', C.c_int64), ('idx_is_float', C.c_int32),
                ('rows_per_ray', C.c_int32), ('n_rows', C.c_int64), ('n_units_dev', C.c_void_p),
                ('rows_per_unit', C.c_int32), ('work_counter', C.c_void_p), ('grad', ModelGrads), ('dd_in', C.c_void_p)]


class MlpLaunch(C.Structure):
    """struct mnr_mlp_launch"""
    _fields_ = [('packed_dev', C.c_void_p), ('desc', C.POINTER(ModelDesc)), ('io', C.POINTER(MlpIO)), ('tape_dev', C.c_void_p),
                ('tape_rows', C.c_int64), ('tape_row0', C.c_int64)]


class MlpCellsLaunch(C.Structure):
    """struct mnr_mlp_cells_launch"""
    _fields_ = [('desc', C.POINTER(ModelDesc)), ('cells_dev', C.c_void_p), ('n_cells', C.c_int32), ('io', C.POINTER(MlpIO))]


class MlpGradLaunch(C.Structure):
    """struct mnr_mlp_grad_launch"""
    _fields_ = [('packed_fwd_dev', C.c_void_p), ('packed_bwd_dev', C.c_void_p), ('desc', C.POINTER(ModelDesc)),
                ('io', C.POINTER(MlpGradIO))]


class WgradRegion(C.Structure):
    """struct mnr_wgrad_region"""
    _fields_ = [('desc', C.POINTER(ModelDesc)), ('tape', C.c_void_p), ('gtape', C.c_void_p), ('tape_rows', C.c_int64),
                ('n_ranges', C.c_int32), ('row0', C.c_int64 * 2), ('n_rows', C.c_int64 * 2),
                ('n_units_dev', C.c_void_p * 2), ('rows_per_unit', C.c_int32 * 2), ('grad', ModelGrads)]


class TGemm(C.Structure):
    """struct mnr_tgemm"""
    _fields_ = [('a', C.c_void_p * 2), ('lda', C.c_int64 * 2), ('b', C.c_void_p * 2), ('ldb', C.c_int64 * 2),
                ('k', C.c_int32 * 2), ('n_phases', C.c_int32), ('b_kslow', C.c_int32), ('relu', C.c_int32),
                ('c', C.c_void_p), ('ldc', C.c_int64), ('m', C.c_int64), ('n', C.c_int32), ('bias', C.c_void_p),
                ('gate', C.c_void_p), ('ldgate', C.c_int64), ('r1_row', C.c_void_p), ('r1_stride', C.c_int64),
                ('r1_col', C.c_void_p)]


class WgradJob(C.Structure):
    """struct mnr_wgrad_job"""
    _fields_ = [('dz', C.c_void_p), ('ldz', C.c_int64), ('in_', C.c_void_p), ('ldin', C.c_int64), ('in_cols', C.c_int32),
                ('in_block', C.c_int32), ('dw', C.c_void_p), ('ldw', C.c_int64), ('db', C.c_void_p)]


WGRAD_MAX_JOBS = 24          # MNR_WGRAD_MAX_JOBS


class CompositeGradIO(C.Structure):
    """struct mnr_composite_grad_io"""
    _fields_ = [('z', C.c_void_p), ('raw', C.c_void_p), ('last_delta', C.c_void_p), ('zmax_src', C.c_void_p),
                ('zmax_S', C.c_int32), ('flip', C.c_int32), ('N', C.c_int64), ('n_units_dev', C.c_void_p),
                ('S', C.c_int32), ('d_rgb', C.c_void_p), ('d_bg_lambda', C.c_void_p), ('d_raw', C.c_void_p)]


MNR_STEP_MAX_CELLS = 16
MNR_STEP_NO_OPTIMIZER = 1
MNR_STEP_STICKY_NONFINITE, MNR_STEP_STICKY_OUTSIDE = 1, 2
MNR_STEP_SPANS = 9
STEP_SPAN_NAMES = ('samples', 'fwd_c', 'mid', 'fwd_f', 'tail', 'bwd', 'head_grads', 'wgrad', 'adam_pack')


class StepModel(C.Structure):
    """struct mnr_step_model"""
    _fields_ = [('desc', ModelDesc), ('grad', ModelGrads), ('adam_m', ModelGrads), ('adam_v', ModelGrads),
                ('adam_steps_dev', C.c_void_p),
                ('packed_dev', C.c_void_p), ('packed_bwd_dev', C.c_void_p), ('packed_h2_dev', C.c_void_p), ('packed_bwd_h2_dev', C.c_void_p)]


class StepCfg(C.Structure):
    """struct mnr_step_cfg"""
    _fields_ = [('n_cells', C.c_int32), ('n_rays', C.c_int32), ('coarse_samples', C.c_int32), ('fine_samples', C.c_int32),
                ('perturb', C.c_float), ('sigma_noise', C.c_int32), ('sphere_center', C.c_float * 3), ('sphere_radius', C.c_float * 3),
                ('grad_floats_per_cell', C.c_int64), ('adam_beta1', C.c_float), ('adam_beta2', C.c_float), ('adam_eps', C.c_float),
                ('t_coarse', c_float_p), ('t_bg_coarse', c_float_p), ('t_fine', c_float_p), ('t_bg_fine', c_float_p), ('split_precision', C.c_int32)]


class StepLayout(C.Structure):
    """struct mnr_step_layout"""
    _fields_ = [('workspace_bytes', C.c_size_t), ('grad_offset', C.c_size_t), ('grad_stride', C.c_size_t), ('loss_offset', C.c_size_t),
                ('rgb_offset', C.c_size_t), ('depth_var_offset', C.c_size_t), ('bg_lambda_offset', C.c_size_t),
                ('n_bg_offset', C.c_size_t), ('err_offset', C.c_size_t), ('tape_fg_offset', C.c_size_t), ('tape_bg_offset', C.c_size_t),
                ('tape_fg_rows', C.c_int64), ('tape_bg_rows', C.c_int64), ('gtape_fg_offset', C.c_size_t), ('gtape_bg_offset', C.c_size_t),
                ('sticky_offset', C.c_size_t)]


class StepBatch(C.Structure):
    """struct mnr_step_batch"""
    _fields_ = [('rays', C.c_void_p), ('idx', C.c_void_p), ('idx_is_float', C.c_int32), ('target', C.c_void_p),
                ('select', C.c_void_p), ('target_u8', C.c_void_p), ('u8_table', C.c_void_p), ('rng_cell_plus1', C.c_int64)]


class StepRandoms(C.Structure):
    """struct mnr_step_randoms"""
    _fields_ = [('fg_perturb', C.c_void_p), ('bg_perturb', C.c_void_p), ('fg_noise_coarse', C.c_void_p), ('fg_noise_fine', C.c_void_p),
                ('bg_noise_coarse', C.c_void_p), ('bg_noise_fine', C.c_void_p), ('fg_u', C.c_void_p), ('bg_u', C.c_void_p)]


class RenderIO(C.Structure):
    """struct mnr_render_io"""
    _fields_ = [('fg', C.POINTER(ModelDesc)), ('bg', C.POINTER(ModelDesc)), ('fg_packed', C.c_void_p), ('bg_packed', C.c_void_p),
                ('rays', C.c_void_p), ('idx', C.c_void_p), ('idx_is_float', C.c_int32), ('n_rays', C.c_int64),
                ('coarse_samples', C.c_int32), ('fine_samples', C.c_int32), ('split_precision', C.c_int32),
                ('sphere_center', C.c_float * 3), ('sphere_radius', C.c_float * 3),
                ('t_coarse_dev', C.c_void_p), ('t_bg_coarse_dev', C.c_void_p), ('t_fine_dev', C.c_void_p), ('t_bg_fine_dev', C.c_void_p),
                ('rgb', C.c_void_p), ('depth', C.c_void_p), ('fg_rgb', C.c_void_p), ('bg_rgb', C.c_void_p), ('fg_depth', C.c_void_p),
                ('bg_depth', C.c_void_p), ('bg_lambda', C.c_void_p), ('n_bg', C.c_void_p), ('err', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('side', C.c_void_p),
                ('n_cells', C.c_int32), ('fg_cell_packed', C.POINTER(C.c_void_p)), ('bg_cell_packed', C.POINTER(C.c_void_p)),
                ('fg_cell_emb', C.POINTER(C.c_void_p)), ('bg_cell_emb', C.POINTER(C.c_void_p)), ('centroids_host', c_float_p),
                ('boundary_margin', C.c_float), ('cluster_2d', C.c_int32), ('route_workspace', C.c_void_p), ('route_workspace_bytes', C.c_size_t)]


class Calibration(C.Structure):
    """struct mnr_calibration"""
    _fields_ = [('cu_count', C.c_int32), ('nominal_sclk_mhz', C.c_float), ('nominal_mclk_mhz', C.c_float), ('l2_bytes', C.c_int64),
                ('mfma_f32_tflops', C.c_float), ('sclk_mhz_under_mfma_load', C.c_float), ('mfma_wg_ms_min', C.c_float), ('mfma_wg_ms_median', C.c_float),
                ('mfma_wg_ms_max', C.c_float), ('mfma_xcd_ms_fastest', C.c_float), ('mfma_xcd_ms_slowest', C.c_float), ('mfma_slowest_wg_where', C.c_int32),
                ('mfma_start_skew_us', C.c_float), ('sclk_mhz_fma_chain', C.c_float),
                ('sclk_mhz_mfma_chain', C.c_float), ('dma_stream_gbps', C.c_float), ('dma_chunk_round_trip_us', C.c_float),
                ('dma_chunk_round_trip_alone_us', C.c_float), ('chase_l1_ns', C.c_float), ('chase_l2_ns', C.c_float), ('chase_mall_ns', C.c_float),
                ('chase_hbm_ns', C.c_float), ('hbm_read_gbps', C.c_float), ('hbm_write_gbps', C.c_float)]


EXPORTS = [
    'mnr_version', 'mnr_last_error', 'mnr_device_available', 'mnr_ray_directions', 'mnr_get_rays',
    'mnr_packed_model_bytes', 'mnr_pack_model', 'mnr_layout_src_col', 'mnr_layout_num_steps', 'mnr_layout_parts',
    'mnr_mlp_forward', 'mnr_ray_setup', 'mnr_fg_samples', 'mnr_fg_points', 'mnr_bg_samples', 'mnr_sample_pdf',
    'mnr_sample_fine', 'mnr_merge_sorted', 'mnr_sort_rows', 'mnr_composite', 'mnr_bg_blend',
    'mnr_tape_floats_per_row', 'mnr_mlp_forward_train', 'mnr_packed_bwd_bytes', 'mnr_pack_model_bwd',
    'mnr_mlp_backward_data', 'mnr_mlp_backward_weights', 'mnr_composite_backward', 'mnr_merge_backward',
    'mnr_bg_blend_backward', 'mnr_route', 'mnr_route_indexed', 'mnr_route_combine_indexed', 'mnr_route_accumulate', 'mnr_embed', 'mnr_gather_rows', 'mnr_linear',
    'mnr_fused_supported', 'mnr_cluster_min_ratios', 'mnr_gemm', 'mnr_act_grad', 'mnr_col_sum', 'mnr_scatter_rows',
    'mnr_sh_apply', 'mnr_sh_backward', 'mnr_fused_train_supported', 'mnr_image_metrics', 'mnr_get_rays_indexed', 'mnr_mlp_forward_cells', 'mnr_tape_plane_offset',
    'mnr_wgrad_workspace_bytes', 'mnr_mlp_backward_weights_multi', 'mnr_mlp_forward_multi', 'mnr_mlp_backward_data_multi', 'mnr_affine_apply', 'mnr_affine_backward',
    'mnr_tgemm_run', 'mnr_wgrad_jobs', 'mnr_mlp_backward_chain_multi', 'mnr_mlp_head_grads_multi',
    'mnr_step_query', 'mnr_step_create', 'mnr_step_destroy', 'mnr_step_repack', 'mnr_train_step', 'mnr_step_profile', 'mnr_step_kernel_times',
    'mnr_packed_model_h2_bytes', 'mnr_pack_model_h2', 'mnr_mlp_forward_multi_h2', 'mnr_render_workspace_bytes', 'mnr_render_fwd', 'mnr_packed_bwd_h2_bytes', 'mnr_pack_model_bwd_h2',
    'mnr_mlp_backward_weights_multi_h2', 'mnr_mlp_forward_cells_h2', 'mnr_side_create', 'mnr_side_destroy', 'mnr_mlp_forward_cells_multi',
    'mnr_calibrate', 'mnr_calibrate_scratch_bytes', 'mnr_calibrate_hog', 'mnr_render_route_workspace_bytes',
]

_lib: Optional[C.CDLL] = None


class NativeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError('libmeganerf_hip.so not found at {} -- run `python __graft_entry__.py` (build()) '
                              'or `make -C mega-nerf_amd/csrc`; there is no CPU fallback'.format(LIB_PATH))
        _lib = C.CDLL(str(LIB_PATH))
        _lib.mnr_last_error.restype = C.c_char_p
        _lib.mnr_packed_model_bytes.restype = C.c_size_t
        _lib.mnr_packed_model_bytes.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_pack_model.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(ModelDesc), C.c_void_p]
        _lib.mnr_layout_src_col.argtypes = [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int]
        _lib.mnr_layout_num_steps.argtypes = [C.POINTER(ModelDesc), C.c_int]
        _lib.mnr_layout_parts.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_mlp_forward.argtypes = [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(MlpIO), C.c_void_p]
        _lib.mnr_ray_directions.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_int, C.c_void_p]
        _lib.mnr_get_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                      C.c_float, c_float_p, C.c_void_p]
        _lib.mnr_ray_setup.argtypes = [C.c_void_p, C.c_int64, c_float_p, c_float_p] + [C.c_void_p] * 7
        _lib.mnr_fg_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_fg_points.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_bg_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float,
                                        C.c_void_p, C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_sample_pdf.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_sample_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_merge_sorted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_sort_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        _lib.mnr_composite.argtypes = [C.POINTER(CompositeIO), C.c_void_p]
        _lib.mnr_bg_blend.argtypes = [C.c_void_p] * 6 + [C.c_int64] + [C.c_void_p] * 5
        _lib.mnr_tape_floats_per_row.restype = C.c_int64
        _lib.mnr_tape_floats_per_row.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_tape_plane_offset.restype = C.c_int64
        _lib.mnr_tape_plane_offset.argtypes = [C.POINTER(ModelDesc), C.c_int]
        _lib.mnr_mlp_forward_train.argtypes = [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(MlpIO), C.c_void_p, C.c_int64,
                                               C.c_int64, C.c_void_p]
        _lib.mnr_packed_bwd_bytes.restype = C.c_size_t
        _lib.mnr_packed_bwd_bytes.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_pack_model_bwd.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(ModelDesc), C.c_void_p]
        _lib.mnr_mlp_backward_data.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ModelDesc), C.POINTER(MlpGradIO),
                                               C.c_void_p]
        _lib.mnr_mlp_backward_weights.argtypes = [C.POINTER(ModelDesc), C.POINTER(MlpGradIO), C.c_void_p]
        _lib.mnr_mlp_forward_multi.argtypes = [C.POINTER(MlpLaunch), C.c_int, C.c_void_p]
        _lib.mnr_mlp_backward_data_multi.argtypes = [C.POINTER(MlpGradLaunch), C.c_int, C.c_void_p]
        _lib.mnr_mlp_backward_chain_multi.argtypes = [C.POINTER(MlpGradLaunch), C.c_int, C.c_void_p]
        _lib.mnr_mlp_head_grads_multi.argtypes = [C.POINTER(MlpGradLaunch), C.c_int, C.c_void_p]
        _lib.mnr_affine_apply.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_int64, C.c_int64, C.c_void_p]
        _lib.mnr_affine_backward.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
        _lib.mnr_wgrad_workspace_bytes.restype = C.c_size_t
        _lib.mnr_wgrad_workspace_bytes.argtypes = []
        _lib.mnr_mlp_backward_weights_multi.argtypes = [C.POINTER(WgradRegion), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.mnr_mlp_backward_weights_multi_h2.argtypes = [C.POINTER(WgradRegion), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.mnr_tgemm_run.argtypes = [C.POINTER(TGemm), C.c_void_p]
        _lib.mnr_wgrad_jobs.argtypes = [C.POINTER(WgradJob), C.c_int, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.mnr_composite_backward.argtypes = [C.POINTER(CompositeGradIO), C.c_void_p]
        _lib.mnr_merge_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        _lib.mnr_route.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_route_indexed.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int,
                                           C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.mnr_route_combine_indexed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        _lib.mnr_route_accumulate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        _lib.mnr_fused_supported.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_mlp_forward_cells.argtypes = [C.POINTER(ModelDesc), C.c_void_p, C.c_int, C.POINTER(MlpIO), C.c_void_p]
        _lib.mnr_mlp_forward_cells_h2.argtypes = [C.POINTER(ModelDesc), C.c_void_p, C.c_int, C.POINTER(MlpIO), C.c_void_p]
        _lib.mnr_mlp_forward_cells_multi.argtypes = [C.POINTER(MlpCellsLaunch), C.c_int, C.c_void_p]
        _lib.mnr_fused_train_supported.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_embed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
        _lib.mnr_gather_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                         C.c_int64, C.c_int64, C.c_void_p]
        _lib.mnr_linear.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        _lib.mnr_bg_blend_backward.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3
        _lib.mnr_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        _lib.mnr_act_grad.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                      C.c_int, C.c_void_p]
        _lib.mnr_col_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        _lib.mnr_scatter_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                          C.c_int64, C.c_int64, C.c_void_p]
        _lib.mnr_sh_apply.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                      C.c_int64, C.c_void_p]
        _lib.mnr_sh_backward.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_int64, C.c_int, C.c_int64, C.c_void_p]
        _lib.mnr_get_rays_indexed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64,
                                              C.c_float, C.c_float, c_float_p, C.c_void_p, C.c_void_p]
        _lib.mnr_image_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_float,
                                           C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib.mnr_cluster_min_ratios.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_int, C.c_int, C.c_float, C.c_void_p]
        _lib.mnr_step_query.argtypes = [C.POINTER(StepCfg), C.POINTER(ModelDesc), C.POINTER(ModelDesc), C.POINTER(StepLayout)]
        _lib.mnr_step_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(StepCfg), C.POINTER(StepModel), C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.mnr_step_destroy.argtypes = [C.c_void_p]
        _lib.mnr_side_create.argtypes = [C.POINTER(C.c_void_p)]
        _lib.mnr_side_destroy.argtypes = [C.c_void_p]
        _lib.mnr_side_destroy.restype = None
        _lib.mnr_step_destroy.restype = None
        _lib.mnr_step_repack.argtypes = [C.c_void_p, C.c_void_p]
        _lib.mnr_packed_model_h2_bytes.restype = C.c_size_t
        _lib.mnr_packed_model_h2_bytes.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_pack_model_h2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(ModelDesc), C.c_void_p]
        _lib.mnr_mlp_forward_multi_h2.argtypes = [C.POINTER(MlpLaunch), C.c_int, C.c_void_p]
        _lib.mnr_packed_bwd_h2_bytes.restype = C.c_size_t
        _lib.mnr_packed_bwd_h2_bytes.argtypes = [C.POINTER(ModelDesc)]
        _lib.mnr_pack_model_bwd_h2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(ModelDesc), C.c_void_p]
        _lib.mnr_render_workspace_bytes.restype = C.c_size_t
        _lib.mnr_render_workspace_bytes.argtypes = [C.c_int64, C.c_int, C.c_int]
        _lib.mnr_render_fwd.argtypes = [C.POINTER(RenderIO), C.c_void_p]
        _lib.mnr_render_route_workspace_bytes.restype = C.c_size_t
        _lib.mnr_render_route_workspace_bytes.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.mnr_step_profile.argtypes = [C.c_void_p, C.c_int]
        _lib.mnr_step_kernel_times.argtypes = [C.c_void_p, C.c_int, c_float_p]
        _lib.mnr_train_step.argtypes = [C.c_void_p, C.POINTER(StepBatch), C.POINTER(StepRandoms), C.c_double, C.c_int64, C.c_uint64, C.c_int,
                                        C.c_void_p]
        _lib.mnr_calibrate_scratch_bytes.restype = C.c_size_t
        _lib.mnr_calibrate.argtypes = [C.POINTER(Calibration), C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.mnr_calibrate_hog.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError('libmeganerf_hip: {} (code {})'.format(lib().mnr_last_error().decode(), rc))


def require_device(t: torch.Tensor, name: str = 'tensor') -> None:
    if not t.is_cuda:
        raise NativeError('{} must live on the HIP device (got {}); the MI355X kernels have no CPU fallback'.format(
            name, t.device))


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def host3(v) -> Optional[C.Array]:
    """3 host floats (sphere centre / radius) as a ctypes array; accepts tensors, lists, None."""
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().tolist()
    return (C.c_float * 3)(*[float(x) for x in v])


_WGRAD_WS: dict = {}


def wgrad_workspace(dev):
    """Scratch of the batched weight-gradient launches (partial-sum slabs, mnr_wgrad_workspace_bytes()), one per device and stream."""
    import torch
    dev = torch.device(dev)
    # one per (device, stream): two renders enqueued on different streams must not share the queue heads / slabs
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = _WGRAD_WS[key] = torch.empty(lib().mnr_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
    return ws


def calibrate(dev, scratch: Optional[torch.Tensor] = None) -> dict:
    """mnr_calibrate on the current stream of ``dev``: what this GPU delivers right now (fp32 MFMA rate, L2 -> LDS weight-stream shape,
    dependent-load latencies, HBM streams, clocks).  Synchronises.  ``scratch``: a uint8 device tensor to reuse (>= 64 MiB)."""
    dev = torch.device(dev)
    if scratch is None:
        scratch = torch.empty(lib().mnr_calibrate_scratch_bytes(), dtype=torch.uint8, device=dev)
    require_device(scratch, 'scratch')
    out = Calibration()
    with torch.cuda.device(dev):
        check(lib().mnr_calibrate(C.byref(out), scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream(dev).cuda_stream))
    return {k: (round(getattr(out, k), 3) if isinstance(getattr(out, k), float) else int(getattr(out, k))) for k, _ in Calibration._fields_}
