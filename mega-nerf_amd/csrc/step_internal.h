// step_internal.h -- entry points shared between the translation units of libmeganerf_hip.so for the fused training / rendering
// step (csrc/step.hip): the multi-segment MLP launches with several cells' rows side by side in every segment.
#pragma once
#include "common.h"
#include "mlp_device.h"
#include "route_internal.h"

namespace mnr {

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m);
int bwd_layout_from_desc(const mnr_model_desc *d, BwdLayout &b);

// Multi-cell form of a segment: `dcells` = device table with one MlpCellSeg per cell, `cell_rows` = row capacity per cell
// (a multiple of 64); the segment's mnr_mlp_io / mnr_mlp_grad_io then describes arrays that hold the cells' rows back to back
// (n_rows = n_cells * cell_rows; packed pointers / tape_row0 / n_units_dev of the io are ignored in favour of the table).
struct CellTable {
    const MlpCellSeg *dcells;
    long cell_rows;
};

int mlp_forward_multi_impl(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s);
// 512-wide default architectures (csrc/mlp_fwd_pair.hip); MNR_E_UNSUPPORTED for anything else
int mlp_forward_pair_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                              const mnr_mlp_cell *cells, int n_cells, float *tape, long tape_rows, long tape_row0);
int mlp_backward_chain_multi_impl(const mnr_mlp_grad_launch *segs, int n_segs, const CellTable *cells, hipStream_t s);
// split-precision forms (csrc/mlp_fwd_h2.hip, csrc/mlp_bwd_h2.hip): packed pointers = the (hi, lo) f16 images
int h2_layout(const mnr_model_desc *d, ModelLayout &m);
int mlp_forward_multi_h2_impl(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s);
int mlp_backward_chain_multi_h2_impl(const mnr_mlp_grad_launch *segs, int n_segs, const CellTable *cells, hipStream_t s);

// sigma / rgb head weight gradients of several (tape, row range) jobs in ONE launch
struct HeadJob {
    const float *dheads, *a_last, *dact;
    long row0, n_rows;
    const int32_t *n_units_dev;
    int rows_per_unit, n_blocks;
    float *d_sigma_w, *d_sigma_b, *d_rgb_w, *d_rgb_b;
};
constexpr int HEAD_MAX_JOBS = 32;
int head_grads_jobs(const HeadJob *jobs, int n_jobs, int W, hipStream_t s);
// ... the job of one mnr_mlp_grad_io (all rows [tape_row0, tape_row0 + n_rows) of its tape)
int head_job_of(const mnr_model_desc *d, const mnr_mlp_grad_io *io, HeadJob &job);

// spherical-harmonics colour head backward (k_sh_head_bwd, csrc/mlp_bwd.hip): one job per (cell, branch, pass)
struct ShHeadJob {
    const float *d_out, *out;      // [..][4]: dL/d(rgb after the sigmoid, sigma), the forward's output rows -- row = out_row0 + r
    const float *dirs;             // ray directions: dirs + ((out_row0 + r) / rows_per_ray) * dir_stride
    long dir_stride;
    int rows_per_ray, sh_deg;
    const float *dact;             // tape plane of the dir_a output [tape rows][128]: row = tape_row0 + r
    float *dd;                     // out [..][128] dL/d(dir_a output), row = out_row0 + r (mnr_mlp_grad_io::dd_in of the chain launch)
    const float *rgb_w;            // nn.Linear(128, 3 nb).weight
    float *d_rgb_w, *d_rgb_b;      // += (zeroed by the caller)
    long out_row0, tape_row0, n_rows;
    const int32_t *n_units_dev;    // NULL, or the device-side unit count: rows = *n_units_dev * rows_per_unit
    int rows_per_unit, n_blocks;
};
constexpr int SH_HEAD_MAX_JOBS = 16;
int sh_head_bwd_jobs(const ShHeadJob *jobs, int n_jobs, hipStream_t s);

// weight gradients with caller-placed control words (so that the step's single memset can clear them): as
// mnr_mlp_backward_weights_multi, but counters_dev (256 bytes, ZEROED by the caller), ep_job_dev and slab_dev are separate
int wgrad_regions_launch(const mnr_wgrad_region *regions, int n_regions, int32_t *counters_dev, int32_t *ep_job_dev, float *slab_dev,
                         hipStream_t s, const int32_t *const *zexp = nullptr);
size_t wgrad_ep_job_bytes();
size_t wgrad_slab_bytes();

}  // namespace mnr
