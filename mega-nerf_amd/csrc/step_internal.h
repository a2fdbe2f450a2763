// step_internal.h -- entry points shared between the translation units of libmeganerf_hip.so for the fused training / rendering
// step (csrc/step.hip): the multi-segment MLP launches with several cells' rows side by side in every segment.
#pragma once
#include "common.h"
#include "mlp_device.h"

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

// weight gradients with caller-placed control words (so that the step's single memset can clear them): as
// mnr_mlp_backward_weights_multi, but counters_dev (256 bytes, ZEROED by the caller), ep_job_dev and slab_dev are separate
int wgrad_regions_launch(const mnr_wgrad_region *regions, int n_regions, int32_t *counters_dev, int32_t *ep_job_dev, float *slab_dev,
                         hipStream_t s, const int32_t *const *zexp = nullptr);
size_t wgrad_ep_job_bytes();
size_t wgrad_slab_bytes();

}  // namespace mnr
