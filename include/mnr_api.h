/*
 * mnr_api.h -- C ABI of libmeganerf_hip.so: the MI355X (gfx950) implementation of the Mega-NeRF
 * hot path  ray_utils.get_rays + rendering.render_rays + models.NeRF forward/backward.
 *
 * The reference (cmusatyalab/mega-nerf) has no FFI layer: its "plugin boundary" is a set of Python
 * call signatures that bottom out in torch ATen ops.  Each entry point below replaces one group of
 * those ops; the reference location it replaces is cited as  file:line  (relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer marked "dev" is a device (HBM) pointer owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only ENQUEUE work:
 *     they never synchronise the stream, never allocate device memory, keep no global mutable state;
 *   - return 0 on success, a negative MNR_E_* code otherwise; mnr_last_error() gives a thread-local
 *     message.  Entry points are re-entrant and may be called from several host threads on different
 *     streams (the reference calls ray generation from a prefetch thread: filesystem_dataset.py:70-77);
 *   - all matrices are row-major fp32; "N" = rays, "S" = samples per ray.
 */
#ifndef MNR_API_H
#define MNR_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNR_VERSION 1

#define MNR_OK 0
#define MNR_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define MNR_E_LAUNCH (-2)    /* HIP launch failure */
#define MNR_E_UNSUPPORTED (-3)

#define MNR_MAX_LAYERS 16

int mnr_version(void);
const char *mnr_last_error(void);
/* 1 if a HIP device is usable by this process, else 0 (never fails). */
int mnr_device_available(void);

/* ------------------------------------------------------------------------------------------------
 * Ray generation -- mega_nerf/ray_utils.py
 * ---------------------------------------------------------------------------------------------- */

/* get_ray_directions (ray_utils.py:6-18): out[H][W][3] = normalize([(i+c-cx)/fx, -(j+c-cy)/fy, -1]),
 * c = 0.5 if center_pixels. */
int mnr_ray_directions(float *out_dev, int W, int H, float fx, float fy, float cx, float cy,
                       int center_pixels, void *stream);

/* get_rays / get_rays_batch (ray_utils.py:21-41) + _get_rays_inner (:44-62) +
 * _truncate_with_plane_intersection (:65-84).
 *   dirs_dev : [n_dirs_sets][P][3]; n_dirs_sets is 1 (get_rays: one direction image shared) or n_poses
 *   c2w_dev  : [n_poses][3][4]
 *   out_dev  : [n_poses][P][8] = (origin3, dir3, near, far)
 *   alt_range: host pointer to 2 floats (already normalised) or NULL. */
int mnr_get_rays(float *out_dev, const float *dirs_dev, int64_t P, int n_dirs_sets, const float *c2w_dev,
                 int n_poses, float near, float far, const float *alt_range_host, void *stream);

/* Rays of a shuffled training chunk (filesystem_dataset.py:96-124): out[t] = ray of pixel pixel_idx[t] (row of the shared
 * direction table dirs_dev [n_dirs][3]) seen from pose img_idx[t] (c2w_dev [n_poses][12]); same arithmetic as mnr_get_rays.
 * Out-of-range indices are clamped and set *err_flag_dev (nullable) to 1. */
int mnr_get_rays_indexed(float *out_dev, const float *dirs_dev, int64_t n_dirs, const int32_t *pixel_idx_dev, const float *c2w_dev,
                         int n_poses, const int32_t *img_idx_dev, int64_t M, float near, float far, const float *alt_range_host,
                         int32_t *err_flag_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * NeRF MLP -- mega_nerf/models/nerf.py:45-160
 * ---------------------------------------------------------------------------------------------- */

/* Architecture + device pointers to the nn.Module parameters (nn.Linear.weight = [out][in] row-major,
 * y = x W^T + b), i.e. exactly the tensors of the reference state_dict (runner.py:521-536):
 *   layer_w[i]/layer_b[i] = xyz_encodings.{i}.0.{weight,bias}; final_* = xyz_encoding_final;
 *   dir_a_* = dir_a_encoding.0; sigma_*; rgb_*; embedding_a = embedding_a.weight. */
typedef struct mnr_model_desc {
    int32_t xyz_dim;          /* 3 foreground, 4 background (nerf.py:51,  model_utils.py:12-17) */
    int32_t pos_xyz_dim;      /* frequency bands for xyz (opts.py:42)  */
    int32_t pos_dir_dim;      /* frequency bands for direction, 0 = no view dependence (opts.py:44) */
    int32_t layers;           /* opts.py:46 */
    int32_t skip_mask;        /* bit i set: layer i consumes cat([embedding, h])  (nerf.py:128-129) */
    int32_t layer_dim;        /* W (opts.py:48-49) */
    int32_t appearance_dim;   /* opts.py:50; 0 = none */
    int32_t appearance_count; /* rows of embedding_a */
    int32_t rgb_dim;          /* 3, or 3*(sh_deg+1)^2 (model_utils.py:57) */
    int32_t sigma_activation; /* 0 = ReLU, 1 = ShiftedSoftplus (nerf.py:28-39) */
    int32_t mfma_tile;        /* samples per wavefront: 0 = auto (32 for W<=256 else 16), 32 or 16 */
    const float *layer_w[MNR_MAX_LAYERS];
    const float *layer_b[MNR_MAX_LAYERS];
    const float *final_w, *final_b;   /* NULL when the model has neither dir nor appearance input */
    const float *dir_a_w, *dir_a_b;
    const float *sigma_w, *sigma_b;
    const float *rgb_w, *rgb_b;
    const float *embedding_a;         /* NULL when appearance_dim == 0 */
} mnr_model_desc;

/* Bytes of the packed (MFMA-fragment-ordered, chunked) weight image for this architecture. 0 + error
 * string if the architecture is not supported by the fused kernels. */
size_t mnr_packed_model_bytes(const mnr_model_desc *desc);

/* Re-pack the module parameters into `packed_dev` (device->device, one launch per layer). Call again
 * whenever the optimiser has stepped. */
int mnr_pack_model(void *packed_dev, size_t packed_bytes, const mnr_model_desc *desc, void *stream);

/* Host-side, no GPU needed: source column of nn.Linear `layer` (0..layers-1 trunk, layers = final,
 * layers+1 = dir_a) that K-step `step` / lane-part `part` of the packed image multiplies; -1 = zero pad.
 * Exposed so the layout can be unit-tested without a device. */
int mnr_layout_src_col(const mnr_model_desc *desc, int layer, int step, int part);
int mnr_layout_num_steps(const mnr_model_desc *desc, int layer);
int mnr_layout_parts(const mnr_model_desc *desc);

/* One batched MLP evaluation = NeRF.forward(x, sigma_only, sigma_noise)  (nerf.py:115-160).
 * Row r of the logical input x is  [ xyz(r) | dir(r / rows_per_ray) | idx(r / rows_per_ray) ]:
 * with rows_per_ray == 1 and the three pointers aimed into one [B][ncols] matrix this is exactly the
 * reference's x; with rows_per_ray == S it is the repeat/cat of rendering.py:280-319 without
 * materialising it. */
typedef struct mnr_mlp_io {
    const float *xyz;  int64_t xyz_stride;        /* dev [n_rows][>=xyz_dim], row stride in floats */
    const float *dir;  int64_t dir_stride;        /* dev, 3 floats per ray (NULL if pos_dir_dim==0 and no SH) */
    const void  *idx;  int64_t idx_stride;        /* dev, image index per ray: float or int32 */
    int32_t idx_is_float;
    int32_t rows_per_ray;
    const float *sigma_noise;                     /* dev [n_rows] or NULL (rendering.py:294,321; nerf.py:133-134) */
    float *out;        int64_t out_stride;        /* dev [n_rows][out_stride]; writes rgb_dim+1 (or 1) floats */
    int64_t n_rows;                               /* upper bound on rows (grid size) */
    const int32_t *n_units_dev;                   /* optional dev scalar: actual rows = *n_units_dev * rows_per_unit */
    int32_t rows_per_unit;
    int32_t sigma_only;
    int32_t apply_sh_deg;                         /* -1: raw output; >=0: rgb = sigmoid(eval_sh(deg, coeffs, dir))
                                                     (rendering.py:301-306), output is 4 floats */
    const int32_t *row_index;                     /* optional gather: logical row r reads the inputs (xyz, dir, idx,
                                                     sigma_noise) of source row row_index[r]; the output stays compact
                                                     at out[r] (per-cell evaluation under the MegaNeRF router) */
} mnr_mlp_io;

/* Register-chained evaluation of nerf.py:115-160.  Kernel families behind it (all exact fp32 MFMA, same packed image): layer_dim <= 256:
 * one wavefront owns 16 samples x all features, two workgroups per CU (csrc/mlp_fwd_kernels.h); layer_dim 512 with the default
 * encodings (README "Larger models", Building): a wavefront PAIR owns the 16 samples and splits every layer's output features, two
 * wavefronts per SIMD (csrc/mlp_fwd_pair.hip: whole render at 0.82 of the fp32-MFMA peak; MNR_NO_PAIR_KERNEL=1 selects the
 * one-wavefront-per-SIMD instantiation for comparison).  MNR_E_UNSUPPORTED: no instantiation -- use the per-layer entry points below. */
int mnr_mlp_forward(const void *packed_dev, const mnr_model_desc *desc, const mnr_mlp_io *io, void *stream);
/* All cells of a routed MegaNeRF evaluation in ONE launch (the per-cell launches of mega_nerf.py:28-49 are individually
 * too small to fill 256 CUs).  cells_dev: DEVICE array; every cell shares the architecture of `desc` (its weight pointers
 * are ignored).  Cell c evaluates the rows row_index[0 .. *count) of the shared inputs in `io` (xyz / dir / idx / noise,
 * strides, rows_per_ray, sigma_only, apply_sh_deg; io->n_rows = capacity of every row list) into out[k * io->out_stride]
 * for its k-th listed row.  io->out, io->row_index and io->n_units_dev are ignored. */
typedef struct mnr_mlp_cell {
    const void *packed_dev;        /* mnr_pack_model image of this cell */
    const float *embedding_a;      /* its appearance table (NULL without appearance input) */
    const int32_t *row_index;      /* compact list of the rows routed to it (mnr_route) */
    const int32_t *count;          /* device-side length of that list */
    float *out;                    /* [>= *count][out_stride] */
} mnr_mlp_cell;
int mnr_mlp_forward_cells(const mnr_model_desc *desc, const mnr_mlp_cell *cells_dev, int n_cells, const mnr_mlp_io *io,
                          void *stream);
/* The same routed launch on the 16-bit matrix pipe (opt-in split precision, csrc/mlp_fwd_h2.hip): mnr_mlp_cell::packed_dev are
 * mnr_pack_model_h2 images; default 8x256 architectures, no sigma_only / SH (mega_nerf.py:28-49 under rendering.SPLIT_PRECISION). */
int mnr_mlp_forward_cells_h2(const mnr_model_desc *desc, const mnr_mlp_cell *cells_dev, int n_cells, const mnr_mlp_io *io,
                             void *stream);
/* Host-side query (no GPU work): 1 if mnr_mlp_forward has a fused kernel for this architecture, else 0. */
int mnr_fused_supported(const mnr_model_desc *desc);
/* ... and 1 if the fused training kernels (mnr_mlp_forward_train / mnr_mlp_backward_*) cover it. */
int mnr_fused_train_supported(const mnr_model_desc *desc);

/* ---- generic-width fallback (layer_dim > 512 or architectures without a fused instantiation) ---------------
 * One launch per nn.Linear with activations in HBM; same exact-fp32 MFMA arithmetic.  The host sequences them like
 * nerf.py:115-160 (mega_nerf/models/nerf.py::_evaluate_layerwise). */
/* out[r][:] = [x, sin(2^0 x), cos(2^0 x), ...] (nerf.py:20-25) of x = src[(r / rows_per_src)][0..D) */
int mnr_embed(float *out_dev, int64_t ldo, const float *x_dev, int64_t ldx, int D, int L, int64_t rows_per_src, int64_t B,
              void *stream);
/* out[r][0..width) = table[idx[r / rows_per_ray]][:]  (nerf.py:149) */
int mnr_gather_rows(float *out_dev, int64_t ldo, const float *table_dev, int width, int count, const void *idx_dev,
                    int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t B, void *stream);
/* Y[b][n] = act( [X1 | X2][b] . W[n] + bias[n] + row_add[b] ), act: 0 none, 1 ReLU, 2 sigmoid, 3 softplus(x-1) */
int mnr_linear(float *Y_dev, int64_t ldy, const float *X1_dev, int64_t ldx1, int K1, const float *X2_dev, int64_t ldx2, int K2,
               const float *W_dev, int64_t ldw, const float *bias_dev, const float *row_add_dev, int64_t B, int N, int act,
               void *stream);

/* Adjoint of the layer-by-layer path (training of the generic-width architectures; autograd of nerf.py:115-160).
 * C[m][n] (op)= sum_k A(m,k) B(n,k) with A(m,k) = A[m sam + k sak], B(n,k) = B[n sbn + k sbk] (exact fp32 MFMA):
 *   data gradient    dX = G W       : A = G (sam = ldg, sak = 1),  B = W (sbn = 1, sbk = ldw)
 *   weight gradient  dW += G^T X    : A = G (sam = 1, sak = ldg),  B = X (sbn = 1, sbk = ldx),  K = rows
 * accumulate 0: C = ..., 1: C += ...; split_k > 1 splits K over workgroups (atomic adds, needs accumulate = 1), 0 = auto */
int mnr_gemm(float *C_dev, int64_t ldc, const float *A_dev, int64_t sam, int64_t sak, const float *B_dev, int64_t sbn,
             int64_t sbk, int64_t M, int N, int64_t K, int accumulate, int split_k, void *stream);
/* G = dY * act'(Y), act' written through the layer output Y (act codes of mnr_linear); G may alias dY */
int mnr_act_grad(float *G_dev, int64_t ldg, const float *dY_dev, int64_t ldd, const float *Y_dev, int64_t ldy, int64_t R, int N,
                 int act, void *stream);
/* out[n] += sum_r G[r][n]  (bias gradients) */
int mnr_col_sum(float *out_dev, const float *G_dev, int64_t ldg, int64_t R, int N, void *stream);
/* table_grad[idx[r / rows_per_ray]][0..width) += src[r][0..width)  (gradient of mnr_gather_rows) */
int mnr_scatter_rows(float *table_grad_dev, int width, int count, const void *idx_dev, int64_t idx_stride, int idx_is_float,
                     int64_t rows_per_ray, const float *src_dev, int64_t ld_src, int64_t R, void *stream);
/* ---- wide layers of the layer-by-layer path: tiled GEMM with fused epilogues (csrc/tgemm.hip) and batched weight
 * gradients (csrc/wgrad.hip); what nerf.py:115-160 + autograd get from cuBLAS, for layer widths that are multiples of 256.
 *   C[m][n] = gate( relu( sum_p sum_k a[p][m][k] B_p(n,k) + bias[n] + r1_row[m] r1_col[n] ) )
 * b_kslow 0: B_p(n,k) = b[p][n * ldb[p] + k] (nn.Linear weights, forward);  1: b[p][k * ldb[p] + n] (data gradient dZ . W).
 * gate (optional): C is zeroed where gate[m][n] <= 0 -- the ReLU adjoint through the previous layer's output.
 * Requirements: n % 256 == 0, k[p] % 32 == 0, all pointers 16-byte aligned and all pitches multiples of 4 floats
 * (the caller zero-pads odd input widths, e.g. the 63 embedding columns to 64); anything else -> MNR_E_INVALID. */
typedef struct mnr_tgemm {
    const float *a[2];  int64_t lda[2];
    const float *b[2];  int64_t ldb[2];
    int32_t k[2];
    int32_t n_phases;            /* 1 or 2 */
    int32_t b_kslow;
    int32_t relu;
    float *c;  int64_t ldc;
    int64_t m;  int32_t n;
    const float *bias;           /* [n] or NULL */
    const float *gate;  int64_t ldgate;
    const float *r1_row;  int64_t r1_stride;  const float *r1_col;    /* NULL or rank-1 addend */
} mnr_tgemm;
int mnr_tgemm_run(const mnr_tgemm *g, void *stream);

/* dw[m * ldw + n] += sum_r dz[r][m] * in[r][n] (m < 256, n < in_cols),  db[m] += sum_r dz[r][m]  for a list of jobs over
 * the same `rows` rows (rows % 32 == 0), one launch + one reduction launch (the kernel of mnr_mlp_backward_weights_multi).
 * dz: 256 consecutive columns of a [rows][ldz] gradient.  in_block 256: 256 consecutive columns of a [rows][ldin] matrix;
 * in_block 32/64/96/128: a dense [rows][in_block] matrix (ldin == in_block, columns >= in_cols ignored).
 * At most MNR_WGRAD_MAX_JOBS jobs per call. */
#define MNR_WGRAD_MAX_JOBS 24
typedef struct mnr_wgrad_job {
    const float *dz;  int64_t ldz;
    const float *in;  int64_t ldin;
    int32_t in_cols, in_block;
    float *dw;  int64_t ldw;
    float *db;                   /* NULL: no bias gradient from this job */
} mnr_wgrad_job;
int mnr_wgrad_jobs(const mnr_wgrad_job *jobs, int n_jobs, int64_t rows, void *workspace_dev, size_t workspace_bytes, void *stream);

/* Spherical-harmonics colour outside the fused epilogue (rendering.py:300-305, spherical_harmonics.py:55-107):
 * out[r] = [sigmoid(eval_sh(deg, coef[r] viewed (3, (deg+1)^2), dir[r / rows_per_ray])), coef[r][3 (deg+1)^2]] and its adjoint */
int mnr_sh_apply(float *out_dev, int64_t ldo, const float *coef_dev, int64_t ldc, const float *dirs_dev, int64_t dir_stride,
                 int64_t rows_per_ray, int deg, int64_t R, void *stream);
int mnr_sh_backward(float *d_coef_dev, int64_t ldc, const float *d_out_dev, int64_t ldd, const float *out_dev, int64_t ldo,
                    const float *dirs_dev, int64_t dir_stride, int64_t rows_per_ray, int deg, int64_t R, void *stream);

/* Affine appearance (nerf.py:87-89,156-158): out[r][0..3) = sigmoid(A[:, :3] . raw[r] + A[:, 3]) with A = table[idx[r / rows_per_ray]]
 * viewed (3, 4); table [count][12] = affine(embedding_a.weight) (one mnr_linear per weight version).  The adjoint returns d_raw and,
 * per row, the 12 partial derivatives with respect to its A (reduce per appearance index with mnr_scatter_rows). */
int mnr_affine_apply(float *out_dev, int64_t ldo, const float *raw_dev, int64_t ldr, const float *table_dev, int count,
                     const void *idx_dev, int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t R, void *stream);
int mnr_affine_backward(float *d_raw_dev, int64_t ldr, float *d_affine_rows_dev, const float *d_out_dev, int64_t ldd,
                        const float *out_dev, int64_t ldo, const float *raw_dev, int64_t ldri, const float *table_dev, int count,
                        const void *idx_dev, int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t R, void *stream);

/* ---- training (the reference obtains all of this from torch autograd over nerf.py:115-160) -------------
 * Forward pass that additionally writes the activation tape (post-ReLU output of every layer, the two
 * positional encodings in reference column order and the gathered appearance rows) as dense row-major
 * planes [tape_rows][width]; mnr_tape_floats_per_row() * tape_rows floats in total. */
int64_t mnr_tape_floats_per_row(const mnr_model_desc *desc);
int mnr_mlp_forward_train(const void *packed_dev, const mnr_model_desc *desc, const mnr_mlp_io *io, float *tape_dev,
                          int64_t tape_rows, int64_t tape_row0, void *stream);

/* Several independent MLP evaluations in ONE launch (the foreground and the background model of one pass of a training /
 * rendering step; the compacted background rows alone cannot fill 256 CUs).  Every segment is what mnr_mlp_forward (tape_dev
 * NULL) or mnr_mlp_forward_train would take; all segments of a call are of the same kind.  Covers the default 8x256
 * foreground / background architectures (MNR_E_UNSUPPORTED otherwise: launch the segments one by one). */
typedef struct mnr_mlp_launch {
    const void *packed_dev;
    const mnr_model_desc *desc;
    const mnr_mlp_io *io;
    float *tape_dev;                 /* training: activation tape of this segment's model, else NULL */
    int64_t tape_rows, tape_row0;
} mnr_mlp_launch;
int mnr_mlp_forward_multi(const mnr_mlp_launch *segs, int n_segs, void *stream);
/* Routed evaluations of SEVERAL merged models in one launch (mega_nerf.py:28-49 for the foreground container and the background
 * container of one render pass, rendering.py:275-331): segment s = mnr_mlp_forward_cells(desc, cells_dev, n_cells, io) of one
 * container.  The background's routed rows alone fill a fraction of the chip; side by side with the foreground's they only lengthen
 * its tail.  Inference, the default 8x256 foreground / background architectures (MNR_E_UNSUPPORTED otherwise: one
 * mnr_mlp_forward_cells per container). */
typedef struct mnr_mlp_cells_launch {
    const mnr_model_desc *desc;      /* architecture shared by the container's cells */
    const mnr_mlp_cell *cells_dev;   /* device array [n_cells] */
    int32_t n_cells;                 /* 1 .. 64 */
    const mnr_mlp_io *io;            /* as for mnr_mlp_forward_cells */
} mnr_mlp_cells_launch;
int mnr_mlp_forward_cells_multi(const mnr_mlp_cells_launch *segs, int n_segs, void *stream);

/* ---- opt-in split-precision inference (csrc/mlp_fwd_h2.hip) ------------------------------------------------
 * Same contract as mnr_mlp_forward_multi (inference segments only, default 8 x 256 fg / bg architectures), computed on the
 * 16-bit matrix pipe: every fp32 operand is split into two f16 halves and every layer is three v_mfma_f32_16x16x32_f16
 * products accumulated in fp32 (w_hi x_hi + w_lo x_hi + w_hi x_lo; heads, biases, activations, encodings stay fp32 VALU).
 * Measured 4.6e-7 relative error per layer against fp64 -- the fp32 kernel's class -- at ~2.5x its speed; it needs its own weight
 * image (mnr_pack_model_h2: (hi, lo) fragment pairs, same size as the fp32 image).  NOT the default: the fp32 kernels are. */
size_t mnr_packed_model_h2_bytes(const mnr_model_desc *desc);
int mnr_pack_model_h2(void *packed_dev, size_t packed_bytes, const mnr_model_desc *desc, void *stream);
/* ... and the transposed image of the split-precision data-gradient chain (training through mnr_train_step only) */
size_t mnr_packed_bwd_h2_bytes(const mnr_model_desc *desc);
int mnr_pack_model_bwd_h2(void *packed_dev, size_t packed_bytes, const mnr_model_desc *desc, void *stream);
int mnr_mlp_forward_multi_h2(const mnr_mlp_launch *segs, int n_segs, void *stream);

/* Transposed weight image for the data-gradient chain (re-pack after every optimiser step). */
size_t mnr_packed_bwd_bytes(const mnr_model_desc *desc);
int mnr_pack_model_bwd(void *packed_dev, size_t packed_bytes, const mnr_model_desc *desc, void *stream);

/* Gradient buffers with the shapes of the nn.Module parameters (= param.grad); gradients are ACCUMULATED. */
typedef struct mnr_model_grads {
    float *layer_w[MNR_MAX_LAYERS];
    float *layer_b[MNR_MAX_LAYERS];
    float *final_w, *final_b, *dir_a_w, *dir_a_b, *sigma_w, *sigma_b, *rgb_w, *rgb_b;
    float *embedding_a;          /* [appearance_count][appearance_dim] or NULL */
} mnr_model_grads;

typedef struct mnr_mlp_grad_io {
    const float *tape;           /* written by mnr_mlp_forward_train for the same rows */
    float *gtape;                /* scratch of the same size: dL/d(pre-activation) of every layer */
    int64_t tape_rows;           /* row capacity of every plane */
    int64_t tape_row0;           /* tape row of this launch's row 0 (several passes share one tape) */
    const float *d_out;  int64_t d_out_stride;   /* dL/d(out) [n_rows][>=4] (rgb after sigmoid, sigma after activation) */
    const float *out;    int64_t out_stride;     /* the forward output itself */
    float *dheads;               /* scratch [n_rows][4] */
    const void *idx;  int64_t idx_stride;  int32_t idx_is_float;
    int32_t rows_per_ray;
    int64_t n_rows;
    const int32_t *n_units_dev;  int32_t rows_per_unit;
    int32_t *work_counter;       /* scratch: one device int32 (item queue head of the weight-gradient launch) */
    mnr_model_grads grad;
    const float *dd_in;          /* models whose colour head is not 3 sigmoid outputs (spherical harmonics, rgb_dim > 3):
                                    dL/d(output of dir_a_encoding) [n_rows][layer_dim/2], produced by the caller from the
                                    colour epilogue (mnr_sh_backward) and the rgb layer (mnr_gemm); the chain then starts
                                    there and the rgb.* gradients are the caller's.  NULL for rgb_dim == 3. */
} mnr_mlp_grad_io;

/* Offset (in floats per row; plane base = tape + offset * tape_rows) of a tape plane: which = 0: post-ReLU output of
 * dir_a_encoding (width layer_dim/2), 1: output of xyz_encoding_final, 2: post-ReLU output of the last trunk layer.  < 0 on error. */
int64_t mnr_tape_plane_offset(const mnr_model_desc *desc, int which);

/* Backward, step 1: data-gradient chain (fused, register-chained like the forward) for the rows of one
 * forward pass; fills gtape / dheads rows [tape_row0, tape_row0 + n_rows) and accumulates the appearance
 * embedding gradient.  No gradient w.r.t. xyz / directions is produced (rays are data). */
int mnr_mlp_backward_data(const void *packed_fwd_dev, const void *packed_bwd_dev, const mnr_model_desc *desc,
                          const mnr_mlp_grad_io *io, void *stream);
/* Backward, step 1, batched: the data-gradient chains of several segments (coarse + fine rows of the foreground and the
 * background model of a training step) in ONE launch, followed by the head gradients of every segment.  Each segment is what
 * mnr_mlp_backward_data would take.  Default 8x256 fg / bg architectures (MNR_E_UNSUPPORTED otherwise). */
typedef struct mnr_mlp_grad_launch {
    const void *packed_fwd_dev;
    const void *packed_bwd_dev;
    const mnr_model_desc *desc;
    const mnr_mlp_grad_io *io;
} mnr_mlp_grad_launch;
int mnr_mlp_backward_data_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream);
/* ... its two halves, for callers that time or schedule them separately: the chain launch (k_mlp_bwd_multi: fills gtape / dheads,
 * accumulates the appearance-embedding gradient) and the head gradients (sigma / rgb weights and biases from dheads + tape). */
int mnr_mlp_backward_chain_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream);
int mnr_mlp_head_grads_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream);

/* Backward, step 2: weight + bias gradients of every layer over tape rows [tape_row0, tape_row0 + n_rows) in one
 * launch (so several forward passes that share a tape are reduced together).  d_out/out/idx are not read. */
int mnr_mlp_backward_weights(const mnr_model_desc *desc, const mnr_mlp_grad_io *io, void *stream);

/* Backward, step 2, batched: the weight + bias gradients of SEVERAL models (foreground + background of one training step)
 * in ONE launch + one reduction launch.  Each region names a model's tape / gradient tape and up to two row ranges of them
 * (coarse rows, fine rows; device-side counts for the compacted background).  Contract: every range starts on a multiple of
 * 4 rows, the tape capacity covers the range padded to 32 rows, and padding rows hold dZ = 0 and finite activations (the
 * fused forward / data-gradient kernels write whole 64-row tiles that way).  workspace_dev: mnr_wgrad_workspace_bytes()
 * bytes of scratch (partial-sum slabs; no initialisation needed).  Gradients are ACCUMULATED into region.grad. */
typedef struct mnr_wgrad_region {
    const mnr_model_desc *desc;
    const float *tape;
    const float *gtape;
    int64_t tape_rows;
    int32_t n_ranges;
    int64_t row0[2];
    int64_t n_rows[2];
    const int32_t *n_units_dev[2];
    int32_t rows_per_unit[2];
    mnr_model_grads grad;
} mnr_wgrad_region;
size_t mnr_wgrad_workspace_bytes(void);
int mnr_mlp_backward_weights_multi(const mnr_wgrad_region *regions, int n_regions, void *workspace_dev, size_t workspace_bytes,
                                   void *stream);
/* The same weight gradients on the 16-bit matrix pipe (opt-in, csrc/wgrad.hip H2 path): both operands of every product split into
 * f16 (hi, lo) halves on the fly, three v_mfma_f32_32x32x16_f16 products per block, fp32 accumulation; every dZ plane is scaled by a
 * power of two first (found by one extra pass over the plane here; the fused split-precision step -- mnr_train_step -- gets the
 * exponents from its data-gradient chain).  Same arguments, workspace and results (within fp32 rounding of the sums) as above. */
int mnr_mlp_backward_weights_multi_h2(const mnr_wgrad_region *regions, int n_regions, void *workspace_dev, size_t workspace_bytes,
                                      void *stream);

/* ------------------------------------------------------------------------------------------------
 * Volume rendering stages -- mega_nerf/rendering.py
 * ---------------------------------------------------------------------------------------------- */

/* rendering.py:33-45 + _intersect_sphere (:396-417): per ray
 *   fg_far = max(sphere_exit, near); has_bg = far > fg_far; far_out = min(far, fg_far);
 *   last_delta = has_bg ? fg_far : 1e10.
 * Then a stable (ascending ray index) compaction of the has_bg rays:
 *   bg_list[k] = ray, bg_slot[ray] = k or -1, *n_bg.
 * err_flag (dev int) is set to 1 if any camera lies outside the unit ellipsoid (the reference raises,
 * rendering.py:412-414; the host shim raises the same text at its next sync).
 * sphere_center/radius are host pointers to 3 floats (radius may be NULL: plain unit sphere). */
int mnr_ray_setup(const float *rays_dev, int64_t N, const float *sphere_center_host,
                  const float *sphere_radius_host, float *far_out_dev, float *last_delta_dev,
                  int32_t *bg_list_dev, int32_t *bg_slot_dev, int32_t *n_bg_dev, int32_t *err_flag_dev,
                  void *stream);

/* rendering.py:82-87 (+ _expand_and_perturb_z_vals :472-483): z = near*(1-t)+far*t, optional
 * stratified jitter with caller-supplied uniforms, then xyz = o + d*z.
 *   far_dev  : [N] (from mnr_ray_setup) or NULL to use rays[:,7]
 *   t_dev    : [S] the torch.linspace(0,1,S) table (values are data)
 *   rand_dev : [N][S] uniforms or NULL (perturb == 0)
 *   z_out [N][S], xyz_out [N][S][3] */
int mnr_fg_samples(const float *rays_dev, const float *far_dev, int64_t N, int S, const float *t_dev,
                   float perturb, const float *rand_dev, float *z_out_dev, float *xyz_out_dev, void *stream);

/* xyz = o + d*z for caller-provided z [N][S]  (rendering.py:100 xyz_fine_fn). */
int mnr_fg_points(const float *rays_dev, int64_t N, int S, const float *z_dev, float *xyz_out_dev,
                  void *stream);

/* Background samples (rendering.py:47-56, 70-75): for k < *n_bg, ray = bg_list[k]:
 *   z[k][s] = t[s] (+ jitter)  when z_in_dev == NULL, else z_in_dev[k][s] (fine pass)
 *   pts/depth_real = _depth2pts_outside (rendering.py:420-469); pts has 4 columns, or 7 when
 *   include_xyz_real (container / train_mega_nerf: :52-53, :457-464). */
int mnr_bg_samples(const float *rays_dev, const int32_t *bg_list_dev, const int32_t *n_bg_dev, int64_t N_max,
                   int S, const float *t_dev, float perturb, const float *rand_dev, const float *z_in_dev,
                   const float *sphere_center_host, const float *sphere_radius_host, int include_xyz_real,
                   int cluster_2d, float *z_out_dev, float *pts_out_dev, float *depth_real_out_dev,
                   void *stream);

/* _sample_pdf + _sample_cdf (rendering.py:486-536) on caller-provided bins/weights:
 *   bins [N][nb+1], weights [N][nb] (row strides given), u: either u_dev [N][nf] (det=0) or the shared
 *   table t_dev [nf] (det=1).  samples_out [N][nf]; inds_out [N][nf] int32 (optional, for parity).
 * The normaliser and cdf reproduce the reference CPU association order, so indices are
 * bit-exact for identical inputs. */
int mnr_sample_pdf(const float *bins_dev, int64_t bins_stride, const float *weights_dev, int64_t weights_stride,
                   int64_t N, const int32_t *n_units_dev, int nb, int nf, int det, const float *u_dev,
                   float *samples_out_dev, int32_t *inds_out_dev, void *stream);

/* The importance-sampling step of _get_results (rendering.py:212-216): bins = mid-points of z [N][S],
 * weights = w[:,1:-1]  ->  nf samples per ray.  flip is irrelevant here (quirk Q1 is reproduced by the
 * caller passing the flipped-order weights with ascending z, exactly as the reference does). */
int mnr_sample_fine(const float *z_dev, const float *weights_dev, int64_t N, const int32_t *n_units_dev, int S,
                    int nf, int det, const float *u_dev, float *samples_out_dev, int32_t *inds_out_dev,
                    void *stream);

/* Merge coarse and fine samples (rendering.py:336-350): stable sort of cat([z_fine, z_coarse]) along the
 * ray (descending when flip) and gather of raw rgb/sigma (and depth_real).
 *   raw_* are [N][S*][4] MLP outputs; outputs z [N][Sf+Sc], raw [N][Sf+Sc][4], depth_real [N][Sf+Sc]. */
int mnr_merge_sorted(const float *z_fine_dev, const float *raw_fine_dev, const float *dr_fine_dev, int Sf,
                     const float *z_coarse_dev, const float *raw_coarse_dev, const float *dr_coarse_dev, int Sc,
                     int64_t N, const int32_t *n_units_dev, int flip, float *z_out_dev, float *raw_out_dev,
                     float *dr_out_dev, int32_t *order_out_dev, void *stream);

/* Sort z only (cascade: rendering.py:218-219). */
int mnr_sort_rows(const float *a_dev, int Sa, const float *b_dev, int Sb, int64_t N, const int32_t *n_units_dev,
                  float *out_dev, void *stream);

/* Volume compositing (rendering.py:353-393) of one ray per wavefront:
 *   delta_k = z_{k+1}-z_k (z_k - z_{k+1} when flip), last = last_delta[ray] - (last_delta<1e10 ? zmax_sub[ray] : 0)
 *   alpha = 1-exp(-delta*sigma); T = cumprod(1-alpha+1e-8); w = alpha * T_shifted
 * outputs (each optional / NULL): weights [N][S], rgb [N][3], depth [N], depth_var [N], bg_lambda [N].
 * depth uses depth_real when given (bg), depth_var always uses z (rendering.py:392). */
typedef struct mnr_composite_io {
    const float *z;          /* [N][S] */
    const float *raw;        /* [N][S][4] rgb,sigma */
    const float *depth_real; /* [N][S] or NULL */
    const float *last_delta; /* [N] or NULL (=1e10) */
    const float *zmax_src;   /* [N][zmax_S]: last_delta -= max_s zmax_src[ray][s] where last_delta < 1e10
                                (rendering.py:192-193, 224-225); NULL = no subtraction */
    int32_t zmax_S;
    int32_t flip;
    int64_t N;
    const int32_t *n_units_dev;
    int32_t S;
    float *weights;
    float *rgb;
    float *depth;
    float *depth_var;
    float *bg_lambda;
} mnr_composite_io;
int mnr_composite(const mnr_composite_io *io, void *stream);

/* MegaNeRF routing (mega_nerf.py:19-49).  For every row (first 3 floats of pos = world position) the blend weight
 * of every cell: hard arg-min for boundary_margin == 1, else w_i = (1/(d_i+1e-8)) [d_i <= margin * d_min], normalised.
 *   centroids_host [n_sub][3]; cluster_dim_start = 1 drops the altitude axis (cluster_2d)
 *   weights_out [n_sub][B]; lists_out [n_sub][B] compacted row ids of the rows routed to cell i (device-side append,
 *   order unspecified); counts_out [n_sub] (zeroed by the call).  No host synchronisation. */
int mnr_route(const float *pos_dev, int64_t pos_stride, int64_t B, const int32_t *n_units_dev, int rows_per_unit,
              const float *centroids_host, int n_sub, int cluster_dim_start, float boundary_margin,
              float *weights_out_dev, int32_t *lists_out_dev, int32_t *counts_out_dev, void *stream);

/* mnr_route + the INVERSE of the lists: inverse_out [n_sub][B], inverse[i][row] = position of `row` in cell i's list, -1 where the row
 * was not routed to cell i (written for every row below the device-side count) -- what mnr_route_combine_indexed reads. */
int mnr_route_indexed(const float *pos_dev, int64_t pos_stride, int64_t B, const int32_t *n_units_dev, int rows_per_unit,
                      const float *centroids_host, int n_sub, int cluster_dim_start, float boundary_margin,
                      float *weights_out_dev, int32_t *lists_out_dev, int32_t *counts_out_dev, int32_t *inverse_out_dev, void *stream);

/* out[list[r]][c] (+)= sub_out[r][c] * (weights ? weights[list[r]] : 1) for r < *count  (mega_nerf.py:45-49). */
int mnr_route_accumulate(float *out_dev, int64_t out_stride, const float *sub_out_dev, int64_t sub_stride, int n_cols,
                         const int32_t *list_dev, const int32_t *count_dev, int64_t B_max, const float *weights_dev,
                         int assign, void *stream);
/* All cells at once: out[row] = sum_i w_i[row] * sub_i[inverse_i[row]] in cell order (identical roundings to n_sub
 * mnr_route_accumulate calls on a zeroed output: mega_nerf.py:43-49).  sub_all: cell i's compact outputs start at
 * sub_all + i * cell_stride, rows sub_stride apart; inverse / weights [n_sub][B] as written by mnr_route_indexed (weights NULL = hard
 * routing).  One launch; rows at or past *n_units_dev * rows_per_unit are ZEROED (the caller need not clear `out`). */
int mnr_route_combine_indexed(float *out_dev, int64_t out_stride, const float *sub_all_dev, int64_t cell_stride, int64_t sub_stride,
                              int n_cols, const int32_t *inverse_dev, const float *weights_dev, int n_sub, int64_t B,
                              const int32_t *n_units_dev, int rows_per_unit, void *stream);

/* fg/bg blend (rendering.py:102-139): for every ray, slot = bg_slot[ray]:
 *   bg_rgb = slot>=0 ? lambda*bg_rgb_c[slot] : 0;  rgb = fg + bg_rgb  (same for depth).
 * Optional outputs fg_/bg_ copies (get_bg_fg_rgb). In-place on rgb/depth. */
int mnr_bg_blend(float *rgb_dev, float *depth_dev, const float *bg_lambda_dev, const int32_t *bg_slot_dev,
                 const float *bg_rgb_c_dev, const float *bg_depth_c_dev, int64_t N, float *fg_rgb_out,
                 float *bg_rgb_out, float *fg_depth_out, float *bg_depth_out, void *stream);

/* ---- cluster masks (scripts/create_cluster_masks.py:157-187) ------------------------------------------------
 * Per ray (rays_dev [n_rays][8] = o, d, near, far): n_samples points z = near (1 - t) + far t with t = z_steps_dev[s]
 * (the CPU torch.linspace(0, 1, n_samples) table), distance of each point to every centroid exactly as torch.cdist
 * computes it (matmul formulation, :174-175), ratio = dist / (min over centroids + 1e-8) (:184), and the minimum ratio
 * over the samples of the ray for every centroid.
 *   ratios_out [n_rays][n_centroids] float (nullable)  == min_dist_ratio, :184
 *   masks_out  [n_centroids][n_rays] uint8 (nullable)  == ratio <= boundary_margin, :203-204 (one plane per cell)
 *   centroids_dev [n_centroids][3]; cluster_2d != 0 ignores the altitude axis (:111). At most 64 centroids. */
int mnr_cluster_min_ratios(float *ratios_out, uint8_t *masks_out, const float *rays_dev, int64_t n_rays,
                           const float *z_steps_dev, int n_samples, const float *centroids_dev, int n_centroids,
                           int cluster_2d, float boundary_margin, void *stream);

/* ---- validation metrics (metrics.py:8-10 PSNR, :51-121 SSIM; runner.py:413-436) --------------------------------
 * pred / target: [H][W][3] fp32 images on the device, rows `row_stride` floats apart (so the right-half views of
 * runner.py:413-414 need no copy).  filter_dev: the normalised 1-D Gaussian (filter_size taps, odd, <= 33) as metrics.py:77-82
 * builds it.  Adds to acc_dev[0] the sum of squared errors over H*W*3 values and to acc_dev[1] the sum of the SSIM map
 * (zero-padded separable blur, variance clamps and covariance limit of :96-111); the caller zeroes acc_dev and divides. */
int mnr_image_metrics(const float *pred_dev, const float *target_dev, int H, int W, int64_t row_stride, const float *filter_dev,
                      int filter_size, float max_val, float k1, float k2, double *acc_dev, void *stream);

/* ---- backward of the rendering stages (training; autograd over rendering.py:102-131,336-393) --------- */

/* Gradient of mnr_composite's rgb (and bg_lambda) output w.r.t. the raw MLP outputs. Inputs as in the forward
 * call; d_rgb [N][3], d_bg_lambda [N] or NULL; writes d_raw [N][S][4] (d rgb, d sigma). */
typedef struct mnr_composite_grad_io {
    const float *z, *raw, *last_delta, *zmax_src;
    int32_t zmax_S, flip;
    int64_t N;
    const int32_t *n_units_dev;
    int32_t S;
    const float *d_rgb;
    const float *d_bg_lambda;
    float *d_raw;
} mnr_composite_grad_io;
int mnr_composite_backward(const mnr_composite_grad_io *io, void *stream);

/* Inverse of mnr_merge_sorted for gradients: d_merged [N][Sa+Sb][4] + order (as exported by the forward) ->
 * d_a [N][Sa][4] (fine), d_b [N][Sb][4] (coarse). */
int mnr_merge_backward(const float *d_merged_dev, const int32_t *order_dev, int Sa, int Sb, int64_t N,
                       const int32_t *n_units_dev, float *d_a_dev, float *d_b_dev, void *stream);

/* Backward of the rgb part of mnr_bg_blend: d_lambda [N], d_bg_rgb_c [slots][3]. */
int mnr_bg_blend_backward(const float *d_rgb_dev, const float *bg_lambda_dev, const int32_t *bg_slot_dev,
                          const float *bg_rgb_c_dev, int64_t N, float *d_lambda_dev, float *d_bg_rgb_c_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * One whole training step per call -- runner.py:244-277 + rendering.py:15-173
 *
 * What the reference's trainer does per iteration (runner.py:347-358 render_rays with the training flags, :370 mse_loss,
 * :263-277 backward + Adam step on the foreground and the background model), for ONE OR SEVERAL independent submodules
 * ("cells": parscripts/run_8.txt runs one trainer per cell; a rank that owns several cells steps all of them here), enqueued
 * on one stream as a fixed sequence of 12 kernel launches + one memset for one cell, 2 more per further cell (csrc/step.hip):
 *     memset (gradients, counters) | k_step_begin (batch copy, _intersect_sphere, background compaction) | k_step_samples
 *     (coarse samples of both branches, random numbers) | MLP coarse pass, all cells, fg + bg rows | k_step_mid (coarse
 *     compositing weights -> _sample_pdf -> fine points) | MLP fine pass | k_step_tail (merge, compositing, fg/bg blend, MSE,
 *     and the adjoints of all of these) | data-gradient chains | head gradients | weight gradients (+ reduction, per cell) |
 *     Adam | re-pack of every weight image.
 * Default architectures only (8 x 256 fg / bg models with appearance embedding, no cascade, fine_samples > 0); anything else:
 * MNR_E_UNSUPPORTED, and the caller sequences the stage entry points above.  All device memory is the caller's: one workspace
 * (mnr_step_query tells its size and where the gradient area sits inside it), the parameters, Adam moments and packed images.
 * ---------------------------------------------------------------------------------------------- */
#define MNR_STEP_MAX_CELLS 16

typedef struct mnr_step_model {
    mnr_model_desc desc;             /* the nn.Module parameters (updated in place by the optimiser) */
    mnr_model_grads grad;            /* param.grad: views INTO the workspace's gradient area (zeroed at the start of every step) */
    mnr_model_grads adam_m, adam_v;  /* torch.optim.Adam's exp_avg / exp_avg_sq, same shapes (caller-owned, zero before step 1) */
    int32_t *adam_steps_dev;         /* DEVICE int32: optimiser steps this model has taken (torch.optim.Adam's per-parameter `step`; caller-owned,
                                        0 before step 1, set from a checkpoint on resume).  The step reads it for the bias corrections and
                                        advances it -- for a background model only on batches with background rays, which are the only
                                        ones the reference steps that optimiser on (runner.py:268-272) */
    void *packed_dev;                /* mnr_packed_model_bytes(desc) bytes: re-packed at the end of every step */
    void *packed_bwd_dev;            /* mnr_packed_bwd_bytes(desc) bytes */
    void *packed_h2_dev;             /* cfg.split_precision: mnr_packed_model_h2_bytes / mnr_packed_bwd_h2_bytes bytes INSTEAD of the */
    void *packed_bwd_h2_dev;         /*   two fp32 images above (which may then be NULL) */
} mnr_step_model;

typedef struct mnr_step_cfg {
    int32_t n_cells;                 /* 1 .. MNR_STEP_MAX_CELLS */
    int32_t n_rays;                  /* rays per cell and step (opts.py:74 batch_size) */
    int32_t coarse_samples, fine_samples;       /* opts.py:32-35; multiples of 2, (n_rays * samples) % 128 == 0 */
    float perturb;                   /* opts.py:80: > 0 = stratified jitter + random u (training mode); 0 = deterministic */
    int32_t sigma_noise;             /* 1: uniform noise on sigma before its activation (rendering.py:294,321, training mode) */
    float sphere_center[3], sphere_radius[3];   /* runner.py:96-106 */
    int64_t grad_floats_per_cell;    /* size of one cell's gradient area (fg + bg, the caller's layout), in floats */
    float adam_beta1, adam_beta2, adam_eps;     /* 0.9, 0.999, 1e-8 = torch defaults (runner.py:169-171) */
    const float *t_coarse;           /* HOST tables: torch.linspace(0, 1, n) as the CPU kernel computes it (values are data: */
    const float *t_bg_coarse;        /*   DESIGN.md) for n = coarse_samples, coarse_samples / 2, fine_samples, fine_samples / 2 */
    const float *t_fine, *t_bg_fine;
    int32_t split_precision;         /* opt-in: the tape-writing forward and the data-gradient chain on the 16-bit matrix pipe with
                                        split-precision operands (csrc/mlp_fwd_h2.hip, mlp_bwd_h2.hip); tapes stay fp32, the
                                        weight gradients, heads, ray stages and the optimiser are the same fp32 kernels */
} mnr_step_cfg;

typedef struct mnr_step_layout {
    size_t workspace_bytes;
    size_t grad_offset, grad_stride; /* cell c's gradient area: workspace + grad_offset + c * grad_stride */
    size_t loss_offset;              /* float  [n_cells]          mean squared error of the step */
    size_t rgb_offset;               /* float  [n_cells][n_rays][3]   rgb_fine */
    size_t depth_var_offset;         /* float  [n_cells][n_rays]      depth_variance_fine */
    size_t bg_lambda_offset;         /* float  [n_cells][n_rays]      bg_lambda_fine */
    size_t n_bg_offset, err_offset;  /* int32  [n_cells]          rays with a background segment / camera-outside-sphere flag */
    size_t tape_fg_offset, tape_bg_offset;   /* the activation tapes (mlp_layout.h TapeLayout planes, tape_*_rows rows each): cell c's coarse */
    int64_t tape_fg_rows, tape_bg_rows;      /*   rows start at row c * (tape_*_rows / n_cells), its fine rows follow (tests read the ReLU masks here) */
    size_t gtape_fg_offset, gtape_bg_offset; /* the gradient tapes (same planes: dL/d(pre-activation) of every layer) */
    size_t sticky_offset;            /* int32  [n_cells]          MNR_STEP_STICKY_* bits OR-ed over all optimising steps since mnr_step_create
                                        (never cleared by a step): lets a trainer check "Train metrics not finite" (runner.py:260-261) and
                                        the camera-outside-sphere error (rendering.py:412-414) every k steps instead of synchronising every step */
} mnr_step_layout;
#define MNR_STEP_STICKY_NONFINITE 1  /* a step's loss was NaN / inf */
#define MNR_STEP_STICKY_OUTSIDE   2  /* a step's batch had a camera outside the unit ellipsoid */
int mnr_step_query(const mnr_step_cfg *cfg, const mnr_model_desc *fg_arch, const mnr_model_desc *bg_arch, mnr_step_layout *out);

typedef struct mnr_step_plan mnr_step_plan;
/* models: [n_cells][2] = fg of cell 0, bg of cell 0, fg of cell 1, ...  Uploads the (step-invariant) device tables into the
 * workspace and packs every weight image on `stream`.  The plan keeps host copies of everything it was given. */
int mnr_step_create(mnr_step_plan **out, const mnr_step_cfg *cfg, const mnr_step_model *models, void *workspace_dev,
                    size_t workspace_bytes, void *stream);
void mnr_step_destroy(mnr_step_plan *plan);
/* re-pack all weight images (after the caller changed parameters behind the plan's back, e.g. loaded a checkpoint) */
int mnr_step_repack(mnr_step_plan *plan, void *stream);

typedef struct mnr_step_batch {      /* one cell's batch: device pointers, read (copied into the workspace) by the first kernel */
    const float *rays;               /* [n_rays][8] */
    const void *idx;                 /* [n_rays] image indices, int32 or float */
    int32_t idx_is_float;
    const float *target;             /* [n_rays][3] ground-truth colours */
    /* Gathered form -- a device-resident training set (memory_dataset.py, filesystem_dataset.py chunks): with select != NULL the batch is
     * rows select[0 .. n_rays) of `rays` / `idx` / `target` (or `target_u8`), which then point at the WHOLE set; the gather that the
     * reference's DataLoader collation does on the host (runner.py:228-238) happens inside the step's first kernel. */
    const int64_t *select;           /* [n_rays] row numbers, or NULL */
    const uint8_t *target_u8;        /* instead of `target`: uint8 colours [rows][3] (dataset_utils.py:30 keeps them as bytes) ... */
    const float *u8_table;           /* ... converted through this DEVICE table of 256 floats (the CPU's i / 255. values) */
    /* Which random streams the cell draws from: 0 = keyed by its position in the plan (seed + position); k > 0 = seed + (k - 1).  A cell
     * trained in a plan of its own (position 0) and the same cell trained side by side with a rank's other cells then see the same
     * numbers (tools/train_cells.py; the reference seeds every per-cell process alike, parscripts/run_8.txt + opts.py:101). */
    int64_t rng_cell_plus1;
} mnr_step_batch;
typedef struct mnr_step_randoms {    /* optional injected uniforms of one cell (parity tests); NULL members are generated */
    const float *fg_perturb, *bg_perturb;            /* [n_rays][coarse], [n_bg][coarse / 2] */
    const float *fg_noise_coarse, *fg_noise_fine;    /* [n_rays * coarse], [n_rays * fine] */
    const float *bg_noise_coarse, *bg_noise_fine;    /* [n_bg * coarse / 2], [n_bg * fine / 2] (compacted background rays) */
    const float *fg_u, *bg_u;                        /* [n_rays][fine], [n_bg][fine / 2] */
} mnr_step_randoms;
#define MNR_STEP_NO_OPTIMIZER 1      /* flags: stop after the gradients (no Adam, no re-pack) */
/* batches [n_cells]; randoms NULL or [n_cells]; lr = this step's learning rate (the caller applies ExponentialLR,
 * runner.py:173-176; a double, as torch.optim.Adam computes lr / bias_correction1 before it rounds to fp32); iteration = 1, 2, ...:
 * seed + iteration key the counter-based generator (the bias corrections come from each model's adam_steps_dev). */
int mnr_train_step(mnr_step_plan *plan, const mnr_step_batch *batches, const mnr_step_randoms *randoms, double lr, int64_t iteration,
                   uint64_t seed, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------
 * One inference render per call -- rendering.render_rays with the evaluation flags (runner.py:569-578: get_depth,
 * get_bg_fg_rgb), default 8 x 256 fg + bg models, no cascade: six launches on one stream (k_step_begin, k_step_samples, MLP coarse
 * pass, k_step_mid, MLP fine pass, k_render_tail), stateless, all memory the caller's.  split_precision != 0: the MLP passes run on
 * mnr_mlp_forward_multi_h2 and fg_packed / bg_packed must be mnr_pack_model_h2 images.  t_*_dev: DEVICE copies of the CPU
 * torch.linspace(0, 1, n) tables for n = coarse, coarse / 2, fine, fine / 2.  Outputs [n_rays][3] / [n_rays]; depth, fg_*, bg_* may
 * be NULL.  *n_bg / *err: device scalars (background-ray count; 1 if a camera lies outside the unit ellipsoid, rendering.py:412-414).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mnr_render_io {
    const mnr_model_desc *fg, *bg;
    const void *fg_packed, *bg_packed;
    const float *rays;  const void *idx;  int32_t idx_is_float;
    int64_t n_rays;
    int32_t coarse_samples, fine_samples, split_precision;
    float sphere_center[3], sphere_radius[3];
    const float *t_coarse_dev, *t_bg_coarse_dev, *t_fine_dev, *t_bg_fine_dev;
    float *rgb, *depth, *fg_rgb, *bg_rgb, *fg_depth, *bg_depth, *bg_lambda;
    int32_t *n_bg, *err;
    void *workspace;  size_t workspace_bytes;
    void *side;                        /* optional side handle of mnr_side_create: the background branch runs on its stream beside the foreground's passes; NULL = one stream */
    /* Merged containers (mega_nerf.py:19-61 behind rendering.py:275-331, model_utils.py:22-29): with n_cells > 0 BOTH models are MegaNeRF
     * routers over the same n_cells centroids.  fg / bg then describe
     * the cells' shared architectures, fg_packed / bg_packed are ignored, and every MLP pass becomes: route the pass's rows (k_route:
     * blend weights, per-cell row lists, on the device) -> ONE gather-mode launch of all cells of both containers -> blend in cell order
     * (k_route_combine).  The background's routing position is its ray's sphere-exit point (rendering.py:463-464), or the sample's own
     * far-away position under cluster_2d. */
    int32_t n_cells;                   /* 0 = plain models; 1 .. 64 */
    const void *const *fg_cell_packed; /* HOST arrays [n_cells]: device pointers of the cells' mnr_pack_model images ... */
    const void *const *bg_cell_packed;
    const float *const *fg_cell_emb;   /* ... and of their appearance tables */
    const float *const *bg_cell_emb;
    const float *centroids_host;       /* [n_cells][3] */
    float boundary_margin;             /* >= 1 (1 = hard routing) */
    int32_t cluster_2d;                /* container.cluster_2d (mega_nerf.py:16): distances over y, z; a background row then routes on o + d * depth_real (rendering.py:458-461) */
    void *route_workspace;  size_t route_workspace_bytes;    /* mnr_render_route_workspace_bytes() */
} mnr_render_io;
/* A side stream + fork / join events a caller may lend to mnr_render_fwd (host objects; create once per device / thread, destroy at exit). */
typedef struct mnr_side mnr_side;
int mnr_side_create(mnr_side **out);
void mnr_side_destroy(mnr_side *side);
size_t mnr_render_workspace_bytes(int64_t n_rays, int coarse_samples, int fine_samples);
/* out_cols: columns the cells write per row -- 4, or rgb_dim + 1 for spherical-harmonics cells under a soft blend (the reference blends
 * the RAW coefficients and evaluates eval_sh + sigmoid on the blend: mega_nerf.py:45-49, rendering.py:300-306) */
size_t mnr_render_route_workspace_bytes(int64_t n_rays, int coarse_samples, int fine_samples, int n_cells, int out_cols);
int mnr_render_fwd(const mnr_render_io *io, void *stream);

/* Kernel-level timing without a profiler (bench.py's roofline): after mnr_step_profile(plan, n) every step records HIP events on
 * its launch stream around its kernels, into slot (step index mod n); mnr_step_kernel_times reads a finished slot (the caller
 * synchronises first): ms[MNR_STEP_SPANS] = { samples stage (begin + samples), MLP coarse pass, mid stage, MLP fine pass, tail
 * stage, data-gradient chains, head gradients, weight gradients (all cells, incl. reductions), Adam + re-pack }.  n = 0 stops
 * recording.  The events are host objects owned by the plan. */
#define MNR_STEP_SPANS 9
int mnr_step_profile(mnr_step_plan *plan, int n_slots);
int mnr_step_kernel_times(mnr_step_plan *plan, int slot, float *ms_out);

/* ------------------------------------------------------------------------------------------------
 * Device calibration (bench.py `calibration`; no reference counterpart: the reference has no kernels of its own).  What THIS GPU
 * delivers, right now, on the resources the register-chained MLP kernels use -- so that a slow benchmark line can be told apart
 * from a slow box.  The ONE entry point that synchronises the stream (it times its own launches).  scratch_dev: at least 64 MiB,
 * mnr_calibrate_scratch_bytes() (1 GiB) for an HBM-resident pointer chase; contents are overwritten.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mnr_calibration {
    int32_t cu_count;
    float nominal_sclk_mhz, nominal_mclk_mhz;  /* hipDeviceProp */
    int64_t l2_bytes;
    float mfma_f32_tflops;                /* 512 workgroups x 4 wavefronts of v_mfma_f32_16x16x4_f32 (peak 157.3 at 2.4 GHz) */
    float sclk_mhz_under_mfma_load;       /* ... = the clock the matrix pipes held with every CU busy */
    /* the same launch, workgroup by workgroup (wall_clock64 at both ends of each): a CU or an XCD that runs behind the others delays every
     * launch whose workgroups are dealt statically -- the register-chained MLP kernels at 1024-ray sizes -- and none of the persistent,
     * work-stealing ones (k_wgrad2, k_tgemm) */
    float mfma_wg_ms_min, mfma_wg_ms_median, mfma_wg_ms_max;
    float mfma_xcd_ms_fastest, mfma_xcd_ms_slowest;   /* mean workgroup duration of the fastest / slowest XCD */
    int32_t mfma_slowest_wg_where;        /* (xcc_id << 16) | (HW_ID & 0xffff) of the slowest workgroup */
    float mfma_start_skew_us;             /* last workgroup start - first workgroup start */
    float sclk_mhz_fma_chain;             /* ONE wavefront: dependent v_fma_f32 chain (4 cycles each) against the 100 MHz counter */
    float sclk_mhz_mfma_chain;            /* ONE wavefront: dependent v_mfma_f32_32x32x2_f32 chain (64 cycles each) */
    float dma_stream_gbps;                /* 512 workgroups streaming the same 2.4 MB image L2 -> LDS (global_load_lds_dwordx4, 32 KiB
                                             chunks, two in flight): aggregate GB/s */
    float dma_chunk_round_trip_us;        /* the same with ONE chunk in flight: request, vmcnt(0), barrier */
    float dma_chunk_round_trip_alone_us;  /* ... with one workgroup on the chip */
    float chase_l1_ns, chase_l2_ns, chase_mall_ns, chase_hbm_ns;   /* dependent-load latency: 8 KiB / 256 KiB / 64 MiB / half-scratch line sets */
    float hbm_read_gbps, hbm_write_gbps;  /* streaming, non-temporal, whole scratch */
} mnr_calibration;
size_t mnr_calibrate_scratch_bytes(void);
int mnr_calibrate(mnr_calibration *out, void *scratch_dev, size_t scratch_bytes, void *stream);
/* A memory hog for contention experiments (enqueue only, like every other entry point): `workgroups` workgroups stream-write, then
 * stream-read `bytes` of scratch, `passes` times.  Launched on a side stream next to a training step it shows which kernels lose most
 * when HBM / fabric latency rises (tools/probe_contention.py). */
int mnr_calibrate_hog(void *scratch_dev, size_t bytes, int workgroups, int passes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MNR_API_H */
