// mlp_fwd.hip -- fused NeRF MLP forward for gfx950 (replaces nerf.py:115-160 + the repeat/cat/
// chunk loop of rendering.py:275-331).
//
// One wavefront = TILE samples x all W features, activations chained layer to layer in registers
// (mlp_layout.h).  A workgroup is 4 wavefronts (one per SIMD, up to 512 VGPR+AGPR each); they share
// the weight stream: 32 KiB chunks staged global -> registers -> LDS, double buffered, one barrier
// per chunk.  Arithmetic is exact fp32: v_mfma_f32_32x32x2_f32 (k-ordered fmaf chain), so results
// agree with the fp32 reference to GEMM-reassociation error (~1e-6 relative).
//
// Per-sample work: positional encoding (accurate sincosf, arguments 2^f * x formed exactly),
// 8 trunk layers (+skip), sigma head (VALU dot + cross-lane add), xyz_encoding_final, dir/appearance
// layer, rgb head, sigmoid / shifted softplus -- nothing but the inputs (<= 36 B) and the 16 B result
// touches HBM.
#include <stdlib.h>

#include "mlp_fwd_kernels.h"

#ifdef MNR_PROBE_TRAIN      // codegen probes (not part of the library): -DMNR_PROBE_TRAIN=1 the foreground training kernel,
                            // =2 / =3 the two-model launch in its eval / training instantiation (seconds instead of minutes)
#if MNR_PROBE_TRAIN == 2
template __global__ void mnr::k_mlp_fwd_multi<mnr::MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>, mnr::MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>, false>(mnr::MlpFwdMulti);
#elif MNR_PROBE_TRAIN == 3
template __global__ void mnr::k_mlp_fwd_multi<mnr::MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>, mnr::MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>, true>(mnr::MlpFwdMulti);
#else
template __global__ void mnr::k_mlp_fwd<mnr::MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>, true>(mnr::MlpFwdArgs);
#endif
#else
using namespace mnr;

namespace mnr {
// mlp_fwd_pair.hip: 512-wide default architectures, two wavefronts per SIMD (a wavefront pair splits the output features)
int mlp_forward_pair_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                              const mnr_mlp_cell *cells, int n_cells, float *tape, long tape_rows, long tape_row0);
// mlp_fwd_wide.hip / mlp_fwd_variants.hip: the remaining inference instantiations (their own translation units: the instantiations of this
// file alone compiled for ten minutes); 1 = "not mine", otherwise the launch's return code
int mlp_forward_wide_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                              const mnr_mlp_cell *cells, int n_cells);
int mlp_forward_variants_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                                  const mnr_mlp_cell *cells, int n_cells);
// mlp_fwd_train.hip: launches the tape-writing instantiation for this architecture (MNR_E_UNSUPPORTED if there is none)
int mlp_forward_train_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                               float *tape, long tape_rows, long tape_row0);
}

static int mlp_forward_impl(const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, void *stream,
                            float *tape, long tape_rows, long tape_row0, const mnr_mlp_cell *cells = nullptr, int n_cells = 0);

// 1 if mnr_mlp_forward has a register-chained instantiation for this architecture, else 0 (host-side query).
extern "C" int mnr_fused_supported(const mnr_model_desc *d) {
    ModelLayout m;
    if (layout_from_desc(d, m) != MNR_OK) return 0;
    mnr_mlp_io io{};
    float dummy;
    io.xyz = &dummy; io.out = &dummy; io.dir = &dummy; io.idx = &dummy; io.rows_per_ray = 1; io.n_rows = 0;   // n_rows = 0: nothing launches
    io.apply_sh_deg = -1;
    static float packed_stub;
    mnr_model_desc dd = *d;
    float e = 0.f;
    if (dd.appearance_dim > 0 && !dd.embedding_a) dd.embedding_a = &e;
    return mlp_forward_impl(&packed_stub, &dd, &io, nullptr, nullptr, 0, 0) == MNR_OK ? 1 : 0;
}

// 1 if mnr_mlp_forward_train / mnr_mlp_backward_* have instantiations for this architecture (host-side query).
extern "C" int mnr_fused_train_supported(const mnr_model_desc *d) {
    ModelLayout m;
    if (layout_from_desc(d, m) != MNR_OK) return 0;
    // 512-wide models have a tape-writing FORWARD (k_mlp_fwd_pair<.., true>: activation planes for the tiled per-layer backward of
    // models/layerwise.py) but no register-chained data-gradient chain: not "fused training" in the sense of this query
    if (d->layer_dim > 256) return 0;
    mnr_mlp_io io{};
    float dummy;
    io.xyz = &dummy; io.out = &dummy; io.dir = &dummy; io.idx = &dummy; io.rows_per_ray = 1; io.n_rows = 0;
    io.apply_sh_deg = -1;
    static float packed_stub;
    mnr_model_desc dd = *d;
    float e = 0.f;
    if (dd.appearance_dim > 0 && !dd.embedding_a) dd.embedding_a = &e;
    return mlp_forward_impl(&packed_stub, &dd, &io, nullptr, &dummy, 0, 0) == MNR_OK ? 1 : 0;
}

extern "C" int mnr_mlp_forward(const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, void *stream) {
    return mlp_forward_impl(packed_dev, d, io, stream, nullptr, 0, 0);
}

extern "C" int mnr_mlp_forward_train(const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io,
                                     float *tape_dev, int64_t tape_rows, int64_t tape_row0, void *stream) {
    MNR_REQUIRE(tape_dev && io && tape_row0 >= 0 && tape_rows >= tape_row0 + io->n_rows, "tape buffer missing or too small");
    MNR_REQUIRE(!io->sigma_only, "sigma_only has no training variant");
    return mlp_forward_impl(packed_dev, d, io, stream, tape_dev, (long)tape_rows, (long)tape_row0);
}

extern "C" int64_t mnr_tape_floats_per_row(const mnr_model_desc *d) {
    ModelLayout m;
    if (layout_from_desc(d, m) != MNR_OK) return -1;
    return tape_layout(ArchDims{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim,
                                d->appearance_dim, d->rgb_dim, d->mfma_tile}).floats_per_row;
}

extern "C" int64_t mnr_tape_plane_offset(const mnr_model_desc *d, int which) {
    ModelLayout m;
    if (layout_from_desc(d, m) != MNR_OK) return -1;
    const TapeLayout t = tape_layout(ArchDims{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim,
                                              d->appearance_dim, d->rgb_dim, d->mfma_tile});
    if (which == 0) return t.dact_off;
    if (which == 1) return t.fin_off;
    if (which == 2) return t.act_off[d->layers - 1];
    set_err(MNR_E_INVALID, "unknown tape plane %d", which);
    return -1;
}

static int mlp_forward_impl(const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, void *stream,
                            float *tape, long tape_rows, long tape_row0, const mnr_mlp_cell *cells, int n_cells) {
    ModelLayout m;
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(io && io->xyz && (cells ? n_cells > 0 : (packed_dev && io->out)), "NULL pointer argument");
    MNR_REQUIRE(io->rows_per_ray >= 1, "rows_per_ray must be >= 1");
    MNR_REQUIRE(io->n_rows >= 0, "negative n_rows");
    const bool need_dir = !io->sigma_only && (d->pos_dir_dim > 0 || (d->rgb_dim > 3 && io->apply_sh_deg >= 0));
    MNR_REQUIRE(!need_dir || io->dir, "dir pointer required");
    MNR_REQUIRE(d->appearance_dim == 0 || io->sigma_only || (io->idx && (cells || d->embedding_a)), "idx / embedding_a required");
    if (io->apply_sh_deg >= 0) MNR_REQUIRE(3 * (io->apply_sh_deg + 1) * (io->apply_sh_deg + 1) == d->rgb_dim,
                                           "apply_sh_deg does not match rgb_dim");
    hipStream_t s = as_stream(stream);
#define MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL)                                                     \
    if (d->xyz_dim == XYZ && d->pos_xyz_dim == LX && d->pos_dir_dim == LD && d->appearance_dim == APP &&       \
        d->layer_dim == W && d->layers == NL && d->skip_mask == SKIP && d->rgb_dim == RGB && m.tile == TL && !tape) \
        return launch_fwd<MlpCfg<XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL>>(m, packed_dev, d, io, s, nullptr, 0, 0, cells, n_cells);
#define MNR_TRY(XYZ, LX, LD, APP, W, NL, SKIP, RGB) MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, tile_for_width(W))
    // configs/mega-nerf/*.yaml (opts.py defaults): fg / bg
    MNR_TRY(3, 12, 4, 48, 256, 8, 16, 3)
    MNR_TRY(4, 12, 4, 48, 256, 8, 16, 3)
    // training variants (forward pass that also writes the activation tape): instantiated in mlp_fwd_train.hip
    if (tape && d->layer_dim == 512) {       // activation planes only (no sign-bit / encoding planes): the forward of the tiled training path
        const int prc = mlp_forward_pair_dispatch(m, packed_dev, d, io, s, nullptr, 0, tape, tape_rows, tape_row0);
        if (prc != MNR_E_UNSUPPORTED) return prc;
    }
    if (tape) return mlp_forward_train_dispatch(m, packed_dev, d, io, s, tape, tape_rows, tape_row0);
    // 32-samples-per-wave variants (v_mfma_f32_32x32x2_f32, one workgroup per CU)
    MNR_TRY_T(3, 12, 4, 48, 256, 8, 16, 3, 32)
    MNR_TRY_T(4, 12, 4, 48, 256, 8, 16, 3, 32)
    // configs/mega-nerf Building: 512 channels -- the wavefront-pair kernel (two wavefronts per SIMD); MNR_NO_PAIR_KERNEL=1 keeps the
    // one-wavefront-per-SIMD instantiations below (comparison runs)
    if (d->layer_dim == 512 && !tape && !getenv("MNR_NO_PAIR_KERNEL")) {
        const int prc = mlp_forward_pair_dispatch(m, packed_dev, d, io, s, cells, n_cells, nullptr, 0, 0);
        if (prc != MNR_E_UNSUPPORTED) return prc;
    }
    {
        int orc = mlp_forward_wide_dispatch(m, packed_dev, d, io, s, cells, n_cells);          // one wavefront per SIMD at 512 channels (comparison runs)
        if (orc != 1) return orc;
        orc = mlp_forward_variants_dispatch(m, packed_dev, d, io, s, cells, n_cells);          // SH heads, 64-wide test models, no appearance
        if (orc != 1) return orc;
    }
#undef MNR_TRY
#undef MNR_TRY_T
    return set_err(MNR_E_UNSUPPORTED,
                   "no fused MLP kernel for xyz_dim=%d pos_xyz_dim=%d pos_dir_dim=%d appearance_dim=%d layer_dim=%d "
                   "layers=%d skip_mask=%d rgb_dim=%d",
                   d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->appearance_dim, d->layer_dim, d->layers,
                   d->skip_mask, d->rgb_dim);
}

// Routed evaluation of all cells of a MegaNeRF in ONE launch (mega_nerf.py:28-49 evaluates cell after cell): every cell
// brings its packed weights, appearance table, compact row list (+ device-side count) and output buffer.
extern "C" int mnr_mlp_forward_cells(const mnr_model_desc *d, const mnr_mlp_cell *cells_dev, int n_cells, const mnr_mlp_io *io,
                                     void *stream) {
    MNR_REQUIRE(d && cells_dev && n_cells > 0 && n_cells <= 64 && io, "bad arguments to mnr_mlp_forward_cells");
    return mlp_forward_impl(nullptr, d, io, stream, nullptr, 0, 0, cells_dev, n_cells);
}

#endif   // MNR_PROBE_TRAIN
