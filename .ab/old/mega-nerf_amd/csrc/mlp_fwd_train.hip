// mlp_fwd_train.hip -- the tape-writing (training) instantiations of the register-chained forward kernel; split from
// mlp_fwd.hip so that the two sets compile in parallel.  Kernel and design notes: mlp_fwd_kernels.h / mlp_fwd.hip.
#include "mlp_fwd_kernels.h"

namespace mnr {

int mlp_forward_train_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                               float *tape, long tape_rows, long tape_row0) {
#define MNR_TRY_TRAIN(XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL)                                                 \
    if (d->xyz_dim == XYZ && d->pos_xyz_dim == LX && d->pos_dir_dim == LD && d->appearance_dim == APP &&       \
        d->layer_dim == W && d->layers == NL && d->skip_mask == SKIP && d->rgb_dim == RGB && m.tile == TL && tape) \
        return launch_fwd<MlpCfg<XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL>, true>(m, packed_dev, d, io, s, tape, tape_rows, tape_row0);
    MNR_TRY_TRAIN(3, 12, 4, 48, 256, 8, 16, 3, 16)
    MNR_TRY_TRAIN(4, 12, 4, 48, 256, 8, 16, 3, 16)
#ifdef MNR_ALL_VARIANTS
    MNR_TRY_TRAIN(3, 12, 4, 0, 256, 8, 16, 3, 16)     // configs/mega-nerf-no-embed
    MNR_TRY_TRAIN(4, 12, 4, 0, 256, 8, 16, 3, 16)
    MNR_TRY_TRAIN(3, 12, 0, 48, 256, 8, 16, 27, 16)   // configs/mega-nerf-sh-3 (colour epilogue + rgb layer adjoint: caller)
    MNR_TRY_TRAIN(4, 12, 0, 48, 256, 8, 16, 27, 16)
    MNR_TRY_TRAIN(3, 12, 0, 48, 256, 8, 16, 48, 16)   // sh_deg 3
    MNR_TRY_TRAIN(4, 12, 0, 48, 256, 8, 16, 48, 16)
#endif
#undef MNR_TRY_TRAIN
    return set_err(MNR_E_UNSUPPORTED,
                   "no fused training kernel for xyz_dim=%d pos_xyz_dim=%d pos_dir_dim=%d appearance_dim=%d layer_dim=%d "
                   "layers=%d skip_mask=%d rgb_dim=%d",
                   d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->appearance_dim, d->layer_dim, d->layers, d->skip_mask, d->rgb_dim);
}

}  // namespace mnr
