// mlp_fwd_variants.hip -- the remaining inference instantiations of the register-chained forward kernel: spherical-harmonics heads
// (configs/mega-nerf-sh-3), the 64-wide models of the cascade tests, appearance_dim 0 (configs/mega-nerf-no-embed, configs/npp).
// Own translation unit so that they compile beside mlp_fwd.hip.  Dispatch contract: mlp_fwd.hip.
#include "mlp_fwd_kernels.h"

namespace mnr {

#define MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL)                                                     \
    if (d->xyz_dim == XYZ && d->pos_xyz_dim == LX && d->pos_dir_dim == LD && d->appearance_dim == APP &&       \
        d->layer_dim == W && d->layers == NL && d->skip_mask == SKIP && d->rgb_dim == RGB && m.tile == TL)     \
        return launch_fwd<MlpCfg<XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL>>(m, packed_dev, d, io, s, nullptr, 0, 0, cells, n_cells);
#define MNR_TRY(XYZ, LX, LD, APP, W, NL, SKIP, RGB) MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, tile_for_width(W))

int mlp_forward_variants_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                                  const mnr_mlp_cell *cells, int n_cells) {
#ifdef MNR_ALL_VARIANTS
    // configs/mega-nerf-sh-3 (sh_deg 2, pos_dir_dim 0)
    MNR_TRY(3, 12, 0, 48, 256, 8, 16, 27)
    MNR_TRY(4, 12, 0, 48, 256, 8, 16, 27)
    // sh_deg 3 (BASELINE.json's wording of configs[4]): 48 colour coefficients
    MNR_TRY(3, 12, 0, 48, 256, 8, 16, 48)
    MNR_TRY(4, 12, 0, 48, 256, 8, 16, 48)
    // small-width models used by the cascade tests
    MNR_TRY(3, 12, 4, 0, 64, 8, 16, 3)
    MNR_TRY(3, 12, 4, 48, 64, 8, 16, 3)
    MNR_TRY(4, 12, 4, 48, 64, 8, 16, 3)
    // appearance_dim 0 (configs/mega-nerf-no-embed, configs/npp)
    MNR_TRY(3, 12, 4, 0, 256, 8, 16, 3)
    MNR_TRY(4, 12, 4, 0, 256, 8, 16, 3)
#endif
    return 1;
}

}  // namespace mnr
