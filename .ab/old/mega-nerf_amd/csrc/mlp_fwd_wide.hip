// mlp_fwd_wide.hip -- inference instantiations of the register-chained forward kernel for 512-wide models with ONE wavefront per SIMD
// (128 input + 128 accumulator registers per lane).  The default for these models is the wavefront-pair kernel (mlp_fwd_pair.hip);
// these stay as its cross-check (MNR_NO_PAIR_KERNEL=1, tests/test_gpu_parity_extra.py::test_pair_kernel_equals_...).  Own translation
// unit: two minutes of compile time each.  Dispatch contract: mlp_fwd.hip.
#include "mlp_fwd_kernels.h"

namespace mnr {

#define MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL)                                                     \
    if (d->xyz_dim == XYZ && d->pos_xyz_dim == LX && d->pos_dir_dim == LD && d->appearance_dim == APP &&       \
        d->layer_dim == W && d->layers == NL && d->skip_mask == SKIP && d->rgb_dim == RGB && m.tile == TL)     \
        return launch_fwd<MlpCfg<XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL>>(m, packed_dev, d, io, s, nullptr, 0, 0, cells, n_cells);
#define MNR_TRY(XYZ, LX, LD, APP, W, NL, SKIP, RGB) MNR_TRY_T(XYZ, LX, LD, APP, W, NL, SKIP, RGB, tile_for_width(W))

int mlp_forward_wide_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                              const mnr_mlp_cell *cells, int n_cells) {
    MNR_TRY(3, 12, 4, 48, 512, 8, 16, 3)
    MNR_TRY(4, 12, 4, 48, 512, 8, 16, 3)
    return 1;
}

}  // namespace mnr
