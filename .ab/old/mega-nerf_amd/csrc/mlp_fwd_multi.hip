// mlp_fwd_multi.hip -- mnr_mlp_forward_multi: the foreground AND the background model's rows of one pass in ONE launch
// (k_mlp_fwd_multi, mlp_fwd_kernels.h); its own translation unit so that it compiles beside mlp_fwd.hip.
#include <stdlib.h>

#include "mlp_fwd_multi_impl.h"

using namespace mnr;

// the foreground / background pair of the reference's default configuration (configs/mega-nerf/*.yaml)
using CfgFG = MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>;
using CfgBG = MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>;

// 0: not a multi-launch architecture; 1: the default pair; 2 / 3: the spherical-harmonics pair of that degree (rgb_dim 27 / 48, no direction encoding)
static int pair_of(const mnr_model_desc *d) {
    const bool trunk = (d->xyz_dim == 3 || d->xyz_dim == 4) && d->pos_xyz_dim == 12 && d->appearance_dim == 48 && d->layer_dim == 256 &&
                       d->layers == 8 && d->skip_mask == 16 && (d->mfma_tile == 0 || d->mfma_tile == 16);
    if (!trunk) return 0;
    if (d->pos_dir_dim == 4 && d->rgb_dim == 3) return 1;
    if (d->pos_dir_dim == 0 && d->rgb_dim == 27) return 2;
    if (d->pos_dir_dim == 0 && d->rgb_dim == 48) return 3;
    return 0;
}

int mnr::mlp_forward_multi_impl(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= MLP_MAX_SEGS, "1..%d segments per launch", MLP_MAX_SEGS);
    int pair = -1;
    for (int i = 0; i < n_segs; ++i) {
        MNR_REQUIRE(segs[i].desc, "segment %d: NULL pointer argument", i);
        const int p = pair_of(segs[i].desc);
        if (p == 0 || (pair >= 0 && p != pair))
            return set_err(MNR_E_UNSUPPORTED, "mnr_mlp_forward_multi covers the default 8x256 fg / bg models and their spherical-harmonics (sh_deg 2 / 3) forms");
        pair = p;
    }
#ifdef MNR_EXPERIMENT_8WAVES
    // Experiment (round 4, -DMNR_EXPERIMENT_8WAVES + MNR_FWD_8WAVES=1): eight wavefronts per workgroup share one weight stream (128 rows
    // per pass, one workgroup per CU: half the stream traffic and half the barriers per CU).  Measured on the benchmark step: eval 2.08 ->
    // 2.38 ms, training forward 0.84 + 1.45 -> 1.02 + 1.73 ms, the 8-cell set 47.5 -> 49.4 ms: two independent four-wavefront workgroups
    // per CU, whose chunk barriers interleave, beat one barrier domain of eight -- also after the asm fragment reads of run_segment removed
    // the per-chunk vmcnt(0) wait (re-measured: eval 2.04 vs 2.34 ms, training forward 0.81 + 1.42 vs 0.97 + 1.60 ms).  Not instantiated by default.
    if (pair == 1 && getenv("MNR_FWD_8WAVES")) {
        bool ok = true;
        for (int i = 0; i < n_segs; ++i) ok = ok && (!cells || cells[i].cell_rows % 128 == 0);
        if (ok) return mlp_forward_multi_pair<CfgFG, CfgBG, 8>(segs, n_segs, cells, s);
    }
#endif
    // (pair codes 2 / 3 are the spherical-harmonics pairs of rgb_dim 27 / 48, i.e. of degree 2 / 3: passed on as the DEGREE)
    const int sh_deg = pair == 2 ? 2 : 3;
    return pair == 1 ? mlp_forward_multi_pair<CfgFG, CfgBG>(segs, n_segs, cells, s) : mlp_forward_multi_sh(segs, n_segs, cells, sh_deg, s);
}

extern "C" int mnr_mlp_forward_multi(const mnr_mlp_launch *segs, int n_segs, void *stream) {
    return mlp_forward_multi_impl(segs, n_segs, nullptr, as_stream(stream));
}

// Routed evaluations of several merged models in one launch: segment = one container's cells (mnr_mlp_forward_cells semantics)
extern "C" int mnr_mlp_forward_cells_multi(const mnr_mlp_cells_launch *segs, int n_segs, void *stream) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= MLP_MAX_SEGS, "1..%d segments per launch", MLP_MAX_SEGS);
    mnr_mlp_launch L[MLP_MAX_SEGS] = {};
    RoutedSeg R[MLP_MAX_SEGS];
    for (int i = 0; i < n_segs; ++i) {
        MNR_REQUIRE(segs[i].desc && segs[i].cells_dev && segs[i].io, "segment %d: NULL pointer argument", i);
        if (pair_of(segs[i].desc) != 1)
            return set_err(MNR_E_UNSUPPORTED, "mnr_mlp_forward_cells_multi covers containers of the default 8x256 fg / bg models");
        L[i].desc = segs[i].desc;
        L[i].io = segs[i].io;
        R[i] = RoutedSeg{segs[i].cells_dev, segs[i].n_cells};
    }
    return mlp_forward_multi_pair<CfgFG, CfgBG>(L, n_segs, nullptr, as_stream(stream), R);
}
