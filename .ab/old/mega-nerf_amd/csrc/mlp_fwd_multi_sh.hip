// mlp_fwd_multi_sh.hip -- the spherical-harmonics pair of mnr_mlp_forward_multi (configs/mega-nerf-sh-3/*.yaml: sh_deg 2, pos_dir_dim 0:
// rgb head of 27 coefficients, dir_a_encoding over [features | appearance], colour = sigmoid(eval_sh(coefficients, ray direction)) in the
// kernel's epilogue -- spherical_harmonics.py:55-107, rendering.py:301-306); its own translation unit: the pairs compile in parallel.
#include "mlp_fwd_multi_impl.h"

using namespace mnr;

int mnr::mlp_forward_multi_sh(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, int sh_deg, hipStream_t s) {
#ifdef MNR_ALL_VARIANTS
    if (sh_deg == 3)      // 48 coefficients: the degree BASELINE.json's configs[4] words (the shipped yaml files say 2)
        return mlp_forward_multi_pair<MlpCfg<3, 12, 0, 48, 256, 8, 16, 48, 16>, MlpCfg<4, 12, 0, 48, 256, 8, 16, 48, 16>>(segs, n_segs, cells, s);
    using CfgFG = MlpCfg<3, 12, 0, 48, 256, 8, 16, 27, 16>;
    using CfgBG = MlpCfg<4, 12, 0, 48, 256, 8, 16, 27, 16>;
    return mlp_forward_multi_pair<CfgFG, CfgBG>(segs, n_segs, cells, s);
#else
    return set_err(MNR_E_UNSUPPORTED, "built without MNR_ALL_VARIANTS: no spherical-harmonics multi-segment kernels");
#endif
}
