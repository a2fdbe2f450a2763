// mlp_fwd_multi_impl.h -- body of mnr_mlp_forward_multi for one (foreground, background) configuration pair; included by the
// translation units that instantiate a pair (mlp_fwd_multi.hip: the default models; mlp_fwd_multi_sh.hip: the spherical-harmonics
// models of configs/mega-nerf-sh-3) so that the pairs compile in parallel.
#pragma once
#include <stdlib.h>
#include "mlp_fwd_kernels.h"
#include "step_internal.h"

namespace mnr {

static inline long n_cells_of(const mnr_mlp_launch &L, const CellTable &c) { return c.cell_rows > 0 ? L.io->n_rows / c.cell_rows : 0; }

// routed: per segment the device table of a merged container's cells (mnr_mlp_forward_cells_multi), else NULL
struct RoutedSeg { const mnr_mlp_cell *cells; int n_cells; };

// one feature-split workgroup per CU at most (the partial quantum it is meant for); 0 CUs known -> 256
static inline int split_workgroups_max() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) n = cus;
        else n = 256;
    }
    return n;
}

template <class CfgFG, class CfgBG, int NW = 4>
static int mlp_forward_multi_pair(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s, const RoutedSeg *routed = nullptr) {
    constexpr int ROWS_WG = NW * CfgFG::TILE;
    MlpFwdMulti mm{};
    mm.split_seg = -1;
    const bool train = segs[0].tape_dev != nullptr;
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_launch &L = segs[i];
        MNR_REQUIRE(L.desc && L.io && L.io->xyz && (routed || (L.packed_dev && L.io->out)), "segment %d: NULL pointer argument", i);
        if (routed) MNR_REQUIRE(routed[i].cells && routed[i].n_cells >= 1 && routed[i].n_cells <= 64 && !cells && !train, "segment %d: bad cell table", i);
        MNR_REQUIRE((L.tape_dev != nullptr) == train, "segments must be all training or all inference launches");
        MNR_REQUIRE(!L.io->row_index && !L.io->sigma_only, "segment %d: gather / sigma_only are single-launch features", i);
        // spherical-harmonics pair: the colour epilogue (eval_sh + sigmoid) must be ON -- a multi-segment launch writes 4 floats per row
        // ... of exactly the degree the instantiated head has coefficients for: rgb_dim = 3 (deg + 1)^2 (27 <-> 2, 48 <-> 3)
        MNR_REQUIRE(CfgFG::RGB == 3 ? L.io->apply_sh_deg < 0 : (L.io->apply_sh_deg >= 0 && 3 * (L.io->apply_sh_deg + 1) * (L.io->apply_sh_deg + 1) == CfgFG::RGB),
                    "segment %d: apply_sh_deg %d does not fit an rgb head of %d outputs", i, L.io->apply_sh_deg, CfgFG::RGB);
        MNR_REQUIRE(L.io->rows_per_ray >= 1 && L.io->n_rows >= 0, "segment %d: bad row counts", i);
        MNR_REQUIRE(L.io->dir && L.io->idx && L.desc->embedding_a, "segment %d: dir / idx / embedding_a required", i);
        if (train) MNR_REQUIRE(L.tape_row0 >= 0 && L.tape_rows >= L.tape_row0 + L.io->n_rows, "segment %d: tape buffer too small", i);
        ModelLayout m;
        int rc = layout_from_desc(L.desc, m);
        if (rc != MNR_OK) return rc;
        const bool is_bg = L.desc->xyz_dim == 4;
        const mnr_mlp_cell *rcells = routed ? routed[i].cells : nullptr;
        const int rn = routed ? routed[i].n_cells : 0;
        rc = !is_bg ? fill_fwd_args<CfgFG>(mm.seg[i], m, L.packed_dev, L.desc, L.io, L.tape_dev, (long)L.tape_rows, (long)L.tape_row0, rcells, rn)
                    : fill_fwd_args<CfgBG>(mm.seg[i], m, L.packed_dev, L.desc, L.io, L.tape_dev, (long)L.tape_rows, (long)L.tape_row0, rcells, rn);
        if (rc != MNR_OK) return rc;
        if (cells && cells[i].dcells) {
            MNR_REQUIRE(cells[i].cell_rows > 0 && cells[i].cell_rows % ROWS_WG == 0 && L.io->n_rows % cells[i].cell_rows == 0,
                        "segment %d: rows per cell must be a multiple of %d", i, ROWS_WG);
            mm.seg[i].dcells = cells[i].dcells;
            mm.seg[i].cell_rows = cells[i].cell_rows;
        }
        mm.is_b[i] = is_bg ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        // the background segment of a single cell's pass may run as feature-split workgroups of 32 rows (mlp_fwd_split.h): the kernel decides
        // on the device-side row count; the grid covers the finer layout.  (Several cells fill each other's launch tails: not split.)
        const long ncell = cells ? n_cells_of(L, cells[i]) : 1;
        const bool may_split = is_bg && NW == 4 && split_capable<CfgBG>() && !routed && ncell == 1 && i == n_segs - 1 && !getenv("MNR_NO_SPLIT_TAIL");
        if (may_split) {
            mm.split_seg = i;
            mm.split_max = split_workgroups_max();
            mm.split_rpu = L.io->rows_per_unit;
            mm.split_units = L.io->n_units_dev;
            mm.split_dcells = mm.seg[i].dcells;
            mm.split_fixed = mm.seg[i].dcells ? mm.seg[i].cell_rows : (long)L.io->n_rows;
        }
        const long rows_wg = may_split ? 32 : ROWS_WG;
        if (cells) {                       // grid = (workgroups per cell, cells): every segment spans the same cells
            MNR_REQUIRE(cells[i].dcells && n_cells_of(L, cells[i]) == n_cells_of(segs[0], cells[0]) && n_cells_of(L, cells[i]) >= 1,
                        "multi-cell launch: every segment needs a cell table over the same number of cells");
            wg += cells[i].cell_rows / rows_wg;
        } else
        // (routed: the worst case -- every row routed to every cell; workgroups past the device-side counts exit at once)
        wg += routed ? routed_grid((L.io->n_rows + rows_wg - 1) / rows_wg * routed[i].n_cells) : (L.io->n_rows + rows_wg - 1) / rows_wg;
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one MLP launch");
    }
    for (int i = n_segs; i <= MLP_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    mm.nseg = n_segs;
    if (wg == 0) return MNR_OK;
    const unsigned ny = cells ? (unsigned)n_cells_of(segs[0], cells[0]) : 1u;
    constexpr size_t LDS = fwd_lds_bytes<CfgFG, NW>() > fwd_lds_bytes<CfgBG, NW>() ? fwd_lds_bytes<CfgFG, NW>() : fwd_lds_bytes<CfgBG, NW>();
    const int lrc = allow_lds(train ? reinterpret_cast<const void *>(k_mlp_fwd_multi<CfgFG, CfgBG, true, NW>)
                                    : reinterpret_cast<const void *>(k_mlp_fwd_multi<CfgFG, CfgBG, false, NW>), LDS);
    if (lrc != MNR_OK) return lrc;
    if (train) hipLaunchKernelGGL((k_mlp_fwd_multi<CfgFG, CfgBG, true, NW>), dim3((unsigned)wg, ny), dim3(64 * NW), LDS, s, mm);
    else hipLaunchKernelGGL((k_mlp_fwd_multi<CfgFG, CfgBG, false, NW>), dim3((unsigned)wg, ny), dim3(64 * NW), LDS, s, mm);
    return check_launch("k_mlp_fwd_multi");
}

// the pairs of the spherical-harmonics configurations, sh_deg 2 or 3 (mlp_fwd_multi_sh.hip)
int mlp_forward_multi_sh(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, int sh_deg, hipStream_t s);

}  // namespace mnr
