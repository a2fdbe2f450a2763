// mlp_fwd_multi.hip -- mnr_mlp_forward_multi: the foreground AND the background model's rows of one pass in ONE launch
// (k_mlp_fwd_multi, mlp_fwd_kernels.h); its own translation unit so that it compiles beside mlp_fwd.hip.
#include "mlp_fwd_kernels.h"
#include "step_internal.h"

using namespace mnr;

// the foreground / background pair of the reference's default configuration (configs/mega-nerf/*.yaml)
using CfgFG = MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>;
using CfgBG = MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>;

static bool desc_is(const mnr_model_desc *d, int xyz) {
    return d->xyz_dim == xyz && d->pos_xyz_dim == 12 && d->pos_dir_dim == 4 && d->appearance_dim == 48 && d->layer_dim == 256 &&
           d->layers == 8 && d->skip_mask == 16 && d->rgb_dim == 3;
}

static long n_cells_of(const mnr_mlp_launch &L, const CellTable &c) { return c.cell_rows > 0 ? L.io->n_rows / c.cell_rows : 0; }

int mnr::mlp_forward_multi_impl(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= MLP_MAX_SEGS, "1..%d segments per launch", MLP_MAX_SEGS);
    MlpFwdMulti mm{};
    const bool train = segs[0].tape_dev != nullptr;
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_launch &L = segs[i];
        MNR_REQUIRE(L.packed_dev && L.desc && L.io && L.io->xyz && L.io->out, "segment %d: NULL pointer argument", i);
        MNR_REQUIRE((L.tape_dev != nullptr) == train, "segments must be all training or all inference launches");
        MNR_REQUIRE(!L.io->row_index && !L.io->sigma_only && L.io->apply_sh_deg < 0, "segment %d: gather / sigma_only / SH are single-launch features", i);
        MNR_REQUIRE(L.io->rows_per_ray >= 1 && L.io->n_rows >= 0, "segment %d: bad row counts", i);
        MNR_REQUIRE(L.io->dir && L.io->idx && L.desc->embedding_a, "segment %d: dir / idx / embedding_a required", i);
        if (train) MNR_REQUIRE(L.tape_row0 >= 0 && L.tape_rows >= L.tape_row0 + L.io->n_rows, "segment %d: tape buffer too small", i);
        ModelLayout m;
        int rc = layout_from_desc(L.desc, m);
        if (rc != MNR_OK) return rc;
        const bool is_fg = desc_is(L.desc, 3) && m.tile == 16, is_bg = desc_is(L.desc, 4) && m.tile == 16;
        if (!is_fg && !is_bg) return set_err(MNR_E_UNSUPPORTED, "mnr_mlp_forward_multi covers the default 8x256 fg / bg models");
        rc = is_fg ? fill_fwd_args<CfgFG>(mm.seg[i], m, L.packed_dev, L.desc, L.io, L.tape_dev, (long)L.tape_rows, (long)L.tape_row0, nullptr, 0)
                   : fill_fwd_args<CfgBG>(mm.seg[i], m, L.packed_dev, L.desc, L.io, L.tape_dev, (long)L.tape_rows, (long)L.tape_row0, nullptr, 0);
        if (rc != MNR_OK) return rc;
        if (cells && cells[i].dcells) {
            MNR_REQUIRE(cells[i].cell_rows > 0 && cells[i].cell_rows % CfgFG::ROWS_PER_WG == 0 && L.io->n_rows % cells[i].cell_rows == 0,
                        "segment %d: rows per cell must be a multiple of %d", i, CfgFG::ROWS_PER_WG);
            mm.seg[i].dcells = cells[i].dcells;
            mm.seg[i].cell_rows = cells[i].cell_rows;
        }
        mm.is_b[i] = is_bg ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        if (cells) {                       // grid = (workgroups per cell, cells): every segment spans the same cells
            MNR_REQUIRE(cells[i].dcells && n_cells_of(L, cells[i]) == n_cells_of(segs[0], cells[0]) && n_cells_of(L, cells[i]) >= 1,
                        "multi-cell launch: every segment needs a cell table over the same number of cells");
            wg += cells[i].cell_rows / CfgFG::ROWS_PER_WG;
        } else
        wg += (L.io->n_rows + CfgFG::ROWS_PER_WG - 1) / CfgFG::ROWS_PER_WG;
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one MLP launch");
    }
    for (int i = n_segs; i <= MLP_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    mm.nseg = n_segs;
    if (wg == 0) return MNR_OK;
    const unsigned ny = cells ? (unsigned)n_cells_of(segs[0], cells[0]) : 1u;
    if (train) hipLaunchKernelGGL((k_mlp_fwd_multi<CfgFG, CfgBG, true>), dim3((unsigned)wg, ny), dim3(256), 2 * CHUNK_BYTES, s, mm);
    else hipLaunchKernelGGL((k_mlp_fwd_multi<CfgFG, CfgBG, false>), dim3((unsigned)wg, ny), dim3(256), 2 * CHUNK_BYTES, s, mm);
    return check_launch("k_mlp_fwd_multi");
}

extern "C" int mnr_mlp_forward_multi(const mnr_mlp_launch *segs, int n_segs, void *stream) {
    return mlp_forward_multi_impl(segs, n_segs, nullptr, as_stream(stream));
}
