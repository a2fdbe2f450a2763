// mlp_fwd_h2.hip -- opt-in SPLIT-PRECISION variant of the fused NeRF MLP forward (inference only).
//
// The fp32 kernels (mlp_fwd_kernels.h) run on the fp32 matrix pipe: 157 TFLOP/s, 1/16 of the 16-bit pipe.  Here every
// fp32 operand is split into two f16 halves, x = x_hi + x_lo (x_hi = x with the mantissa cut to 10 bits, x_lo = x - x_hi: exact in
// fp32, then rounded to f16), likewise every weight at pack time, and each layer is three v_mfma_f32_16x16x32_f16 products with
// fp32 accumulation:
//     acc += w_hi x_hi + w_lo x_hi + w_hi x_lo            (dropped: w_lo x_lo, 2^-22 relative)
// Measured on the MI355X before this was built (tools/micro/split_probe.hip, profiles/r03_split_probe.jsonl): an 8-layer
// 256 -> 256 chain incl. the re-split of the activations between layers runs at 335 TFLOP/s fp32-equivalent (the fp32 kernel:
// ~130) with 1.1e-6 error against fp64 after 8 layers, 4.6e-7 per layer -- fp32-class accuracy (bf16 halves: 1.4e-5 per layer).
//
// Structure = the fp32 kernel's: a wavefront owns 16 samples x all features; accumulator register i of lane-part p is feature
// hid_src(4, i, p) (mlp_layout.h), and K-step S of the next layer (32 features) consumes registers 8S .. 8S+7 of every part -- the
// fp32 kernel's step order, 8 steps per MFMA instead of one, so the SAME segment tables (ModelLayout) drive the packer.  8 waves
// (2 per SIMD) share a 2 x 64 KiB LDS ring of fragment-ordered (hi, lo) weight pairs: 128 rows per pass over the 2.5 MB image.
// Value range: |activation| and |weight| < 65504 (f16); smaller than 6e-5 they keep 2^-24 absolute precision.
// Heads (sigma, rgb), biases, activations, positional encodings: fp32 VALU exactly as in the fp32 kernel.
#include "h2_device.h"
#include "mlp_fwd_kernels.h"
#include "step_internal.h"

namespace mnr {

// (layout, packer bodies, weight stream, K-step runner: h2_device.h)

__global__ void k_pack_model_h2(ModelLayout m, uint4v *__restrict__ chunks, float *__restrict__ aux, long n_u4) {
    pack_model_h2_thread(m, chunks, aux, n_u4, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- kernel ------------------------------------------------------------------------------------------------------------------
struct H2Args {
    const uint4v *chunks;
    const float *aux;
    const float *emb_a;
    mnr_mlp_io io;
    int32_t bias_off[MAX_MFMA_LAYERS];
    int32_t sigma_off, rgb_off, sigma_act, app_count;
    // training (tape-writing) launches: the fp32 kernel's activation tape, same planes (mlp_layout.h TapeLayout)
    float *tape;
    long tape_rows, tape_row0;
    TapeLayout tl;
    // several cells' rows side by side in the segment (csrc/step.hip): device table, blockIdx.y = cell
    const MlpCellSeg *dcells;
    long cell_rows, aux_byte_off;
    // routed evaluation (mnr_mlp_forward_cells_h2): workgroups laid out cell after cell, device-side row lists / counts
    const mnr_mlp_cell *cells;
    int n_cells;
};

template <class C, bool TRAIN>
__device__ __forceinline__ void mlp_fwd_h2_body(const H2Args &a, long blk, int cidx) {
    static_assert(C::TILE == 16 && C::W == 256 && C::HAS_FINAL && C::RGB == 3, "split-precision kernel: default 8x256 architectures");
    constexpr int P = C::P, H = C::H, NOB = C::NOB, NOB2 = C::NOB2, H2 = C::H2, RPB = C::RPB;
    constexpr int KE = (C::EX + 7) / 8, KH = H / 8, KD = (C::ED + 7) / 8;
    constexpr int FG = TRAIN ? 2 : H2_FRAG_GROUP;        // fragment read-ahead: the training instantiation has no registers to spare
    extern __shared__ uint4v h2_ring[];
    const mnr_mlp_io &io = a.io;
    const uint4v *chunks = a.chunks;
    const float *aux = a.aux, *emb_a = a.emb_a;
    long n_rows, row_base = 0, tape_row0 = a.tape_row0;
    const int32_t *row_index = nullptr;
    float *outp = io.out;
    if (a.cells) {
        // as the fp32 kernel's routed mode (mlp_fwd_kernels.h): ceil(count_c / rows per workgroup) workgroups per cell, in cell order
        int c = 0;
        n_rows = 0;
        for (; c < a.n_cells; ++c) {
            const long n = *a.cells[c].count, t = (n + H2_ROWS - 1) / H2_ROWS;
            if (blk < t) { n_rows = n; break; }
            blk -= t;
        }
        if (c == a.n_cells) return;
        const mnr_mlp_cell cell = a.cells[c];
        chunks = reinterpret_cast<const uint4v *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed_dev)));
        aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(chunks) + a.aux_byte_off);
        emb_a = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(cell.embedding_a)));
        row_index = cell.row_index;
        outp = cell.out;
    } else if (a.dcells) {
        const MlpCellSeg cell = a.dcells[cidx];
        n_rows = cell.n_units ? (long)__builtin_amdgcn_readfirstlane(*cell.n_units) * io.rows_per_unit : a.cell_rows;
        if (blk * H2_ROWS >= n_rows) return;
        chunks = reinterpret_cast<const uint4v *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed)));
        aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(chunks) + a.aux_byte_off);
        emb_a = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(cell.emb_a)));
        row_base = (long)cidx * a.cell_rows;
        tape_row0 = uniform_long(cell.tape_row0);
    } else {
        n_rows = io.n_units_dev ? (long)(*io.n_units_dev) * io.rows_per_unit : (long)io.n_rows;
        if (blk * H2_ROWS >= n_rows) return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const long lrow = (blk * H2_WAVES + wave) * 16 + (lane & 15);
    const bool valid = lrow < n_rows;
    const long row = row_base + lrow;
    const long rc = row_base + (valid ? lrow : n_rows - 1);
    const long src = row_index ? (long)row_index[rc] : rc;            // gathered evaluation (MegaNeRF router)
    const long ray = src / io.rows_per_ray;
    const unsigned trow0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((blk * H2_WAVES + wave) * 16 + tape_row0));

    H2Stream st;
    st.init(chunks, h2_ring);

    float x[C::XYZ];
#pragma unroll
    for (int d = 0; d < C::XYZ; ++d) x[d] = io.xyz[src * io.xyz_stride + d];
    float ex[C::EX];
    embed<C::XYZ, C::LX, P>(ex, x, part);
    if constexpr (TRAIN) {
        if (valid) tape_store_emb<C::XYZ, C::LX, P>(a.tape + a.tl.embx_off * a.tape_rows, tape_row<16>(trow0), a.tl.embx_w, ex, part);
    }

    float h[H];
    floatx4 acc[NOB];
    // Training: the output plane of a layer goes to the tape right behind the NEXT layer's first chunk boundary (a boundary drains
    // vmcnt, so stores issued just before one would cost the wavefront a write round trip; the registers are the next layer's B
    // operands and stay live).  16 x `global_store_dwordx4 v_off, v[data], s[plane]` per lane (mlp_device.h gstore4).
    const unsigned trow_off = (unsigned)((tape_row<16>(trow0) * C::W + 4 * part) * 4);        // this lane's row in a W-wide plane, bytes (< 2^32: checked by the host)
    auto store_prev = [&](auto planec) {            // hook: h -> activation plane `plane` + its sign bits
        return [&](auto cc) {
            constexpr int ci = decltype(cc)::value, pl = decltype(planec)::value;
            if constexpr (TRAIN && ci == 0) {
                if (valid) {
#ifndef H2_EXPERIMENT_NO_TAPE
                    tape_store_regs_part<P, 0, 16>(a.tape + a.tl.act_off[pl] * a.tape_rows, trow_off, h);
#endif
                    tape_store_mask<P>(a.tape + a.tl.mask_off[pl] * a.tape_rows, tape_row<16>(trow0), a.tl.mask_w, h, part);
                }
            }
        };
    };
    static_for<0, C::NL>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        init_acc<NOB, RPB>(acc, aux + a.bias_off[l] + part * H);
        if constexpr (l == 0) {
            h2_segment_g<NOB, KE, 0, FG, true>(acc, ex, st, lane);
        } else if constexpr ((C::SKIP >> l) & 1) {
            h2_segment_g<NOB, KE, 0, FG, true>(acc, ex, st, lane, store_prev(std::integral_constant<int, l - 1>{}));
            h2_segment_g<NOB, KH, KE, FG, true>(acc, h, st, lane, store_prev(std::integral_constant<int, l - 1>{}));
        } else {
            h2_segment_g<NOB, KH, 0, FG, true>(acc, h, st, lane, store_prev(std::integral_constant<int, l - 1>{}));
        }
        acc_to_regs<NOB, RPB, true>(h, acc);
    });

    // sigma head (nerf.py:132-136): fp32 VALU
    float sigma;
    {
        const float *ws = aux + a.sigma_off;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(ws + part * H + 4 * q);
            s = fmaf(h[4 * q + 0], w4.x, s); s = fmaf(h[4 * q + 1], w4.y, s);
            s = fmaf(h[4 * q + 2], w4.z, s); s = fmaf(h[4 * q + 3], w4.w, s);
        }
        s = reduce_parts<P>(s) + ws[P * H];
        if (io.sigma_noise) s += io.sigma_noise[src];
        sigma = a.sigma_act ? softplus_shifted(s) : fmaxf(s, 0.f);
    }

    // xyz_encoding_final (no activation), then dir_a_encoding over [final | dir embedding | appearance]
    init_acc<NOB, RPB>(acc, aux + a.bias_off[C::NL] + part * H);
    h2_segment_g<NOB, KH, 0, FG, true>(acc, h, st, lane, store_prev(std::integral_constant<int, C::NL - 1>{}));      // (the last trunk layer's plane)
    acc_to_regs<NOB, RPB, false>(h, acc);
    floatx4 acc2[NOB2];
    init_acc<NOB2, RPB>(acc2, aux + a.bias_off[C::NL + 1] + part * H2);
    // ... and xyz_encoding_final's plane during dir_a's first segment (two chunks: half the plane behind each boundary)
    h2_segment_g<NOB2, KH, 0, FG, true>(acc2, h, st, lane, [&](auto cc) {
        if constexpr (TRAIN && decltype(cc)::value == 0) {
            if (valid) tape_store_regs_part<P, 0, 16>(a.tape + a.tl.fin_off * a.tape_rows, trow_off, h);
        }
    });
    {
        float dv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) dv[d] = io.dir[ray * io.dir_stride + d];
        float ed[C::ED];
        embed<3, C::LD, P>(ed, dv, part);
        if constexpr (TRAIN) {
            if (valid) tape_store_emb<3, C::LD, P>(a.tape + a.tl.embd_off * a.tape_rows, tape_row<16>(trow0), a.tl.embd_w, ed, part);
        }
        h2_segment_g<NOB2, KD, KH, FG, true>(acc2, ed, st, lane);
        long idx = io.idx_is_float ? (long)reinterpret_cast<const float *>(io.idx)[ray * io.idx_stride]
                                   : (long)reinterpret_cast<const int32_t *>(io.idx)[ray * io.idx_stride];
        idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
        const float *ea = emb_a + idx * C::APP + part * (C::APP / P);
        float ap[C::AP];
#pragma unroll
        for (int i = 0; i < C::AP; ++i) ap[i] = (i < C::APP / P) ? ea[i] : 0.f;
        if constexpr (TRAIN) {
            if (valid) {
                float *r = a.tape + a.tl.app_off * a.tape_rows + tape_row<16>(trow0) * a.tl.app_w + part * (C::APP / P);
#pragma unroll
                for (int i = 0; i < C::APP / P; ++i) r[i] = ap[i];
            }
        }
        h2_segment_g<NOB2, (C::AP + 7) / 8, KH + KD, FG, true>(acc2, ap, st, lane);
    }
    float dreg[H2];
    acc_to_regs<NOB2, RPB, true>(dreg, acc2);
    if constexpr (TRAIN) {
        if (valid) {
            tape_store_regs<P>(a.tape + a.tl.dact_off * a.tape_rows, tape_row<16>(trow0), C::W / 2, dreg, part);
            tape_store_mask<P>(a.tape + a.tl.dmask_off * a.tape_rows, tape_row<16>(trow0), a.tl.dmask_w, dreg, part);
        }
    }
    float rgbraw[3];
    const float *wr = aux + a.rgb_off;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < H2 / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wr + (c * P + part) * H2 + 4 * q);
            s = fmaf(dreg[4 * q + 0], w4.x, s); s = fmaf(dreg[4 * q + 1], w4.y, s);
            s = fmaf(dreg[4 * q + 2], w4.z, s); s = fmaf(dreg[4 * q + 3], w4.w, s);
        }
        rgbraw[c] = reduce_parts<P>(s) + wr[3 * P * H2 + c];
    }
    if (!(valid && part == 0)) return;
    float *o = outp + row * io.out_stride;
    o[0] = sigmoidf_(rgbraw[0]); o[1] = sigmoidf_(rgbraw[1]); o[2] = sigmoidf_(rgbraw[2]); o[3] = sigma;
}

constexpr int H2_MAX_SEGS = 4;
struct H2Multi {
    H2Args seg[H2_MAX_SEGS];
    int32_t wg0[H2_MAX_SEGS + 1];
    int32_t is_b[H2_MAX_SEGS];
};
template <class CA, class CB, bool TRAIN>
__global__ __launch_bounds__(H2_THREADS, 1) void k_mlp_fwd_h2(H2Multi m) {
    const int blk = blockIdx.x;
    const int s = (blk >= m.wg0[1]) + (blk >= m.wg0[2]) + (blk >= m.wg0[3]);
    if (m.is_b[s]) mlp_fwd_h2_body<CB, TRAIN>(m.seg[s], blk - m.wg0[s], blockIdx.y);
    else mlp_fwd_h2_body<CA, TRAIN>(m.seg[s], blk - m.wg0[s], blockIdx.y);
}

}  // namespace mnr

using namespace mnr;

using H2FG = MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>;
using H2BG = MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>;

int mnr::h2_layout(const mnr_model_desc *d, ModelLayout &m) {
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    const bool ok = (d->xyz_dim == 3 || d->xyz_dim == 4) && d->pos_xyz_dim == 12 && d->pos_dir_dim == 4 && d->appearance_dim == 48 &&
                    d->layer_dim == 256 && d->layers == 8 && d->skip_mask == 16 && d->rgb_dim == 3 && m.tile == 16;
    if (!ok) return set_err(MNR_E_UNSUPPORTED, "the split-precision kernels cover the default 8x256 foreground / background models");
    return MNR_OK;
}

extern "C" size_t mnr_packed_model_h2_bytes(const mnr_model_desc *d) {
    ModelLayout m;
    if (h2_layout(d, m) != MNR_OK) return 0;
    return (size_t)h2_total_chunks(m) * H2_CHUNK_BYTES + (size_t)m.aux_floats * 4;
}

extern "C" int mnr_pack_model_h2(void *packed_dev, size_t bytes, const mnr_model_desc *d, void *stream) {
    ModelLayout m;
    int rc = h2_layout(d, m);
    if (rc != MNR_OK) return rc;
    const size_t need = (size_t)h2_total_chunks(m) * H2_CHUNK_BYTES + (size_t)m.aux_floats * 4;
    MNR_REQUIRE(packed_dev && bytes >= need, "packed buffer missing or too small: %zu < %zu", bytes, need);
    for (int i = 0; i < m.n_mfma_layers; ++i) MNR_REQUIRE(m.layer[i].w && m.layer[i].b, "missing weight/bias pointer for MFMA layer %d", i);
    MNR_REQUIRE(m.sigma_w && m.sigma_b && m.rgb_w && m.rgb_b, "missing sigma/rgb head pointers");
    const long n_u4 = (long)h2_total_chunks(m) * H2_CHUNK_U4, total = n_u4 + m.aux_floats;
    hipLaunchKernelGGL(k_pack_model_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), m,
                       reinterpret_cast<uint4v *>(packed_dev), reinterpret_cast<float *>(reinterpret_cast<char *>(packed_dev) + (size_t)n_u4 * 16), n_u4);
    return check_launch("k_pack_model_h2");
}

static int h2_enable_lds() {
    static bool lds_enabled_dev[MAX_DEVICES] = {};       // raise the dynamic-LDS cap once per device (benign if raced)
    bool &lds_enabled = lds_enabled_dev[device_slot()];
    if (!lds_enabled) {
        for (const void *f : {reinterpret_cast<const void *>(k_mlp_fwd_h2<H2FG, H2BG, false>), reinterpret_cast<const void *>(k_mlp_fwd_h2<H2FG, H2BG, true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H2_CHUNK_BYTES);
            if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_mlp_fwd_h2): %s", hipGetErrorString(e));
        }
        lds_enabled = true;
    }
    return MNR_OK;
}

// all cells of a routed evaluation in one launch (one architecture; inference)
extern "C" int mnr_mlp_forward_cells_h2(const mnr_model_desc *d, const mnr_mlp_cell *cells_dev, int n_cells, const mnr_mlp_io *io, void *stream) {
    MNR_REQUIRE(d && cells_dev && n_cells > 0 && n_cells <= 64 && io && io->xyz && io->dir && io->idx, "bad arguments to mnr_mlp_forward_cells_h2");
    MNR_REQUIRE(!io->sigma_only && io->apply_sh_deg < 0 && io->rows_per_ray >= 1 && io->n_rows >= 0, "sigma_only / SH are not covered by the split-precision kernel");
    ModelLayout m;
    int rc = h2_layout(d, m);
    if (rc != MNR_OK) return rc;
    H2Multi mm{};
    H2Args &a = mm.seg[0];
    a.aux_byte_off = (long)h2_total_chunks(m) * H2_CHUNK_BYTES;
    a.io = *io;
    a.io.row_index = nullptr; a.io.n_units_dev = nullptr;
    for (int k = 0; k < MAX_MFMA_LAYERS; ++k) a.bias_off[k] = k < m.n_mfma_layers ? m.layer[k].bias_off : 0;
    a.sigma_off = m.sigma_off; a.rgb_off = m.rgb_off; a.sigma_act = d->sigma_activation; a.app_count = d->appearance_count;
    a.cells = cells_dev; a.n_cells = n_cells;
    mm.is_b[0] = d->xyz_dim == 4 ? 1 : 0;
    // the worst case (every row routed to every cell); workgroups past the device-side counts exit at once
    const long wg = (io->n_rows + H2_ROWS - 1) / H2_ROWS * n_cells;
    MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one launch");
    for (int i = 1; i <= H2_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    if (wg == 0) return MNR_OK;
    rc = h2_enable_lds();
    if (rc != MNR_OK) return rc;
    hipLaunchKernelGGL((k_mlp_fwd_h2<H2FG, H2BG, false>), dim3((unsigned)wg), dim3(H2_THREADS), 2 * H2_CHUNK_BYTES, as_stream(stream), mm);
    return check_launch("k_mlp_fwd_h2 (cells)");
}

int mnr::mlp_forward_multi_h2_impl(const mnr_mlp_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= H2_MAX_SEGS, "1..%d segments per launch", H2_MAX_SEGS);
    H2Multi mm{};
    const bool train = segs[0].tape_dev != nullptr;
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_launch &L = segs[i];
        MNR_REQUIRE(L.packed_dev && L.desc && L.io && L.io->xyz && L.io->out, "segment %d: NULL pointer argument", i);
        MNR_REQUIRE((L.tape_dev != nullptr) == train, "segments must be all training or all inference launches");
        MNR_REQUIRE(!L.io->row_index && !L.io->sigma_only && L.io->apply_sh_deg < 0, "segment %d: gather / sigma_only / SH are not covered", i);
        MNR_REQUIRE(L.io->rows_per_ray >= 1 && L.io->n_rows >= 0 && L.io->dir && L.io->idx && L.desc->embedding_a, "segment %d: dir / idx / embedding_a required", i);
        if (train) MNR_REQUIRE(L.tape_row0 >= 0 && L.tape_rows >= L.tape_row0 + L.io->n_rows, "segment %d: tape buffer too small", i);
        ModelLayout m;
        int rc = h2_layout(L.desc, m);
        if (rc != MNR_OK) return rc;
        H2Args &a = mm.seg[i];
        a.chunks = reinterpret_cast<const uint4v *>(L.packed_dev);
        a.aux_byte_off = (long)h2_total_chunks(m) * H2_CHUNK_BYTES;
        a.aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(L.packed_dev) + a.aux_byte_off);
        a.emb_a = L.desc->embedding_a;
        a.io = *L.io;
        for (int k = 0; k < MAX_MFMA_LAYERS; ++k) a.bias_off[k] = k < m.n_mfma_layers ? m.layer[k].bias_off : 0;
        a.sigma_off = m.sigma_off; a.rgb_off = m.rgb_off; a.sigma_act = L.desc->sigma_activation; a.app_count = L.desc->appearance_count;
        MNR_REQUIRE(!L.tape_dev || (long)L.tape_rows * L.desc->layer_dim * 4 < (1ll << 32), "tape capacity: a plane must stay below 4 GiB");
        a.tape = L.tape_dev; a.tape_rows = L.tape_rows; a.tape_row0 = L.tape_row0;
        a.tl = tape_layout(ArchDims{L.desc->xyz_dim, L.desc->pos_xyz_dim, L.desc->pos_dir_dim, L.desc->layers, L.desc->skip_mask, L.desc->layer_dim,
                                    L.desc->appearance_dim, L.desc->rgb_dim, L.desc->mfma_tile});
        mm.is_b[i] = L.desc->xyz_dim == 4 ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        if (cells) {
            MNR_REQUIRE(cells[i].dcells && cells[i].cell_rows > 0 && cells[i].cell_rows % H2_ROWS == 0 && L.io->n_rows % cells[i].cell_rows == 0 &&
                        L.io->n_rows / cells[i].cell_rows == segs[0].io->n_rows / cells[0].cell_rows,
                        "segment %d: multi-cell launch needs rows per cell in multiples of %d and the same cells in every segment", i, H2_ROWS);
            a.dcells = cells[i].dcells; a.cell_rows = cells[i].cell_rows;
            wg += cells[i].cell_rows / H2_ROWS;
        } else {
            wg += (L.io->n_rows + H2_ROWS - 1) / H2_ROWS;
        }
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one launch");
    }
    for (int i = n_segs; i <= H2_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    if (wg == 0) return MNR_OK;
    {
        const int rc_lds = h2_enable_lds();
        if (rc_lds != MNR_OK) return rc_lds;
    }
    const unsigned ny = cells ? (unsigned)(segs[0].io->n_rows / cells[0].cell_rows) : 1u;
    if (train) hipLaunchKernelGGL((k_mlp_fwd_h2<H2FG, H2BG, true>), dim3((unsigned)wg, ny), dim3(H2_THREADS), 2 * H2_CHUNK_BYTES, s, mm);
    else hipLaunchKernelGGL((k_mlp_fwd_h2<H2FG, H2BG, false>), dim3((unsigned)wg, ny), dim3(H2_THREADS), 2 * H2_CHUNK_BYTES, s, mm);
    return check_launch("k_mlp_fwd_h2");
}

extern "C" int mnr_mlp_forward_multi_h2(const mnr_mlp_launch *segs, int n_segs, void *stream) {
    MNR_REQUIRE(segs && n_segs >= 1, "NULL argument");
    for (int i = 0; i < n_segs; ++i) MNR_REQUIRE(!segs[i].tape_dev, "mnr_mlp_forward_multi_h2 is the inference entry (the step owns the training form)");
    return mlp_forward_multi_h2_impl(segs, n_segs, nullptr, as_stream(stream));
}
