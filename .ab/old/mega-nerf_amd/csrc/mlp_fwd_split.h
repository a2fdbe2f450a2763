// mlp_fwd_split.h -- the register-chained forward with a 16-row tile's OUTPUT FEATURES split over a wavefront pair: the body the
// multi-segment launches give the LAST PARTIAL QUANTUM of their workgroups (the background segment of a single cell's pass).
// Included by mlp_fwd_kernels.h.
//
// Why.  A workgroup of mlp_fwd_body is four wavefronts x 16 rows and takes ~0.14 ms whatever else runs (one wavefront chains 9 632 MFMAs);
// 256 of them -- one per CU -- are a quantum.  The foreground rows of a 1024-ray coarse pass are exactly four quanta; the 69 workgroups of
// its background rows then run alone on 69 CUs for a fifth: 0.14 ms for 0.27 of a quantum's work (the launch quantum,
// measured directly as 0.632 vs 0.775 ms).  Here a workgroup is four wavefronts x 8 rows: wavefronts w and w + 2 own the same 16 rows
// and each computes HALF of every layer's output blocks (half the MFMAs: ~0.07 ms), then the halves swap their results through LDS --
// by the K ordering of mlp_layout.h a half's accumulators are a contiguous half of the next layer's B operands.  Twice the workgroups,
// half as long: the background segment of the coarse pass becomes 138 workgroups on 138 CUs for ~0.08 ms.
//
// Same packed weight image, same chunk stream (a half reads its 8 of a group's 16 fragments), same aux block, same tape layout; every
// output feature's K loop is the fmaf chain mlp_fwd_body runs, and the heads see the full activation: results are BIT-IDENTICAL to
// mlp_fwd_body's.  The kernel decides per launch, on the device-side row count, which body a segment takes (k_mlp_fwd_multi).
#pragma once

namespace mnr {

// The halves of a pair (wavefronts w, w ^ 2) swap their NH output registers, lane for lane, through the 8 KB behind the weight ring
// (mlp_fwd_body's direction-encoding stash: this body keeps that encoding in registers) in rounds of 8 registers:
// full[HALF * NH ..] = own, full[(1 - HALF) * NH ..] = the partner's.  Raw s_barrier + hand-counted lgkmcnt: __syncthreads would put a
// vmcnt(0) -- a wait for the weight DMA in flight -- in front of every one of the 2 * NH / 8 barriers.
template <int NH, int HALF, int NF>
__device__ __forceinline__ void split_exchange(float (&full)[NF], const float (&own)[NH], unsigned xaddr_mine, unsigned xaddr_theirs) {
    static_assert(NH % 8 == 0 && NF >= 2 * NH, "exchange in rounds of eight registers");
    static_for<0, NH / 8>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        floatx4 w0 = {own[8 * r], own[8 * r + 1], own[8 * r + 2], own[8 * r + 3]}, w1 = {own[8 * r + 4], own[8 * r + 5], own[8 * r + 6], own[8 * r + 7]};
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" ::"v"(xaddr_mine), "v"(w0), "v"(w1) : "memory");
        __builtin_amdgcn_s_barrier();
        floatx4 v0, v1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(xaddr_theirs) : "memory");
        constexpr int o = (1 - HALF) * NH + 8 * r;
        full[o] = v0[0]; full[o + 1] = v0[1]; full[o + 2] = v0[2]; full[o + 3] = v0[3];
        full[o + 4] = v1[0]; full[o + 5] = v1[1]; full[o + 6] = v1[2]; full[o + 7] = v1[3];
        __builtin_amdgcn_s_barrier();                          // (the partner has read before the next round overwrites)
    });
#pragma unroll
    for (int i = 0; i < NH; ++i) full[HALF * NH + i] = own[i];
}

template <class C, bool TRAIN, int HALF>
__device__ __forceinline__ void mlp_fwd_split_body(const MlpFwdArgs &a, long blk, int cidx) {
    static_assert(C::TILE == 16 && C::HAS_FINAL && C::NOB % 8 == 0 && C::NOB2 % 8 == 0, "feature split: 16-row tiles, whole four-block batches per half");
    constexpr int TILE = 16, NW = 4, P = C::P, H = C::H, HH = H / 2, NOB = C::NOB, NOBH = NOB / 2, RPB = C::RPB, ROWS_WG = 2 * TILE;
    constexpr int NOB2 = C::NOB2, NOB2H = NOB2 / 2, H2 = C::H2, H2H = H2 / 2;
    extern __shared__ float4 lds_ring[];

    const mnr_mlp_io &io = a.io;
    const float4 *chunks = a.chunks;
    const float *aux = a.aux, *emb_a = a.emb_a;
    float *outp = io.out;
    long n_rows, row_base = 0, tape_row0 = a.tape_row0;
    if (a.dcells) {
        const MlpCellSeg cell = a.dcells[cidx];
        n_rows = cell.n_units ? (long)__builtin_amdgcn_readfirstlane(*cell.n_units) * io.rows_per_unit : a.cell_rows;
        if (blk * ROWS_WG >= n_rows) return;
        chunks = reinterpret_cast<const float4 *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed)));
        aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(chunks) + a.aux_byte_off);
        emb_a = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(cell.emb_a)));
        row_base = (long)cidx * a.cell_rows;
        tape_row0 = uniform_long(cell.tape_row0);
    } else {
        n_rows = io.n_units_dev ? (long)(*io.n_units_dev) * io.rows_per_unit : (long)io.n_rows;
        if (blk * ROWS_WG >= n_rows) return;
    }
    aux = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(aux)));
    emb_a = reinterpret_cast<decltype(emb_a)>(uniform_ptr(reinterpret_cast<const char *>(emb_a)));
    outp = reinterpret_cast<float *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(outp))));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 1;                                 // wavefronts w and w ^ 2 share their 16 rows; HALF = w >> 1
    const int part = lane / TILE;
    const long lrow = (blk * 2 + pair) * TILE + (lane % TILE);
    const bool valid = lrow < n_rows;
    const long row = row_base + lrow;
    const unsigned trow0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((blk * 2 + pair) * TILE + tape_row0));
    const long rc = row_base + (valid ? lrow : n_rows - 1);
    const long src = rc;
    const long ray = src / io.rows_per_ray;
    // exchange slots: 2 KB per wavefront (two float4 per lane) in the 8 KB behind the ring
    const unsigned xbase = lds_addr(lds_ring + 2 * CHUNK_F4);
    const unsigned xmine = xbase + (unsigned)(wave * 2048 + lane * 16), xtheirs = xbase + (unsigned)((wave ^ 2) * 2048 + lane * 16);

    WStreamT<64 * NW> st;
    st.g = reinterpret_cast<const float4 *>(uniform_ptr(reinterpret_cast<const char *>(chunks)));
    st.lds = lds_ring;
    st.cur = 1;
    st.issue();

    float x[C::XYZ];
#pragma unroll
    for (int d = 0; d < C::XYZ; ++d) x[d] = io.xyz[src * io.xyz_stride + d];
    float ex[C::EX];
    embed<C::XYZ, C::LX, P>(ex, x, part);
    if constexpr (TRAIN) {
        if (HALF == 0 && valid) tape_store_emb<C::XYZ, C::LX, P>(a.tape + a.tl.embx_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.embx_w, ex, part);
    }
    float ed[C::ED > 0 ? C::ED : 1];
    if constexpr (C::ED > 0) {
        float dv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) dv[d] = io.dir[ray * io.dir_stride + d];
        float e0[C::ED];
        embed<3, C::LD, P>(e0, dv, part);
#pragma unroll
        for (int i = 0; i < C::ED; ++i) ed[i] = e0[i];
        if constexpr (TRAIN) {
            if (HALF == 1 && valid) tape_store_emb<3, C::LD, P>(a.tape + a.tl.embd_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.embd_w, e0, part);
        }
    }

    float h[H];
    floatx4 acc[NOBH];
    float4 *bias_slot = lds_ring + 2 * CHUNK_F4 + fwd_stash_f4<C, NW>();           // (same LDS map as mlp_fwd_body: ring | >= 8 KB | bias slots)
    auto bias_at = [&](int layer, int regs_per_part, int own_regs) {
        return lds_addr(bias_slot + (layer & 1) * (C::W / 4)) + (unsigned)((part * regs_per_part + HALF * own_regs) * 4);
    };
    bias_dma<C::W, 64 * NW>(aux + a.bias_off[0], bias_slot);
    constexpr bool PUB_PLAIN = seg_weaves<TILE, NOBH, H / 4, C::GPC, 0>();
    constexpr bool PUB_SKIP = seg_weaves<TILE, NOBH, H / 4, C::GPC, C::EX / 4>();
    constexpr bool PUB_L0 = seg_weaves<TILE, NOBH, C::EX / 4, C::GPC, 0>();
    auto publishes = [](int l) constexpr { return l == 0 ? PUB_L0 : (((C::SKIP >> l) & 1) ? PUB_SKIP : PUB_PLAIN); };
    // a layer's output plane: every half stores the float4 pieces of its own blocks (from the full register array, after the exchange);
    // the sign-bit plane needs the whole row and is written by the lower half
    auto store_plane = [&](int act_plane, int mask_plane) __attribute__((always_inline)) {
        if (valid) {
            tape_store_regs_part<P, HALF * (H / 8), H / 8>(a.tape + (long)act_plane * a.tape_rows, tape_row_off<TILE>(trow0, C::W, part), h);
            if (HALF == 0 && mask_plane >= 0) tape_store_mask<P>(a.tape + (long)mask_plane * a.tape_rows, tape_row<TILE>(trow0), a.tl.mask_w, h, part);
        }
    };

    // ---- trunk ----------------------------------------------------------------------------------------------------------------
    static_for<0, C::NL>([&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        constexpr bool PUB = publishes(l);
        if constexpr (l == 0 || !publishes(l > 0 ? l - 1 : 0)) st.next_chunk();
        bias_dma<C::W, 64 * NW>(aux + a.bias_off[l + 1], bias_slot + ((l + 1) & 1) * (C::W / 4));
        init_acc_lds<NOBH, RPB>(acc, bias_at(l, H, HH));
        if constexpr (TRAIN && l > 0) store_plane(a.tl.act_off[l - 1], a.tl.mask_off[l - 1]);
        if constexpr (l == 0) {
            run_segment<TILE, NOBH, C::EX / 4, C::GPC, 0, PUB, NOB, HALF * NOBH>(acc, ex, st, lane);
        } else if constexpr ((C::SKIP >> l) & 1) {
            run_segment<TILE, NOBH, C::EX / 4, C::GPC, 0, false, NOB, HALF * NOBH>(acc, ex, st, lane);
            run_segment<TILE, NOBH, H / 4, C::GPC, C::EX / 4, PUB, NOB, HALF * NOBH>(acc, h, st, lane);
        } else {
            run_segment<TILE, NOBH, H / 4, C::GPC, 0, PUB, NOB, HALF * NOBH>(acc, h, st, lane);
        }
        float o[HH];
        acc_to_regs<NOBH, RPB, true>(o, acc);
        split_exchange<HH, HALF>(h, o, xmine, xtheirs);
    });

    // ---- sigma head (both halves hold the full activation: computed twice, written once) ---------------------------------------------
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

    // ---- xyz_encoding_final (no activation) ----------------------------------------------------------------------------------------
    if constexpr (!publishes(C::NL - 1)) st.next_chunk();
    bias_dma<C::W / 2, 64 * NW>(aux + a.bias_off[C::NL + 1], bias_slot + ((C::NL + 1) & 1) * (C::W / 4));
    init_acc_lds<NOBH, RPB>(acc, bias_at(C::NL, H, HH));
    if constexpr (TRAIN) store_plane(a.tl.act_off[C::NL - 1], a.tl.mask_off[C::NL - 1]);
    run_segment<TILE, NOBH, H / 4, C::GPC, 0, PUB_PLAIN, NOB, HALF * NOBH>(acc, h, st, lane);
    {
        float o[HH];
        acc_to_regs<NOBH, RPB, false>(o, acc);
        split_exchange<HH, HALF>(h, o, xmine, xtheirs);
    }

    // ---- dir_a_encoding ------------------------------------------------------------------------------------------------------------
    floatx4 acc2[NOB2H];
    if constexpr (!PUB_PLAIN) st.next_chunk();
    init_acc_lds<NOB2H, RPB>(acc2, bias_at(C::NL + 1, H2, H2H));
    if constexpr (TRAIN) store_plane(a.tl.fin_off, -1);
    run_segment<TILE, NOB2H, H / 4, C::GPC2, 0, false, NOB2, HALF * NOB2H>(acc2, h, st, lane);
    if constexpr (C::ED > 0) {
        float e1[C::ED];
#pragma unroll
        for (int i = 0; i < C::ED; ++i) e1[i] = ed[i];
        run_segment<TILE, NOB2H, C::ED / 4, C::GPC2, H / 4, false, NOB2, HALF * NOB2H>(acc2, e1, st, lane);
    }
    if constexpr (C::AP > 0) {
        long idx = io.idx_is_float ? (long)reinterpret_cast<const float *>(io.idx)[ray * io.idx_stride]
                                   : (long)reinterpret_cast<const int32_t *>(io.idx)[ray * io.idx_stride];
        idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
        const float *ea = emb_a + idx * C::APP + part * (C::APP / P);
        float ap[C::AP];
#pragma unroll
        for (int i = 0; i < C::AP; ++i) ap[i] = (i < C::APP / P) ? ea[i] : 0.f;
        if constexpr (TRAIN) {
            if (HALF == 1 && valid) {
                float *r = a.tape + a.tl.app_off * a.tape_rows + tape_row<TILE>(trow0) * a.tl.app_w + part * (C::APP / P);
#pragma unroll
                for (int i = 0; i < C::APP / P; ++i) r[i] = ap[i];
            }
        }
        run_segment<TILE, NOB2H, C::AP / 4, C::GPC2, H / 4 + C::ED / 4, false, NOB2, HALF * NOB2H>(acc2, ap, st, lane);
    }
    float dreg[H2];
    {
        float o2[H2H];
        acc_to_regs<NOB2H, RPB, true>(o2, acc2);
        split_exchange<H2H, HALF>(dreg, o2, xmine, xtheirs);
    }
    if constexpr (TRAIN) {
        if (valid) {
            tape_store_regs_part<P, HALF * (H2 / 8), H2 / 8>(a.tape + a.tl.dact_off * a.tape_rows, tape_row_off<TILE>(trow0, C::W / 2, part), dreg);
            if (HALF == 1) tape_store_mask<P>(a.tape + a.tl.dmask_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.dmask_w, dreg, part);
        }
    }
    // ---- rgb head (both halves; written once) ---------------------------------------------------------------------------------------
    float rgbraw[C::RGB];
    const float *wr = aux + a.rgb_off;
#pragma unroll
    for (int c = 0; c < C::RGB; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < H2 / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wr + (c * P + part) * H2 + 4 * q);
            s = fmaf(dreg[4 * q + 0], w4.x, s); s = fmaf(dreg[4 * q + 1], w4.y, s);
            s = fmaf(dreg[4 * q + 2], w4.z, s); s = fmaf(dreg[4 * q + 3], w4.w, s);
        }
        rgbraw[c] = reduce_parts<P>(s) + wr[C::RGB * P * H2 + c];
    }
    if (!(HALF == 0 && valid && part == 0)) return;
    float *o = outp + row * io.out_stride;
    if constexpr (C::RGB == 3) {
        o[0] = sigmoidf_(rgbraw[0]); o[1] = sigmoidf_(rgbraw[1]); o[2] = sigmoidf_(rgbraw[2]); o[3] = sigma;
    } else {
        if (io.apply_sh_deg >= 0) {
            constexpr int NB = C::RGB / 3;
            const float dx = io.dir[ray * io.dir_stride], dy = io.dir[ray * io.dir_stride + 1], dz = io.dir[ray * io.dir_stride + 2];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = sigmoidf_(eval_sh_channel(io.apply_sh_deg, rgbraw + c * NB, dx, dy, dz));
            o[3] = sigma;
        } else {
#pragma unroll
            for (int c = 0; c < C::RGB; ++c) o[c] = rgbraw[c];
            o[C::RGB] = sigma;
        }
    }
}

}  // namespace mnr
