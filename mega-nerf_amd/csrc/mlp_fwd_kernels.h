// mlp_fwd_kernels.h -- the register-chained forward kernel templates (k_mlp_fwd, k_mlp_fwd_multi) and their launch helpers.
// Included by the three translation units that instantiate them -- mlp_fwd.hip (inference), mlp_fwd_train.hip (tape-writing
// instantiations), mlp_fwd_multi.hip (two-model launches) -- so that the instantiations compile in parallel (one unit took
// 7.5 minutes).  Design notes: mlp_fwd.hip.
#pragma once
#include <stdlib.h>
#include "mlp_device.h"
#include "sh_device.h"

namespace mnr {

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m);

struct MlpFwdArgs {
    const float4 *chunks;
    const float *aux;
    const float *emb_a;
    mnr_mlp_io io;
    int32_t bias_off[MAX_MFMA_LAYERS];
    int32_t sigma_off, rgb_off;
    int32_t sigma_act, app_count;
    const mnr_mlp_cell *cells;   // batched routed evaluation: per-cell weights / row lists / outputs (device array), else NULL
    int n_cells;
    int xcd_order;               // routed evaluation: every XCD takes a CONTIGUOUS eighth of the cell-after-cell workgroup sequence (see xcd_contiguous)
    long aux_byte_off;           // offset of the aux block inside a packed image (same for all cells of one architecture)
    const MlpCellSeg *dcells;    // several cells' rows side by side in one segment (device table), else NULL
    long cell_rows;              // ... rows per cell (capacity; a multiple of the rows per workgroup)
    float *tape;              // training only: activation tape (TapeLayout planes), else NULL
    long tape_rows;           // row capacity of every tape plane
    long tape_row0;           // tape row of this launch's row 0
    TapeLayout tl;
};

// The tape row of a lane, re-derived at every store site from an SGPR (first tape row of the wave) and a freshly read lane
// id (v_mbcnt: needs no input register): nothing row-related stays live in VGPRs across the layers.  Kept as a value computed
// once, the row offset was spilled (the kernel sits at its 256-register budget), and every layer's reload (`scratch_load` +
// vmcnt(0)) drained the weight chunk that had just been requested.
template <int TILE>
__device__ __forceinline__ long tape_row(unsigned wave_row0) {
    unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return (long)(wave_row0 + (l & (TILE - 1)));
}

// byte offset of this lane's 16-byte column group inside a `width`-float-wide plane (rows < 2^32 / (4 width): checked by the host);
// re-derived at every use like tape_row, one 32-bit VGPR while it lives
template <int TILE>
__device__ __forceinline__ unsigned tape_row_off(unsigned wave_row0, int width, int part) {
    return (unsigned)(((unsigned)tape_row<TILE>(wave_row0) * (unsigned)width + 4u * (unsigned)part) * 4u);
}

// tape stores (training): flat register i of a C-layout array <-> feature 4P*(i/4) + 4*part + i%4
template <int P, int NH>
__device__ __forceinline__ void tape_store_regs(float *plane, long row, int width, const float (&h)[NH], int part) {
    float *r = plane + row * width + 4 * part;
#pragma unroll
    for (int q = 0; q < NH / 4; ++q)
        *reinterpret_cast<float4 *>(r + 4 * P * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
}
// ... float4 pieces Q0 .. Q0 + NQ - 1 of a row only, addressed as uniform plane + 32-bit row offset in bytes (mlp_device.h gstore4): the
// split-precision kernels spread a plane's stores over the chunk periods of the following layer
template <int P, int Q0, int NQ, int NH>
__device__ __forceinline__ void tape_store_regs_part(const float *plane, unsigned row_byte_off, const float (&h)[NH]) {
    static_assert(4 * (Q0 + NQ) <= NH, "piece range");
    static_for<Q0, Q0 + NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        gstore4<16 * P * q>(plane, row_byte_off, make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]));
    });
}
// ReLU sign bits of a C-layout register array, packed per lane (TapeLayout mask planes)
template <int P, int NH>
__device__ __forceinline__ void tape_store_mask(float *plane, long row, int width, const float (&h)[NH], int part) {
#ifdef MNR_EXPERIMENT_NO_MASK          // timing experiment only (results invalid): what do the sign-bit planes cost?
    return;
#endif
    constexpr int NW = (NH + 31) / 32;
    uint32_t *r = reinterpret_cast<uint32_t *>(plane) + row * width + part * NW;
    uint32_t w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = 0u;
#pragma unroll
    // h is a ReLU output (>= +0, never -0: relu_bits), so h > 0 <=> its bit pattern is non-zero: min(bits, 1) << i, OR-ed in with
    // one v_lshl_or_b32 -- 2 VALU instructions per value (compare + select + or: 3)
    // (inline asm: LLVM turns the min back into compare + select)
    for (int i = 0; i < NH; ++i) {
        unsigned t;
        asm("v_min_u32 %1, 1, %2\n\tv_lshl_or_b32 %0, %1, %3, %0" : "+v"(w[i / 32]), "=&v"(t) : "v"(h[i]), "n"(i % 32));
    }
    if constexpr (NW == 2) *reinterpret_cast<uint2 *>(r) = make_uint2(w[0], w[1]);
    else {
#pragma unroll
        for (int i = 0; i < NW; ++i) r[i] = w[i];
    }
}
// positional-encoding registers -> reference column order (nerf.py:20-25)
template <int D, int L, int P, int NE>
__device__ __forceinline__ void tape_store_emb(float *plane, long row, int width, const float (&e)[NE], int part) {
    constexpr int NP = emb_pairs(D, L, P);
    float *r = plane + row * width;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int col = D + (part * (L / P) + i / D) * 2 * D + i % D;
        r[col] = e[2 * i];
        r[col + D] = e[2 * i + 1];
    }
#pragma unroll
    for (int j = 0; j < cdiv(D, P); ++j) {
        const int dim = j * P + part;
        if (dim < D) r[dim] = e[2 * NP + j];
    }
}

// Direction-encoding stash.  `embed` is sincosf -- its large-argument path alone wants ~40 registers -- and the dir_a layer, where the
// encoding is consumed, is where a lane holds the 64 feature registers, 32 accumulators and 32 fragment registers in flight: evaluated
// there it pushed 51-63 registers to scratch (rounds 1-4; the spherical-harmonics pairs, which have no direction encoding, never
// spilled).  It is evaluated at the start of the kernel instead, next to the position encoding where nothing else is live, parked in
// a lane-private LDS slot behind the weight ring (ED floats per lane; slot i of thread t at (i * NT + t) * 4: conflict-free) and read
// back in front of its K segment.  asm on both sides: a compiler-visible access to the ring's array would get a vmcnt(0) (mlp_device.h).
// the region behind the weight ring: mlp_fwd_body's direction-encoding stash, mlp_fwd_split_body's exchange slots (8 KB) -- float4 units
template <class C, int NW>
constexpr int fwd_stash_f4() { return (C::ED * 64 * NW * 4 >= 8192 ? C::ED * 64 * NW * 4 : 8192) / 16; }
template <class C, int NW>
constexpr size_t fwd_lds_bytes() { return (size_t)2 * CHUNK_BYTES + (size_t)fwd_stash_f4<C, NW>() * 16 + (size_t)2 * C::W * 4; }
template <int I, int NT>
__device__ __forceinline__ void stash_put(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(I * NT * 4) : "memory"); }
template <int I, int NT>
__device__ __forceinline__ float stash_get(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(I * NT * 4) : "memory");
    return v;
}

// Biases through LDS.  A layer's accumulators start as its bias (64 registers per lane at W = 256); fetched from the aux block with vector
// loads at the top of every layer, they put an L2 round trip in front of the layer's first MFMA (1.5 % of the forward kernels, measured
// with a zero-bias build in round 4) and, being FLAT loads, counted on lgkmcnt as well.  Now the NEXT layer's bias row (W floats = one
// 16-byte piece per lane of W / 4 lanes) rides the LDS-DMA engine like the weight chunks do: requested at the top of a layer into one
// of two slots behind the encoding stash, published by the chunk barriers in between, read with W / 64 broadcast ds_read_b128 per lane.
template <int W, int NT>
__device__ __forceinline__ void bias_dma(const float *aux_bias, float4 *slot) {
    if ((int)threadIdx.x < W / 4) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        unsigned lo = threadIdx.x * 16u;
        asm("" : "+v"(lo));
        __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(reinterpret_cast<const char *>(aux_bias)) + lo),
                                         (lds_void_t *)(slot + wave * 64), 16, 0, 0);
    }
}
template <int NOB, int RPB, class AccT>
__device__ __forceinline__ void init_acc_lds(AccT (&acc)[NOB], unsigned addr) {      // addr: this lane-part's first bias float in the slot
    // eight quads per round (a 32-block layer in one round left the whole kernel's arrays in scratch: hipcc gave up promoting them)
    constexpr int NQ = NOB * RPB / 4, QR = NQ < 8 ? NQ : 8;
    static_assert(NQ % QR == 0, "bias quads per round");
    static_for<0, NQ / QR>([&](auto rc) {
        constexpr int q0 = decltype(rc)::value * QR;
        floatx4 t[QR];
        static_for<0, QR>([&](auto qc) { t[decltype(qc)::value] = lds_ld4<(q0 + decltype(qc)::value) * 16>(addr); });
        wait_lgkm<0>();
        static_for<0, QR>([&](auto qc) {                        // (uses stay behind the wait: the compiler does not see an asm read's latency)
            constexpr int q = decltype(qc)::value;
            pin(t[q]);
            static_for<0, 4>([&](auto ec) {
                constexpr int e = 4 * (q0 + q) + decltype(ec)::value;          // flat accumulator register
                acc[e / RPB][e % RPB] = t[q][decltype(ec)::value];
            });
        });
    });
}

// Routed evaluation, workgroup order.  The hardware deals workgroups to the 8 XCDs round-robin (flat id % 8), so with the cells laid out one
// after another every XCD's L2 sees the weight streams of ALL the cells in flight (2-5 of them: 2.4 MB each at 256 channels, 9.3 MB at 512, against
// 4 MB of L2).  Here XCD x takes logical workgroups [x * ceil(T / 8), (x + 1) * ceil(T / 8)) of the T the device-side counts add up to, in
// order: one cell's stream per L2 at a time, as in a single-cell launch.  `blk` = index inside the segment (whose first workgroup is
// blockIdx.x - blk); returns the logical index, or -1 for a surplus workgroup.  (The grid is the worst case rounded up to 8 plus 8, so
// every residue has at least ceil(T / 8) workgroups.)
__device__ __forceinline__ long xcd_contiguous(long blk, long T) {
    const int x = blockIdx.x & 7;
    const int o = (int)(((long)blockIdx.x - blk) & 7);
    const long j = (blk - ((x - o) & 7)) >> 3;
    const long per = (T + 7) >> 3;
    return j < per ? x * per + j : -1;
}

// NW = wavefronts per workgroup sharing one weight stream (4: two workgroups per CU; 8: one -- half the stream traffic and barriers per CU)
template <class C, bool TRAIN, int NW = 4>
__device__ __forceinline__ void mlp_fwd_body(const MlpFwdArgs &a, long blk, int cidx = 0) {
    constexpr int TILE = C::TILE, P = C::P, H = C::H, NOB = C::NOB, RPB = C::RPB, ROWS_WG = NW * TILE;
    using AccT = typename std::conditional<TILE == 32, floatx16, floatx4>::type;
    extern __shared__ float4 lds_ring[];

    const mnr_mlp_io &io = a.io;
    const float4 *chunks = a.chunks;
    const float *aux = a.aux, *emb_a = a.emb_a;
    const int32_t *row_index = io.row_index;
    float *outp = io.out;
    long n_rows, row_base = 0, tape_row0 = a.tape_row0;
    if (a.cells) {
        // One launch for all cells of a routed evaluation: workgroups are laid out cell after cell, ceil(count_c / rows
        // per workgroup) each; everything below is uniform per workgroup, so the per-cell pointers stay in SGPRs.
        if (a.xcd_order) {
            long T = 0;
            for (int c = 0; c < a.n_cells; ++c) T += ((long)*a.cells[c].count + ROWS_WG - 1) / ROWS_WG;
            blk = xcd_contiguous(blk, T);
            if (blk < 0) return;
        }
        int c = 0;
        n_rows = 0;
        for (; c < a.n_cells; ++c) {
            const long n = *a.cells[c].count, t = (n + ROWS_WG - 1) / ROWS_WG;
            if (blk < t) { n_rows = n; break; }
            blk -= t;
        }
        if (c == a.n_cells) return;
        const mnr_mlp_cell cell = a.cells[c];
        chunks = reinterpret_cast<const float4 *>(cell.packed_dev);
        aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(cell.packed_dev) + a.aux_byte_off);
        emb_a = cell.embedding_a;
        row_index = cell.row_index;
        outp = cell.out;
    } else if (a.dcells) {
        // Training step of several submodules: cell c = blockIdx.y owns rows [c * cell_rows, (c + 1) * cell_rows) of the
        // segment's arrays; `blk` is the workgroup index inside the cell, `row_base` the cell's offset into the shared arrays.
        // (cidx = blockIdx.y: no division; the table entry comes through vector loads -> everything is moved to SGPRs at once)
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
        if (blk * ROWS_WG >= n_rows) return;             // uniform per workgroup
    }

    // the per-cell pointers come out of a device table (vector loads), so the merged values would live in VGPR pairs for the
    // whole kernel -- and get spilled: every layer then reloaded `aux` from scratch and waited vmcnt(0) for it, draining the
    // weight prefetch in the middle of the layer.  They are uniform: move them to SGPRs.
    if constexpr (TRAIN) {          // (the eval instantiation fits its 252 registers without this and spills 26 with it)
        aux = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(aux)));
        emb_a = reinterpret_cast<decltype(emb_a)>(uniform_ptr(reinterpret_cast<const char *>(emb_a)));
        outp = reinterpret_cast<float *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(outp))));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane / TILE;
    const long lrow = (blk * NW + wave) * TILE + (lane % TILE);          // row inside the segment (inside the cell: dcells)
    const bool valid = lrow < n_rows;
    const long row = row_base + lrow;
    // first tape row of this wave (training; uniform -> SGPR; tapes hold < 2^32 rows)
    const unsigned trow0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((blk * NW + wave) * TILE + tape_row0));
    const long rc = row_base + (valid ? lrow : n_rows - 1);
    const long src = row_index ? (long)row_index[rc] : rc;           // gathered evaluation (MegaNeRF router)
    const long ray = src / io.rows_per_ray;

    WStreamT<64 * NW> st;
    st.g = reinterpret_cast<const float4 *>(uniform_ptr(reinterpret_cast<const char *>(chunks)));      // into SGPRs once: the stream pointer arithmetic stays scalar
    st.lds = lds_ring;
    st.cur = 1;
    st.issue();                                   // chunk 0 in flight while we encode

    float x[C::XYZ];
#pragma unroll
    for (int d = 0; d < C::XYZ; ++d) x[d] = io.xyz[src * io.xyz_stride + d];
    float ex[C::EX];
    embed<C::XYZ, C::LX, P>(ex, x, part);
    if constexpr (TRAIN) {
        if (valid) tape_store_emb<C::XYZ, C::LX, P>(a.tape + a.tl.embx_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.embx_w, ex, part);
    }

    if constexpr (C::ED > 0) {
        if (!io.sigma_only) {                                // (a density-only launch has no directions: io.dir may be NULL)
            float dv[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) dv[d] = io.dir[ray * io.dir_stride + d];
            float ed[C::ED];
            embed<3, C::LD, P>(ed, dv, part);
            if constexpr (TRAIN) {
                if (valid) tape_store_emb<3, C::LD, P>(a.tape + a.tl.embd_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.embd_w, ed, part);
            }
            const unsigned sa = lds_addr(lds_ring + 2 * CHUNK_F4) + threadIdx.x * 4u;
            static_for<0, C::ED>([&](auto ic) { stash_put<decltype(ic)::value, 64 * NW>(sa, ed[decltype(ic)::value]); });
        }
    }

    float h[H];
    AccT acc[NOB];

    // bias rows: two LDS slots behind the encoding stash, the next layer's row requested at the top of every layer (bias_dma above)
    float4 *bias_slot = lds_ring + 2 * CHUNK_F4 + fwd_stash_f4<C, NW>();
    auto bias_at = [&](int layer, int regs_per_part) {
        return lds_addr(bias_slot + (layer & 1) * (C::W / 4)) + (unsigned)(part * regs_per_part * 4);
    };
    bias_dma<C::W, 64 * NW>(aux + a.bias_off[0], bias_slot);
    // a layer whose last K segment runs as the woven pipeline publishes the NEXT layer's first chunk itself (run_segment PUB_END)
    constexpr bool PUB_PLAIN = seg_weaves<TILE, NOB, H / 4, C::GPC, 0>();
    constexpr bool PUB_SKIP = seg_weaves<TILE, NOB, H / 4, C::GPC, C::EX / 4>();
    constexpr bool PUB_L0 = seg_weaves<TILE, NOB, C::EX / 4, C::GPC, 0>();
    auto publishes = [](int l) constexpr {             // does trunk layer l publish for its successor?
        if (l + 1 >= C::NL && !C::HAS_FINAL) return false;
        return l == 0 ? PUB_L0 : (((C::SKIP >> l) & 1) ? PUB_SKIP : PUB_PLAIN);
    };

    // ---- trunk: nerf.py:127-130 ------------------------------------------------------------------
    // (always_inline: left to the inliner's cost model, the 512-wide instantiation kept the skip layer's body as a FUNCTION, and everything the
    // lambda captures by reference -- accumulators, activations, stream -- then lived in scratch: 58 000 scratch instructions)
    static_for<0, C::NL>([&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        constexpr bool PUB = publishes(l);
        if constexpr (l == 0 || !publishes(l > 0 ? l - 1 : 0)) st.next_chunk();
        if constexpr (l + 1 < C::NL || C::HAS_FINAL) bias_dma<C::W, 64 * NW>(aux + a.bias_off[l + 1], bias_slot + ((l + 1) & 1) * (C::W / 4));
        init_acc_lds<NOB, RPB>(acc, bias_at(l, H));
        if constexpr (TRAIN && l > 0) {
            // Tape stores of the previous layer's output go out in front of the layer's first MFMAs: the next chunk barrier (which
            // drains vmcnt) is six batches away, so a store issued here has ~3 000 cycles to retire.
            if (valid) {
                tape_store_regs_part<P, 0, H / 4>(a.tape + a.tl.act_off[l - 1] * a.tape_rows, tape_row_off<TILE>(trow0, C::W, part), h);
                tape_store_mask<P>(a.tape + a.tl.mask_off[l - 1] * a.tape_rows, tape_row<TILE>(trow0), a.tl.mask_w, h, part);
            }
        }
        if constexpr (l == 0) {
            run_segment<TILE, NOB, C::EX / 4, C::GPC, 0, PUB>(acc, ex, st, lane);
        } else if constexpr ((C::SKIP >> l) & 1) {
            run_segment<TILE, NOB, C::EX / 4, C::GPC, 0>(acc, ex, st, lane);
            run_segment<TILE, NOB, H / 4, C::GPC, C::EX / 4, PUB>(acc, h, st, lane);
        } else {
            run_segment<TILE, NOB, H / 4, C::GPC, 0, PUB>(acc, h, st, lane);
        }
        acc_to_regs<NOB, RPB, true>(h, acc);
    });

    // ---- sigma head: nerf.py:132-136 -------------------------------------------------------------
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
    if (io.sigma_only) {
        if (valid && part == 0) outp[row * io.out_stride] = sigma;
        return;
    }

    // ---- colour branch: nerf.py:141-152 ----------------------------------------------------------
    float rgbraw[C::RGB];
    const float *wr = aux + a.rgb_off;
    if constexpr (C::HAS_FINAL) {
        if constexpr (!publishes(C::NL - 1)) st.next_chunk();
        bias_dma<C::W / 2, 64 * NW>(aux + a.bias_off[C::NL + 1], bias_slot + ((C::NL + 1) & 1) * (C::W / 4));
        init_acc_lds<NOB, RPB>(acc, bias_at(C::NL, H));
        if constexpr (TRAIN) {                                   // deferred store of the last trunk layer (see above)
            if (valid) {
                tape_store_regs_part<P, 0, H / 4>(a.tape + a.tl.act_off[C::NL - 1] * a.tape_rows, tape_row_off<TILE>(trow0, C::W, part), h);
                tape_store_mask<P>(a.tape + a.tl.mask_off[C::NL - 1] * a.tape_rows, tape_row<TILE>(trow0), a.tl.mask_w, h, part);
            }
        }
        run_segment<TILE, NOB, H / 4, C::GPC, 0, PUB_PLAIN>(acc, h, st, lane);
        acc_to_regs<NOB, RPB, false>(h, acc);                    // xyz_encoding_final: no activation

        constexpr int NOB2 = C::NOB2, H2 = C::H2;
        AccT acc2[NOB2];
        if constexpr (!PUB_PLAIN) st.next_chunk();
        init_acc_lds<NOB2, RPB>(acc2, bias_at(C::NL + 1, H2));
        if constexpr (TRAIN) {
            if (valid) tape_store_regs_part<P, 0, H / 4>(a.tape + a.tl.fin_off * a.tape_rows, tape_row_off<TILE>(trow0, C::W, part), h);
        }
        run_segment<TILE, NOB2, H / 4, C::GPC2, 0>(acc2, h, st, lane);
        if constexpr (C::ED > 0) {
            float ed[C::ED];
            unsigned sa = lds_addr(lds_ring + 2 * CHUNK_F4) + threadIdx.x * 4u;
            asm volatile("" : "+v"(sa));        // formed here, not kept in a register since the start of the kernel
            static_for<0, C::ED>([&](auto ic) { ed[decltype(ic)::value] = stash_get<decltype(ic)::value, 64 * NW>(sa); });
            wait_lgkm<0>();
#pragma unroll
            for (int i = 0; i < C::ED; ++i) pin(ed[i]);
            run_segment<TILE, NOB2, C::ED / 4, C::GPC2, H / 4>(acc2, ed, st, lane);
        }
        if constexpr (C::AP > 0) {
            long idx = io.idx_is_float ? (long)reinterpret_cast<const float *>(io.idx)[ray * io.idx_stride]
                                       : (long)reinterpret_cast<const int32_t *>(io.idx)[ray * io.idx_stride];
            idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);   // reference would raise; stay in bounds
            const float *ea = emb_a + idx * C::APP + part * (C::APP / P);
            float ap[C::AP];
#pragma unroll
            for (int i = 0; i < C::AP; ++i) ap[i] = (i < C::APP / P) ? ea[i] : 0.f;
            if constexpr (TRAIN) {
                if (valid) {
                    float *r = a.tape + a.tl.app_off * a.tape_rows + tape_row<TILE>(trow0) * a.tl.app_w + part * (C::APP / P);
#pragma unroll
                    for (int i = 0; i < C::APP / P; ++i) r[i] = ap[i];
                }
            }
            run_segment<TILE, NOB2, C::AP / 4, C::GPC2, H / 4 + C::ED / 4>(acc2, ap, st, lane);
        }
        float dreg[H2];
        acc_to_regs<NOB2, RPB, true>(dreg, acc2);
        if constexpr (TRAIN) {
            if (valid) {
                tape_store_regs_part<P, 0, H2 / 4>(a.tape + a.tl.dact_off * a.tape_rows, tape_row_off<TILE>(trow0, C::W / 2, part), dreg);
                tape_store_mask<P>(a.tape + a.tl.dmask_off * a.tape_rows, tape_row<TILE>(trow0), a.tl.dmask_w, dreg, part);
            }
        }
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
    } else {
#pragma unroll
        for (int c = 0; c < C::RGB; ++c) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < H / 4; ++q) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wr + (c * P + part) * H + 4 * q);
                s = fmaf(h[4 * q + 0], w4.x, s); s = fmaf(h[4 * q + 1], w4.y, s);
                s = fmaf(h[4 * q + 2], w4.z, s); s = fmaf(h[4 * q + 3], w4.w, s);
            }
            rgbraw[c] = reduce_parts<P>(s) + wr[C::RGB * P * H + c];
        }
    }

    if (!(valid && part == 0)) return;
    float *o = outp + row * io.out_stride;
    if constexpr (C::RGB == 3) {
        o[0] = sigmoidf_(rgbraw[0]); o[1] = sigmoidf_(rgbraw[1]); o[2] = sigmoidf_(rgbraw[2]); o[3] = sigma;
    } else {
        if (io.apply_sh_deg >= 0) {
            constexpr int NB = C::RGB / 3;
            const float dx = io.dir[ray * io.dir_stride], dy = io.dir[ray * io.dir_stride + 1],
                        dz = io.dir[ray * io.dir_stride + 2];
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

template <class C, bool TRAIN>
__global__ __launch_bounds__(256, C::TILE == 16 && C::W <= 256 ? 2 : 1) void k_mlp_fwd(MlpFwdArgs a) {
    mlp_fwd_body<C, TRAIN>(a, blockIdx.x);
}

// Several independent evaluations (the foreground and the background model of one pass of a training / rendering step) in
// ONE launch: workgroups [wg0[s], wg0[s+1]) belong to segment s, which runs configuration CA or CB.  The compacted
// background rows alone fill half the chip at best; side by side with the foreground rows they only lengthen its tail.
constexpr int MLP_MAX_SEGS = 4;
struct MlpFwdMulti {
    MlpFwdArgs seg[MLP_MAX_SEGS];
    int32_t wg0[MLP_MAX_SEGS + 1];
    int32_t is_b[MLP_MAX_SEGS];
    // Feature-split tail (mlp_fwd_split.h): segment split_seg (a CB segment, the launch's last; -1: none) takes the 32-row body when its
    // device-side row count needs at most split_max such workgroups.  The count's sources are repeated here as top-level scalars: read
    // through seg[s], hipcc merges the alternatives into a select between ADDRESSES of argument fields and generic pointers, which pins
    // the whole argument block into a private copy.
    int32_t split_seg, split_max, split_rpu;
    const int32_t *split_units;        // io.n_units_dev of that segment, or NULL
    const MlpCellSeg *split_dcells;    // its cell table (one cell), or NULL
    long split_fixed;                  // the row count when neither holds a device-side count
    int32_t nseg;
};
}  // namespace mnr
#include "mlp_fwd_split.h"
namespace mnr {
// can configuration C's segments run as feature-split workgroups (mlp_fwd_split.h)?
template <class C>
constexpr bool split_capable() { return C::TILE == 16 && C::HAS_FINAL && C::NOB % 8 == 0 && C::NOB2 % 8 == 0 && C::W <= 256; }

template <class CA, class CB, bool TRAIN, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_mlp_fwd_multi(MlpFwdMulti m) {
    const int blk = blockIdx.x;
    const int s = (blk >= m.wg0[1]) + (blk >= m.wg0[2]) + (blk >= m.wg0[3]);
    if (m.is_b[s]) {
        if constexpr (split_capable<CB>() && NW == 4) {
            // The launch's LAST segment -- the background rows -- decides here, on its device-side row count, whether it is a partial quantum
            // worth splitting: at most split_max (= one per CU) half-length workgroups.  The grid covers either layout.
            if (s == m.split_seg) {
                const int32_t *nu = m.split_units;
                if (m.split_dcells) nu = *const_cast<const int32_t *const volatile *>(&m.split_dcells[0].n_units);   // (volatile: not merged with the argument read above)
                const long n = nu ? (long)*nu * m.split_rpu : m.split_fixed;
                if ((n + 31) / 32 <= m.split_max) {
                    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 7)) mlp_fwd_split_body<CB, TRAIN, 1>(m.seg[s], blk - m.wg0[s], blockIdx.y);
                    else mlp_fwd_split_body<CB, TRAIN, 0>(m.seg[s], blk - m.wg0[s], blockIdx.y);
                    return;
                }
            }
        }
        mlp_fwd_body<CB, TRAIN, NW>(m.seg[s], blk - m.wg0[s], blockIdx.y);
    } else mlp_fwd_body<CA, TRAIN, NW>(m.seg[s], blk - m.wg0[s], blockIdx.y);
}

// dynamic LDS beyond 64 KiB has to be granted per kernel function
static inline int allow_lds(const void *fn, size_t bytes) {
    if (bytes <= 65536) return MNR_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %zu): %s", bytes, hipGetErrorString(e));
    return MNR_OK;
}

// grid of a routed segment with `worst` workgroups in the worst case (xcd_contiguous: every residue of 8 needs its share)
static inline long routed_grid(long worst) { return (worst + 7) / 8 * 8 + 8; }

template <class C>
static int fill_fwd_args(MlpFwdArgs &a, const ModelLayout &m, const void *packed, const mnr_model_desc *d, const mnr_mlp_io *io,
                         float *tape, long tape_rows, long tape_row0, const mnr_mlp_cell *cells, int n_cells) {
    // the template's static structure must agree with the runtime layout the packer used
    if (m.tile != C::TILE || m.layer[0].nsteps != C::EX || m.layer[0].gpc != C::GPC || m.has_final != (int)C::HAS_FINAL ||
        m.rgb_in_regs != C::H2 || m.n_mfma_layers != C::NL + (C::HAS_FINAL ? 2 : 0))
        return set_err(MNR_E_INVALID, "internal: kernel template / layout mismatch");
    if (C::HAS_FINAL && m.layer[C::NL + 1].nsteps != C::H + C::ED + C::AP)
        return set_err(MNR_E_INVALID, "internal: dir_a layer layout mismatch");
    a.chunks = reinterpret_cast<const float4 *>(packed);
    a.aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(packed) + (size_t)m.total_chunks * CHUNK_BYTES);
    a.emb_a = d->embedding_a;
    a.cells = cells;
    a.n_cells = n_cells;
    a.xcd_order = cells && !getenv("MNR_NO_XCD_ORDER");
    a.dcells = nullptr;
    a.cell_rows = 0;
    a.aux_byte_off = (long)m.total_chunks * CHUNK_BYTES;
    a.io = *io;
    for (int i = 0; i < MAX_MFMA_LAYERS; ++i) a.bias_off[i] = i < m.n_mfma_layers ? m.layer[i].bias_off : 0;
    a.sigma_off = m.sigma_off;
    a.rgb_off = m.rgb_off;
    a.sigma_act = d->sigma_activation;
    a.app_count = d->appearance_count;
    a.tape = tape;
    if (tape && (long)tape_rows * d->layer_dim * 4 >= (1ll << 32)) return set_err(MNR_E_INVALID, "tape capacity: a plane must stay below 4 GiB (32-bit row offsets in the store addressing)");
    a.tape_rows = tape_rows;
    a.tape_row0 = tape_row0;
    a.tl = tape_layout(ArchDims{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim,
                                d->appearance_dim, d->rgb_dim, d->mfma_tile});
    return MNR_OK;
}

template <class C, bool TRAIN = false>
static int launch_fwd(const ModelLayout &m, const void *packed, const mnr_model_desc *d, const mnr_mlp_io *io,
                      hipStream_t stream, float *tape = nullptr, long tape_rows = 0, long tape_row0 = 0,
                      const mnr_mlp_cell *cells = nullptr, int n_cells = 0) {
    MlpFwdArgs a;
    const int rc = fill_fwd_args<C>(a, m, packed, d, io, tape, tape_rows, tape_row0, cells, n_cells);
    if (rc != MNR_OK) return rc;
    // cells: the worst case (every row routed to every cell); workgroups past the device-side counts exit at once
    long nwg = (io->n_rows + C::ROWS_PER_WG - 1) / C::ROWS_PER_WG * (cells ? n_cells : 1);
    if (nwg <= 0) return MNR_OK;
    if (cells) nwg = routed_grid(nwg);
    if (nwg > 0x7fffffffL) return set_err(MNR_E_INVALID, "too many rows for one MLP launch");
    constexpr size_t LDS = fwd_lds_bytes<C, 4>();
    const int lrc = allow_lds(reinterpret_cast<const void *>(k_mlp_fwd<C, TRAIN>), LDS);
    if (lrc != MNR_OK) return lrc;
    hipLaunchKernelGGL((k_mlp_fwd<C, TRAIN>), dim3((unsigned)nwg), dim3(256), LDS, stream, a);
    return check_launch("k_mlp_fwd");
}

}  // namespace mnr
