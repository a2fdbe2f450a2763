// mlp_fwd_pair.hip -- register-chained forward for layer_dim 512 (configs/mega-nerf Building, README "Larger models") with TWO wavefronts
// per SIMD: a wavefront PAIR owns 16 samples and splits the 512 output features of every layer between its halves.
//
// Why: with one wavefront owning all 512 features of its 16 samples (k_mlp_fwd<MlpCfg<.., 512, ..>>) a lane holds 128 input + 128
// accumulator registers -- one wavefront per SIMD, nothing to hide its LDS waits and chunk barriers behind (0.63 of the fp32-MFMA peak).
// Here a lane holds the 128 input registers of the FULL previous layer but only the 64 accumulators of its half of the output blocks:
// ~230 VGPRs, two wavefronts per SIMD (one 8-wavefront workgroup per CU), the shape the 256-wide kernel runs at 0.80-0.83.  By the K
// ordering of mlp_layout.h a half's accumulator registers are a contiguous half of the next layer's B-operand registers, lane for lane,
// so the exchange between the halves of a pair is a lane-wise copy through LDS: 16 KB per wavefront and layer against 2 048 MFMAs.
// Same packed image, same chunk stream (every wavefront reads its half of each group's A fragments), same aux block, same numerics as
// the one-wavefront kernel up to the order in which nothing is summed differently: a feature's K loop is the same fmaf chain.
//
// Inference only (plain launches and the routed gather mode of merged containers); training of 512-wide models stays on the tiled
// per-layer GEMMs (csrc/tgemm.hip).
#include "lds_asm.h"
#include "mlp_fwd_kernels.h"

namespace mnr {

constexpr int PAIR_THREADS = 512;
constexpr int PAIR_XBUF_F4 = 8 * 512;                                   // exchange space: 8 wavefronts x 32 registers x 64 lanes (float4 units)
constexpr size_t PAIR_LDS_BYTES = (size_t)2 * CHUNK_BYTES + (size_t)PAIR_XBUF_F4 * 16 + 4 * 16 * 4 * sizeof(float);

struct WStream8 {      // WStream (mlp_device.h) for a 512-thread workgroup
    const float4 *g;
    float4 *lds;
    int cur;
    __device__ __forceinline__ void issue() {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        float4 *dst = lds + (cur ^ 1) * CHUNK_F4 + wave * 64;
        const unsigned lane_off = threadIdx.x * 16u;
#pragma unroll
        for (int i = 0; i < CHUNK_F4 / PAIR_THREADS; ++i) {
            unsigned lo = lane_off;
            asm("" : "+v"(lo));
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(reinterpret_cast<const char *>(g + i * PAIR_THREADS)) + lo),
                                             (lds_void_t *)(dst + i * PAIR_THREADS), 16, 0, 0);
        }
        g += CHUNK_F4;
    }
    __device__ __forceinline__ void next_chunk() {
        __syncthreads();
        cur ^= 1;
        issue();
    }
};

// One K segment for one half of the output blocks: NOBH blocks starting at block ob0 of the layer's NOB_FULL.
// The A fragments are read with inline-asm ds_read_b128 and hand-counted waits (lds_asm.h), two batches of four blocks in flight:
// left to the compiler, all 16 reads of a group are hoisted in front of its MFMAs (64 registers on top of 128 inputs + 64 accumulators:
// 31-78 spilled VGPRs), and scheduling fences / group barriers either spilled more or did not finish compiling.
// the lane id, re-read where it is needed (v_mbcnt needs no input register) instead of living in a VGPR for the whole kernel
__device__ __forceinline__ int fresh_lane() {
    unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return (int)l;
}

struct NoHook { template <class G> __device__ __forceinline__ void operator()(G) const {} };

// BSTASH: the B operands (a positional encoding) are not a register array but sit in LDS, one float4 per K group and lane at
// bst + g * BST_STRIDE bytes (this wavefront's slice of the exchange buffer), and are fetched group by group: four registers live instead
// of 20-28 next to the 128 input + 64 accumulator + 32 fragment registers of the skip layer (which is where rounds 3-4 spilled).
constexpr int BST_STRIDE = 512 * 16;
template <int NOBH, int NOB_FULL, int NG, int GPC, int G0, int NB, class Hook = NoHook, bool BSTASH = false>
__device__ __forceinline__ void run_segment_half(floatx4 (&acc)[NOBH], const float (&b)[NB], WStream8 &st, int lane, int ob0, Hook hook = Hook(),
                                                 unsigned bst = 0) {
    static_assert(BSTASH || NB >= 4 * NG, "B register array too small");
    static_assert(NOBH % 8 == 0, "blocks per half: whole pairs of four-block batches");
#ifdef MNR_PAIR_SETPRIO                // comparison builds (mlp_device.h::run_segment has it by default)
    __builtin_amdgcn_s_setprio(2);
    struct PrioGuard { __device__ ~PrioGuard() { __builtin_amdgcn_s_setprio(0); } } prio_guard;
#endif
    // ONE two-deep pipeline over the segment's batches (mlp_device.h: SegSched, frag_load, frag_mfmas -- the accumulator pins behind every
    // batch keep the MFMAs above the reads that follow them).  A chunk boundary does not restart it: when batch t + 2 opens a new weight
    // chunk, the barrier is taken at the START of batch t, as soon as the reads of the old chunk (batches t, t + 1) have landed; the
    // wavefront then stands at the barrier with 32 MFMAs in hand, the DMA it releases overwrites a buffer nobody reads any more, and
    // the first reads of the new chunk go out behind batch t.  (W = 512: a chunk is ONE group = four batches -- a restart per chunk
    // left the matrix pipe idle for an LDS round trip every 64 MFMAs.)
    constexpr int NBATCH = NOBH / 4, T = NG * NBATCH;
    static_assert(GPC * NOB_FULL * 1024 <= 65536, "fragment offsets must fit the ds_read immediate");
    static_assert(GPC * NBATCH >= 2, "a chunk holds at least two batches");
    using S = SegSched<NBATCH, GPC, G0, T>;
    floatx4 a0[4], a1[4], bq = floatx4(0.f);
    if constexpr (S::chunk_start(0)) { st.next_chunk(); hook(std::integral_constant<int, G0>{}); }      // (hook: right behind a chunk barrier)
    unsigned addr = lds_addr(st.lds + st.cur * CHUNK_F4 + ob0 * 64 + fresh_lane());
    frag_load<G0 % GPC, 0, NOB_FULL>(a0, addr);
    if constexpr (T > 1) frag_load<(G0 + 1 / NBATCH) % GPC, (1 % NBATCH) * 4, NOB_FULL>(a1, addr);
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, u = t + 2, gl = t / NBATCH;
        constexpr bool early = u < T && S::chunk_start(u), group_start = BSTASH && t % NBATCH == 0;
        if constexpr (group_start) bq = lds_ld4<gl * BST_STRIDE>(bst);
        if constexpr (early) {
            wait_lgkm<0>();
            st.next_chunk();
            hook(std::integral_constant<int, G0 + u / NBATCH>{});
            addr = lds_addr(st.lds + st.cur * CHUNK_F4 + ob0 * 64 + fresh_lane());
        } else if constexpr (group_start || t + 1 >= T) {
            wait_lgkm<0>();
        } else {
            wait_lgkm<4>();
        }
        float b0, b1, b2, b3;
        if constexpr (BSTASH) { pin(bq); b0 = bq[0]; b1 = bq[1]; b2 = bq[2]; b3 = bq[3]; }
        else { b0 = b[4 * gl]; b1 = b[4 * gl + 1]; b2 = b[4 * gl + 2]; b3 = b[4 * gl + 3]; }
        if constexpr (t % 2 == 0) {
            frag_mfmas<(t % NBATCH) * 4>(acc, a0, b0, b1, b2, b3);
            if constexpr (u < T) frag_load<(G0 + u / NBATCH) % GPC, (u % NBATCH) * 4, NOB_FULL>(a0, addr);
        } else {
            frag_mfmas<(t % NBATCH) * 4>(acc, a1, b0, b1, b2, b3);
            if constexpr (u < T) frag_load<(G0 + u / NBATCH) % GPC, (u % NBATCH) * 4, NOB_FULL>(a1, addr);
        }
    });
}

// a positional encoding -> this wavefront's slice of the exchange buffer, one float4 per K group (BSTASH above)
template <int OFF>
__device__ __forceinline__ void lds_st4(unsigned addr, floatx4 v) { asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
template <int NE>
__device__ __forceinline__ void stash_encoding(unsigned bst, const float (&e)[NE]) {
    static_assert(NE % 4 == 0 && (NE / 4) * BST_STRIDE <= PAIR_XBUF_F4 * 16, "encoding groups must fit the exchange buffer");
    static_for<0, NE / 4>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        lds_st4<g * BST_STRIDE>(bst, floatx4{e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]});
    });
}

// The halves of a pair swap their NH output registers (lane for lane) through LDS in two rounds of NH / 2 registers:
// h[own0 .. own0 + NH) = own, h[oth0 .. oth0 + NH) = the partner's.
template <int NH, bool UPPER, int NHF>
__device__ __forceinline__ void pair_exchange(float (&h)[NHF], const float (&o)[NH], float4 *xb, int wave, int lane) {
    static_assert(NH % 8 == 0 && NH <= 64 && NHF >= 2 * NH, "exchange in two rounds of at most 32 registers");
    constexpr int Q = NH / 8;                 // float4 pieces per round
    const int ln = fresh_lane();
    float4 *mine = xb + wave * 512 + ln, *theirs = xb + (wave ^ 4) * 512 + ln;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = 4 * (round * Q + q);
            mine[q * 64] = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = 4 * (round * Q + q);
            const float4 v = theirs[q * 64];
            // which half a wavefront is, is a TEMPLATE parameter of the kernel body (k_mlp_fwd_pair branches once, at its top): own registers
            // move by renaming, the partner's land where they belong -- no selects.  (Rounds 4-5 selected by value on a run-time `upper`:
            // 128 v_cndmask per layer and, in the tape-writing instantiation, 50 registers of scratch in the last exchange.)
            if constexpr (UPPER) {
                h[i] = v.x; h[i + 1] = v.y; h[i + 2] = v.z; h[i + 3] = v.w;
                h[NH + i] = o[i]; h[NH + i + 1] = o[i + 1]; h[NH + i + 2] = o[i + 2]; h[NH + i + 3] = o[i + 3];
            } else {
                h[i] = o[i]; h[i + 1] = o[i + 1]; h[i + 2] = o[i + 2]; h[i + 3] = o[i + 3];
                h[NH + i] = v.x; h[NH + i + 1] = v.y; h[NH + i + 2] = v.z; h[NH + i + 3] = v.w;
            }
        }
        __syncthreads();
    }
}

// TRAIN: additionally writes the activation planes of the tape (TapeLayout act[l], fin, dact -- dense row-major [rows][width], what the
// tiled data-gradient / weight-gradient kernels of the layer-by-layer training path read: models/layerwise.py); every half stores the
// 256 (128) columns it computed.  The stores of a layer's output are issued from the NEXT layer's input registers, a quarter behind
// each of that layer's first four chunk barriers: a barrier drains vmcnt, so a store issued in front of one (and the exchange is four
// barriers) would stall the wavefront for a write round trip.
template <int Q0, int NQ, bool UPPER, int NH2>
__device__ __forceinline__ void pair_store_own(const float *plane, unsigned row_byte_off, const float (&h)[NH2]) {
    constexpr int NH = NH2 / 2;
    static_for<Q0, Q0 + NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value, i = 4 * q;
        constexpr int j = UPPER ? NH + i : i;
        gstore4<64 * q>(plane, row_byte_off, make_float4(h[j], h[j + 1], h[j + 2], h[j + 3]));
    });
}

template <class C, bool TRAIN, int HALF>
__device__ __forceinline__ void pair_body(const MlpFwdArgs &a) {
    static_assert(C::TILE == 16 && C::W == 512 && C::HAS_FINAL && C::RGB == 3, "pair kernel: the 512-wide default architectures");
    constexpr int P = C::P, H = C::H, NOB = C::NOB, NOBH = NOB / 2, HH = H / 2;          // H = 128 input registers, HH = 64 own outputs
    constexpr int NOB2 = C::NOB2, NOB2H = NOB2 / 2, H2 = C::H2, H2H = H2 / 2;             // dir_a: 256 outputs -> 64 registers, 32 own
    extern __shared__ float4 lds_ring[];
    float4 *xb = lds_ring + 2 * CHUNK_F4;
    float *rsum = reinterpret_cast<float *>(xb + PAIR_XBUF_F4);                          // [4 pairs][16 rows][4]

    const mnr_mlp_io &io = a.io;
    const float4 *chunks = a.chunks;
    const float *aux = a.aux, *emb_a = a.emb_a;
    const int32_t *row_index = io.row_index;
    float *outp = io.out;
    long n_rows, blk = blockIdx.x;
    if (a.cells) {          // routed evaluation: workgroups laid out cell after cell (see mlp_fwd_body)
        if (a.xcd_order) {
            long T = 0;
            for (int c = 0; c < a.n_cells; ++c) T += ((long)*a.cells[c].count + C::ROWS_PER_WG - 1) / C::ROWS_PER_WG;
            blk = xcd_contiguous(blk, T);
            if (blk < 0) return;
        }
        int c = 0;
        n_rows = 0;
        for (; c < a.n_cells; ++c) {
            const long n = *a.cells[c].count, t = (n + C::ROWS_PER_WG - 1) / C::ROWS_PER_WG;
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
    } else {
        n_rows = io.n_units_dev ? (long)(*io.n_units_dev) * io.rows_per_unit : (long)io.n_rows;
        if (blk * C::ROWS_PER_WG >= n_rows) return;
    }
    // Everything uniform goes to SGPRs, everything per-row is RE-DERIVED where it is used (the row from a fresh lane id, the gathered source
    // row by re-reading the index list): with 128 input + 64 accumulator + 32 fragment registers a lane has ~30 registers for all the
    // rest, and whatever stays live from here to the epilogue is parked in scratch (rounds 3-4: 30 dwords spilled at the top of the kernel).
    aux = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(aux)));
    emb_a = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(emb_a)));
    outp = reinterpret_cast<float *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(outp))));
    row_index = reinterpret_cast<const int32_t *>(uniform_ptr(reinterpret_cast<const char *>(row_index)));
    n_rows = uniform_long(n_rows);
    blk = uniform_long(blk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 3;                              // waves w and w + 4 share their 16 rows
    constexpr int half = HALF;                               // (0: wavefronts 0-3, the lower output blocks; 1: wavefronts 4-7)
    const int part = lane / 16;
    auto row_of = [&]() -> long {                            // this lane's row inside the launch (the cell's list)
        unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return (blk * 4 + pair) * 16 + (long)(l & 15u);
    };
    auto src_of = [&](long lrow_) -> long {                  // ... and the row of the input arrays it evaluates (gathered launches)
        const long rc = lrow_ < n_rows ? lrow_ : n_rows - 1;
        return row_index ? (long)row_index[rc] : rc;
    };
    const bool valid = row_of() < n_rows;
    // training: byte offset of this lane's first own column inside a 512-wide (256-wide) plane row
    // (re-derived at every store site, like the row: one VGPR while it lives instead of two for the whole kernel)
    auto off_of = [&](unsigned width) -> unsigned {
        unsigned l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        const unsigned trow = (unsigned)((blk * 4 + pair) * 16 + a.tape_row0) + (l & 15u);
        return (trow * width + 4u * (l >> 4)) * 4u + 2u * width * (unsigned)half;
    };
    constexpr bool upper = HALF != 0;

    WStream8 st;
    st.g = reinterpret_cast<const float4 *>(uniform_ptr(reinterpret_cast<const char *>(chunks)));
    st.lds = lds_ring;
    st.cur = 1;
    st.issue();

    float h[H];
    floatx4 acc[NOBH];
    const int ob0 = half * NOBH;
    // ---- trunk -------------------------------------------------------------------------------------
    static_for<0, C::NL>([&](auto lc) __attribute__((always_inline)) {
        constexpr int l = decltype(lc)::value;
        constexpr bool ENC = l == 0 || ((C::SKIP >> l) & 1);
        const int ln = fresh_lane();
        const float *bias = aux + a.bias_off[l] + (ln >> 4) * H + half * HH;
        const unsigned bst = lds_addr(xb + wave * 64 + ln);
        if constexpr (ENC) {
            // The positional encoding is evaluated where it is consumed (layer 0 and the skip layer) -- BEFORE this layer's accumulators
            // exist (sincosf wants ~40 registers of its own) -- and parked in this wavefront's slice of the exchange buffer, which is idle
            // between two layers' exchanges; the K segment then fetches it group by group (run_segment_half BSTASH).
            float x[C::XYZ];
            const long src = src_of(row_of());
#pragma unroll
            for (int d = 0; d < C::XYZ; ++d) x[d] = io.xyz[src * io.xyz_stride + d];
            float ex[C::EX];
            embed<C::XYZ, C::LX, P>(ex, x, ln >> 4);
            stash_encoding(bst, ex);
            asm volatile("" : "+v"(bias));            // the bias loads below stay below (they would hold 64 registers across the sincosf code)
        }
        init_acc<NOBH, 4>(acc, bias);
        st.next_chunk();
        // deferred tape stores of layer l - 1 (its output = this layer's input registers): pieces behind chunk barriers 0 .. 3
        auto hook = [&](auto gc) {
            if constexpr (TRAIN && l > 0) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g < 4) { if (valid) pair_store_own<4 * g, 4, upper>(a.tape + a.tl.act_off[l - 1] * a.tape_rows, off_of(512u), h); }
            }
        };
        hook(std::integral_constant<int, 0>{});
        if constexpr (ENC) {
            const float none[4] = {0.f, 0.f, 0.f, 0.f};
            run_segment_half<NOBH, NOB, C::EX / 4, C::GPC, 0, 4, decltype(hook), true>(acc, none, st, lane, ob0, hook, bst);
            if constexpr (l > 0) run_segment_half<NOBH, NOB, H / 4, C::GPC, C::EX / 4>(acc, h, st, lane, ob0, hook);
        } else {
            run_segment_half<NOBH, NOB, H / 4, C::GPC, 0>(acc, h, st, lane, ob0, hook);
        }
        float o[HH];
        acc_to_regs<NOBH, 4, true>(o, acc);
        // layer 0 ends in its encoding segment: a wavefront that is done must not start writing exchange data over the slice another one
        // still fetches its last encoding group from (in the skip layer the hidden-state segment's chunk barriers lie in between)
        if constexpr (l == 0) __syncthreads();
        pair_exchange<HH, upper>(h, o, xb, wave, lane);
    });

    // ---- sigma head (both halves hold the full activation: computed twice, written once) ------------
    float sigma;
    {
        const float *ws = aux + a.sigma_off;
        float s = 0.f;
        // (the head's weight loads stay behind the last exchange: hoisted into it, eight quads sat on top of its 64 own + 128 input
        // + 32 partner registers and the tape-writing instantiation spilled 50)
        asm volatile("" : "+v"(s)::"memory");
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            // eight weight quads in flight at most: left alone, hipcc requests all 32 up front (128 registers beside the 128 inputs)
            if (q % 8 == 0 && q > 0) asm volatile("" : "+v"(s)::"memory");
            const float4 w4 = *reinterpret_cast<const float4 *>(ws + part * H + 4 * q);
            s = fmaf(h[4 * q + 0], w4.x, s); s = fmaf(h[4 * q + 1], w4.y, s);
            s = fmaf(h[4 * q + 2], w4.z, s); s = fmaf(h[4 * q + 3], w4.w, s);
        }
        s = reduce_parts<P>(s) + ws[P * H];
        if (io.sigma_noise) s += io.sigma_noise[src_of(row_of())];
        sigma = a.sigma_act ? softplus_shifted(s) : fmaxf(s, 0.f);
    }
    if (io.sigma_only) {
        if (valid && part == 0 && half == 0) outp[row_of() * io.out_stride] = sigma;
        return;
    }

    // ---- xyz_encoding_final (no activation) -----------------------------------------------------------
    {
        init_acc<NOBH, 4>(acc, aux + a.bias_off[C::NL] + part * H + half * HH);
        st.next_chunk();
        auto hook = [&](auto gc) {
            if constexpr (TRAIN) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g < 4) { if (valid) pair_store_own<4 * g, 4, upper>(a.tape + a.tl.act_off[C::NL - 1] * a.tape_rows, off_of(512u), h); }
            }
        };
        hook(std::integral_constant<int, 0>{});
        run_segment_half<NOBH, NOB, H / 4, C::GPC, 0>(acc, h, st, lane, ob0, hook);
        float o[HH];
        acc_to_regs<NOBH, 4, false>(o, acc);
        pair_exchange<HH, upper>(h, o, xb, wave, lane);
    }

    // ---- dir_a_encoding: 256 outputs, 128 per half --------------------------------------------------------
    floatx4 acc2[NOB2H];
    init_acc<NOB2H, 4>(acc2, aux + a.bias_off[C::NL + 1] + part * H2 + half * H2H);
    st.next_chunk();
    auto hook_fin = [&](auto gc) {              // (two groups per chunk here: a barrier every other group)
        if constexpr (TRAIN) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g % 2 == 0 && g < 8) { if (valid) pair_store_own<2 * g, 4, upper>(a.tape + a.tl.fin_off * a.tape_rows, off_of(512u), h); }
        }
    };
    hook_fin(std::integral_constant<int, 0>{});
    run_segment_half<NOB2H, NOB2, H / 4, C::GPC2, 0>(acc2, h, st, lane, half * NOB2H, hook_fin);
    if constexpr (C::ED > 0) {
        float dv[3];
        const long ray_d = src_of(row_of()) / io.rows_per_ray;
#pragma unroll
        for (int d = 0; d < 3; ++d) dv[d] = io.dir[ray_d * io.dir_stride + d];
        float ed[C::ED];
        embed<3, C::LD, P>(ed, dv, part);
        run_segment_half<NOB2H, NOB2, C::ED / 4, C::GPC2, H / 4>(acc2, ed, st, lane, half * NOB2H);
    }
    if constexpr (C::AP > 0) {
        const long ray = src_of(row_of()) / io.rows_per_ray;
        long idx = io.idx_is_float ? (long)reinterpret_cast<const float *>(io.idx)[ray * io.idx_stride]
                                   : (long)reinterpret_cast<const int32_t *>(io.idx)[ray * io.idx_stride];
        idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
        const float *ea = emb_a + idx * C::APP + part * (C::APP / P);
        float ap[C::AP];
#pragma unroll
        for (int i = 0; i < C::AP; ++i) ap[i] = (i < C::APP / P) ? ea[i] : 0.f;
        run_segment_half<NOB2H, NOB2, C::AP / 4, C::GPC2, H / 4 + C::ED / 4>(acc2, ap, st, lane, half * NOB2H);
    }
    float dreg[H2H];
    acc_to_regs<NOB2H, 4, true>(dreg, acc2);
    if constexpr (TRAIN) {
        if (valid) {
            const unsigned o256 = off_of(256u);
            static_for<0, H2H / 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                gstore4<64 * q>(a.tape + a.tl.dact_off * a.tape_rows, o256, make_float4(dreg[4 * q], dreg[4 * q + 1], dreg[4 * q + 2], dreg[4 * q + 3]));
            });
        }
    }

    // ---- rgb head: each half sums its 128 features; the upper half hands its partial sums over through LDS ----
    const float *wr = aux + a.rgb_off;
    float rgbp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < H2H / 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wr + (c * P + part) * H2 + half * H2H + 4 * q);
            s = fmaf(dreg[4 * q + 0], w4.x, s); s = fmaf(dreg[4 * q + 1], w4.y, s);
            s = fmaf(dreg[4 * q + 2], w4.z, s); s = fmaf(dreg[4 * q + 3], w4.w, s);
        }
        rgbp[c] = reduce_parts<P>(s);
    }
    float *rs = rsum + (pair * 16 + (lane % 16)) * 4;
    if (half == 1 && part == 0) { rs[0] = rgbp[0]; rs[1] = rgbp[1]; rs[2] = rgbp[2]; }
    __syncthreads();
    if (!(valid && part == 0 && half == 0)) return;
    float *o = outp + row_of() * io.out_stride;
    o[0] = sigmoidf_(rgbp[0] + rs[0] + wr[3 * P * H2 + 0]);
    o[1] = sigmoidf_(rgbp[1] + rs[1] + wr[3 * P * H2 + 1]);
    o[2] = sigmoidf_(rgbp[2] + rs[2] + wr[3 * P * H2 + 2]);
    o[3] = sigma;
}

// the two halves of every pair run two specialisations of the body (the branch is uniform per wavefront; both sides execute the same
// barriers in the same order, which is all s_barrier counts)
template <class C, bool TRAIN>
__global__ __launch_bounds__(PAIR_THREADS, 1) void k_mlp_fwd_pair(MlpFwdArgs a) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) pair_body<C, TRAIN, 1>(a);
    else pair_body<C, TRAIN, 0>(a);
}

template <class C, bool TRAIN>
static int launch_fwd_pair(const ModelLayout &m, const void *packed, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t stream,
                           const mnr_mlp_cell *cells, int n_cells, float *tape, long tape_rows, long tape_row0) {
    MlpFwdArgs a;
    const int rc = fill_fwd_args<C>(a, m, packed, d, io, tape, tape_rows, tape_row0, cells, n_cells);
    if (rc != MNR_OK) return rc;
    long nwg = (io->n_rows + C::ROWS_PER_WG - 1) / C::ROWS_PER_WG * (cells ? n_cells : 1);
    if (nwg <= 0) return MNR_OK;
    if (cells) nwg = routed_grid(nwg);
    if (nwg > 0x7fffffffL) return set_err(MNR_E_INVALID, "too many rows for one MLP launch");
    static bool lds_enabled_dev[MAX_DEVICES] = {};
    bool &lds_enabled = lds_enabled_dev[device_slot()];
    if (!lds_enabled) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_fwd_pair<C, TRAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PAIR_LDS_BYTES) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_mlp_fwd_pair): %s", hipGetErrorString(hipGetLastError()));
        lds_enabled = true;
    }
    hipLaunchKernelGGL((k_mlp_fwd_pair<C, TRAIN>), dim3((unsigned)nwg), dim3(PAIR_THREADS), PAIR_LDS_BYTES, stream, a);
    return check_launch("k_mlp_fwd_pair");
}

// ONE instantiation per translation unit (each is two specialised bodies and compiles for ~6 minutes): this file is MNR_PAIR_TU 0 (foreground,
// inference, + the dispatcher); mlp_fwd_pair_bg.hip / mlp_fwd_pair_train.hip / mlp_fwd_pair_train_bg.hip include it with MNR_PAIR_TU 1 / 2 / 3.
#ifndef MNR_PAIR_TU
#define MNR_PAIR_TU 0
#endif
using PairFG = MlpCfg<3, 12, 4, 48, 512, 8, 16, 3, 16>;
using PairBG = MlpCfg<4, 12, 4, 48, 512, 8, 16, 3, 16>;
#define MNR_PAIR_LAUNCHER(name) \
    int name(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s, const mnr_mlp_cell *cells, \
             int n_cells, float *tape, long tape_rows, long tape_row0)
MNR_PAIR_LAUNCHER(launch_pair_fg_eval);
MNR_PAIR_LAUNCHER(launch_pair_bg_eval);
MNR_PAIR_LAUNCHER(launch_pair_fg_train);
MNR_PAIR_LAUNCHER(launch_pair_bg_train);
#if MNR_PAIR_TU == 0
MNR_PAIR_LAUNCHER(launch_pair_fg_eval) { return launch_fwd_pair<PairFG, false>(m, packed_dev, d, io, s, cells, n_cells, tape, tape_rows, tape_row0); }
// launch of a 512-wide default architecture through the pair kernel; MNR_E_UNSUPPORTED for anything else
int mlp_forward_pair_dispatch(const ModelLayout &m, const void *packed_dev, const mnr_model_desc *d, const mnr_mlp_io *io, hipStream_t s,
                              const mnr_mlp_cell *cells, int n_cells, float *tape, long tape_rows, long tape_row0) {
    const bool arch = d->pos_xyz_dim == 12 && d->pos_dir_dim == 4 && d->appearance_dim == 48 && d->layer_dim == 512 && d->layers == 8 &&
                      d->skip_mask == 16 && d->rgb_dim == 3 && m.tile == 16;
    if (tape && (cells || io->sigma_only)) return set_err(MNR_E_INVALID, "the tape-writing pair kernel takes plain launches");
    if (arch && d->xyz_dim == 3) return tape ? launch_pair_fg_train(m, packed_dev, d, io, s, nullptr, 0, tape, tape_rows, tape_row0)
                                             : launch_pair_fg_eval(m, packed_dev, d, io, s, cells, n_cells, nullptr, 0, 0);
    if (arch && d->xyz_dim == 4) return tape ? launch_pair_bg_train(m, packed_dev, d, io, s, nullptr, 0, tape, tape_rows, tape_row0)
                                             : launch_pair_bg_eval(m, packed_dev, d, io, s, cells, n_cells, nullptr, 0, 0);
    return set_err(MNR_E_UNSUPPORTED, "the pair kernel covers the 512-wide default fg / bg architectures");
}
#elif MNR_PAIR_TU == 1
MNR_PAIR_LAUNCHER(launch_pair_bg_eval) { return launch_fwd_pair<PairBG, false>(m, packed_dev, d, io, s, cells, n_cells, tape, tape_rows, tape_row0); }
#elif MNR_PAIR_TU == 2
MNR_PAIR_LAUNCHER(launch_pair_fg_train) { return launch_fwd_pair<PairFG, true>(m, packed_dev, d, io, s, cells, n_cells, tape, tape_rows, tape_row0); }
#else
MNR_PAIR_LAUNCHER(launch_pair_bg_train) { return launch_fwd_pair<PairBG, true>(m, packed_dev, d, io, s, cells, n_cells, tape, tape_rows, tape_row0); }
#endif
#undef MNR_PAIR_LAUNCHER

}  // namespace mnr
