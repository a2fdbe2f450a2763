// tgemm.hip -- tiled exact-fp32 GEMM for the wide layers of the layer-by-layer path (layer_dim 512 of the Building /
// Rubble configs, 2048 of configs/nerf ...: reference nerf.py:115-160 runs one nn.Linear per layer through cuBLAS).
//
//   C[m][n] = epi( sum_p sum_k A_p[m][k] * B_p(n, k) ),   epi(v) = gate( relu( v + bias[n] + r1_row[m] r1_col[n] ) )
//
// forward   : A = layer input(s) (two K phases for [embedding | hidden] and [features | direction, appearance]),
//             B_p(n, k) = W[n][koff_p + k]                       (weights as nn.Linear stores them, k contiguous)
// data grad : A = dZ of the layer, B(n, k) = W[k][col0 + n]      ("k-slow": the contraction runs along the rows of W),
//             gate = output of the previous layer (ReLU adjoint), r1 = the sigma head's rank-1 contribution
//
// Structure (same machinery as wgrad.hip):
//  * one persistent workgroup per CU, 8 waves (2 per SIMD), 256 x 256 output tile, K tiles of 32;
//    wave (wf, wr) of the 4 x 2 wave grid owns 2 feature blocks x 4 row blocks of 32 x 32 (v_mfma_f32_32x32x2_f32,
//    128 accumulator registers).  The MFMA "A" operand is the WEIGHT block, so a lane owns one output row and four
//    consecutive features per register quad: the epilogue stores (and the gate loads) are 16-byte accesses.
//  * both operand tiles of K tile t+1 are requested as one LDS-DMA burst (global_load_lds_dwordx4) at the start of K tile t
//    and waited for at the tile boundary only (s_waitcnt vmcnt(0) + s_barrier); the stream runs across output tiles, so
//    the epilogue of tile i overlaps the first fetch of tile i+1.
//  * k-contiguous operands sit in LDS as 16-byte pieces, piece (row, q) at slot row * 8 + (q ^ (row & 7)): the XOR swizzle
//    is applied on the GLOBAL side of the DMA (each lane chooses which 16 bytes it fetches), and makes the ds_read_b128
//    fragment reads (lane = row) conflict-free.  One b128 read feeds four MFMAs (k = 8 g + 4 (lane / 32) + q).
//    k-slow weights are stored row-major and read with conflict-free ds_read_b32 (lanes along the feature dimension).
//  * all LDS reads are inline asm with hand-counted lgkmcnt waits (lds_asm.h explains why).
//  * consecutive tile ids share an XCD (the n tiles of one row block reuse the activation rows from that XCD's L2).
#include <stdlib.h>

#include "lds_asm.h"

namespace mnr {

constexpr int TG_THREADS = 512;
constexpr int TG_BM = 256, TG_BN = 256, TG_KT = 32;   // (TG_BM: the taller of the two tile heights; sizes the LDS stage)
constexpr int TG_FBW = 2;                           // 32-feature blocks per wave (x RBW 32-row blocks: kernel template parameter)
constexpr int TG_OPER_BYTES = TG_BM * TG_KT * 4;    // one operand tile = 32 KB
constexpr int TG_STAGE_BYTES = 2 * TG_OPER_BYTES;

struct TgArgs {
    const float *a[2];  long lda[2];
    const float *b[2];  long ldb[2];
    int kt[2];                      // K tiles of each phase
    float *c;  long ldc;
    long M;
    int n_tiles, total_tiles;
    const float *bias;  int relu;
    const float *gate;  long ldgate;
    const float *r1_row;  long r1_stride;  const float *r1_col;
};

// BKS: k-slow weights;  BIAS / GATE / R1: which epilogue terms exist (compile-time, so the epilogue is branch-free and its
// loads are issued together)
// RBW: 32-row blocks per wave = 4 (256-row tiles) or 2 (128-row tiles: twice the tiles, for row counts whose 256-row tiling
// leaves a poorly filled last round of workgroups)
template <bool BKS, bool BIAS, bool GATE, bool R1, int RBW>
__global__ __launch_bounds__(TG_THREADS, 2) void k_tgemm(TgArgs a) {
    constexpr int BM = 64 * RBW;
    extern __shared__ float tg_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = wave & 3, wr = wave >> 2;
    const int i32 = lane & 31, kk = lane >> 5;
    const unsigned lds0 = lds_addr(tg_lds);

    int vid = blockIdx.x;
    if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int nkt = a.kt[0] + a.kt[1];
    if (vid >= a.total_tiles) return;

    // ---- fragment read addresses (stage 0) ----
    unsigned xa[4], wa[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const unsigned po = (unsigned)(((2 * g + kk) ^ (i32 & 7)) * 16);
        xa[g] = lds0 + (unsigned)((wr * RBW * 32 + i32) * 128) + po;
        wa[g] = lds0 + TG_OPER_BYTES + (unsigned)((wf * TG_FBW * 32 + i32) * 128) + po;
    }
    const unsigned wb = lds0 + TG_OPER_BYTES + (unsigned)(((4 * kk) * TG_BN + wf * TG_FBW * 32 + i32) * 4);

    // ---- DMA of one K tile of both operands into a stage ----
    // Addresses are (uniform 64-bit base, kept on the scalar unit) + (32-bit per-thread offset that never changes): the
    // loads use the SGPR-base form and the per-tile address arithmetic costs no VALU cycles next to the MFMAs.
    const int prow = tid >> 3;                                      // row of this thread's pieces inside a 64-row group
    const int ppiece = ((tid & 7) ^ (prow & 7)) * 4;                // which 4 floats of the row's 32 it fetches (swizzle)
    unsigned xvo[2], wvo[2];                                        // byte offsets of this thread inside a 64-row group, per phase
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        xvo[p] = (unsigned)((prow * a.lda[p] + ppiece) * 4);
        wvo[p] = BKS ? (unsigned)(lane * 16) : (unsigned)((prow * a.ldb[p] + ppiece) * 4);
    }
    auto issue = [&](int stage, long m0, int n0, int kti) {
        const int p = kti >= a.kt[0] ? 1 : 0;
        const int k0 = (kti - (p ? a.kt[0] : 0)) * TG_KT;
        const float *A = p ? a.a[1] : a.a[0];
        const float *B = p ? a.b[1] : a.b[0];
        const long lda = p ? a.lda[1] : a.lda[0], ldb = p ? a.ldb[1] : a.ldb[0];
        const unsigned xv = p ? xvo[1] : xvo[0], wv = p ? wvo[1] : wvo[0];
        float *dst = tg_lds + stage * (TG_STAGE_BYTES / 4) + wave * 256;        // wave-uniform; HW adds lane * 16 bytes
        if (m0 + BM <= a.M) {
            const char *ub = reinterpret_cast<const char *>(A + m0 * lda + k0);
#pragma unroll
            for (int pi = 0; pi < RBW; ++pi)
                __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(ub + (long)pi * 64 * lda * 4) + xv), (lds_void_t *)(dst + pi * 2048), 16, 0, 0);
        } else {
#pragma unroll
            for (int pi = 0; pi < RBW; ++pi) {
                const long row = min(m0 + pi * 64 + prow, a.M - 1);              // rows past M re-read the last row (never stored)
                const float *src = A + row * lda + k0 + ppiece;
                __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)(dst + pi * 2048), 16, 0, 0);
            }
        }
        float *dstb = dst + TG_OPER_BYTES / 4;
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const char *ub;
            if constexpr (BKS) ub = reinterpret_cast<const char *>(B + (long)(k0 + pi * 8 + wave) * ldb + n0);
            else ub = reinterpret_cast<const char *>(B + (long)(n0 + pi * 64) * ldb + k0);
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(ub) + wv), (lds_void_t *)(dstb + pi * 2048), 16, 0, 0);
        }
    };

    floatx16 acc[TG_FBW][RBW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int f = 0; f < TG_FBW; ++f)
#pragma unroll
            for (int r = 0; r < RBW; ++r) acc[f][r] = floatx16(0.f);
    };
    zero_acc();

    int t = vid;
    int kti = 0, s = 0;
    long m0 = (long)(t / a.n_tiles) * BM;
    int n0 = (t % a.n_tiles) * TG_BN;
    issue(0, m0, n0, 0);
    bool landed = false;                             // the wait + barrier for this K tile was taken in front of the previous tile's epilogue
    // ReLU gate of the tile, one bit per value (4 registers): its 8 x 16 bytes per lane and row block are REQUESTED during the tile's last
    // RBW K tiles -- one row block per K tile, behind that K tile's DMA -- and compressed behind the next vmcnt(0) the loop takes anyway;
    // the epilogue then starts with its stores instead of RBW dependent HBM round trips in front of them (measured at K = 512, where a
    // tile is only 16 K tiles: see DESIGN 3c)
    unsigned gm[RBW];
    float4 gv[TG_FBW][4];
    int gv_rb = -1;                                  // row block whose gate values are in flight in gv (-1: none)
    // (the rank-1 form of the tall tile is at 256 registers without the 32 in flight: it keeps the epilogue-side loads)
    constexpr bool CAN_AHEAD = GATE && !(R1 && RBW == 4);
    const bool gate_ahead = CAN_AHEAD && nkt >= RBW;
    auto gate_request = [&](int r) {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int ie = lane_e & 31, ke = lane_e >> 5;
        const long rowc = min(m0 + (wr * RBW + r) * 32 + ie, a.M - 1);
#pragma unroll
        for (int f = 0; f < TG_FBW; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gv[f][g] = *reinterpret_cast<const float4 *>(a.gate + rowc * a.ldgate + n0 + (wf * TG_FBW + f) * 32 + 8 * g + 4 * ke);
    };
    auto gate_compress = [&]() {
        unsigned m = 0u;
#pragma unroll
        for (int f = 0; f < TG_FBW; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int b0 = (f * 4 + g) * 4;
                m |= (gv[f][g].x > 0.f ? 1u : 0u) << b0 | (gv[f][g].y > 0.f ? 1u : 0u) << (b0 + 1) |
                     (gv[f][g].z > 0.f ? 1u : 0u) << (b0 + 2) | (gv[f][g].w > 0.f ? 1u : 0u) << (b0 + 3);
            }
        return m;
    };
#pragma unroll
    for (int r = 0; r < RBW; ++r) gm[r] = 0u;
    for (;;) {
        if (!landed) {
            wait_vm0();                              // K tile (t, kti) has landed in stage s ...
            __builtin_amdgcn_s_barrier();            // ... for every wave, and everyone is done reading stage s^1
        }
        landed = false;
        if constexpr (CAN_AHEAD) {
            if (gv_rb >= 0) {                        // (uniform) the gate values requested during the previous K tile have landed
                const unsigned m = gate_compress();
#pragma unroll
                for (int r = 0; r < RBW; ++r) gm[r] = r == gv_rb ? m : gm[r];
                gv_rb = -1;
            }
        }
        int tn = t, ktn = kti + 1;
        const bool last_k = ktn == nkt;
        if (last_k) { tn = t + (int)gridDim.x; ktn = 0; }
        const bool have = tn < a.total_tiles;
        const long m0n = have ? (long)(tn / a.n_tiles) * BM : m0;
        const int n0n = have ? (tn % a.n_tiles) * TG_BN : n0;
        if (have) issue(s ^ 1, m0n, n0n, ktn);
        if constexpr (CAN_AHEAD) {
            if (gate_ahead && kti >= nkt - RBW) {    // (uniform) behind the DMA: the loads of row block kti - (nkt - RBW)
                gv_rb = kti - (nkt - RBW);
                gate_request(gv_rb);
            }
        }

        const unsigned so = s ? (unsigned)TG_STAGE_BYTES : 0u;
        unsigned xs[4], ws[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { xs[g] = xa[g] + so; ws[g] = wa[g] + so; }
        const unsigned wbs = wb + so;
        floatx4 xf[2][RBW], wf4[2][TG_FBW];
        float wf1[2][4][TG_FBW];
        auto frag_read = [&](auto gc, auto bufc) {
            constexpr int g = decltype(gc)::value, buf = decltype(bufc)::value;
            static_for<0, RBW>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                xf[buf][r] = lds_ld4<r * 4096>(xs[g]);
            });
            if constexpr (BKS) {
                static_for<0, 4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    static_for<0, TG_FBW>([&](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        wf1[buf][q][f] = lds_ld<(8 * g + q) * TG_BN * 4 + f * 128>(wbs);
                    });
                });
            } else {
                static_for<0, TG_FBW>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    wf4[buf][f] = lds_ld4<f * 4096>(ws[g]);
                });
            }
        };
        constexpr int NREADS = RBW + (BKS ? 4 * TG_FBW : TG_FBW);
        frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value, cur = g & 1;
            if constexpr (g + 1 < 4) {
                frag_read(std::integral_constant<int, g + 1>{}, std::integral_constant<int, cur ^ 1>{});
                wait_lgkm<NREADS>();
            } else {
                wait_lgkm<0>();
            }
#pragma unroll
            for (int r = 0; r < RBW; ++r) pin(xf[cur][r]);
            if constexpr (BKS) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int f = 0; f < TG_FBW; ++f) pin(wf1[cur][q][f]);
            } else {
#pragma unroll
                for (int f = 0; f < TG_FBW; ++f) pin(wf4[cur][f]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int f = 0; f < TG_FBW; ++f)
#pragma unroll
                    for (int r = 0; r < RBW; ++r)
                        acc[f][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(BKS ? wf1[cur][q][f] : wf4[cur][f][q], xf[cur][r][q],
                                                                         acc[f][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);       // keep the software pipeline as written (reads one group ahead)
        });

        if (last_k) {
            // The next tile's first K tile was requested a whole K tile ago: wait for it (and take its barrier) HERE, with nothing else
            // outstanding, instead of at the top of the next iteration -- where the same s_waitcnt vmcnt(0) would also wait for the 32
            // stores per lane the epilogue is about to issue (256 KB per workgroup, every CU at the same moment: the drain of that burst
            // was exposed in front of every tile; now it hides behind the next tile's first 128 MFMAs per wavefront).
            if (have) {
                wait_vm0();
                __builtin_amdgcn_s_barrier();
                landed = true;
            }
            // ---- epilogue: lane -> output row, register quad -> 4 consecutive features ----
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));         // keeps the address arithmetic of the epilogue out of the main loop
            const int ie = lane_e & 31, ke = lane_e >> 5;
            // Pass 1: every load of the epilogue (gfx9 counts loads and stores in one vmcnt queue and orders them only among
            // themselves, so a load behind a store costs a full drain: no load may follow the first store).  The ReLU gate
            // of a row block is 8 x 16 bytes per lane; it is compressed to one bit per value (4 registers in all).
            const float lo = a.relu ? 0.f : -__builtin_inff();
            float4 colv[TG_FBW][4];                  // bias or rank-1 column factor of this lane's 2 x 16 features
            if constexpr (BIAS || R1) {
                const float *cp = BIAS ? a.bias : a.r1_col;
#pragma unroll
                for (int f = 0; f < TG_FBW; ++f)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        colv[f][g] = *reinterpret_cast<const float4 *>(cp + n0 + (wf * TG_FBW + f) * 32 + 8 * g + 4 * ke);
            }
            float r1[RBW];
            if constexpr (CAN_AHEAD) {
                if (gate_ahead) {                    // the last row block's values were requested at the top of this K tile
                    wait_vm0();
                    const unsigned m = gate_compress();
#pragma unroll
                    for (int r = 0; r < RBW; ++r) gm[r] = r == gv_rb ? m : gm[r];
                    gv_rb = -1;
                }
            }
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
                const long rowc = min(m0 + (wr * RBW + r) * 32 + ie, a.M - 1);
                r1[r] = 0.f;
                if constexpr (R1) r1[r] = a.r1_row[rowc * a.r1_stride];
                if constexpr (GATE) {
                    if (!gate_ahead) {               // (fewer K tiles than row blocks: the gate is read here, one row block at a time)
                        gate_request(r);
                        gm[r] = gate_compress();
                        asm volatile("" : "+v"(gm[r]));          // materialise the mask HERE (LLVM would sink the compares into pass 2 and
                        __builtin_amdgcn_sched_barrier(0);      //  keep all 128 loaded values alive); one row block's 8 loads in flight at a time
                    }
                }
            }
            // Pass 2: arithmetic + 16-byte stores
#pragma unroll
            for (int r = 0; r < RBW; ++r) {
                const long row = m0 + (wr * RBW + r) * 32 + ie;
                if (row < a.M) {
#pragma unroll
                    for (int f = 0; f < TG_FBW; ++f) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int col = n0 + (wf * TG_FBW + f) * 32 + 8 * g + 4 * ke;
                            float v[4] = {acc[f][r][4 * g], acc[f][r][4 * g + 1], acc[f][r][4 * g + 2], acc[f][r][4 * g + 3]};
                            const float cv[4] = {colv[f][g].x, colv[f][g].y, colv[f][g].z, colv[f][g].w};
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if constexpr (BIAS) v[c] += cv[c];
                                if constexpr (R1) v[c] = fmaf(r1[r], cv[c], v[c]);
                                v[c] = fmaxf(v[c], lo);
                                if constexpr (GATE) v[c] = (gm[r] >> ((f * 4 + g) * 4 + c)) & 1u ? v[c] : 0.f;
                            }
                            // (plain stores: the tile's 256 KB are absorbed by L2; non-temporal stores measured 0.565 -> 0.665 ms at 131072 x 512 x 512)
                            *reinterpret_cast<float4 *>(a.c + row * a.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
            zero_acc();
        }
        if (!have) break;
        t = tn; kti = ktn; m0 = m0n; n0 = n0n;
        s ^= 1;
    }
}

constexpr int TG_VARIANTS = 8;                       // epilogue form (4) x tile height (2)
static const void *tgemm_variant(int v) {
    switch (v) {
        case 0: return reinterpret_cast<const void *>(k_tgemm<false, true, false, false, 4>);
        case 1: return reinterpret_cast<const void *>(k_tgemm<true, false, false, false, 4>);
        case 2: return reinterpret_cast<const void *>(k_tgemm<true, false, true, false, 4>);
        case 3: return reinterpret_cast<const void *>(k_tgemm<true, false, true, true, 4>);
        case 4: return reinterpret_cast<const void *>(k_tgemm<false, true, false, false, 2>);
        case 5: return reinterpret_cast<const void *>(k_tgemm<true, false, false, false, 2>);
        case 6: return reinterpret_cast<const void *>(k_tgemm<true, false, true, false, 2>);
        default: return reinterpret_cast<const void *>(k_tgemm<true, false, true, true, 2>);
    }
}

}  // namespace mnr

using namespace mnr;

extern "C" int mnr_tgemm_run(const mnr_tgemm *g, void *stream) {
    MNR_REQUIRE(g && g->c && g->a[0] && g->b[0], "mnr_tgemm_run: missing operand");
    MNR_REQUIRE(g->n_phases == 1 || g->n_phases == 2, "mnr_tgemm_run: 1 or 2 K phases");
    MNR_REQUIRE(g->n > 0 && g->n % TG_BN == 0, "mnr_tgemm_run: n must be a multiple of %d", TG_BN);
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    TgArgs a{};
    for (int p = 0; p < g->n_phases; ++p) {
        MNR_REQUIRE(g->a[p] && g->b[p] && g->k[p] > 0 && g->k[p] % TG_KT == 0, "mnr_tgemm_run: phase %d: k must be a positive multiple of %d", p, TG_KT);
        MNR_REQUIRE(al16(g->a[p]) && al16(g->b[p]) && g->lda[p] % 4 == 0 && g->ldb[p] % 4 == 0, "mnr_tgemm_run: phase %d: operands must be 16-byte aligned", p);
        MNR_REQUIRE(g->lda[p] > 0 && g->ldb[p] > 0 && g->lda[p] < (1 << 22) && g->ldb[p] < (1 << 22), "mnr_tgemm_run: phase %d: pitch out of range", p);
        a.a[p] = g->a[p]; a.lda[p] = g->lda[p]; a.b[p] = g->b[p]; a.ldb[p] = g->ldb[p]; a.kt[p] = g->k[p] / TG_KT;
    }
    MNR_REQUIRE(al16(g->c) && g->ldc % 4 == 0, "mnr_tgemm_run: output must be 16-byte aligned");
    MNR_REQUIRE(!g->bias || al16(g->bias), "mnr_tgemm_run: bias must be 16-byte aligned");
    MNR_REQUIRE(!g->gate || (al16(g->gate) && g->ldgate % 4 == 0), "mnr_tgemm_run: gate must be 16-byte aligned");
    MNR_REQUIRE(!g->r1_row || (g->r1_col && al16(g->r1_col)), "mnr_tgemm_run: rank-1 addend needs an aligned column vector");
    if (g->m <= 0) return MNR_OK;
    a.c = g->c; a.ldc = g->ldc; a.M = g->m;
    a.n_tiles = g->n / TG_BN;
    a.bias = g->bias; a.relu = g->relu; a.gate = g->gate; a.ldgate = g->ldgate;
    a.r1_row = g->r1_row; a.r1_stride = g->r1_stride; a.r1_col = g->r1_col;
    static int n_cu_dev[MAX_DEVICES] = {};
    static bool lds_enabled_dev[MAX_DEVICES] = {};               // per device: function attributes and the CU count are
    const int slot = device_slot();
    int &n_cu = n_cu_dev[slot];
    bool &lds_enabled = lds_enabled_dev[slot];
    if (!lds_enabled) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipGetDeviceProperties");
        n_cu = prop.multiProcessorCount;
        for (int v = 0; v < TG_VARIANTS; ++v) {
            hipError_t e = hipFuncSetAttribute(tgemm_variant(v), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TG_STAGE_BYTES);
            if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_tgemm): %s", hipGetErrorString(e));
        }
        lds_enabled = true;
    }
    const char *ev = getenv("MNR_TGEMM_WGS");
    const long wgs = ev ? atol(ev) : n_cu;
    // Tile height: 256 rows unless that tiling leaves the last round of workgroups poorly filled and 128-row tiles (same
    // kernel, half the accumulators, ~10 % less efficient per tile: twice the weight-tile traffic per FLOP) fill it better.
    auto fill = [&](long tiles) { const double r = (double)tiles / (double)wgs; return r / (double)((tiles + wgs - 1) / wgs); };
    const long tiles256 = (g->m + 255) / 256 * (long)a.n_tiles, tiles128 = (g->m + 127) / 128 * (long)a.n_tiles;
    const char *eh = getenv("MNR_TGEMM_TILE_ROWS");
    const bool half = eh ? atoi(eh) == 128 : 0.9 * fill(tiles128) > fill(tiles256);
    const long total = half ? tiles128 : tiles256;
    MNR_REQUIRE(total < (1l << 30), "mnr_tgemm_run: too many output tiles");
    a.total_tiles = (int)total;
    long grid = wgs;
    if (grid > a.total_tiles) grid = a.total_tiles;
    if (grid < 1) grid = 1;
    hipStream_t s = as_stream(stream);
    // forward: bias (+ ReLU); data gradient: plain, gated, gated + rank-1 addend
    int variant;
    if (!g->b_kslow) {
        MNR_REQUIRE(g->bias && !g->gate && !g->r1_row, "mnr_tgemm_run: the forward form takes a bias and neither gate nor rank-1 addend");
        variant = 0;
    } else {
        MNR_REQUIRE(!g->bias && !g->relu && (g->gate || !g->r1_row), "mnr_tgemm_run: the k-slow form takes no bias / ReLU; a rank-1 addend needs a gate");
        variant = g->gate ? (g->r1_row ? 3 : 2) : 1;
    }
    if (half) variant += 4;
    void *params[] = {&a};
    if (hipLaunchKernel(tgemm_variant(variant), dim3((unsigned)grid), dim3(TG_THREADS), params, 2 * TG_STAGE_BYTES, s) != hipSuccess)
        return set_err(MNR_E_LAUNCH, "hipLaunchKernel(k_tgemm)");
    return check_launch("k_tgemm");
}
