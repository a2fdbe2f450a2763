// wgrad.hip -- weight gradients of the fused NeRF MLPs for gfx950 (training; the reference gets these from torch
// autograd over nerf.py:115-160):  dW_l += dZ_l^T . IN_l,  db_l += colsum(dZ_l)  for EVERY layer of EVERY model of a
// training step (foreground + background) in ONE launch, followed by one small reduction launch.
//
// Structure:
//  * operands are the dense row-major planes of the activation tape (IN_l) and of the gradient tape (dZ_l); a tile is
//    32 consecutive rows of both planes = two contiguous byte ranges, streamed global -> LDS by LDS-DMA
//    (global_load_lds_dwordx4) into a 2-stage ring.  The next tile's DMA is issued piecewise between the MFMAs of the
//    current tile and only waited for at the tile boundary (s_waitcnt vmcnt(0) + raw s_barrier): hipcc would otherwise
//    put a vmcnt(0) in front of the first ds_read behind every LDS-DMA, i.e. no overlap at all (round-1 kernel: 23 % of the
//    wave cycles parked).  That is why every LDS read below is inline asm with hand-counted lgkmcnt waits.
//  * 8 waves per workgroup (2 per SIMD) own a GM x GN x KS decomposition of the (M x N) product per job shape;
//    v_mfma_f32_32x32x2_f32, fragments by conflict-free ds_read_b32 (lanes run along the feature dimension), the reads
//    of k-pair p+1 in flight during the MFMAs of k-pair p.
//  * scheduling: every job is cut into items of `tiles_per_item` tiles (about equal cost); every job has its own item
//    counter.  A workgroup starts on its "home" job (proportional share), pulls items of that job until it is exhausted
//    (the next item is pulled while the current one computes), then steals from the job with the most items left.  One
//    (workgroup, job) episode = ONE accumulator flush, written with plain coalesced stores to a private slab; the reduce
//    kernel sums the slabs of a job into the nn.Parameter gradients.  No fp32 atomics on the gradients (round 1: 86 M
//    per launch), run-to-run differences only from the summation order of ~20 partials.
//  * rows: a job covers up to two row ranges of its tapes (coarse + fine rows of a branch; device-side counts for the
//    compacted background).  Contract with the data-gradient kernel: rows are padded to multiples of 32 inside the
//    allocated capacity, padding rows carry dZ = 0 and finite activations, so tiles are never ragged.
#include <stdlib.h>

#include <initializer_list>

#include "lds_asm.h"
#include "step_internal.h"

#ifndef MNR_WGRAD_LOAD_AUX
#define MNR_WGRAD_LOAD_AUX 2          // cache policy of the operand stream: 2 = non-temporal (the tapes are read once; fp32 launch unchanged at 1.933 ms,
                                      // split-precision launch -- HBM-bound -- 0.871 -> 0.860 ms; 0 = default policy, 1 = sc0, 16 = sc1: comparison builds)
#endif
namespace mnr {

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m);

constexpr int W2_THREADS = 512;
constexpr int W2_KT = 32;                       // rows per tile
constexpr int W2_MAX_JOBS = 24;                 // 2 models x 10 jobs (+ slack)
constexpr int W2_MAX_REGIONS = 2;
constexpr int W2_MAX_EPISODES = 768;            // slab slots; past that a flush falls back to atomics
constexpr int W2_EP_FLOATS = 256 * 384 + 512;   // largest job tile (KS x M x NP: skip layer, 12 blocks wide) + bias partials (KS x M)
constexpr int W2_CTRL_FLOATS = 256;             // LDS control area (mailboxes, per-job item counts)

// ---- job shapes ------------------------------------------------------------------------------------------------------
// A job is  dW[M][sum N_s] += dZ[rows][M]^T . [IN_0 | IN_1 | IN_2][rows][..]  with up to three input planes ("segments":
// the skip layer reads [embedding | hidden], dir_a reads [final features | direction embedding | appearance row]), so a
// layer is ONE job and its dZ plane is streamed once.  LD_s = row pitch of segment s, NB_s = its 32-column blocks.
// (GM x GN x KS) = wave grid (8 waves): GM splits the M row blocks, GN the N blocks (single-segment shapes only), KS > 1
// lets the wave groups take alternate k-pairs of a tile (M = 128: only four row blocks).  MBW x NBW = blocks per wave.
template <int M_, int GM_, int GN_, int KS_, int MBW_, int LD0_, int NB0_, int LD1_ = 0, int NB1_ = 0, int LD2_ = 0, int NB2_ = 0>
struct WShape {
    static constexpr int M = M_, GM = GM_, GN = GN_, KS = KS_, MBW = MBW_;
    static constexpr int NSEG = LD2_ ? 3 : (LD1_ ? 2 : 1);
    // the 256 x 256 shape takes its two operands with run-time row pitches (column windows of wider matrices)
    static constexpr bool ZPITCH = M_ == 256;                               // dz: run-time pitch (WJob::ldz)
    static constexpr bool PITCHED = M_ == 256 && LD0_ == 256 && LD1_ == 0;   // ... and in[0] too (WJob::ldin0)
    static constexpr int LD[3] = {LD0_, LD1_, LD2_};
    static constexpr int NB[3] = {NB0_, NB1_, NB2_};
    static constexpr int NBT = NB0_ + NB1_ + NB2_;              // N blocks in total
    static constexpr int NBW = NBT / GN_;
    static constexpr int NP = NBT * 32;                         // padded output width held in registers
    static constexpr int LDSUM = LD0_ + LD1_ + LD2_;
    static constexpr int TILE_F4 = W2_KT * (M_ + LDSUM) / 4;    // float4 per tile (dz rows, then the rows of every segment)
    static constexpr int PIECES = (TILE_F4 + W2_THREADS - 1) / W2_THREADS;
    static constexpr int STAGE_FLOATS = (TILE_F4 * 4 + 255) / 256 * 256 + 256;   // + slack: B fragments may read past LD
    static constexpr int SEG_OFF[3] = {W2_KT * M_, W2_KT * (M_ + LD0_), W2_KT * (M_ + LD0_ + LD1_)};   // floats inside a stage
    static_assert(GM_ * GN_ * KS_ == 8, "8 waves");
    static_assert(GM_ * MBW_ * 32 == M_, "M decomposition");
    static_assert(NBT % GN_ == 0 && (GN_ == 1 || NSEG == 1), "N decomposition");
    static_assert(GN_ >= MBW_, "bias sums: one row block per wave column");
    static_assert(NB0_ * 32 >= LD0_ - 3 && NB1_ * 32 >= LD1_ - 3 && NB2_ * 32 >= LD2_ - 3, "segment blocks");
    static_assert(KS_ * M_ * NP + KS_ * M_ <= W2_EP_FLOATS, "slab slot too small");
    static_assert((W2_CTRL_FLOATS + 2 * STAGE_FLOATS) * 4 <= 160 * 1024, "LDS");
    // segment / local block of N block nb (compile-time)
    static constexpr int seg_of(int nb) { return nb < NB0_ ? 0 : (nb < NB0_ + NB1_ ? 1 : 2); }
    static constexpr int nb0_of(int s) { return s == 0 ? 0 : (s == 1 ? NB0_ : NB0_ + NB1_); }
};
using WS_BIG = WShape<256, 2, 4, 1, 4, 256, 8>;                       // hidden layers, xyz_encoding_final
using WS_L0F = WShape<256, 8, 1, 1, 1, 76, 3>;                        // layer 0, foreground (75 embedding columns)
using WS_L0B = WShape<256, 8, 1, 1, 1, 100, 4>;                       // layer 0, background (100)
using WS_SKF = WShape<256, 8, 1, 1, 1, 76, 3, 256, 8>;                // skip layer [embedding | hidden], foreground (the background's is two jobs:
                                                                      //  12 blocks per wave = 192 accumulator registers would spill)
using WS_DIR = WShape<128, 4, 1, 2, 1, 256, 8, 28, 1, 48, 2>;         // dir_a [final | direction embedding (27) | appearance (48)]
using WS_DIR_NOAPP = WShape<128, 4, 1, 2, 1, 256, 8, 28, 1>;          // appearance_dim 0
using WS_DIR_NODIR = WShape<128, 4, 1, 2, 1, 256, 8, 48, 2>;          // spherical harmonics: no direction input
// dense zero-padded inputs of the layer-by-layer path (mnr_wgrad_jobs): embeddings / direction + appearance columns
using WS_D32 = WShape<256, 8, 1, 1, 1, 32, 1>;
using WS_D64 = WShape<256, 8, 1, 1, 1, 64, 2>;
using WS_D96 = WShape<256, 8, 1, 1, 1, 96, 3>;
using WS_D128 = WShape<256, 8, 1, 1, 1, 128, 4>;
enum WShapeId : int32_t { WSI_BIG = 0, WSI_L0F, WSI_L0B, WSI_SKF, WSI_DIR, WSI_DIR_NOAPP, WSI_DIR_NODIR, WSI_D32, WSI_D64, WSI_D96,
                          WSI_D128, WSI_COUNT };

struct WShapeInfo { int M, NP, KS, nseg, ld[3], nb[3], blocks_per_wave, tile_bytes; };
template <class S>
__host__ __device__ inline WShapeInfo wshape_info_of() {
    return {S::M, S::NP, S::KS, S::NSEG, {S::LD[0], S::LD[1], S::LD[2]}, {S::NB[0], S::NB[1], S::NB[2]}, S::MBW * S::NBW,
            S::TILE_F4 * 16};
}
__host__ __device__ inline WShapeInfo wshape_info(int id) {
    switch (id) {
        case WSI_BIG: return wshape_info_of<WS_BIG>();
        case WSI_L0F: return wshape_info_of<WS_L0F>();
        case WSI_L0B: return wshape_info_of<WS_L0B>();
        case WSI_SKF: return wshape_info_of<WS_SKF>();
        case WSI_DIR: return wshape_info_of<WS_DIR>();
        case WSI_DIR_NOAPP: return wshape_info_of<WS_DIR_NOAPP>();
        case WSI_DIR_NODIR: return wshape_info_of<WS_DIR_NODIR>();
        case WSI_D32: return wshape_info_of<WS_D32>();
        case WSI_D64: return wshape_info_of<WS_D64>();
        case WSI_D96: return wshape_info_of<WS_D96>();
        default: return wshape_info_of<WS_D128>();
    }
}

struct WRegion {                    // the row ranges of one model's tapes
    int64_t row0[2];
    int64_t n_rows[2];              // host-side bound
    const int32_t *n_units[2];      // optional device-side count (rows = *n_units * rows_per_unit)
    int32_t rows_per_unit[2];
    int32_t n_ranges, pad;
};
struct WJob {
    const float *dz, *in[3];        // plane bases (tape row 0)
    float *dw, *db;                 // gradient view dw[m * ldw + col0[s] + n] (n < N[s]) per segment, bias gradient or NULL
    int32_t ldw;
    int32_t ldz, ldin0;             // row pitches of dz (M = 256 shapes) and of in[0] (PITCHED shape); everything else is dense
    int16_t col0[3], N[3];
    int16_t shape, region;
    int32_t tiles_per_item;
    int32_t red_block0;             // first block of this job in the reduce launch
    const int32_t *zexp;            // split-precision launches: exponent word of the dz plane (mlp_device.h ZEXP_*), else NULL
};
struct WArgs {
    WJob job[W2_MAX_JOBS];
    WRegion region[W2_MAX_REGIONS];
    int32_t njobs;
    int32_t *counters;              // [W2_MAX_JOBS] per-job item queue heads, then [1] episode counter (all zeroed per launch)
    int32_t *ep_job;                // [W2_MAX_EPISODES]
    float *slab;                    // [W2_MAX_EPISODES][W2_EP_FLOATS]
    long long *prof;                // optional [grid][8] cycle accumulators (MNR_WGRAD_PROF diagnostics), else NULL
    int32_t max_episodes;           // slab slots in use (<= W2_MAX_EPISODES; tests lower it to exercise the atomic fallback)
};

// rows of a range (device-side count when given), padded to whole tiles
__device__ __forceinline__ int range_tiles(const WRegion &r, int i) {
    const long rows = r.n_units[i] ? (long)(*r.n_units[i]) * r.rows_per_unit[i] : (long)r.n_rows[i];
    return (int)((rows + W2_KT - 1) / W2_KT);
}

struct WCtl {                        // LDS control area layout (word offsets)
    static constexpr int MAILBOX = 0;        // 2 alternating slots
    static constexpr int ITEMS = 4;          // [W2_MAX_JOBS] items per job
    static constexpr int TILES0 = 36;        // [W2_MAX_JOBS] tiles in range 0
    static constexpr int TILES = 68;         // [W2_MAX_JOBS] tiles in total
    static constexpr int PICK = 100;         // steal-scan result
};

// ---- one (workgroup, job) episode ------------------------------------------------------------------------------------
// Processes `item` and every further item of job j this workgroup manages to pull; flushes once.
typedef _Float16 half8w __attribute__((ext_vector_type(8)));
typedef unsigned uint4w __attribute__((ext_vector_type(4)));
// x * scale = hi + lo as packed f16 (csrc/h2_device.h h2_split8: round-towards-zero pack, remainder by one mixed-precision FMA)
__device__ __forceinline__ void w2_split8(const float (&x)[8], float scale, uint4w &hi, uint4w &lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(x[2 * q] * scale, x[2 * q + 1] * scale);
        const float l0 = __builtin_fmaf(x[2 * q], scale, -(float)h[0]);
        const float l1 = __builtin_fmaf(x[2 * q + 1], scale, -(float)h[1]);
        hi[q] = __builtin_bit_cast(unsigned, h);
        lo[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    }
}
__device__ __forceinline__ floatx16 w2_mfma(uint4w a, uint4w b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8w, a), __builtin_bit_cast(half8w, b), c, 0, 0, 0);
}

// H2 = split-precision form (opt-in): the same tiles, schedule and flush; a tile's 32 rows are two K-steps of
// v_mfma_f32_32x32x16_f16, both operands split into f16 (hi, lo) halves on the fly -- three products per block, fp32 accumulate --,
// dZ pre-scaled by the plane's power of two (WJob::zexp; undone at the flush).  The kernel is then bound by the tape stream (HBM).
template <class S, bool H2 = false>
__device__ __forceinline__ void wgrad_episode(const WArgs &a, int j, int item, float *lds_all) {
    constexpr int M = S::M, MBW = S::MBW, NBW = S::NBW, KS = S::KS;
    const WJob &J = a.job[j];
    const WRegion &R = a.region[J.region];
    // the thread index is made opaque once per episode: everything derived from it (fragment / flush addresses) is then formed here,
    // inside the episode loop.  Left visible, those expressions are loop-invariant, get hoisted to the top of the kernel and -- with
    // 128 accumulators + the fragment registers live in the episodes -- are parked in scratch until each flush (11-16 spilled VGPRs).
    unsigned tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int ks = wave / (S::GM * S::GN), wq = wave % (S::GM * S::GN);
    const int wr = wq / S::GN, wc = wq % S::GN;
    const int i32 = lane & 31, kk = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned ctl = lds_addr(lds_all);
    float *stage0 = lds_all + W2_CTRL_FLOATS;
    const int n_items = lds_ld_u(ctl + 4 * (WCtl::ITEMS + j));
    const int tiles0 = lds_ld_u(ctl + 4 * (WCtl::TILES0 + j));
    const int tiles_all = lds_ld_u(ctl + 4 * (WCtl::TILES + j));
    const int tpi = J.tiles_per_item;
    float zscale = 1.f, zdescale = 1.f, one = 1.f;
    if constexpr (H2) {
        const int zb = J.zexp ? __builtin_amdgcn_readfirstlane(*J.zexp) : 0;
        if (zb > 0) { zscale = ldexpf(1.f, ZEXP_TARGET + ZEXP_BIAS - zb); zdescale = ldexpf(1.f, zb - ZEXP_BIAS - ZEXP_TARGET); }
        asm volatile("" : "+s"(one));
    }

    floatx16 acc[MBW][NBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[m][n] = floatx16(0.f);
    float bsum = 0.f;
    const int bm = wc % MBW;                   // fragment m of this wave = row block (m + bm) % MBW of its wave row

    // per-lane fragment read addresses of the two stages
    // wave group ks of a K-split shape takes the k-pairs  kp * KS + ks  (rows 2 (kp KS + ks) + kk)
    unsigned a_off[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m) a_off[m] = (unsigned)(((2 * ks + kk) * M + (wr * MBW + (m + bm) % MBW) * 32 + i32) * 4);
    // B fragments: N block n of this wave is block wc * NBW + n; with several segments (GN == 1) block -> segment is static
    unsigned b_off[S::NSEG];
#pragma unroll
    for (int sg = 0; sg < S::NSEG; ++sg)
        b_off[sg] = (unsigned)((S::SEG_OFF[sg] + (2 * ks + kk) * S::LD[sg] + (S::GN > 1 ? wc * NBW * 32 : 0) + i32) * 4);
    const unsigned st_base0 = lds_addr(stage0), st_base1 = lds_addr(stage0 + S::STAGE_FLOATS);

    // pitched 256-wide operands: float offset of this wave's row inside a tile, per piece (uniform, once per episode)
    unsigned zrow_off[4], irow_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        zrow_off[p] = S::ZPITCH ? (unsigned)((p * 8 + wave_u) * J.ldz) : 0u;
        irow_off[p] = S::PITCHED ? (unsigned)((p * 8 + wave_u) * J.ldin0) : 0u;
    }
    // A tile in LDS = [dz rows | segment 0 rows | segment 1 rows | ...], each a contiguous byte range of its plane.
    struct Src { const float *z, *i[3]; };
    auto tile_src = [&](int t, Src &q) {       // tile index in the job's concatenated row space
        const long row = t < tiles0 ? R.row0[0] + (long)t * W2_KT : R.row0[1] + (long)(t - tiles0) * W2_KT;
        q.z = J.dz + row * (S::ZPITCH ? (long)J.ldz : (long)M);
        if constexpr (S::PITCHED) q.i[0] = J.in[0] + row * J.ldin0;
        else {
#pragma unroll
            for (int sg = 0; sg < S::NSEG; ++sg) q.i[sg] = J.in[sg] + row * S::LD[sg];
        }
    };
    auto dma_piece = [&](auto pc, const Src &q, int stage) {
        constexpr int p = decltype(pc)::value;
        constexpr int F0 = W2_KT * M / 4, F1 = S::SEG_OFF[1] / 4, F2 = S::SEG_OFF[2] / 4;     // float4 boundaries of the segments
        constexpr int lo = p * W2_THREADS, hi = lo + W2_THREADS;
        const int t = lo + (int)tid;
        if (hi <= S::TILE_F4 || t < S::TILE_F4) {
            const float *src;
            // 256-wide operands with a run-time pitch: 64 float4 per row, so piece p = rows 8 p .. 8 p + 7 (one per wave)
            if constexpr (S::ZPITCH && hi <= F0) src = q.z + zrow_off[p] + lane * 4;
            else if constexpr (S::PITCHED) src = q.i[0] + irow_off[p - F0 / W2_THREADS] + lane * 4;
            else if constexpr (hi <= F0) src = q.z + t * 4;
            else if constexpr (lo >= F0 && (S::NSEG == 1 || hi <= F1)) src = q.i[0] + (t - F0) * 4;
            else if constexpr (S::NSEG >= 2 && lo >= F1 && (S::NSEG == 2 || hi <= F2)) src = q.i[1] + (t - F1) * 4;
            else if constexpr (S::NSEG == 3 && lo >= F2) src = q.i[2] + (t - F2) * 4;
            else {                                     // the piece straddles a boundary: per-lane choice
                src = t < F0 ? q.z + t * 4 : q.i[0] + (t - F0) * 4;
                if constexpr (S::NSEG >= 2) src = t >= F1 ? q.i[1] + (t - F1) * 4 : src;
                if constexpr (S::NSEG == 3) src = t >= F2 ? q.i[2] + (t - F2) * 4 : src;
            }
            float *dst = stage0 + stage * S::STAGE_FLOATS + (lo + wave * 64) * 4;   // wave-uniform; HW adds lane*16
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)dst, 16, 0, MNR_WGRAD_LOAD_AUX);
        }
    };

    // item bookkeeping (uniform across the workgroup).  A pull for the NEXT item is started with every item; its value
    // lands (through the mailbox) at the first tile boundary of the current item, i.e. before it can be needed.
    int t = item * tpi, t_end = min(t + tpi, tiles_all);
    int pend = 0, slot = 0, next_item = 0, pulled = 0;
    auto start_pull = [&]() {
        if (tid == 0) pulled = atomic_inc_async(a.counters + j);
        pend = 1;
    };
    start_pull();
    Src q;
    tile_src(t, q);
    static_for<0, S::PIECES>([&](auto pc) { dma_piece(pc, q, 0); });
    int s = 0;
    long long pc_vm = 0, pc_bar = 0, pc_cmp = 0, pc_tiles = 0;
    const long long pc_t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    for (;;) {
        // ---- tile boundary: tile t has landed in stage s; everyone is done reading stage s^1 ----
        const long long c0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
        wait_vm0();
        const long long c1 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
        if (pend && tid == 0) lds_st_i(ctl + 4 * (WCtl::MAILBOX + slot), pulled);
        __builtin_amdgcn_s_barrier();
        const long long c2 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
        if (pend) { next_item = lds_ld_u(ctl + 4 * (WCtl::MAILBOX + slot)); slot ^= 1; pend = 0; }
        int t_next = t + 1;
        bool have_next = true;
        if (t_next == t_end) {
            if (next_item < n_items) { t_next = next_item * tpi; t_end = min(t_next + tpi, tiles_all); start_pull(); }
            else have_next = false;
        }
        tile_src(have_next ? t_next : t, q);     // no successor: re-fetch this tile into the idle stage (keeps the loop branch-free)
        // ---- compute tile t from stage s; the DMA of tile t_next into stage s^1 is issued between the MFMAs ----
        const unsigned sb = s ? st_base1 : st_base0;
        unsigned ab[MBW];
#pragma unroll
        for (int m = 0; m < MBW; ++m) ab[m] = sb + a_off[m];
        unsigned bb[S::NSEG];
#pragma unroll
        for (int sg = 0; sg < S::NSEG; ++sg) bb[sg] = sb + b_off[sg];
        if constexpr (H2) {
            // lane (i32, kk) of a fragment: feature i32 of its block, tile rows 16 (q KS + ks) + 8 kk + j, j = 0..7 (consecutive k)
            constexpr int NQ = W2_KT / 16 / KS;
            unsigned ab2[MBW], bb2[S::NSEG];
#pragma unroll
            for (int m = 0; m < MBW; ++m) ab2[m] = ab[m] + (unsigned)(((14 * ks + 7 * kk) * M) * 4);          // (2 ks + kk) -> (16 ks + 8 kk)
#pragma unroll
            for (int sg = 0; sg < S::NSEG; ++sg) bb2[sg] = bb[sg] + (unsigned)(((14 * ks + 7 * kk) * S::LD[sg]) * 4);
            static_for<0, S::PIECES>([&](auto pc) { dma_piece(pc, q, s ^ 1); });
            static_for<0, NQ>([&](auto qc) {
                constexpr int kq = decltype(qc)::value;
                // dZ fragments of the K-step: all MBW row blocks, split once (they meet every input block).  Every fragment is read
                // and waited for in one statement (lds_asm.h lds_ld8_wait); the co-resident wavefront covers the LDS latency, and
                // the kernel has HBM time to spare.
                uint4w ah[MBW], al[MBW];
                static_for<0, MBW>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    float ar[8];
                    lds_ld8_wait<(16 * kq * KS * M) * 4, M * 4>(ab2[m], ar);
                    if constexpr (m == 0) {
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) bsum += ar[jj];           // bias gradient: fp32, unscaled (fragment 0 only, see below)
                    }
                    w2_split8(ar, zscale, ah[m], al[m]);
                });
                static_for<0, NBW>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int sg = S::GN > 1 ? 0 : S::seg_of(n), nl = S::GN > 1 ? n : n - S::nb0_of(sg);
                    float br[8];
                    lds_ld8_wait<(16 * kq * KS * S::LD[sg] + nl * 32) * 4, S::LD[sg] * 4>(bb2[sg], br);
                    uint4w bh, bl;
                    w2_split8(br, one, bh, bl);
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        acc[m][n] = w2_mfma(ah[m], bh, acc[m][n]);
                        acc[m][n] = w2_mfma(al[m], bh, acc[m][n]);
                        acc[m][n] = w2_mfma(ah[m], bl, acc[m][n]);
                    }
                });
            });
        } else {
            float af[2][MBW], bf[2][NBW];
            constexpr int KP = W2_KT / 2 / KS;      // k-pairs per wave per tile (the KS wave groups interleave k-pairs)
            auto frag_read = [&](auto kpc, auto bufc) {
                constexpr int kp = decltype(kpc)::value, buf = decltype(bufc)::value;
                static_for<0, MBW>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    af[buf][m] = lds_ld<((2 * kp * KS) * M) * 4>(ab[m]);
                });
                static_for<0, NBW>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    constexpr int sg = S::GN > 1 ? 0 : S::seg_of(n), nl = S::GN > 1 ? n : n - S::nb0_of(sg);
                    bf[buf][n] = lds_ld<((2 * kp * KS) * S::LD[sg] + nl * 32) * 4>(bb[sg]);
                });
            };
            // the whole next tile is requested up front (measured on the structural microbenchmark tools/micro/mfma_probe.hip:
            // 147 vs 137 TFLOP/s for one piece per k-pair), so it has the full tile time to land
            static_for<0, S::PIECES>([&](auto pc) { dma_piece(pc, q, s ^ 1); });
            frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<0, KP>([&](auto kpc) {
                constexpr int kp = decltype(kpc)::value;
                constexpr int cur = kp & 1;
                if constexpr (kp + 1 < KP) {
                    frag_read(std::integral_constant<int, kp + 1>{}, std::integral_constant<int, cur ^ 1>{});
                    wait_lgkm<MBW + NBW>();
                } else {
                    wait_lgkm<0>();
                }
    #pragma unroll
                for (int m = 0; m < MBW; ++m) pin(af[cur][m]);
    #pragma unroll
                for (int n = 0; n < NBW; ++n) pin(bf[cur][n]);
                // bias gradient = column sums of dZ: VALU adds between MFMAs cost MFMA issue slots (measured: one add per A
                // fragment = -10 % on the 256 x 256 shape), so every wave sums only ONE of its MBW row blocks -- its fragment 0,
                // which is row block (wc % MBW) of the wave row thanks to the rotated block order (the GN >= MBW wave columns
                // of a wave row cover all blocks between them)
    #pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    if (m == 0) bsum += af[cur][0];
    #pragma unroll
                    for (int n = 0; n < NBW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m], bf[cur][n], acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the software pipeline as written (reads one k-pair ahead)
            });
        }
        if (a.prof) { const long long c3 = __builtin_amdgcn_s_memtime(); pc_vm += c1 - c0; pc_bar += c2 - c1; pc_cmp += c3 - c2; pc_tiles += 1; }
        if (!have_next) break;
        t = t_next;
        s ^= 1;
    }
    wait_vm0();                                   // the idle-stage fetch of the last tile must not outlive the episode

    // ---- flush: one slab slot per episode (plain coalesced stores); atomics only when the slots ran out ----
    // (lane coordinates re-derived behind an opaque asm: otherwise hipcc hoists the ~130 store offsets of every shape's
    //  flush to the kernel prologue and spills them)
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const int i32f = lane_f & 31, kkf = lane_f >> 5;
    int ep = 0;
    if (threadIdx.x == 0) ep = atomicAdd(a.counters + W2_MAX_JOBS, 1);
    __builtin_amdgcn_s_barrier();                 // everyone is past its last mailbox read
    if (threadIdx.x == 0) lds_st_i(ctl + 4 * (WCtl::MAILBOX + 2), ep);
    __builtin_amdgcn_s_barrier();
    ep = lds_ld_u(ctl + 4 * (WCtl::MAILBOX + 2));
    if (ep < a.max_episodes) {
        float *base = a.slab + (long)ep * W2_EP_FLOATS;
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const int col = (wc * NBW + n) * 32 + i32f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = ks * M + (wr * MBW + (m + bm) % MBW) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kkf;
                    base[row * S::NP + col] = H2 ? acc[m][n][r] * zdescale : acc[m][n][r];
                }
            }
        }
        {
            const float sm = bsum + __shfl_xor(bsum, 32);
            if (wc < MBW && kkf == 0) base[KS * M * S::NP + ks * M + (wr * MBW + bm) * 32 + i32f] = sm;
        }
        if (threadIdx.x == 0) a.ep_job[ep] = j;
        if (a.prof && threadIdx.x == 0) {
            const long long c4 = __builtin_amdgcn_s_memtime();
            long long *pr = a.prof + (long)blockIdx.x * 8;
            pr[0] += pc_vm; pr[1] += pc_bar; pr[2] += pc_cmp; pr[3] += pc_tiles; pr[4] += c4 - pc_t0; pr[5] += 1;
        }
    } else {
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const int nb = wc * NBW + n;
                const int sg = nb < S::NB[0] ? 0 : (nb < S::NB[0] + S::NB[1] ? 1 : 2);
                const int col = (nb - S::nb0_of(sg)) * 32 + i32f;
                if (col < J.N[sg]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wr * MBW + (m + bm) % MBW) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kkf;
                        atomicAdd(J.dw + (long)row * J.ldw + J.col0[sg] + col, H2 ? acc[m][n][r] * zdescale : acc[m][n][r]);
                    }
                }
            }
        }
        const float sm = bsum + __shfl_xor(bsum, 32);
        if (J.db && wc < MBW && kkf == 0) atomicAdd(J.db + (wr * MBW + bm) * 32 + i32f, sm);
    }
}

// broadcast a value of thread 0 to the workgroup through an LDS word (no LDS-DMA may be in flight)
__device__ __forceinline__ int wg_broadcast(unsigned word_addr, int v) {
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) lds_st_i(word_addr, v);
    __builtin_amdgcn_s_barrier();
    return lds_ld_u(word_addr);
}

// FAMILY 0: the shapes of the fused models' tapes (mnr_mlp_backward_weights_multi); FAMILY 1: the job form of the layer-by-layer
// path (mnr_wgrad_jobs).  Two kernels rather than one: with all eleven episode instantiations in one kernel the 256 x 256
// shape ran 4 % slower (2.02 vs 1.94 ms on the benchmark step, same instruction mix -- code placement).
template <int FAMILY, bool H2 = false>
__global__ __launch_bounds__(W2_THREADS, 2) void k_wgrad2(WArgs a) {
    extern __shared__ float lds_all[];
    const unsigned ctl = lds_addr(lds_all);
    // per-job tile / item counts from the device-side row counts
    if ((int)threadIdx.x < a.njobs) {
        const int j = threadIdx.x;
        const WJob &J = a.job[j];
        const WRegion &R = a.region[J.region];
        const int t0 = range_tiles(R, 0), t1 = R.n_ranges > 1 ? range_tiles(R, 1) : 0;
        lds_st_i(ctl + 4 * (WCtl::TILES0 + j), t0);
        lds_st_i(ctl + 4 * (WCtl::TILES + j), t0 + t1);
        lds_st_i(ctl + 4 * (WCtl::ITEMS + j), (t0 + t1 + J.tiles_per_item - 1) / J.tiles_per_item);
    }
    __builtin_amdgcn_s_barrier();
    // home job: the job that holds global item  blockIdx * total / gridDim  (items cost about the same by construction)
    long total = 0;
    for (int j = 0; j < a.njobs; ++j) total += lds_ld_u(ctl + 4 * (WCtl::ITEMS + j));
    if (total == 0) return;
    const long long k_t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    long g = (long)blockIdx.x * total / gridDim.x;
    int cur = 0;
    for (int j = 0; j < a.njobs; ++j) {
        const int n = lds_ld_u(ctl + 4 * (WCtl::ITEMS + j));
        if (g < n) { cur = j; break; }
        g -= n;
    }
    for (;;) {
        int item = 0;
        if (threadIdx.x == 0) item = atomicAdd(a.counters + cur, 1);
        item = wg_broadcast(ctl + 4 * WCtl::PICK, item);
        if (item < lds_ld_u(ctl + 4 * (WCtl::ITEMS + cur))) {
            if constexpr (FAMILY == 0) {
                switch (a.job[cur].shape) {
                    case WSI_BIG: wgrad_episode<WS_BIG, H2>(a, cur, item, lds_all); break;
                    case WSI_L0F: wgrad_episode<WS_L0F, H2>(a, cur, item, lds_all); break;
                    case WSI_L0B: wgrad_episode<WS_L0B, H2>(a, cur, item, lds_all); break;
                    case WSI_SKF: wgrad_episode<WS_SKF, H2>(a, cur, item, lds_all); break;
                    case WSI_DIR: wgrad_episode<WS_DIR, H2>(a, cur, item, lds_all); break;
#ifdef MNR_ALL_VARIANTS
                    case WSI_DIR_NOAPP: wgrad_episode<WS_DIR_NOAPP, H2>(a, cur, item, lds_all); break;
                    case WSI_DIR_NODIR: wgrad_episode<WS_DIR_NODIR, H2>(a, cur, item, lds_all); break;
#endif
                    default: break;
                }
            } else {
                switch (a.job[cur].shape) {
                    case WSI_BIG: wgrad_episode<WS_BIG>(a, cur, item, lds_all); break;
                    case WSI_D32: wgrad_episode<WS_D32>(a, cur, item, lds_all); break;
                    case WSI_D64: wgrad_episode<WS_D64>(a, cur, item, lds_all); break;
                    case WSI_D96: wgrad_episode<WS_D96>(a, cur, item, lds_all); break;
                    case WSI_D128: wgrad_episode<WS_D128>(a, cur, item, lds_all); break;
                    default: break;
                }
            }
        }
        // this job is exhausted (for us): steal from the job with the most items left
        int pick = -1;
        if (threadIdx.x < 64) {
            int key = -1;
            if ((int)threadIdx.x < a.njobs) {
                const int left = lds_ld_i(ctl + 4 * (WCtl::ITEMS + threadIdx.x)) -
                                 __hip_atomic_load(a.counters + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (left > 0) key = left * 64 + (63 - (int)threadIdx.x);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o));
            pick = key < 0 ? -1 : 63 - (key & 63);
        }
        cur = wg_broadcast(ctl + 4 * (WCtl::PICK + 1), pick);
        if (cur < 0) {
            if (a.prof && threadIdx.x == 0) a.prof[(long)blockIdx.x * 8 + 6] += __builtin_amdgcn_s_memtime() - k_t0;
            return;
        }
    }
}

// Sum the slab slots of every job into the parameter gradients (gradients are ACCUMULATED, like autograd).
// One thread per float4 of a job's (M x NP) tile (+ one block per job for the bias); the job's episode list is compacted
// into LDS first; partials are read four at a time so that every thread keeps several 16-byte loads in flight.
__global__ __launch_bounds__(256) void k_wgrad2_reduce(WArgs a) {
    __shared__ int list[W2_MAX_EPISODES];
    __shared__ int wave_cnt[4];
    int j = 0;
    for (int i = 1; i < a.njobs; ++i)
        if ((int)blockIdx.x >= a.job[i].red_block0) j = i;
    const WJob &J = a.job[j];
    const int b = blockIdx.x - J.red_block0;
    const int n_ep = min(a.counters[W2_MAX_JOBS], a.max_episodes);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cnt = 0;
    for (int e0 = 0; e0 < n_ep; e0 += 256) {          // ordered compaction (slot order) by wave-level prefix sums
        const int e = e0 + threadIdx.x;
        const bool hit = e < n_ep && a.ep_job[e] == j;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = cnt;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (hit) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = e;
        cnt += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (cnt == 0) return;
    const WShapeInfo si = wshape_info(J.shape);
    const int q = si.NP / 4, nblk = si.M * q / 256;
    if (b < nblk) {
        const int e4 = b * 256 + threadIdx.x;
        const int m = e4 / q, n4 = (e4 - m * q) * 4;
        const int nb = n4 / 32, sg = nb < si.nb[0] ? 0 : (nb < si.nb[0] + si.nb[1] ? 1 : 2);
        const int c0 = n4 - 32 * (sg == 0 ? 0 : (sg == 1 ? si.nb[0] : si.nb[0] + si.nb[1]));      // column inside the segment
        if (c0 >= J.N[sg]) return;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        for (int k = 0; k < si.KS; ++k) {
            const long off = (long)(k * si.M + m) * si.NP + n4;
            int i = 0;
            for (; i + 3 < cnt; i += 4) {
                const float4 v0 = *reinterpret_cast<const float4 *>(a.slab + (long)list[i] * W2_EP_FLOATS + off);
                const float4 v1 = *reinterpret_cast<const float4 *>(a.slab + (long)list[i + 1] * W2_EP_FLOATS + off);
                const float4 v2 = *reinterpret_cast<const float4 *>(a.slab + (long)list[i + 2] * W2_EP_FLOATS + off);
                const float4 v3 = *reinterpret_cast<const float4 *>(a.slab + (long)list[i + 3] * W2_EP_FLOATS + off);
                s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
                s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
                s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
                s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
            }
            for (; i < cnt; ++i) {
                const float4 v0 = *reinterpret_cast<const float4 *>(a.slab + (long)list[i] * W2_EP_FLOATS + off);
                s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
            }
        }
        const float r[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                            (s0.w + s1.w) + (s2.w + s3.w)};
        float *d = J.dw + (long)m * J.ldw + J.col0[sg] + c0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c0 + c < J.N[sg]) d[c] += r[c];
    } else if (J.db) {
        for (int m = threadIdx.x; m < si.M; m += 256) {
            float sum = 0.f;
            for (int i = 0; i < cnt; ++i) {
                const float *base = a.slab + (long)list[i] * W2_EP_FLOATS + si.KS * si.M * si.NP;
                for (int k = 0; k < si.KS; ++k) sum += base[k * si.M + m];
            }
            J.db[m] += sum;
        }
    }
}

// Split-precision launches without exponents from the data-gradient chain (stand-alone use, tests): the exponent of every job's dZ
// plane by one pass over it (one more read of the gradient tape; the fused step gets the exponents from k_mlp_bwd_h2 for free).
__global__ __launch_bounds__(256) void k_wgrad_zexp(WArgs a, int32_t *__restrict__ slots) {
    const int j = blockIdx.y;
    const WJob &J = a.job[j];
    const WRegion &R = a.region[J.region];
    const int M = wshape_info(J.shape).M, q = M / 4;
    const long ldz = M == 256 ? (long)J.ldz : (long)M;
    float mx = 0.f;
    for (int i = 0; i < R.n_ranges; ++i) {
        const long rows = (long)range_tiles(R, i) * W2_KT;
        const float *base = J.dz + R.row0[i] * ldz;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * q; e += (long)gridDim.x * 256) {
            const long row = e / q;
            const float4 v = *reinterpret_cast<const float4 *>(base + row * ldz + (e - row * q) * 4);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > 0.f && mx < 3.0e38f) {
        int e = 0;
        (void)frexpf(mx, &e);
        atomicMax(slots + j, (e < -100 ? -100 : e) + ZEXP_BIAS);
    }
}

static ArchDims arch_of2(const mnr_model_desc *d) {
    return ArchDims{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim, d->appearance_dim,
                    d->rgb_dim, d->mfma_tile};
}

struct W2Workspace {
    static constexpr size_t COUNTERS = 0;                                    // (W2_MAX_JOBS + 1) int32, padded
    static constexpr size_t EP_JOB = 256;
    static constexpr size_t SLAB = EP_JOB + (size_t)W2_MAX_EPISODES * 4;
    static constexpr size_t BYTES = SLAB + (size_t)W2_MAX_EPISODES * W2_EP_FLOATS * 4;
};

// the shape whose M and segment pitches match (ld[i] == 0 ends the list)
static int shape_for(int M, const int *ld, int nseg) {
    for (int i = 0; i < WSI_COUNT; ++i) {
        const WShapeInfo si = wshape_info(i);
        if (si.M != M || si.nseg != nseg) continue;
        bool ok = true;
        for (int k = 0; k < nseg; ++k) ok = ok && si.ld[k] == ld[k];
        if (ok) return i;
    }
    return -1;
}

static double env_d(const char *k, double d) { const char *v = getenv(k); return v ? atof(v) : d; }

// per-tile cost model (cycles per SIMD): MFMA time of the two co-resident waves vs LDS-DMA fill time, + a fixed
// boundary term -> tiles per item so that items of all jobs cost about the same (~4 tiles of the 256 x 256 shape)
static double wgrad_tile_cost(int shape, bool h2 = false) {
    const double dma_bpc = env_d("MNR_WGRAD_DMA_BPC", 10.0), fixed = env_d("MNR_WGRAD_FIXED", 600.0);
    const WShapeInfo si = wshape_info(shape);
    const double mfma = h2 ? (double)si.blocks_per_wave * (W2_KT / 16 / si.KS) * 3 * 32.0 * 2.0        // three 32-cycle products per block and K-step
                           : (double)si.blocks_per_wave * (W2_KT / 2 / si.KS) * 64.0 * 2.0;     // two co-resident waves per SIMD
    const double dma = si.tile_bytes / dma_bpc;
    return (mfma > dma ? mfma : dma) + fixed;
}

// shared tail of the two entry points: reduce-block table, workspace carving, the two launches
static int launch_wgrad2(WArgs &wa, int nj, int32_t *counters_dev, int32_t *ep_job_dev, float *slab_dev, bool zero_counters, hipStream_t s, int family,
                         bool h2 = false, bool zexp_pass = false) {
    wa.njobs = nj;
    {
        const int cap = (int)env_d("MNR_WGRAD_MAX_EPISODES", (double)W2_MAX_EPISODES);
        wa.max_episodes = cap < 0 ? 0 : (cap > W2_MAX_EPISODES ? W2_MAX_EPISODES : cap);
    }
    int red_blocks = 0;
    for (int i = 0; i < nj; ++i) {
        const WShapeInfo si = wshape_info(wa.job[i].shape);
        wa.job[i].red_block0 = red_blocks;
        red_blocks += si.M * si.NP / 4 / 256 + (wa.job[i].db ? 1 : 0);
    }
    wa.counters = counters_dev;
    wa.ep_job = ep_job_dev;
    wa.slab = slab_dev;
    wa.prof = getenv("MNR_WGRAD_PROF") ? reinterpret_cast<long long *>(wa.slab + (size_t)(W2_MAX_EPISODES - 1) * W2_EP_FLOATS) : nullptr;   // diagnostics: borrows the last slab slot
    size_t lds = 0;
    for (int i = 0; i < nj; ++i) {
        const size_t need = (W2_CTRL_FLOATS + 2 * ((size_t)wshape_info(wa.job[i].shape).tile_bytes / 4 + 255) / 256 * 256 * 1 + 2 * 256) * sizeof(float);
        lds = need > lds ? need : lds;
    }
    static bool lds_enabled_dev[MAX_DEVICES] = {};       // raise the dynamic-LDS cap once per device (benign if raced)
    bool &lds_enabled = lds_enabled_dev[device_slot()];
    if (!lds_enabled) {
        for (const void *f : {reinterpret_cast<const void *>(k_wgrad2<0>), reinterpret_cast<const void *>(k_wgrad2<1>),
                              reinterpret_cast<const void *>(k_wgrad2<0, true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_wgrad2): %s", hipGetErrorString(e));
        }
        lds_enabled = true;
    }
    if (zero_counters && hipMemsetAsync(wa.counters, 0, 256, s) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipMemsetAsync(wgrad counters)");
    const int grid = (int)env_d("MNR_WGRAD_WGS", 256.0);
    if (zexp_pass) {                            // exponent words = counters[32 + job] (inside the 256 bytes zeroed above)
        MNR_REQUIRE(zero_counters, "internal: the exponent pass needs the zeroed control words");
        hipLaunchKernelGGL(k_wgrad_zexp, dim3(64, nj), dim3(256), 0, s, wa, counters_dev + 32);
        int rc = check_launch("k_wgrad_zexp");
        if (rc) return rc;
    }
    if (h2) hipLaunchKernelGGL((k_wgrad2<0, true>), dim3(grid), dim3(W2_THREADS), lds, s, wa);
    else if (family == 0) hipLaunchKernelGGL(k_wgrad2<0>, dim3(grid), dim3(W2_THREADS), lds, s, wa);
    else hipLaunchKernelGGL(k_wgrad2<1>, dim3(grid), dim3(W2_THREADS), lds, s, wa);
    int rc = check_launch("k_wgrad2");
    if (rc) return rc;
    hipLaunchKernelGGL(k_wgrad2_reduce, dim3(red_blocks), dim3(256), 0, s, wa);
    return check_launch("k_wgrad2_reduce");
}

}  // namespace mnr

using namespace mnr;

extern "C" size_t mnr_wgrad_workspace_bytes(void) { return W2Workspace::BYTES; }

// h2: split-precision kernel; zexp[ri] = the region's exponent words (ZEXP_PLANES, written by k_mlp_bwd_h2) or, when zexp is NULL,
// found by a pass over the planes (k_wgrad_zexp)
static int wgrad_regions_impl(const mnr_wgrad_region *regions, int n_regions, int32_t *counters_dev, int32_t *ep_job_dev, float *slab_dev,
                              bool zero_counters, hipStream_t s, bool h2 = false, const int32_t *const *zexp = nullptr) {
    MNR_REQUIRE(regions && n_regions >= 1 && n_regions <= W2_MAX_REGIONS, "1..%d weight-gradient regions per launch", W2_MAX_REGIONS);
    WArgs wa{};
    int nj = 0;
    long rows_bound = 0;
    // per-tile cost model (cycles per SIMD): MFMA time of the two co-resident waves vs LDS-DMA fill time, + a fixed
    // boundary term -> tiles per item so that items of all jobs cost about the same (~4 tiles of the 256 x 256 shape)
    const double item_tiles = env_d("MNR_WGRAD_ITEM_TILES", 4.0);
    auto cost_of = [&](int shape) { return wgrad_tile_cost(shape, h2); };
    const double cost_big = cost_of(WSI_BIG);
    const int only_shape = (int)env_d("MNR_WGRAD_ONLY_SHAPE", -1.0);
    for (int ri = 0; ri < n_regions; ++ri) {
        const mnr_wgrad_region &rg = regions[ri];
        const mnr_model_desc *d = rg.desc;
        MNR_REQUIRE(d && rg.tape && rg.gtape && rg.n_ranges >= 1 && rg.n_ranges <= 2, "bad weight-gradient region %d", ri);
        ModelLayout m;
        int rc = layout_from_desc(d, m);
        if (rc != MNR_OK) return rc;
        MNR_REQUIRE(m.has_final, "training needs a model with the dir/appearance branch");
        MNR_REQUIRE(d->layer_dim == 256, "weight-gradient kernel supports layer_dim 256");
        WRegion &R = wa.region[ri];
        R.n_ranges = rg.n_ranges;
        for (int i = 0; i < rg.n_ranges; ++i) {
            MNR_REQUIRE(rg.row0[i] >= 0 && rg.n_rows[i] >= 0 && rg.row0[i] % 4 == 0, "range %d of region %d: bad rows", i, ri);
            const long padded = (rg.n_rows[i] + W2_KT - 1) / W2_KT * W2_KT;
            MNR_REQUIRE(rg.row0[i] + padded <= rg.tape_rows, "range %d of region %d: tape capacity must cover the rows padded to %d", i, ri, W2_KT);
            R.row0[i] = rg.row0[i]; R.n_rows[i] = rg.n_rows[i]; R.n_units[i] = rg.n_units_dev[i]; R.rows_per_unit[i] = rg.rows_per_unit[i];
            rows_bound += rg.n_rows[i];
        }
        const TapeLayout tl = tape_layout(arch_of2(d));
        const mnr_model_grads &G = rg.grad;
        const int W = d->layer_dim, L = d->layers;
        const int Ecols = emb_cols(d->xyz_dim, d->pos_xyz_dim), EDcols = emb_cols(3, d->pos_dir_dim);
        const long cap = rg.tape_rows;
        int err = 0;
        struct SegIn { const float *in; int ld, N, col0; };
        auto add = [&](int plane, const float *dz, int M, std::initializer_list<SegIn> segs, float *dw, int ldw, float *db) {
            int ld[3] = {0, 0, 0}, n = 0;
            for (const SegIn &sg : segs) ld[n++] = sg.ld;
            const int shape = shape_for(M, ld, n);
            if (shape < 0 || nj >= W2_MAX_JOBS || !dw) { err = 1; return; }
            if (only_shape >= 0 && shape != only_shape) return;
            WJob &J = wa.job[nj++];
            J.dz = dz; J.dw = dw; J.db = db; J.ldw = ldw;
            J.ldz = M; J.ldin0 = ld[0];
            n = 0;
            for (const SegIn &sg : segs) { J.in[n] = sg.in; J.col0[n] = (int16_t)sg.col0; J.N[n] = (int16_t)sg.N; ++n; }
            J.shape = (int16_t)shape; J.region = (int16_t)ri;
            J.zexp = !h2 ? nullptr : (zexp && zexp[ri] ? zexp[ri] + plane : counters_dev + 32 + (nj - 1));
            int tpi = (int)(item_tiles * cost_big / cost_of(shape) + 0.5);
            J.tiles_per_item = tpi < 1 ? 1 : tpi;
        };
        const float *embx = rg.tape + (long)tl.embx_off * cap;
        for (int l = 0; l < L; ++l) {
            MNR_REQUIRE(G.layer_w[l] && G.layer_b[l], "missing gradient pointer for layer %d", l);
            const float *dz = rg.gtape + (long)tl.act_off[l] * cap;
            const bool skip = (d->skip_mask >> l) & 1;
            const float *prev = l ? rg.tape + (long)tl.act_off[l - 1] * cap : nullptr;
            if (l == 0) add(l, dz, W, {{embx, tl.embx_w, Ecols, 0}}, G.layer_w[l], Ecols, G.layer_b[l]);
            else if (skip && d->xyz_dim == 3) add(l, dz, W, {{embx, tl.embx_w, Ecols, 0}, {prev, W, W, Ecols}}, G.layer_w[l], Ecols + W, G.layer_b[l]);
            else if (skip) {
                add(l, dz, W, {{prev, W, W, Ecols}}, G.layer_w[l], Ecols + W, G.layer_b[l]);
                add(l, dz, W, {{embx, tl.embx_w, Ecols, 0}}, G.layer_w[l], Ecols + W, nullptr);
            }
            else add(l, dz, W, {{prev, W, W, 0}}, G.layer_w[l], W, G.layer_b[l]);
        }
        MNR_REQUIRE(G.final_w && G.final_b && G.dir_a_w && G.dir_a_b, "missing final / dir_a gradient pointers");
        add(L, rg.gtape + (long)tl.fin_off * cap, W, {{rg.tape + (long)tl.act_off[L - 1] * cap, W, W, 0}}, G.final_w, W, G.final_b);
        {
            const float *dz = rg.gtape + (long)tl.dact_off * cap;
            const int ldw = W + EDcols + d->appearance_dim;
            const SegIn fin{rg.tape + (long)tl.fin_off * cap, W, W, 0}, dir{rg.tape + (long)tl.embd_off * cap, tl.embd_w, EDcols, W},
                app{rg.tape + (long)tl.app_off * cap, tl.app_w, d->appearance_dim, W + EDcols};
            if (EDcols && d->appearance_dim) add(L + 1, dz, W / 2, {fin, dir, app}, G.dir_a_w, ldw, G.dir_a_b);
            else if (EDcols) add(L + 1, dz, W / 2, {fin, dir}, G.dir_a_w, ldw, G.dir_a_b);
            else add(L + 1, dz, W / 2, {fin, app}, G.dir_a_w, ldw, G.dir_a_b);
        }
        MNR_REQUIRE(!err, "weight-gradient job table: unsupported layer shape or too many jobs (region %d)", ri);
    }
    if (rows_bound == 0) return MNR_OK;
    MNR_REQUIRE(!h2 || nj <= 24, "split-precision weight gradients: too many jobs");
    return launch_wgrad2(wa, nj, counters_dev, ep_job_dev, slab_dev, zero_counters, s, 0, h2, h2 && !zexp);
}

extern "C" int mnr_mlp_backward_weights_multi(const mnr_wgrad_region *regions, int n_regions, void *workspace_dev,
                                              size_t workspace_bytes, void *stream) {
    MNR_REQUIRE(workspace_dev && workspace_bytes >= W2Workspace::BYTES, "workspace missing or smaller than mnr_wgrad_workspace_bytes()");
    char *ws = reinterpret_cast<char *>(workspace_dev);
    return wgrad_regions_impl(regions, n_regions, reinterpret_cast<int32_t *>(ws + W2Workspace::COUNTERS),
                              reinterpret_cast<int32_t *>(ws + W2Workspace::EP_JOB), reinterpret_cast<float *>(ws + W2Workspace::SLAB), true,
                              as_stream(stream));
}

extern "C" int mnr_mlp_backward_weights_multi_h2(const mnr_wgrad_region *regions, int n_regions, void *workspace_dev,
                                                 size_t workspace_bytes, void *stream) {
    MNR_REQUIRE(workspace_dev && workspace_bytes >= W2Workspace::BYTES, "workspace missing or smaller than mnr_wgrad_workspace_bytes()");
    char *ws = reinterpret_cast<char *>(workspace_dev);
    return wgrad_regions_impl(regions, n_regions, reinterpret_cast<int32_t *>(ws + W2Workspace::COUNTERS),
                              reinterpret_cast<int32_t *>(ws + W2Workspace::EP_JOB), reinterpret_cast<float *>(ws + W2Workspace::SLAB), true,
                              as_stream(stream), true, nullptr);
}

// the step's form (csrc/step.hip): control words placed by the caller and already zeroed by its one memset; zexp != NULL selects the
// split-precision kernel with the exponent words of every region (written by the split-precision data-gradient chain)
int mnr::wgrad_regions_launch(const mnr_wgrad_region *regions, int n_regions, int32_t *counters_dev, int32_t *ep_job_dev, float *slab_dev,
                              hipStream_t s, const int32_t *const *zexp) {
    return wgrad_regions_impl(regions, n_regions, counters_dev, ep_job_dev, slab_dev, false, s, zexp != nullptr, zexp);
}
size_t mnr::wgrad_ep_job_bytes() { return (size_t)W2_MAX_EPISODES * 4; }
size_t mnr::wgrad_slab_bytes() { return (size_t)W2_MAX_EPISODES * W2_EP_FLOATS * 4; }

extern "C" int mnr_wgrad_jobs(const mnr_wgrad_job *jobs, int n_jobs, int64_t rows, void *workspace_dev, size_t workspace_bytes,
                              void *stream) {
    MNR_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= W2_MAX_JOBS, "mnr_wgrad_jobs: 1..%d jobs per call", W2_MAX_JOBS);
    MNR_REQUIRE(workspace_dev && workspace_bytes >= W2Workspace::BYTES, "workspace missing or smaller than mnr_wgrad_workspace_bytes()");
    MNR_REQUIRE(rows >= 0 && rows % W2_KT == 0, "mnr_wgrad_jobs: rows must be a multiple of %d", W2_KT);
    if (rows == 0) return MNR_OK;
    WArgs wa{};
    WRegion &R = wa.region[0];
    R.n_ranges = 1; R.row0[0] = 0; R.n_rows[0] = rows;
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const double item_tiles = env_d("MNR_WGRAD_ITEM_TILES", 4.0);
    const double cost_big = wgrad_tile_cost(WSI_BIG);
    for (int i = 0; i < n_jobs; ++i) {
        const mnr_wgrad_job &q = jobs[i];
        int shape = -1;
        switch (q.in_block) {
            case 256: shape = WSI_BIG; break;
            case 32: shape = WSI_D32; break;
            case 64: shape = WSI_D64; break;
            case 96: shape = WSI_D96; break;
            case 128: shape = WSI_D128; break;
            default: break;
        }
        MNR_REQUIRE(shape >= 0, "mnr_wgrad_jobs: job %d: in_block must be 32, 64, 96, 128 or 256", i);
        MNR_REQUIRE(q.dz && q.in && q.dw && al16(q.dz) && al16(q.in) && q.ldz % 4 == 0 && q.ldin % 4 == 0 && q.ldz >= 256,
                    "mnr_wgrad_jobs: job %d: operands must be 16-byte aligned", i);
        MNR_REQUIRE(q.in_cols >= 1 && q.in_cols <= q.in_block && (q.in_block == 256 ? q.ldin >= 256 : q.ldin == q.in_block),
                    "mnr_wgrad_jobs: job %d: bad input columns / pitch", i);
        MNR_REQUIRE(q.ldz < (1ll << 24) && q.ldin < (1ll << 24) && q.ldw < (1ll << 31), "mnr_wgrad_jobs: job %d: pitch too large", i);
        WJob &J = wa.job[i];
        J.dz = q.dz; J.in[0] = q.in; J.dw = q.dw; J.db = q.db;
        J.ldw = (int32_t)q.ldw; J.ldz = (int32_t)q.ldz; J.ldin0 = (int32_t)q.ldin;
        J.col0[0] = 0; J.N[0] = (int16_t)q.in_cols;
        J.shape = (int16_t)shape; J.region = 0;
        const int tpi = (int)(item_tiles * cost_big / wgrad_tile_cost(shape) + 0.5);
        J.tiles_per_item = tpi < 1 ? 1 : tpi;
    }
    char *ws = reinterpret_cast<char *>(workspace_dev);
    return launch_wgrad2(wa, n_jobs, reinterpret_cast<int32_t *>(ws + W2Workspace::COUNTERS), reinterpret_cast<int32_t *>(ws + W2Workspace::EP_JOB),
                         reinterpret_cast<float *>(ws + W2Workspace::SLAB), true, as_stream(stream), 1);
}
