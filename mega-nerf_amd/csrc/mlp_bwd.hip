// mlp_bwd.hip -- backward pass of the fused NeRF MLP for gfx950 (training; the reference gets this
// from torch autograd over nerf.py:115-160).
//
// Three kernels:
//  * k_pack_bwd      : nn.Linear weights -> transposed MFMA-fragment chunk stream (reverse layer order).
//  * k_mlp_bwd       : data-gradient chain.  Mirrors the forward kernel: one wavefront owns TILE samples and
//                      all W features, dZ_l (C-layout registers) is the B operand of the W_l^T product whose
//                      accumulators, masked by the stored ReLU activations, are dZ_{l-1}.  Every dZ_l is
//                      written to the gradient tape for the weight-gradient GEMMs; the appearance-embedding
//                      gradient is reduced over the wave and added atomically.
//  * k_wgrad         : dW_l += dZ_l^T . IN_l  and  db_l += colsum(dZ_l) for every layer in ONE launch
//                      (job table; split over row ranges; v_mfma_f32_32x32x2_f32; fp32 atomics into .grad).
//  * k_head_grads    : sigma / rgb head weight gradients (M = 1 and 3: VALU).
#include <stdlib.h>

#include "mlp_bwd_device.h"
#include "sh_device.h"
#include "pack_device.h"
#include "step_internal.h"

namespace mnr {

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m);

static ArchDims arch_of(const mnr_model_desc *d) {
    return ArchDims{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim, d->appearance_dim,
                    d->rgb_dim, d->mfma_tile};
}

int bwd_layout_from_desc(const mnr_model_desc *d, BwdLayout &b) {
    ModelLayout m;
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    const char *err = nullptr;
    if (build_bwd_layout(arch_of(d), b, &err)) return set_err(MNR_E_UNSUPPORTED, "backward layout: %s", err);
    if (!b.has_final) return set_err(MNR_E_UNSUPPORTED, "training needs a model with the dir/appearance branch");
    int n = 0;
    b.layer[n++].w = d->dir_a_w;
    b.layer[n++].w = d->final_w;
    for (int l = d->layers - 1; l >= 1; --l) b.layer[n++].w = d->layer_w[l];
    return MNR_OK;
}

__global__ void k_pack_bwd(BwdLayout b, float4 *__restrict__ chunks) {
    pack_bwd_thread(b, chunks, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

template <class C>
__device__ __forceinline__ void mlp_bwd_body(const MlpBwdArgs &a, long blk, int cidx = 0) {
    constexpr int TILE = C::TILE, P = C::P, H = C::H, NOB = C::NOB, RPB = C::RPB, H2 = C::H2, W = C::W;
    static_assert(C::HAS_FINAL, "backward kernel covers the dir/appearance architecture");
    using AccT = typename std::conditional<TILE == 32, floatx16, floatx4>::type;
    // dir_a^T produces W final-feature rows + APP appearance rows, padded to a multiple of 4 blocks
    constexpr int ROWS_D = cdiv(W + C::APP, 4 * TILE) * 4 * TILE, NOBD = ROWS_D / TILE;
    constexpr int GPCD = CHUNK_F4 / (NOBD * 64) < 1 ? 1 : CHUNK_F4 / (NOBD * 64);
    extern __shared__ float4 lds_ring[];

    long n_rows, row_base = 0, tape_row0 = a.tape_row0;
    const float4 *chunks = a.chunks;
    const float *aux = a.aux;
    float *d_emb_a = a.d_emb_a;
    if (a.dcells) {
        // training step of several submodules (see mlp_fwd_body): cell = blockIdx.y, `blk` = workgroup index inside the cell
        const MlpCellSeg cell = a.dcells[cidx];
        n_rows = cell.n_units ? (long)__builtin_amdgcn_readfirstlane(*cell.n_units) * a.rows_per_unit : a.cell_rows;
        if (blk * C::ROWS_PER_WG >= n_rows) return;
        chunks = reinterpret_cast<const float4 *>(cell.packed_bwd);
        aux = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed) + a.aux_byte_off));
        d_emb_a = reinterpret_cast<float *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(cell.d_emb_a))));
        row_base = (long)cidx * a.cell_rows;
        tape_row0 = uniform_long(cell.tape_row0);
    } else {
        n_rows = a.n_units_dev ? (long)(*a.n_units_dev) * a.rows_per_unit : a.n_rows;
        if (blk * C::ROWS_PER_WG >= n_rows) return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane / TILE;
    const long lrow = (blk * 4 + wave) * TILE + (lane % TILE);
    const bool valid = lrow < n_rows;
    const long lrc = valid ? lrow : n_rows - 1;
    const long rc = row_base + lrc;              // row in d_out / out / ray space
    const long cap = a.tape_rows;
    const long trow = lrc + tape_row0;           // row in tape space

    WStream st;
    st.g = reinterpret_cast<const float4 *>(uniform_ptr(reinterpret_cast<const char *>(chunks)));      // into SGPRs once: the stream pointer arithmetic stays scalar
    st.lds = lds_ring;
    st.cur = 1;
    st.issue();

    // ---- output activations backward -------------------------------------------------------------
    float dr[3], ds;
    {
        const float *go = a.d_out + rc * a.d_out_stride, *o = a.out + rc * a.out_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) dr[c] = (valid && C::RGB == 3) ? go[c] * o[c] * (1.f - o[c]) : 0.f;        // sigmoid'
        const float sg = o[3];
        const float da = a.sigma_act ? (1.f - expf(-sg)) : (sg > 0.f ? 1.f : 0.f);           // softplus' = 1 - e^-softplus
        ds = valid ? go[3] * da : 0.f;
        if (valid && part == 0) *reinterpret_cast<float4 *>(a.dheads + trow * 4) = make_float4(dr[0], dr[1], dr[2], ds);
    }

    // ---- rgb head backward -> dZ of dir_a ---------------------------------------------------------
    float dd[H2];
    {
        if constexpr (C::RGB == 3) {
            const float *wr = aux + a.rgb_off;
#pragma unroll
            for (int q = 0; q < H2 / 4; ++q) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wr + (c * P + part) * H2 + 4 * q);
                    s.x = fmaf(dr[c], w4.x, s.x); s.y = fmaf(dr[c], w4.y, s.y);
                    s.z = fmaf(dr[c], w4.z, s.z); s.w = fmaf(dr[c], w4.w, s.w);
                }
                dd[4 * q] = s.x; dd[4 * q + 1] = s.y; dd[4 * q + 2] = s.z; dd[4 * q + 3] = s.w;
            }
        } else {
            // colour epilogue + rgb layer were differentiated by the caller: pick up dL/d(dir_a output) in C layout
            const float *src = a.dd_in + rc * (W / 2) + 4 * part;
#pragma unroll
            for (int q = 0; q < H2 / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(src + 4 * P * q);
                dd[4 * q] = valid ? v.x : 0.f; dd[4 * q + 1] = valid ? v.y : 0.f;
                dd[4 * q + 2] = valid ? v.z : 0.f; dd[4 * q + 3] = valid ? v.w : 0.f;
            }
        }
        const MaskBits<H2> dm = mask_load<H2>(a.tape + a.tl.dmask_off * cap, trow, a.tl.dmask_w, part);
        mask_apply(dd, dm);
    }

    // ---- dir_a^T: d(final features) and d(appearance embedding) -----------------------------------
    float g[H];
    {
        AccT accd[NOBD];
        zero_acc(accd);
        st.next_chunk();
        gtape_store<P>(dd, a.gtape + a.tl.dact_off * cap, (unsigned)((trow * (W / 2) + 4 * part) * 4), valid);
        run_segment<TILE, NOBD, H2 / 4, GPCD, 0, false, NOBD, 0, true>(accd, dd, st, lane);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < RPB; ++r) g[ob * RPB + r] = accd[ob][r];    // dZ of xyz_encoding_final (no activation)
        if constexpr (C::APP > 0) {
            if (d_emb_a) {
                constexpr int NAB = cdiv(C::APP, TILE);          // appearance blocks after the W final-feature rows
                const long ray = rc / a.rows_per_ray;
                long idx = a.idx_is_float ? (long)reinterpret_cast<const float *>(a.idx)[ray * a.idx_stride]
                                          : (long)reinterpret_cast<const int32_t *>(a.idx)[ray * a.idx_stride];
                idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
                const bool uniform = (a.rows_per_ray % TILE) == 0;     // all rows of this wave share one ray
#pragma unroll
                for (int b = 0; b < NAB; ++b)
#pragma unroll
                    for (int r = 0; r < RPB; ++r) {
                        float v = accd[NOB + b][r];
                        // feature of flat register (b, r): C layout of one TILE-row block
                        const int col = TILE == 32 ? b * 32 + (r & 3) + 8 * (r >> 2) + 4 * part : b * 16 + 4 * part + r;
                        if (uniform) {
#pragma unroll
                            for (int o = 1; o < TILE; o <<= 1) v += __shfl_xor(v, o);
                            if ((lane % TILE) == 0 && col < C::APP) atomicAdd(d_emb_a + idx * C::APP + col, v);
                        } else if (valid && col < C::APP) {
                            atomicAdd(d_emb_a + idx * C::APP + col, v);
                        }
                    }
            }
        }
    }

    // ---- final^T (+ sigma head): dZ of trunk layer L-1 ---------------------------------------------
    constexpr bool PUBT = seg_weaves<TILE, NOB, H / 4, C::GPC, 0>();
    AccT acc[NOB];
    {
        const float *ws = aux + a.sigma_off + part * H;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int q = 0; q < RPB / 4; ++q) {
                const float4 w4 = *reinterpret_cast<const float4 *>(ws + ob * RPB + 4 * q);
                acc[ob][4 * q + 0] = ds * w4.x; acc[ob][4 * q + 1] = ds * w4.y;
                acc[ob][4 * q + 2] = ds * w4.z; acc[ob][4 * q + 3] = ds * w4.w;
            }
        const MaskBits<H> bits = mask_load<H>(a.tape + a.tl.mask_off[C::NL - 1] * cap, trow, a.tl.mask_w, part);
        st.next_chunk();
        gtape_store<P>(g, a.gtape + a.tl.fin_off * cap, (unsigned)((trow * W + 4 * part) * 4), valid);
        // (woven pipeline: every W x W layer publishes its successor's first chunk two batches before its own end -- run_segment PUB_END)
        run_segment<TILE, NOB, H / 4, C::GPC, 0, (PUBT && C::NL > 1), NOB, 0, true>(acc, g, st, lane);
        acc_to_regs<NOB, RPB, false>(g, acc);
        mask_apply(g, bits);
    }

    // ---- trunk layers L-1 .. 1 transposed ------------------------------------------------------------
    static_for<0, C::NL - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int l = C::NL - 1 - decltype(jc)::value;       // consumes dZ_l, produces dZ_{l-1}
        zero_acc(acc);
        const MaskBits<H> bits = mask_load<H>(a.tape + a.tl.mask_off[l - 1] * cap, trow, a.tl.mask_w, part);
        if constexpr (!PUBT) st.next_chunk();
        gtape_store<P>(g, a.gtape + a.tl.act_off[l] * cap, (unsigned)((trow * W + 4 * part) * 4), valid);       // dZ_l (deferred, see gtape_store)
        run_segment<TILE, NOB, H / 4, C::GPC, 0, (PUBT && l > 1), NOB, 0, true>(acc, g, st, lane);
        acc_to_regs<NOB, RPB, false>(g, acc);
        mask_apply(g, bits);
    });
    gtape_store<P>(g, a.gtape + a.tl.act_off[0] * cap, (unsigned)((trow * W + 4 * part) * 4), valid);
}

template <class C>
__global__ __launch_bounds__(256, C::TILE == 16 && C::W <= 256 ? 2 : 1) void k_mlp_bwd(MlpBwdArgs a) {
    mlp_bwd_body<C>(a, blockIdx.x);
}

// The data-gradient chains of several forward passes (coarse + fine rows of the foreground and of the background model) in
// ONE launch: workgroups [wg0[s], wg0[s+1]) belong to segment s, which runs configuration CA or CB.
constexpr int MLP_BWD_MAX_SEGS = 4;
struct MlpBwdMulti {
    MlpBwdArgs seg[MLP_BWD_MAX_SEGS];
    int32_t wg0[MLP_BWD_MAX_SEGS + 1];
    int32_t is_b[MLP_BWD_MAX_SEGS];
};
template <class CA, class CB>
__global__ __launch_bounds__(256, 2) void k_mlp_bwd_multi(MlpBwdMulti m) {
    const int blk = blockIdx.x;
    const int s = (blk >= m.wg0[1]) + (blk >= m.wg0[2]) + (blk >= m.wg0[3]);
    if (m.is_b[s]) mlp_bwd_body<CB>(m.seg[s], blk - m.wg0[s], blockIdx.y);
    else mlp_bwd_body<CA>(m.seg[s], blk - m.wg0[s], blockIdx.y);
}

// =================================================================================================
// Weight gradients: dW[M][ldw] (+col0) += dZ[rows][M]^T . IN[rows][N], db[M] += colsum(dZ)
// =================================================================================================
struct WgradJob {
    const float *dz;  int ldz, M;        // ldz == M (planes are dense)
    const float *in;  int ldin, N;
    float *dw;  int ldw, col0;
    float *db;                            // NULL: no bias gradient from this job
    int wg0, nwg;
};
constexpr int WGRAD_MAX_JOBS = 20;
struct WgradArgs {
    WgradJob job[WGRAD_MAX_JOBS];
    int njobs;
    long n_rows;
    const int32_t *n_units_dev;
    int rows_per_unit;
    long row0;                            // first tape row of the region
    int32_t *work_counter;                // device-side item queue head (zeroed before the launch)
};

constexpr int WG_KT = 32;                 // rows (K) per LDS tile
constexpr int WG_THREADS = 512;           // 8 waves: 2 (M) x 4 (N)
constexpr int WG_MIN_ROWS = 512;          // rows per work item, lower bound (multiple of WG_KT)

// One (M x N) weight-gradient job slice.  Wave (wr, wc) of the 2 x 2 wave grid owns MBW x NBW blocks of 32 x 32.
// Tiles of WG_KT rows of dZ and IN are streamed global -> LDS with LDS-DMA into a 2-stage ring; the MFMA loop
// reads both operands with conflict-free ds_read_b32 (lanes run along the feature dimension).
// decode a work item -> (job, row range); returns false when the item is past the end of the table
__device__ __forceinline__ bool wgrad_decode(const WgradArgs &a, int item, long n_rows, int &job, long &r_begin, long &r_end) {
    job = -1;
    for (int i = 0; i < a.njobs; ++i)
        if (item >= a.job[i].wg0 && item < a.job[i].wg0 + a.job[i].nwg) job = i;
    if (job < 0) return false;
    const WgradJob &J = a.job[job];
    long rps = (n_rows + J.nwg - 1) / J.nwg;
    rps = (rps + WG_KT - 1) / WG_KT * WG_KT;
    // the item table is sized on the host for the worst-case row count; when the device-side count is much smaller
    // (background rays), keep items coarse enough that an accumulator flush (up to 64 K atomics) stays amortised
    if (rps < WG_MIN_ROWS) rps = WG_MIN_ROWS;
    r_begin = (long)(item - J.wg0) * rps;
    r_end = min(n_rows, r_begin + rps);
    return true;
}

// workgroup-wide pull of the next item from the device-side queue (mailbox = first word of the LDS block)
__device__ __forceinline__ int wgrad_pull(const WgradArgs &a, float *lds) {
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<int *>(lds)[0] = atomicAdd(a.work_counter, 1);
    __syncthreads();
    return reinterpret_cast<volatile int *>(lds)[0];
}

// Runs consecutive work items of ONE (M x N) job, accumulating in registers, then flushes with atomics.
// The workgroup has 8 waves (2 per SIMD, so one wave's LDS waits hide behind the other's MFMAs); wave (wr, wc) of
// the 2 x 4 wave grid owns MBW x NBW blocks of 32 x 32.  Tiles of WG_KT rows of dZ and IN are
// streamed global -> LDS with LDS-DMA into a 2-stage ring; the MFMA loop reads both operands with conflict-free
// ds_read_b32 (lanes run along the feature dimension).  Returns the first item that belongs to another job.
template <int MBW, int NBW>
__device__ __forceinline__ int wgrad_run(const WgradArgs &a, int item, long n_rows, float *lds_all) {
    float *lds = lds_all + 64;                         // word 0 is the queue mailbox
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int i32 = lane & 31, kk = lane >> 5;
    int job;
    long r_begin, r_end;
    wgrad_decode(a, item, n_rows, job, r_begin, r_end);
    const WgradJob &J = a.job[job];
    const long row0 = a.row0;
    const int M = J.M, ldin = J.ldin;
    const int dz_f4 = WG_KT * M / 4, n_f4 = dz_f4 + WG_KT * ldin / 4;
    const int stage_floats = (WG_KT * (M + ldin) + 255) / 256 * 256 + 256;     // slack: B fragments may read past ldin

    floatx16 acc[MBW][NBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[m][n] = floatx16(0.f);
    float bsum[MBW];
#pragma unroll
    for (int m = 0; m < MBW; ++m) bsum[m] = 0.f;

    int next;
    bool touched = false;
    for (;;) {
        const long ntiles = (r_end - r_begin + WG_KT - 1) / WG_KT;
        touched |= ntiles > 0;
        auto issue = [&](long ti) {
            const long r0 = r_begin + ti * WG_KT;
            const int nr = (int)min((long)WG_KT, r_end - r0);
            float *stage = lds + (ti & 1) * stage_floats;
            if (nr == WG_KT) {
                // full tile: both operand tiles are contiguous blocks of WG_KT rows -> pure pointer arithmetic
                const float *zsrc = J.dz + (row0 + r0) * (long)M, *isrc = J.in + (row0 + r0) * (long)ldin;
                for (int t0 = 0; t0 < n_f4; t0 += WG_THREADS) {
                    const int t = t0 + threadIdx.x;
                    if (t < n_f4) {
                        const float *src = t < dz_f4 ? zsrc + t * 4 : isrc + (t - dz_f4) * 4;
                        float *dst = stage + (t0 + wave * 64) * 4;        // wave-uniform base; HW adds lane*16
                        __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)dst, 16, 0, 0);
                    }
                }
                return;
            }
            for (int t0 = 0; t0 < n_f4; t0 += WG_THREADS) {                     // ragged tile (at most one per region)
                const int t = t0 + threadIdx.x;
                if (t < n_f4) {
                    const float *src;
                    if (t < dz_f4) {
                        const int e = t * 4, r = min(e / M, nr - 1);      // clamp: rows past the region re-read the last row
                        src = J.dz + (row0 + r0 + r) * (long)M + (e % M);
                    } else {
                        const int e = (t - dz_f4) * 4, r = min(e / ldin, nr - 1);
                        src = J.in + (row0 + r0 + r) * (long)ldin + (e % ldin);
                    }
                    float *dst = stage + (t0 + wave * 64) * 4;
                    __builtin_amdgcn_global_load_lds((global_cvoid_t *)src, (lds_void_t *)dst, 16, 0, 0);
                }
            }
        };
        if (ntiles > 0) issue(0);
        for (long ti = 0; ti < ntiles; ++ti) {
            __syncthreads();                               // tile ti landed; everyone is done with tile ti-1
            if (ti + 1 < ntiles) issue(ti + 1);
            float *dz_t = lds + (ti & 1) * stage_floats, *in_t = dz_t + WG_KT * M;
            const int nr = (int)min((long)WG_KT, r_end - (r_begin + ti * WG_KT));
            if (nr < WG_KT) {                              // ragged last tile: rows >= nr must contribute nothing
                for (int e = nr * M + threadIdx.x; e < WG_KT * M; e += WG_THREADS) dz_t[e] = 0.f;
                for (int e = nr * ldin + threadIdx.x; e < WG_KT * ldin; e += WG_THREADS) in_t[e] = 0.f;
                __syncthreads();
            }
            const float *zr = dz_t + kk * M + wr * MBW * 32 + i32;
            const float *ir = in_t + kk * ldin + wc * NBW * 32 + i32;
#pragma unroll 4
            for (int k = 0; k < WG_KT; k += 2) {
                float af[MBW], bf[NBW];
#pragma unroll
                for (int m = 0; m < MBW; ++m) af[m] = zr[k * M + m * 32];
#pragma unroll
                for (int n = 0; n < NBW; ++n) bf[n] = ir[k * ldin + n * 32];
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    bsum[m] += af[m];
#pragma unroll
                    for (int n = 0; n < NBW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
                }
            }
        }
        next = wgrad_pull(a, lds_all);                     // (also fences the last tile's LDS reads)
        int njob;
        if (!wgrad_decode(a, next, n_rows, njob, r_begin, r_end) || njob != job) break;
    }
    if (!touched) return next;                          // only empty row ranges: nothing to add
    // flush: C layout -> atomics into the nn.Parameter gradient
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            const int col = (wc * NBW + n) * 32 + i32;
            if (col < J.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowm = (wr * MBW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    atomicAdd(J.dw + (long)rowm * J.ldw + J.col0 + col, acc[m][n][r]);
                }
            }
        }
        if (J.db && wc == 0) {
            const float sm = bsum[m] + __shfl_xor(bsum[m], 32);
            if (kk == 0) atomicAdd(J.db + (wr * MBW + m) * 32 + i32, sm);
        }
    }
    return next;
}

// Persistent workgroups pulling (job, row-range) items from a device-side queue: perfect load balance across
// jobs of very different shapes, while consecutive items of one job share a single accumulator flush.
// DENSE = row count known on the host (the foreground's single coarse+fine launch); the device-counted form serves the
// compacted background rows.  Two symbols so that a kernel trace reports the dominant launch on its own row.
template <bool DENSE>
__global__ __launch_bounds__(WG_THREADS, 2) void k_wgrad(WgradArgs a) {
    extern __shared__ float wlds[];
    const long n_rows = DENSE ? a.n_rows : (long)(*a.n_units_dev) * a.rows_per_unit;
    int item = wgrad_pull(a, wlds);
    for (;;) {
        int job;
        long rb, re;
        if (!wgrad_decode(a, item, n_rows, job, rb, re)) return;
        if (rb >= re) { item = wgrad_pull(a, wlds); continue; }      // row range past the device-side row count
        const WgradJob &J = a.job[job];
        const int NBW = ((J.N + 31) / 32 + 3) / 4;         // column blocks per wave (4 wave columns)
        if (J.M == 256) {
            if (NBW == 2) item = wgrad_run<4, 2>(a, item, n_rows, wlds);
            else item = wgrad_run<4, 1>(a, item, n_rows, wlds);
        } else {   // M == 128
            if (NBW == 2) item = wgrad_run<2, 2>(a, item, n_rows, wlds);
            else item = wgrad_run<2, 1>(a, item, n_rows, wlds);
        }
    }
}

// sigma / rgb head weight gradients:  d w_sigma[W] = sum_r ds[r] * a_{L-1}[r][:],  d w_rgb[3][W/2] = sum_r dr[r][c] * d[r][:]
// Pure streaming (1.5 KB per row): HBM-bound, so what matters is bytes per load and loads in flight.  A block is 16 wavefronts; a
// wavefront takes PAIRS of rows: a row of the 256-wide plane is one 16-byte load per lane (lane l <-> features 4l .. 4l + 3: the sigma
// head needs no cross-lane sum at all), the two 128-wide rows of the pair are one 16-byte load per lane (lane half <-> row); four
// pairs per iteration are requested before the first is used.  The per-row output gradients are wave-uniform (sigma: scalar loads) or
// half-uniform (rgb: one broadcast 16-byte load).  Wavefronts are combined in LDS, one set of 644 atomics per block (few, long
// blocks: round 1 launched 1024 small blocks per segment and spent most of its time on those same-address atomics).
// (round 3: 0.124 -> 0.111 ms on the benchmark step; the first version read the planes with 4-byte loads, one feature per thread.
// What is left is the launch, 16 short wavefronts per CU and the blocks' same-address atomics, not the stream: 0.33 GB in 0.11 ms.)
constexpr int HG_GROUPS = 4;                 // block = 256 * HG_GROUPS threads = 16 wavefronts
__device__ __forceinline__ void head_grads_body(const float *__restrict__ dheads, const float *__restrict__ a_last, int W,
                                                const float *__restrict__ dact, int W2, long row0, long n_rows,
                                                const int32_t *__restrict__ n_units_dev, int rows_per_unit,
                                                float *__restrict__ d_sigma_w, float *__restrict__ d_sigma_b,
                                                float *__restrict__ d_rgb_w, float *__restrict__ d_rgb_b, int with_rgb, int block, int n_blocks) {
    constexpr int NWAVE = 4 * HG_GROUPS, SLOT = 256 + 3 * 128 + 4;
    __shared__ float red[NWAVE][SLOT];
    const long n = n_units_dev ? (long)(*n_units_dev) * rows_per_unit : n_rows;
    const long per = ((n + n_blocks - 1) / n_blocks + 7) / 8 * 8;            // whole 8-row chunks per block
    const long rb = (long)block * per, re = min(n, rb + per);
    if (rb >= re) return;
    const long r0 = row0 + rb, r1 = row0 + re;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hh = lane >> 5, l5 = lane & 31;
    float4 as = make_float4(0.f, 0.f, 0.f, 0.f);                             // sigma head: features 4 lane .. 4 lane + 3
    float4 ar[3] = {as, as, as};                                             // rgb head: features 4 l5 .. of channel c, rows of parity hh
    float bs = 0.f, br[3] = {0.f, 0.f, 0.f};
    constexpr int U = 4;                                                     // row pairs in flight per wavefront
    for (long p0 = r0 + 2 * wave; p0 < r1; p0 += 2 * NWAVE * U) {
        float4 a0[U], a1[U], d4[U], h4[U];
        float s0[U], s1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = p0 + 2 * NWAVE * u;                               // wave-uniform
            const bool ok0 = r < r1, ok1 = r + 1 < r1;
            const long rr0 = ok0 ? r : r1 - 1, rr1 = ok1 ? r + 1 : r1 - 1;   // clamp: loads stay inside the row range, products are zeroed
            a0[u] = *reinterpret_cast<const float4 *>(a_last + rr0 * W + 4 * lane);
            a1[u] = *reinterpret_cast<const float4 *>(a_last + rr1 * W + 4 * lane);
            s0[u] = ok0 ? dheads[rr0 * 4 + 3] : 0.f;
            s1[u] = ok1 ? dheads[rr1 * 4 + 3] : 0.f;
            if (with_rgb) {
                const long rh = hh ? rr1 : rr0;
                d4[u] = *reinterpret_cast<const float4 *>(dact + rh * W2 + 4 * l5);
                h4[u] = *reinterpret_cast<const float4 *>(dheads + rh * 4);
                if (!(hh ? ok1 : ok0)) h4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            as.x = fmaf(s0[u], a0[u].x, as.x); as.y = fmaf(s0[u], a0[u].y, as.y); as.z = fmaf(s0[u], a0[u].z, as.z); as.w = fmaf(s0[u], a0[u].w, as.w);
            as.x = fmaf(s1[u], a1[u].x, as.x); as.y = fmaf(s1[u], a1[u].y, as.y); as.z = fmaf(s1[u], a1[u].z, as.z); as.w = fmaf(s1[u], a1[u].w, as.w);
            bs += s0[u] + s1[u];
            if (with_rgb) {
                const float hc[3] = {h4[u].x, h4[u].y, h4[u].z};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    ar[c].x = fmaf(hc[c], d4[u].x, ar[c].x); ar[c].y = fmaf(hc[c], d4[u].y, ar[c].y);
                    ar[c].z = fmaf(hc[c], d4[u].z, ar[c].z); ar[c].w = fmaf(hc[c], d4[u].w, ar[c].w);
                    br[c] += hc[c];
                }
            }
        }
    }
    float *mine = red[wave];
    *reinterpret_cast<float4 *>(mine + 4 * lane) = as;
    if (with_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                        // the two row parities of feature quad l5
            float4 v = ar[c];
            v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
            br[c] += __shfl_xor(br[c], 32);
            if (hh == 0) *reinterpret_cast<float4 *>(mine + 256 + c * 128 + 4 * l5) = v;
        }
    }
    if (lane == 0) {
        mine[256 + 384] = bs;
        mine[256 + 384 + 1] = with_rgb ? br[0] : 0.f; mine[256 + 384 + 2] = with_rgb ? br[1] : 0.f; mine[256 + 384 + 3] = with_rgb ? br[2] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SLOT; e += 256 * HG_GROUPS) {
        if (!with_rgb && e >= 256 && e != 256 + 384) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) v += red[w][e];
        if (e < 256) atomicAdd(d_sigma_w + e, v);
        else if (e < 256 + 384) atomicAdd(d_rgb_w + (e - 256), v);
        else if (e == 256 + 384) atomicAdd(d_sigma_b, v);
        else atomicAdd(d_rgb_b + (e - 256 - 384 - 1), v);
    }
}

template <class C>
static int launch_bwd(const BwdLayout &b, const ModelLayout &m, MlpBwdArgs &a, long n_rows_cap, hipStream_t stream) {
    constexpr int ROWS_D = cdiv(C::W + C::APP, 4 * C::TILE) * 4 * C::TILE;
    if (b.tile != C::TILE || b.layer[0].n_rows_pad != ROWS_D || b.layer[1].gpc != C::GPC || b.n_layers != C::NL + 1)
        return set_err(MNR_E_INVALID, "internal: backward kernel template / layout mismatch");
    a.sigma_off = m.sigma_off;
    a.rgb_off = m.rgb_off;
    const long nwg = (n_rows_cap + C::ROWS_PER_WG - 1) / C::ROWS_PER_WG;
    if (nwg <= 0) return MNR_OK;
    hipLaunchKernelGGL(k_mlp_bwd<C>, dim3((unsigned)nwg), dim3(256), 2 * CHUNK_BYTES, stream, a);
    return check_launch("k_mlp_bwd");
}

}  // namespace mnr

using namespace mnr;

extern "C" size_t mnr_packed_bwd_bytes(const mnr_model_desc *d) {
    BwdLayout b;
    if (bwd_layout_from_desc(d, b) != MNR_OK) return 0;
    return packed_bwd_bytes(b);
}

extern "C" int mnr_pack_model_bwd(void *packed_dev, size_t bytes, const mnr_model_desc *d, void *stream) {
    BwdLayout b;
    int rc = bwd_layout_from_desc(d, b);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(packed_dev && bytes >= packed_bwd_bytes(b), "backward packed buffer missing or too small");
    for (int i = 0; i < b.n_layers; ++i) MNR_REQUIRE(b.layer[i].w, "missing weight pointer for backward layer %d", i);
    const long total = (long)b.total_chunks * CHUNK_F4;
    hipLaunchKernelGGL(k_pack_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), b,
                       reinterpret_cast<float4 *>(packed_dev));
    return check_launch("k_pack_bwd");
}

static int check_grad_io(const mnr_model_desc *d, const mnr_mlp_grad_io *io) {
    MNR_REQUIRE(io, "NULL argument");
    MNR_REQUIRE(io->tape && io->gtape && io->dheads, "NULL tape / gradient pointer");
    MNR_REQUIRE(io->tape_row0 >= 0 && io->tape_rows >= io->tape_row0 + io->n_rows && io->rows_per_ray >= 1,
                "bad tape capacity / row offset / rows_per_ray");
    (void)d;
    return MNR_OK;
}

// host side of one data-gradient segment: argument block of the chain kernel
int mnr::fill_bwd_args(MlpBwdArgs &a, const ModelLayout &m, const void *packed_fwd_dev, const void *packed_bwd_dev,
                         const mnr_model_desc *d, const mnr_mlp_grad_io *io) {
    int rc = check_grad_io(d, io);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(packed_fwd_dev && packed_bwd_dev && io->d_out && io->out, "NULL argument");
    MNR_REQUIRE(d->appearance_dim == 0 || io->idx, "image indices required");
    a = MlpBwdArgs{};
    a.chunks = reinterpret_cast<const float4 *>(packed_bwd_dev);
    a.aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(packed_fwd_dev) + (size_t)m.total_chunks * CHUNK_BYTES);
    MNR_REQUIRE((long)io->tape_rows * d->layer_dim * 4 < (1ll << 32), "tape capacity: a plane must stay below 4 GiB (32-bit row offsets in the store addressing)");
    a.tape = io->tape; a.gtape = io->gtape; a.tape_rows = io->tape_rows; a.tl = tape_layout(arch_of(d));
    a.d_out = io->d_out; a.d_out_stride = io->d_out_stride; a.out = io->out; a.out_stride = io->out_stride;
    a.dheads = io->dheads; a.d_emb_a = io->grad.embedding_a;
    a.idx = io->idx; a.idx_stride = io->idx_stride; a.idx_is_float = io->idx_is_float;
    a.rows_per_ray = io->rows_per_ray; a.app_count = d->appearance_count; a.sigma_act = d->sigma_activation;
    a.n_rows = io->n_rows; a.n_units_dev = io->n_units_dev; a.rows_per_unit = io->rows_per_unit;
    a.tape_row0 = io->tape_row0;
    a.dd_in = io->dd_in;
    a.sigma_off = m.sigma_off;
    a.rgb_off = m.rgb_off;
    MNR_REQUIRE(d->rgb_dim == 3 || io->dd_in, "rgb_dim != 3: dd_in (gradient at the dir_a output) is required");
    return MNR_OK;
}

namespace mnr {
__global__ __launch_bounds__(256 * HG_GROUPS) void k_head_grads(const float *__restrict__ dheads, const float *__restrict__ a_last, int W,
                                                    const float *__restrict__ dact, int W2, long row0, long n_rows,
                                                    const int32_t *__restrict__ n_units_dev, int rows_per_unit,
                                                    float *__restrict__ d_sigma_w, float *__restrict__ d_sigma_b,
                                                    float *__restrict__ d_rgb_w, float *__restrict__ d_rgb_b, int with_rgb) {
    head_grads_body(dheads, a_last, W, dact, W2, row0, n_rows, n_units_dev, rows_per_unit, d_sigma_w, d_sigma_b, d_rgb_w, d_rgb_b, with_rgb,
                    (int)blockIdx.x, (int)gridDim.x);
}

// several (tape, row range) jobs in one launch: blockIdx.y = job, blocks past the job's own block count exit
struct HeadJobs { HeadJob job[HEAD_MAX_JOBS]; };
__global__ __launch_bounds__(256 * HG_GROUPS) void k_head_grads_jobs(HeadJobs js, int W) {
    const HeadJob &j = js.job[blockIdx.y];
    if ((int)blockIdx.x >= j.n_blocks) return;
    head_grads_body(j.dheads, j.a_last, W, j.dact, W / 2, j.row0, j.n_rows, j.n_units_dev, j.rows_per_unit, j.d_sigma_w, j.d_sigma_b,
                    j.d_rgb_w, j.d_rgb_b, j.d_rgb_w ? 1 : 0, (int)blockIdx.x, j.n_blocks);
}


// Spherical-harmonics colour head, backward (rendering.py:301-306: rgb = sigmoid(eval_sh(coef, dir)), coef = rgb layer of 3 x nb
// outputs, channel-major): per row  g_c = d_rgb_c * s_c (1 - s_c),  d_coef[c][k] = g_c * basis_k(dir)  and from there
//     dd[j]          = sum_ck d_coef[ck] * W_rgb[ck][j]      dL/d(dir_a output), handed to the data-gradient chain (MlpBwdArgs::dd_in)
//     dW_rgb[ck][j] += d_coef[ck] * a[j],  db_rgb[ck] += d_coef[ck]        (a = dir_a output row on the tape)
// One wavefront per row, lanes over the 128 features (two each): the lane's two columns of W_rgb and of the dW accumulator live in
// registers for the block's whole row range, d_coef is wave-uniform; a block's four wavefronts meet in LDS and add their sums with
// one set of atomics.  HBM: 1 KB per row (a in, dd out) -- ~0.2 GB per benchmark step; few long blocks like k_head_grads.
struct ShHeadJobs { ShHeadJob job[SH_HEAD_MAX_JOBS]; };
// CPP: colour channels per pass over the block's rows.  The per-lane state is 5 x CPP x NB registers (two weight columns, two
// weight-gradient columns, the bias sum): 3 x 9 coefficients fit in one pass, 3 x 16 (sh_deg 3) take one pass per channel -- the rows'
// inputs are read again (1 KB per row and pass) and the data gradient dd accumulates over the passes (each row is written by one lane pair).
template <int NB, int CPP>
__global__ __launch_bounds__(256) void k_sh_head_bwd(ShHeadJobs js) {
    constexpr int NC = CPP * NB, H2 = 128;
    const ShHeadJob &j = js.job[blockIdx.y];
    if ((int)blockIdx.x >= j.n_blocks) return;
    const long n = j.n_units_dev ? (long)(*j.n_units_dev) * j.rows_per_unit : j.n_rows;
    const long per = ((n + j.n_blocks - 1) / j.n_blocks + 15) / 16 * 16;
    const long rb = (long)blockIdx.x * per, re = min(n, rb + per);
    if (rb >= re) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float acc[4][H2];
    for (int c0 = 0; c0 < 3; c0 += CPP) {
        float w0[NC], w1[NC], a0[NC], a1[NC], bsum[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            w0[c] = j.rgb_w[(c0 * NB + c) * H2 + lane]; w1[c] = j.rgb_w[(c0 * NB + c) * H2 + 64 + lane];
            a0[c] = 0.f; a1[c] = 0.f; bsum[c] = 0.f;
        }
        // U rows per wavefront and iteration: all their loads are requested before the first product (the loop is latency-bound otherwise:
        // one row at a time measured 0.30 ms per benchmark step)
        constexpr int U = 4;
        for (long r0 = rb + wave * U; r0 < re; r0 += 4 * U) {
            float4 go[U], o[U];
            float x0[U], x1[U], dx[U], dy[U], dz[U], p0[U], p1[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = r0 + u < re;
                const long r = ok[u] ? r0 + u : re - 1, ro = j.out_row0 + r, rt = j.tape_row0 + r;
                go[u] = *reinterpret_cast<const float4 *>(j.d_out + ro * 4);
                o[u] = *reinterpret_cast<const float4 *>(j.out + ro * 4);
                const float *dv = j.dirs + (ro / j.rows_per_ray) * j.dir_stride;
                dx[u] = dv[0]; dy[u] = dv[1]; dz[u] = dv[2];
                x0[u] = j.dact[rt * H2 + lane]; x1[u] = j.dact[rt * H2 + 64 + lane];
                p0[u] = 0.f; p1[u] = 0.f;
                if (CPP < 3 && c0 > 0) { p0[u] = j.dd[ro * H2 + lane]; p1[u] = j.dd[ro * H2 + 64 + lane]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float b[25];
                sh_basis(j.sh_deg, dx[u], dy[u], dz[u], b);
                const float m = ok[u] ? 1.f : 0.f;
                const float g[3] = {m * go[u].x * (o[u].x * (1.f - o[u].x)), m * go[u].y * (o[u].y * (1.f - o[u].y)), m * go[u].z * (o[u].z * (1.f - o[u].z))};
                float d0 = p0[u], d1 = p1[u];
#pragma unroll
                for (int c = 0; c < CPP; ++c) {
                    const float gc = CPP == 3 ? g[c] : (c0 == 0 ? g[0] : (c0 == 1 ? g[1] : g[2]));
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const float dc = gc * b[k];
                        d0 = fmaf(dc, w0[c * NB + k], d0); d1 = fmaf(dc, w1[c * NB + k], d1);
                        a0[c * NB + k] = fmaf(dc, x0[u], a0[c * NB + k]); a1[c * NB + k] = fmaf(dc, x1[u], a1[c * NB + k]);
                        bsum[c * NB + k] += dc;
                    }
                }
                if (ok[u]) {
                    const long ro = j.out_row0 + r0 + u;
                    j.dd[ro * H2 + lane] = d0; j.dd[ro * H2 + 64 + lane] = d1;
                }
            }
        }
        // combine the four wavefronts: NC x 128 sums in passes of one coefficient row (128 floats per wavefront) to stay inside 64 KB of LDS
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[wave][lane] = a0[c]; acc[wave][64 + lane] = a1[c];
            __syncthreads();
            if (threadIdx.x < H2)
                atomicAdd(j.d_rgb_w + (c0 * NB + c) * H2 + threadIdx.x, acc[0][threadIdx.x] + acc[1][threadIdx.x] + acc[2][threadIdx.x] + acc[3][threadIdx.x]);
            __syncthreads();
        }
        // bias sums are wave-uniform: lane c of every wavefront carries coefficient c
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) mine = lane == c ? bsum[c] : mine;
        if (lane < NC) atomicAdd(j.d_rgb_b + c0 * NB + lane, mine);
    }
}

}  // namespace mnr

int mnr::head_job_of(const mnr_model_desc *d, const mnr_mlp_grad_io *io, HeadJob &job) {
    const mnr_model_grads &G = io->grad;
    MNR_REQUIRE(G.sigma_w && G.sigma_b && G.rgb_w && G.rgb_b, "missing head gradient pointers");
    MNR_REQUIRE(d->layer_dim == 256 && d->rgb_dim == 3, "head-gradient job kernel: layer_dim 256, rgb_dim 3");
    const TapeLayout tl = tape_layout(arch_of(d));
    const long cap = io->tape_rows;
    const long blocks = io->n_units_dev ? 48 : (io->n_rows + 767) / 768;
    job = HeadJob{io->dheads, io->tape + (long)tl.act_off[d->layers - 1] * cap, io->tape + (long)tl.dact_off * cap, (long)io->tape_row0,
                  (long)io->n_rows, io->n_units_dev, io->rows_per_unit, (int)(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks)), G.sigma_w,
                  G.sigma_b, G.rgb_w, G.rgb_b};
    return MNR_OK;
}

int mnr::head_grads_jobs(const HeadJob *jobs, int n_jobs, int W, hipStream_t s) {
    MNR_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= HEAD_MAX_JOBS && W == 256, "1..%d head-gradient jobs per launch (layer_dim 256)", HEAD_MAX_JOBS);
    HeadJobs js{};
    int max_blocks = 1;
    for (int i = 0; i < n_jobs; ++i) { js.job[i] = jobs[i]; max_blocks = jobs[i].n_blocks > max_blocks ? jobs[i].n_blocks : max_blocks; }
    hipLaunchKernelGGL(k_head_grads_jobs, dim3((unsigned)max_blocks, (unsigned)n_jobs), dim3(256 * HG_GROUPS), 0, s, js, W);
    return check_launch("k_head_grads_jobs");
}

int mnr::sh_head_bwd_jobs(const ShHeadJob *jobs, int n_jobs, hipStream_t s) {
    MNR_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= SH_HEAD_MAX_JOBS, "1..%d colour-head jobs per launch", SH_HEAD_MAX_JOBS);
    ShHeadJobs js{};
    int max_blocks = 1;
    for (int i = 0; i < n_jobs; ++i) {
        MNR_REQUIRE(jobs[i].sh_deg == jobs[0].sh_deg && jobs[i].n_blocks >= 1, "colour-head jobs of one launch share the SH degree");
        js.job[i] = jobs[i];
        max_blocks = jobs[i].n_blocks > max_blocks ? jobs[i].n_blocks : max_blocks;
    }
    const dim3 grid((unsigned)max_blocks, (unsigned)n_jobs);
    switch (jobs[0].sh_deg) {
        case 2: hipLaunchKernelGGL((k_sh_head_bwd<9, 3>), grid, dim3(256), 0, s, js); break;
        case 3: hipLaunchKernelGGL((k_sh_head_bwd<16, 1>), grid, dim3(256), 0, s, js); break;
        default: return set_err(MNR_E_UNSUPPORTED, "the fused colour-head adjoint is instantiated for sh_deg 2 (configs/mega-nerf-sh-3) and 3");
    }
    return check_launch("k_sh_head_bwd");
}

// sigma / rgb head weight gradients of the rows of one segment (dheads was just written by the chain kernel)
static int launch_head_grads(const mnr_model_desc *d, const mnr_mlp_grad_io *io, hipStream_t s) {
    if (io->n_rows == 0) return MNR_OK;
    const mnr_model_grads &G = io->grad;
    MNR_REQUIRE(G.sigma_w && G.sigma_b && G.rgb_w && G.rgb_b, "missing head gradient pointers");
    const TapeLayout tl = tape_layout(arch_of(d));
    const long cap = io->tape_rows;
    const int W = d->layer_dim;
    MNR_REQUIRE(W == 256, "head-gradient kernel is written for layer_dim 256 (one thread per sigma-head feature)");
    // every block ends with 644 atomics on the same addresses: few, long blocks (round 1 launched 1024 per segment and spent
    // 60-125 us per launch mostly there)
    const long blocks = io->n_units_dev ? 48 : (io->n_rows + 767) / 768;
    hipLaunchKernelGGL(k_head_grads, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks))), dim3(256 * HG_GROUPS), 0, s, io->dheads, io->tape + (long)tl.act_off[d->layers - 1] * cap, W,
                       io->tape + (long)tl.dact_off * cap, W / 2, (long)io->tape_row0, (long)io->n_rows, io->n_units_dev,
                       io->rows_per_unit, G.sigma_w, G.sigma_b, G.rgb_w, G.rgb_b, d->rgb_dim == 3 ? 1 : 0);
    return check_launch("k_head_grads");
}

extern "C" int mnr_mlp_backward_data(const void *packed_fwd_dev, const void *packed_bwd_dev, const mnr_model_desc *d,
                                     const mnr_mlp_grad_io *io, void *stream) {
    ModelLayout m;
    BwdLayout b;
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    rc = bwd_layout_from_desc(d, b);
    if (rc != MNR_OK) return rc;
    MlpBwdArgs a;
    rc = fill_bwd_args(a, m, packed_fwd_dev, packed_bwd_dev, d, io);
    if (rc != MNR_OK) return rc;
    hipStream_t s = as_stream(stream);
    rc = MNR_E_UNSUPPORTED;
#define MNR_TRY_B(XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL)                                                      \
    if (d->xyz_dim == XYZ && d->pos_xyz_dim == LX && d->pos_dir_dim == LD && d->appearance_dim == APP &&        \
        d->layer_dim == W && d->layers == NL && d->skip_mask == SKIP && d->rgb_dim == RGB && m.tile == TL)      \
        rc = launch_bwd<MlpCfg<XYZ, LX, LD, APP, W, NL, SKIP, RGB, TL>>(b, m, a, io->n_rows, s);
    MNR_TRY_B(3, 12, 4, 48, 256, 8, 16, 3, 16)
    MNR_TRY_B(4, 12, 4, 48, 256, 8, 16, 3, 16)
#ifdef MNR_ALL_VARIANTS
    MNR_TRY_B(3, 12, 4, 0, 256, 8, 16, 3, 16)         // configs/mega-nerf-no-embed
    MNR_TRY_B(4, 12, 4, 0, 256, 8, 16, 3, 16)
    MNR_TRY_B(3, 12, 0, 48, 256, 8, 16, 27, 16)       // configs/mega-nerf-sh-3
    MNR_TRY_B(4, 12, 0, 48, 256, 8, 16, 27, 16)
    MNR_TRY_B(3, 12, 0, 48, 256, 8, 16, 48, 16)       // sh_deg 3
    MNR_TRY_B(4, 12, 0, 48, 256, 8, 16, 48, 16)
#endif
#undef MNR_TRY_B
    if (rc == MNR_E_UNSUPPORTED) return set_err(rc, "no backward kernel for this architecture (training supports the "
                                                   "default 8x256 fg/bg models)");
    if (rc != MNR_OK) return rc;
    return launch_head_grads(d, io, s);
}

// Data-gradient chains of several segments (coarse + fine rows of the foreground and background models) in ONE launch,
// then the head gradients of every segment.  Default 8x256 fg / bg architectures only (MNR_E_UNSUPPORTED otherwise).
template <class CfgFG, class CfgBG>
static int mlp_backward_chain_multi_pair(const mnr_mlp_grad_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MlpBwdMulti mm{};
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_grad_launch &L = segs[i];
        const mnr_model_desc *d = L.desc;
        ModelLayout m;
        BwdLayout b;
        int rc = layout_from_desc(d, m);
        if (rc != MNR_OK) return rc;
        rc = bwd_layout_from_desc(d, b);
        if (rc != MNR_OK) return rc;
        rc = fill_bwd_args(mm.seg[i], m, L.packed_fwd_dev, L.packed_bwd_dev, d, L.io);
        if (rc != MNR_OK) return rc;
        if (cells && cells[i].dcells) {
            MNR_REQUIRE(cells[i].cell_rows > 0 && cells[i].cell_rows % CfgFG::ROWS_PER_WG == 0 && L.io->n_rows % cells[i].cell_rows == 0,
                        "segment %d: rows per cell must be a multiple of %d", i, CfgFG::ROWS_PER_WG);
            mm.seg[i].dcells = cells[i].dcells;
            mm.seg[i].cell_rows = cells[i].cell_rows;
            mm.seg[i].aux_byte_off = (long)m.total_chunks * CHUNK_BYTES;
        }
        mm.is_b[i] = d->xyz_dim == 4 ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        if (cells) {                       // grid = (workgroups per cell, cells)
            MNR_REQUIRE(cells[i].dcells && L.io->n_rows / cells[i].cell_rows == segs[0].io->n_rows / cells[0].cell_rows,
                        "multi-cell launch: every segment needs a cell table over the same number of cells");
            wg += cells[i].cell_rows / CfgFG::ROWS_PER_WG;
        } else
        wg += (L.io->n_rows + CfgFG::ROWS_PER_WG - 1) / CfgFG::ROWS_PER_WG;
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one launch");
    }
    for (int i = n_segs; i <= MLP_BWD_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    if (wg == 0) return MNR_OK;
    const unsigned ny = cells ? (unsigned)(segs[0].io->n_rows / cells[0].cell_rows) : 1u;
    hipLaunchKernelGGL((k_mlp_bwd_multi<CfgFG, CfgBG>), dim3((unsigned)wg, ny), dim3(256), 2 * CHUNK_BYTES, s, mm);
    return check_launch("k_mlp_bwd_multi");
}

// Data-gradient chains of several segments (coarse + fine rows of the foreground and background models) in ONE launch.
// The default 8x256 fg / bg architectures, and their spherical-harmonics form (sh_deg 2: rgb_dim 27, no direction encoding; the
// gradient at the dir_a output comes in through mnr_mlp_grad_io::dd_in); MNR_E_UNSUPPORTED otherwise.
int mnr::mlp_backward_chain_multi_impl(const mnr_mlp_grad_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= MLP_BWD_MAX_SEGS, "1..%d segments per launch", MLP_BWD_MAX_SEGS);
    int pair = -1;
    for (int i = 0; i < n_segs; ++i) {
        MNR_REQUIRE(segs[i].desc && segs[i].io, "segment %d: NULL argument", i);
        const mnr_model_desc *d = segs[i].desc;
        const bool trunk = (d->xyz_dim == 3 || d->xyz_dim == 4) && d->pos_xyz_dim == 12 && d->appearance_dim == 48 && d->layer_dim == 256 &&
                           d->layers == 8 && d->skip_mask == 16 && (d->mfma_tile == 0 || d->mfma_tile == 16);
        const int p = !trunk ? 0 : (d->pos_dir_dim == 4 && d->rgb_dim == 3 ? 1 : (d->pos_dir_dim == 0 && d->rgb_dim == 27 ? 2 : (d->pos_dir_dim == 0 && d->rgb_dim == 48 ? 3 : 0)));
        if (p == 0 || (pair >= 0 && p != pair))
            return set_err(MNR_E_UNSUPPORTED, "mnr_mlp_backward_data_multi covers the default 8x256 fg / bg models and their spherical-harmonics (sh_deg 2 / 3) forms");
        pair = p;
    }
    if (pair == 1)
        return mlp_backward_chain_multi_pair<MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>, MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>>(segs, n_segs, cells, s);
#ifdef MNR_ALL_VARIANTS
    if (pair == 3)
        return mlp_backward_chain_multi_pair<MlpCfg<3, 12, 0, 48, 256, 8, 16, 48, 16>, MlpCfg<4, 12, 0, 48, 256, 8, 16, 48, 16>>(segs, n_segs, cells, s);
    return mlp_backward_chain_multi_pair<MlpCfg<3, 12, 0, 48, 256, 8, 16, 27, 16>, MlpCfg<4, 12, 0, 48, 256, 8, 16, 27, 16>>(segs, n_segs, cells, s);
#else
    return set_err(MNR_E_UNSUPPORTED, "built without MNR_ALL_VARIANTS: no spherical-harmonics multi-segment kernels");
#endif
}

extern "C" int mnr_mlp_backward_chain_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream) {
    return mlp_backward_chain_multi_impl(segs, n_segs, nullptr, as_stream(stream));
}

extern "C" int mnr_mlp_head_grads_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= MLP_BWD_MAX_SEGS, "1..%d segments per launch", MLP_BWD_MAX_SEGS);
    hipStream_t s = as_stream(stream);
    int rc = MNR_OK;
    // segments that continue each other in one tape (coarse + fine rows of the foreground) go out as one launch
    for (int i = 0; i < n_segs && rc == MNR_OK; ++i) {
        MNR_REQUIRE(segs[i].desc && segs[i].io, "segment %d: NULL argument", i);
        mnr_mlp_grad_io io = *segs[i].io;
        while (i + 1 < n_segs && !io.n_units_dev && !segs[i + 1].io->n_units_dev && segs[i + 1].io->tape == io.tape &&
               segs[i + 1].io->dheads == io.dheads && segs[i + 1].io->tape_row0 == io.tape_row0 + io.n_rows &&
               segs[i + 1].io->grad.sigma_w == io.grad.sigma_w) {
            io.n_rows += segs[i + 1].io->n_rows;
            ++i;
        }
        rc = launch_head_grads(segs[i].desc, &io, s);
    }
    return rc;
}

extern "C" int mnr_mlp_backward_data_multi(const mnr_mlp_grad_launch *segs, int n_segs, void *stream) {
    const int rc = mnr_mlp_backward_chain_multi(segs, n_segs, stream);
    return rc != MNR_OK ? rc : mnr_mlp_head_grads_multi(segs, n_segs, stream);
}

extern "C" int mnr_mlp_backward_weights(const mnr_model_desc *d, const mnr_mlp_grad_io *io, void *stream) {
    ModelLayout m;
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    rc = check_grad_io(d, io);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(m.has_final, "training needs a model with the dir/appearance branch");
    hipStream_t s = as_stream(stream);
    const TapeLayout tl = tape_layout(arch_of(d));
    // weight gradients: one launch over a job table
    const mnr_model_grads &G = io->grad;
    const int W = d->layer_dim, L = d->layers;
    const int Ecols = emb_cols(d->xyz_dim, d->pos_xyz_dim), EDcols = emb_cols(3, d->pos_dir_dim);
    const long cap = io->tape_rows;
    WgradArgs wa{};
    int nj = 0;
    auto add = [&](const float *dz, int M, const float *in, int ldin, int N, float *dw, int ldw, int col0, float *db) {
        WgradJob &J = wa.job[nj++];
        J.dz = dz; J.ldz = M; J.M = M; J.in = in; J.ldin = ldin; J.N = N; J.dw = dw; J.ldw = ldw; J.col0 = col0; J.db = db;
    };
    for (int l = 0; l < L; ++l) {
        MNR_REQUIRE(G.layer_w[l] && G.layer_b[l], "missing gradient pointer for layer %d", l);
        const float *dz = io->gtape + (long)tl.act_off[l] * cap;
        const bool skip = (d->skip_mask >> l) & 1;
        if (l == 0) {
            add(dz, W, io->tape + (long)tl.embx_off * cap, tl.embx_w, Ecols, G.layer_w[l], Ecols, 0, G.layer_b[l]);
        } else {
            const int ldw = skip ? Ecols + W : W;
            if (skip) add(dz, W, io->tape + (long)tl.embx_off * cap, tl.embx_w, Ecols, G.layer_w[l], ldw, 0, nullptr);
            add(dz, W, io->tape + (long)tl.act_off[l - 1] * cap, W, W, G.layer_w[l], ldw, skip ? Ecols : 0, G.layer_b[l]);
        }
    }
    MNR_REQUIRE(G.final_w && G.final_b && G.dir_a_w && G.dir_a_b && G.sigma_w && G.sigma_b && G.rgb_w && G.rgb_b,
                "missing head / final gradient pointers");
    add(io->gtape + (long)tl.fin_off * cap, W, io->tape + (long)tl.act_off[L - 1] * cap, W, W, G.final_w, W, 0, G.final_b);
    {
        const float *dz = io->gtape + (long)tl.dact_off * cap;
        const int ldw = W + EDcols + d->appearance_dim;
        add(dz, W / 2, io->tape + (long)tl.fin_off * cap, W, W, G.dir_a_w, ldw, 0, G.dir_a_b);
        if (EDcols) add(dz, W / 2, io->tape + (long)tl.embd_off * cap, tl.embd_w, EDcols, G.dir_a_w, ldw, W, nullptr);
        if (d->appearance_dim) add(dz, W / 2, io->tape + (long)tl.app_off * cap, tl.app_w, d->appearance_dim, G.dir_a_w, ldw,
                                   W + EDcols, nullptr);
    }
    MNR_REQUIRE(W == 256, "weight-gradient kernel supports layer_dim 256");
    // distribute ~256 workgroups (one per CU) over the jobs in proportion to M * N (cost per row)
    // Per-tile cost model (cycles): MFMA time of one wave vs LDS-DMA fill time of the tile, plus a fixed
    // barrier/latency term -- small-N jobs are fill/latency bound, not MFMA bound (calibrated on MI355X, round 1).
    // Work items: every job is cut into row ranges of roughly equal cost (cycles per tile = MFMA time of one wave vs
    // LDS-DMA fill time, plus a fixed barrier/latency term); ~6 items per CU keep the tail short.
    auto env_d = [](const char *k, double d) { const char *v = getenv(k); return v ? atof(v) : d; };
    const double fill_bpc = env_d("MNR_WGRAD_FILL_BPC", 6.0), fixed = env_d("MNR_WGRAD_FIXED", 2500.0);
    const int budget = (int)env_d("MNR_WGRAD_ITEMS", 1024.0);     // tools/sweep_wgrad.py: 768-1024 is the (flat) optimum
    double cost[WGRAD_MAX_JOBS], tot = 0;
    for (int i = 0; i < nj; ++i) {
        const WgradJob &J = wa.job[i];
        const int mbw = J.M / 64, nbw = ((J.N + 31) / 32 + 3) / 4;
        const double mfma = 2 * 16.0 * 64.0 * mbw * nbw, fill = (J.M + J.ldin) * WG_KT * 4.0 / fill_bpc;
        cost[i] = (mfma > fill ? mfma : fill) + fixed;
        tot += cost[i];
    }
    int wg = 0;
    const long tiles = (io->n_rows + WG_KT - 1) / WG_KT;
    size_t lds = 0;
    for (int i = 0; i < nj; ++i) {
        WgradJob &J = wa.job[i];
        int n = (int)(budget * cost[i] / tot + 0.5);
        n = n < 1 ? 1 : n;
        if (n > tiles) n = (int)(tiles < 1 ? 1 : tiles);
        J.wg0 = wg; J.nwg = n; wg += n;                    // wg0 / nwg = first item / item count of the job
        const size_t need = (2 * (size_t)((WG_KT * (J.M + J.ldin) + 255) / 256 * 256 + 256) + 64) * sizeof(float);
        lds = need > lds ? need : lds;
    }
    wa.njobs = nj;
    wa.n_rows = io->n_rows; wa.n_units_dev = io->n_units_dev; wa.rows_per_unit = io->rows_per_unit;
    wa.row0 = io->tape_row0;
    MNR_REQUIRE(io->work_counter, "work_counter (device int32) required");
    wa.work_counter = io->work_counter;
    if (io->n_rows > 0) {
        static size_t lds_enabled_dev[MAX_DEVICES] = {};       // raise the dynamic-LDS cap once per device (monotonic; benign if raced)
        size_t &lds_enabled = lds_enabled_dev[device_slot()];
        if (lds > 64 * 1024 && lds > lds_enabled) {
            for (const void *fn : {reinterpret_cast<const void *>(k_wgrad<true>), reinterpret_cast<const void *>(k_wgrad<false>)}) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_wgrad): %s", hipGetErrorString(e));
            }
            lds_enabled = 160 * 1024;
        }
        if (hipMemsetAsync(io->work_counter, 0, sizeof(int32_t), s) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipMemsetAsync(work_counter)");
        if (io->n_units_dev) hipLaunchKernelGGL(k_wgrad<false>, dim3(wg < 256 ? wg : 256), dim3(WG_THREADS), lds, s, wa);
        else hipLaunchKernelGGL(k_wgrad<true>, dim3(wg < 256 ? wg : 256), dim3(WG_THREADS), lds, s, wa);
        rc = check_launch("k_wgrad");
        if (rc) return rc;
    }
    return rc;
}
