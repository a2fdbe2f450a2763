// mlp_bwd_h2.hip -- the data-gradient chain of the fused NeRF MLP on the 16-bit matrix pipe with split-precision operands: the
// mirror of mlp_fwd_h2.hip (design notes there and in mlp_bwd.hip).  dZ_l in C-layout accumulator registers is re-split into
// f16 (hi, lo) halves K-step by K-step and multiplied with the (hi, lo) image of W_l^T -- three v_mfma_f32_16x16x32_f16 products,
// fp32 accumulation -- masked by the sign bits the forward left on the tape; every dZ_l goes once to the gradient tape (fp32), so the
// fp32 weight-gradient kernel (csrc/wgrad.hip) consumes the tapes unchanged.  8 waves per workgroup, 128 rows per pass over the image.
#include "h2_device.h"
#include "mlp_bwd_device.h"
#include "step_internal.h"

namespace mnr {

__global__ void k_pack_bwd_h2(BwdLayout b, uint4v *__restrict__ chunks) {
    pack_bwd_h2_thread(b, chunks, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// dZ of a layer to the gradient tape, scaled back by the row's power of two (exact): gtape_store_scaled_part below

// publish a plane's row exponent (E + ZEXP_BIAS, 0 for rows without data): largest over the wavefront, one atomic max per wavefront
// and plane, skipped when the published value is already as large (after the first workgroups: nearly always)
__device__ __forceinline__ void zexp_publish(int32_t *zexp, int plane, int E, bool has_data) {
    if (!zexp) return;
    int v = has_data ? E + ZEXP_BIAS : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = max(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0 && v > __hip_atomic_load(zexp + plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(zexp + plane, v);
}

template <int NH>
__device__ __forceinline__ bool h2_renorm(float (&g)[NH], int &E, float &scale_dn) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NH; ++i) mx = fmaxf(mx, fabsf(g[i]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    int e = 0;
    const bool usable = mx > 0.f && mx < 3.0e38f;
    if (usable) (void)frexpf(mx, &e);
    e = E + e < -100 ? -100 - E : e;
    E += e;
    const float up = ldexpf(1.f, -e);
#pragma unroll
    for (int i = 0; i < NH; ++i) g[i] *= up;
    scale_dn = ldexpf(1.f, E);
    return usable;
}

// ... pieces Q0 .. Q0 + NQ - 1 only (a plane's stores are spread over the chunk periods of the product that consumes the registers);
// uniform plane + 32-bit row offset in bytes (mlp_device.h gstore4)
template <int P, int Q0, int NQ, int NH>
__device__ __forceinline__ void gtape_store_scaled_part(const float (&g)[NH], float inv, const float *plane, unsigned row_byte_off, bool valid) {
    static_assert(4 * (Q0 + NQ) <= NH, "piece range");
#ifdef H2_EXPERIMENT_NO_TAPE
    if (NH == 64) return;             // timing experiment only
#endif
    if (!valid) return;
    static_for<Q0, Q0 + NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        gstore4<16 * P * q>(plane, row_byte_off, make_float4(g[4 * q] * inv, g[4 * q + 1] * inv, g[4 * q + 2] * inv, g[4 * q + 3] * inv));
    });
}

template <class C>
__device__ __forceinline__ void mlp_bwd_h2_body(const MlpBwdArgs &a, long blk, int cidx) {
    static_assert(C::TILE == 16 && C::W == 256 && C::HAS_FINAL && C::RGB == 3, "split-precision chain: default 8x256 architectures");
    constexpr int P = C::P, H = C::H, NOB = C::NOB, RPB = C::RPB, H2 = C::H2, W = C::W;
    constexpr int ROWS_D = cdiv(W + C::APP, 4 * 16) * 4 * 16, NOBD = ROWS_D / 16;       // dir_a^T: W final rows + APP appearance rows
    extern __shared__ uint4v h2_ring[];

    long n_rows, row_base = 0, tape_row0 = a.tape_row0;
    const uint4v *chunks = reinterpret_cast<const uint4v *>(a.chunks);
    const float *aux = a.aux;
    float *d_emb_a = a.d_emb_a;
    int32_t *zexp = nullptr;
    if (a.dcells) {
        const MlpCellSeg cell = a.dcells[cidx];
        n_rows = cell.n_units ? (long)__builtin_amdgcn_readfirstlane(*cell.n_units) * a.rows_per_unit : a.cell_rows;
        if (blk * H2_ROWS >= n_rows) return;
        chunks = reinterpret_cast<const uint4v *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed_bwd)));
        aux = reinterpret_cast<const float *>(uniform_ptr(reinterpret_cast<const char *>(cell.packed) + a.aux_byte_off));
        d_emb_a = reinterpret_cast<float *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(cell.d_emb_a))));
        row_base = (long)cidx * a.cell_rows;
        tape_row0 = uniform_long(cell.tape_row0);
        zexp = reinterpret_cast<int32_t *>(const_cast<char *>(uniform_ptr(reinterpret_cast<const char *>(cell.zexp))));
    } else {
        n_rows = a.n_units_dev ? (long)(*a.n_units_dev) * a.rows_per_unit : a.n_rows;
        if (blk * H2_ROWS >= n_rows) return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const long lrow = (blk * H2_WAVES + wave) * 16 + (lane & 15);
    const bool valid = lrow < n_rows;
    const long lrc = valid ? lrow : n_rows - 1;
    const long rc = row_base + lrc;
    const long cap = a.tape_rows;
    const long trow = lrc + tape_row0;

    H2Stream st;
    st.init(chunks, h2_ring);

    // ---- output activations backward (as mlp_bwd_body) ----
    float dr[3], ds;
    {
        const float *go = a.d_out + rc * a.d_out_stride, *o = a.out + rc * a.out_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) dr[c] = valid ? go[c] * o[c] * (1.f - o[c]) : 0.f;
        const float sg = o[3];
        const float da = a.sigma_act ? (1.f - expf(-sg)) : (sg > 0.f ? 1.f : 0.f);
        ds = valid ? go[3] * da : 0.f;
        if (valid && part == 0) *reinterpret_cast<float4 *>(a.dheads + trow * 4) = make_float4(dr[0], dr[1], dr[2], ds);
    }
    // ---- rgb head backward -> dZ of dir_a (fp32 VALU) ----
    float dd[H2];
    {
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
        const MaskBits<H2> dm = mask_load<H2>(a.tape + a.tl.dmask_off * cap, trow, a.tl.dmask_w, part);
        mask_apply(dd, dm);
    }
    gtape_store<P>(dd, a.gtape + a.tl.dact_off * cap, (unsigned)((trow * (W / 2) + 4 * part) * 4), valid);
    // ---- per-row power-of-two scale --------------------------------------------------------------------------------------------
    // Gradients are small (1e-6 .. 1e-12 is ordinary) and f16 ends at 6e-8: the chain is LINEAR per row once the masks are fixed,
    // so every row is scaled by 2^-e (e = exponent of its largest |dZ| entering the chain: max scaled value in [0.5, 1), 16 binades
    // of headroom above, absolute precision 2^-24 of the row's maximum below) and scaled back -- exactly -- wherever a value leaves
    // the chain (gradient tape, embedding gradient).
    float scale_up, scale_dn;
    int E;
    {
        float mx = fabsf(ds);
#pragma unroll
        for (int i = 0; i < H2; ++i) mx = fmaxf(mx, fabsf(dd[i]));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        int e = 0;
        const bool usable = mx > 0.f && mx < 3.0e38f;
        if (usable) (void)frexpf(mx, &e);
        e = e < -100 ? -100 : e;              // rows deep behind a surface carry gradients of 1e-35 and less: 2^-e must stay finite
        E = usable ? e : 0;
        scale_up = ldexpf(1.f, -E);
        scale_dn = ldexpf(1.f, E);
        zexp_publish(zexp, C::NL + 1, E, usable && valid);
#pragma unroll
        for (int i = 0; i < H2; ++i) dd[i] *= scale_up;
    }
    // ---- dir_a^T: d(final features) and d(appearance embedding) ----
    float g[H];
    {
        floatx4 accd[NOBD];
        zero_acc(accd);
        h2_segment<NOBD, H2 / 8, 0>(accd, dd, st, lane);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < RPB; ++r) g[ob * RPB + r] = accd[ob][r];
        if (d_emb_a) {
            constexpr int NAB = cdiv(C::APP, 16);
            const long ray = rc / a.rows_per_ray;
            long idx = a.idx_is_float ? (long)reinterpret_cast<const float *>(a.idx)[ray * a.idx_stride]
                                      : (long)reinterpret_cast<const int32_t *>(a.idx)[ray * a.idx_stride];
            idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
            const bool uniform = (a.rows_per_ray % 16) == 0;
#pragma unroll
            for (int b = 0; b < NAB; ++b)
#pragma unroll
                for (int r = 0; r < RPB; ++r) {
                    float v = accd[NOB + b][r] * scale_dn;
                    const int col = b * 16 + 4 * part + r;
                    if (uniform) {
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
                        if ((lane & 15) == 0 && col < C::APP) atomicAdd(d_emb_a + idx * C::APP + col, v);
                    } else if (valid && col < C::APP) {
                        atomicAdd(d_emb_a + idx * C::APP + col, v);
                    }
                }
        }
    }
    // ---- final^T (+ sigma head): dZ of trunk layer L-1 ----
    { const bool live = h2_renorm(g, E, scale_dn); zexp_publish(zexp, C::NL, E, live && valid); }
    const float ds_s = ds * ldexpf(1.f, -E);          // fp32 accumulator initialiser: may exceed 1, never touches f16
    // dZ planes go to the gradient tape right behind the first chunk boundary of the product that consumes the registers (a boundary
    // drains vmcnt: stores issued just before one cost a write round trip)
    const unsigned grow_off = (unsigned)((trow * W + 4 * part) * 4);        // this lane's row in a W-wide plane, bytes (< 2^32: checked by the host)
    auto store_g = [&](const float *plane) {
        return [&, plane](auto cc) {
            if constexpr (decltype(cc)::value == 0) gtape_store_scaled_part<P, 0, 16>(g, scale_dn, plane, grow_off, valid);
        };
    };
    floatx4 acc[NOB];
    {
        const float *ws = aux + a.sigma_off + part * H;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const float4 w4 = *reinterpret_cast<const float4 *>(ws + ob * RPB);
            acc[ob][0] = ds_s * w4.x; acc[ob][1] = ds_s * w4.y; acc[ob][2] = ds_s * w4.z; acc[ob][3] = ds_s * w4.w;
        }
        const MaskBits<H> bits = mask_load<H>(a.tape + a.tl.mask_off[C::NL - 1] * cap, trow, a.tl.mask_w, part);
        h2_segment<NOB, H / 8, 0>(acc, g, st, lane, store_g(a.gtape + a.tl.fin_off * cap));
        acc_to_regs<NOB, RPB, false>(g, acc);
        mask_apply(g, bits);
        { const bool live = h2_renorm(g, E, scale_dn); zexp_publish(zexp, C::NL - 1, E, live && valid); }
    }
    // ---- trunk layers L-1 .. 1 transposed ----
    static_for<0, C::NL - 1>([&](auto jc) {
        constexpr int l = C::NL - 1 - decltype(jc)::value;
        zero_acc(acc);
        const MaskBits<H> bits = mask_load<H>(a.tape + a.tl.mask_off[l - 1] * cap, trow, a.tl.mask_w, part);
        h2_segment<NOB, H / 8, 0>(acc, g, st, lane, store_g(a.gtape + a.tl.act_off[l] * cap));
        acc_to_regs<NOB, RPB, false>(g, acc);
        mask_apply(g, bits);
        { const bool live = h2_renorm(g, E, scale_dn); zexp_publish(zexp, l - 1, E, live && valid); }
    });
    gtape_store_scaled_part<P, 0, 16>(g, scale_dn, a.gtape + a.tl.act_off[0] * cap, grow_off, valid);
}

constexpr int H2B_MAX_SEGS = 4;
struct H2BwdMulti {
    MlpBwdArgs seg[H2B_MAX_SEGS];
    int32_t wg0[H2B_MAX_SEGS + 1];
    int32_t is_b[H2B_MAX_SEGS];
};
template <class CA, class CB>
__global__ __launch_bounds__(H2_THREADS, 1) void k_mlp_bwd_h2(H2BwdMulti m) {
    const int blk = blockIdx.x;
    const int s = (blk >= m.wg0[1]) + (blk >= m.wg0[2]) + (blk >= m.wg0[3]);
    if (m.is_b[s]) mlp_bwd_h2_body<CB>(m.seg[s], blk - m.wg0[s], blockIdx.y);
    else mlp_bwd_h2_body<CA>(m.seg[s], blk - m.wg0[s], blockIdx.y);
}

}  // namespace mnr

using namespace mnr;
using H2FG = MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>;
using H2BG = MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>;

extern "C" size_t mnr_packed_bwd_h2_bytes(const mnr_model_desc *d) {
    ModelLayout m;
    BwdLayout b;
    if (h2_layout(d, m) != MNR_OK || bwd_layout_from_desc(d, b) != MNR_OK) return 0;
    return (size_t)h2b_total_chunks(b) * H2_CHUNK_BYTES;
}

extern "C" int mnr_pack_model_bwd_h2(void *packed_dev, size_t bytes, const mnr_model_desc *d, void *stream) {
    ModelLayout m;
    BwdLayout b;
    int rc = h2_layout(d, m);
    if (rc != MNR_OK) return rc;
    if ((rc = bwd_layout_from_desc(d, b)) != MNR_OK) return rc;
    const size_t need = (size_t)h2b_total_chunks(b) * H2_CHUNK_BYTES;
    MNR_REQUIRE(packed_dev && bytes >= need, "backward packed buffer missing or too small");
    for (int i = 0; i < b.n_layers; ++i) MNR_REQUIRE(b.layer[i].w, "missing weight pointer for backward layer %d", i);
    const long total = (long)h2b_total_chunks(b) * H2_CHUNK_U4;
    hipLaunchKernelGGL(k_pack_bwd_h2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), b, reinterpret_cast<uint4v *>(packed_dev));
    return check_launch("k_pack_bwd_h2");
}

// segs[i].packed_fwd_dev / packed_bwd_dev: the mnr_pack_model_h2 / mnr_pack_model_bwd_h2 images
int mnr::mlp_backward_chain_multi_h2_impl(const mnr_mlp_grad_launch *segs, int n_segs, const CellTable *cells, hipStream_t s) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= H2B_MAX_SEGS, "1..%d segments per launch", H2B_MAX_SEGS);
    H2BwdMulti mm{};
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_grad_launch &L = segs[i];
        MNR_REQUIRE(L.desc && L.io, "segment %d: NULL argument", i);
        ModelLayout m;
        int rc = h2_layout(L.desc, m);
        if (rc != MNR_OK) return rc;
        rc = fill_bwd_args(mm.seg[i], m, L.packed_fwd_dev, L.packed_bwd_dev, L.desc, L.io);
        if (rc != MNR_OK) return rc;
        // the aux block (head weights, fp32) sits behind the h2 chunk stream of the FORWARD image
        mm.seg[i].aux_byte_off = (long)h2_total_chunks(m) * H2_CHUNK_BYTES;
        mm.seg[i].aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(L.packed_fwd_dev) + mm.seg[i].aux_byte_off);
        mm.is_b[i] = L.desc->xyz_dim == 4 ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        if (cells) {
            MNR_REQUIRE(cells[i].dcells && cells[i].cell_rows > 0 && cells[i].cell_rows % H2_ROWS == 0 && L.io->n_rows % cells[i].cell_rows == 0 &&
                        L.io->n_rows / cells[i].cell_rows == segs[0].io->n_rows / cells[0].cell_rows,
                        "segment %d: multi-cell launch needs rows per cell in multiples of %d and the same cells in every segment", i, H2_ROWS);
            mm.seg[i].dcells = cells[i].dcells; mm.seg[i].cell_rows = cells[i].cell_rows;
            wg += cells[i].cell_rows / H2_ROWS;
        } else {
            wg += (L.io->n_rows + H2_ROWS - 1) / H2_ROWS;
        }
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one launch");
    }
    for (int i = n_segs; i <= H2B_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    if (wg == 0) return MNR_OK;
    static bool lds_enabled_dev[MAX_DEVICES] = {};
    bool &lds_enabled = lds_enabled_dev[device_slot()];
    if (!lds_enabled) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_bwd_h2<H2FG, H2BG>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H2_CHUNK_BYTES);
        if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_mlp_bwd_h2): %s", hipGetErrorString(e));
        lds_enabled = true;
    }
    const unsigned ny = cells ? (unsigned)(segs[0].io->n_rows / cells[0].cell_rows) : 1u;
    hipLaunchKernelGGL((k_mlp_bwd_h2<H2FG, H2BG>), dim3((unsigned)wg, ny), dim3(H2_THREADS), 2 * H2_CHUNK_BYTES, s, mm);
    return check_launch("k_mlp_bwd_h2");
}
