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
#include "mlp_device.h"
#include "pack_device.h"

namespace mnr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int H2_WAVES = 8, H2_THREADS = H2_WAVES * 64, H2_ROWS = H2_WAVES * 16;
constexpr int H2_CHUNK_U4 = 4096;                      // 64 KiB
constexpr int H2_CHUNK_BYTES = H2_CHUNK_U4 * 16;

// ---- image layout (host + device) ----------------------------------------------------------------------------------------------
// per layer: K-steps = sum over its segments of ceil(seg.nsteps / 8) (a segment's registers are padded to whole K-steps with
// zeros); a chunk holds SPC = 4096 / (nob * 2 * 64) K-steps (2 for 16 output blocks, 4 for 8); a layer starts on a chunk boundary;
// inside a chunk: [K-step][output block][hi | lo][lane] x 16 bytes.
MNR_HD int h2_ksteps(const LayerLayout &l) {
    int k = 0;
    for (int i = 0; i < l.nseg; ++i) k += (l.seg[i].nsteps + 7) / 8;
    return k;
}
MNR_HD int h2_spc(const LayerLayout &l) { return H2_CHUNK_U4 / (l.nob * 2 * 64); }
MNR_HD int h2_layer_chunks(const LayerLayout &l) { return (h2_ksteps(l) + h2_spc(l) - 1) / h2_spc(l); }
MNR_HD int h2_total_chunks(const ModelLayout &m) {
    int c = 0;
    for (int i = 0; i < m.n_mfma_layers; ++i) c += h2_layer_chunks(m.layer[i]);
    return c + 1;                                      // + one trailing chunk: the stream prefetches one past the end
}
// source column of slot j of K-step S (lane-part p) of layer l, -1 = zero pad
MNR_HD int h2_src_col(const LayerLayout &l, int P, int S, int p, int j) {
    int s0 = 0;
    for (int i = 0; i < l.nseg; ++i) {
        const Seg &g = l.seg[i];
        const int ks = (g.nsteps + 7) / 8;
        if (S < s0 + ks) {
            const int r = 8 * (S - s0) + j;
            if (r >= g.nsteps) return -1;
            int c = -1;
            if (g.type == SEG_EMB) c = emb_src(g.D, g.L, P, r, p);
            else if (g.type == SEG_HID) c = hid_src(P, r, p);
            else if (g.type == SEG_APP) c = app_src(g.D, P, r, p);
            return c < 0 ? -1 : g.col0 + c;
        }
        s0 += ks;
    }
    return -1;
}

__device__ __forceinline__ void h2_split_weight(float w, unsigned short &hi, unsigned short &lo) {
    // hi = w rounded to 10 mantissa bits (round half up on the magnitude), lo = the rest, rounded the same way
    const unsigned u = (__float_as_uint(w) + 0x1000u) & 0xffffe000u;
    const _Float16 h = (_Float16)__uint_as_float(u);
    const float r = w - (float)h;
    const unsigned v = (__float_as_uint(r) + 0x1000u) & 0xffffe000u;
    const _Float16 l = (_Float16)__uint_as_float(v);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// one thread per 16-byte fragment element of the chunk stream, then one thread per float of the aux image (= the fp32 image's)
__global__ void k_pack_model_h2(ModelLayout m, uint4v *__restrict__ chunks, float *__restrict__ aux, long n_u4) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_u4) {
        pack_model_aux_thread(m, aux, tid - n_u4);
        return;
    }
    const int chunk = (int)(tid / H2_CHUNK_U4), within = (int)(tid % H2_CHUNK_U4);
    uint4v v = {0u, 0u, 0u, 0u};
    int c0 = 0;
    for (int li = 0; li < m.n_mfma_layers; ++li) {
        const LayerLayout &l = m.layer[li];
        const int nc = h2_layer_chunks(l);
        if (chunk >= c0 && chunk < c0 + nc) {
            const int spc = h2_spc(l);
            const int lane = within & 63, frag = within >> 6;                 // frag = (kstep_in_chunk * nob + ob) * 2 + hl
            const int hl = frag & 1, ob = (frag >> 1) % l.nob, kc = (frag >> 1) / l.nob;
            const int S = (chunk - c0) * spc + kc;
            if (kc < spc && S < h2_ksteps(l)) {
                const int row = ob * 16 + (lane & 15), part = lane >> 4;
                unsigned short e[8];
                for (int j = 0; j < 8; ++j) {
                    const int col = h2_src_col(l, m.parts, S, part, j);
                    const float w = (col >= 0 && row < l.n_out) ? l.w[(long)row * l.ld + col] : 0.f;
                    unsigned short hi, lo;
                    h2_split_weight(w, hi, lo);
                    e[j] = hl ? lo : hi;
                }
                v = uint4v{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                           (unsigned)e[6] | ((unsigned)e[7] << 16)};
            }
        }
        c0 += nc;
    }
    chunks[tid] = v;
}

// ---- kernel ------------------------------------------------------------------------------------------------------------------
struct H2Args {
    const uint4v *chunks;
    const float *aux;
    const float *emb_a;
    mnr_mlp_io io;
    int32_t bias_off[MAX_MFMA_LAYERS];
    int32_t sigma_off, rgb_off, sigma_act, app_count;
};

struct H2Stream {
    const uint4v *g;
    uint4v *lds;
    int cur;
    __device__ __forceinline__ void issue() {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        uint4v *dst = lds + (cur ^ 1) * H2_CHUNK_U4 + wave * 64;
        const unsigned lane_off = threadIdx.x * 16u;
#pragma unroll
        for (int i = 0; i < H2_CHUNK_U4 / H2_THREADS; ++i) {
            unsigned lo = lane_off;
            asm("" : "+v"(lo));
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(reinterpret_cast<const char *>(g + i * H2_THREADS)) + lo),
                                             (lds_void_t *)(dst + i * H2_THREADS), 16, 0, 0);
        }
        g += H2_CHUNK_U4;
    }
    __device__ __forceinline__ void next_chunk() {
        __syncthreads();
        cur ^= 1;
        issue();
    }
};

__device__ __forceinline__ void h2_split8(const float (&x)[8], uint4v &hi, uint4v &lo) {
    float h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = __uint_as_float(__float_as_uint(x[j]) & 0xffffe000u);
        l[j] = x[j] - h[j];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2 * q], h[2 * q + 1]));
        lo[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l[2 * q], l[2 * q + 1]));
    }
}

__device__ __forceinline__ floatx4 h2_mfma(uint4v a, uint4v b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// One segment of a layer: NK K-steps whose B operands are src[0 .. NSRC) (zero beyond); K0 = index of the segment's first K-step
// inside the layer (chunk boundaries are static: a new chunk every SPC K-steps, the first at K-step 0 of the layer).
template <int NOB, int NK, int K0, int NSRC>
__device__ __forceinline__ void h2_segment(floatx4 (&acc)[NOB], const float (&src)[NSRC], H2Stream &st, int lane) {
    constexpr int SPC = H2_CHUNK_U4 / (NOB * 2 * 64);
    static_for<0, NK>([&](auto kc) {
        constexpr int kl = decltype(kc)::value, k = K0 + kl;
        if constexpr (k % SPC == 0) st.next_chunk();
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (8 * kl + j < NSRC) ? src[(8 * kl + j < NSRC) ? 8 * kl + j : 0] : 0.f;
        uint4v bh, bl;
        h2_split8(x, bh, bl);
        const uint4v *p = st.lds + st.cur * H2_CHUNK_U4 + (k % SPC) * NOB * 2 * 64 + lane;
#pragma unroll
        for (int o0 = 0; o0 < NOB; o0 += 4) {
            uint4v ah[4], al[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) { ah[o] = p[((o0 + o) * 2) * 64]; al[o] = p[((o0 + o) * 2 + 1) * 64]; }
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o0 + o] = h2_mfma(ah[o], bh, acc[o0 + o]);
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o0 + o] = h2_mfma(al[o], bh, acc[o0 + o]);
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o0 + o] = h2_mfma(ah[o], bl, acc[o0 + o]);
        }
    });
}

template <class C>
__device__ __forceinline__ void mlp_fwd_h2_body(const H2Args &a, long blk) {
    static_assert(C::TILE == 16 && C::W == 256 && C::HAS_FINAL && C::RGB == 3, "split-precision kernel: default 8x256 architectures");
    constexpr int P = C::P, H = C::H, NOB = C::NOB, NOB2 = C::NOB2, H2 = C::H2, RPB = C::RPB;
    constexpr int KE = (C::EX + 7) / 8, KH = H / 8, KD = (C::ED + 7) / 8;
    extern __shared__ uint4v h2_ring[];
    const mnr_mlp_io &io = a.io;
    const long n_rows = io.n_units_dev ? (long)(*io.n_units_dev) * io.rows_per_unit : (long)io.n_rows;
    if (blk * H2_ROWS >= n_rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const long row = (blk * H2_WAVES + wave) * 16 + (lane & 15);
    const bool valid = row < n_rows;
    const long rc = valid ? row : n_rows - 1;
    const long ray = rc / io.rows_per_ray;
    const float *aux = a.aux;

    H2Stream st;
    st.g = a.chunks;
    st.lds = h2_ring;
    st.cur = 1;
    st.issue();

    float x[C::XYZ];
#pragma unroll
    for (int d = 0; d < C::XYZ; ++d) x[d] = io.xyz[rc * io.xyz_stride + d];
    float ex[C::EX];
    embed<C::XYZ, C::LX, P>(ex, x, part);

    float h[H];
    floatx4 acc[NOB];
    static_for<0, C::NL>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        init_acc<NOB, RPB>(acc, aux + a.bias_off[l] + part * H);
        if constexpr (l == 0) {
            h2_segment<NOB, KE, 0>(acc, ex, st, lane);
        } else if constexpr ((C::SKIP >> l) & 1) {
            h2_segment<NOB, KE, 0>(acc, ex, st, lane);
            h2_segment<NOB, KH, KE>(acc, h, st, lane);
        } else {
            h2_segment<NOB, KH, 0>(acc, h, st, lane);
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
        if (io.sigma_noise) s += io.sigma_noise[rc];
        sigma = a.sigma_act ? softplus_shifted(s) : fmaxf(s, 0.f);
    }

    // xyz_encoding_final (no activation), then dir_a_encoding over [final | dir embedding | appearance]
    init_acc<NOB, RPB>(acc, aux + a.bias_off[C::NL] + part * H);
    h2_segment<NOB, KH, 0>(acc, h, st, lane);
    acc_to_regs<NOB, RPB, false>(h, acc);
    floatx4 acc2[NOB2];
    init_acc<NOB2, RPB>(acc2, aux + a.bias_off[C::NL + 1] + part * H2);
    h2_segment<NOB2, KH, 0>(acc2, h, st, lane);
    {
        float dv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) dv[d] = io.dir[ray * io.dir_stride + d];
        float ed[C::ED];
        embed<3, C::LD, P>(ed, dv, part);
        h2_segment<NOB2, KD, KH>(acc2, ed, st, lane);
        long idx = io.idx_is_float ? (long)reinterpret_cast<const float *>(io.idx)[ray * io.idx_stride]
                                   : (long)reinterpret_cast<const int32_t *>(io.idx)[ray * io.idx_stride];
        idx = idx < 0 ? 0 : (idx >= a.app_count ? a.app_count - 1 : idx);
        const float *ea = a.emb_a + idx * C::APP + part * (C::APP / P);
        float ap[C::AP];
#pragma unroll
        for (int i = 0; i < C::AP; ++i) ap[i] = (i < C::APP / P) ? ea[i] : 0.f;
        h2_segment<NOB2, (C::AP + 7) / 8, KH + KD>(acc2, ap, st, lane);
    }
    float dreg[H2];
    acc_to_regs<NOB2, RPB, true>(dreg, acc2);
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
    float *o = io.out + row * io.out_stride;
    o[0] = sigmoidf_(rgbraw[0]); o[1] = sigmoidf_(rgbraw[1]); o[2] = sigmoidf_(rgbraw[2]); o[3] = sigma;
}

constexpr int H2_MAX_SEGS = 4;
struct H2Multi {
    H2Args seg[H2_MAX_SEGS];
    int32_t wg0[H2_MAX_SEGS + 1];
    int32_t is_b[H2_MAX_SEGS];
};
template <class CA, class CB>
__global__ __launch_bounds__(H2_THREADS, 1) void k_mlp_fwd_h2(H2Multi m) {
    const int blk = blockIdx.x;
    const int s = (blk >= m.wg0[1]) + (blk >= m.wg0[2]) + (blk >= m.wg0[3]);
    if (m.is_b[s]) mlp_fwd_h2_body<CB>(m.seg[s], blk - m.wg0[s]);
    else mlp_fwd_h2_body<CA>(m.seg[s], blk - m.wg0[s]);
}

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m);

}  // namespace mnr

using namespace mnr;

using H2FG = MlpCfg<3, 12, 4, 48, 256, 8, 16, 3, 16>;
using H2BG = MlpCfg<4, 12, 4, 48, 256, 8, 16, 3, 16>;

static int h2_layout(const mnr_model_desc *d, ModelLayout &m) {
    int rc = layout_from_desc(d, m);
    if (rc != MNR_OK) return rc;
    const bool ok = (d->xyz_dim == 3 || d->xyz_dim == 4) && d->pos_xyz_dim == 12 && d->pos_dir_dim == 4 && d->appearance_dim == 48 &&
                    d->layer_dim == 256 && d->layers == 8 && d->skip_mask == 16 && d->rgb_dim == 3 && m.tile == 16;
    if (!ok) return set_err(MNR_E_UNSUPPORTED, "the split-precision forward covers the default 8x256 foreground / background models");
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

extern "C" int mnr_mlp_forward_multi_h2(const mnr_mlp_launch *segs, int n_segs, void *stream) {
    MNR_REQUIRE(segs && n_segs >= 1 && n_segs <= H2_MAX_SEGS, "1..%d segments per launch", H2_MAX_SEGS);
    H2Multi mm{};
    long wg = 0;
    for (int i = 0; i < n_segs; ++i) {
        const mnr_mlp_launch &L = segs[i];
        MNR_REQUIRE(L.packed_dev && L.desc && L.io && L.io->xyz && L.io->out && !L.tape_dev, "segment %d: bad arguments (inference only)", i);
        MNR_REQUIRE(!L.io->row_index && !L.io->sigma_only && L.io->apply_sh_deg < 0, "segment %d: gather / sigma_only / SH are not covered", i);
        MNR_REQUIRE(L.io->rows_per_ray >= 1 && L.io->n_rows >= 0 && L.io->dir && L.io->idx && L.desc->embedding_a, "segment %d: dir / idx / embedding_a required", i);
        ModelLayout m;
        int rc = h2_layout(L.desc, m);
        if (rc != MNR_OK) return rc;
        H2Args &a = mm.seg[i];
        a.chunks = reinterpret_cast<const uint4v *>(L.packed_dev);
        a.aux = reinterpret_cast<const float *>(reinterpret_cast<const char *>(L.packed_dev) + (size_t)h2_total_chunks(m) * H2_CHUNK_BYTES);
        a.emb_a = L.desc->embedding_a;
        a.io = *L.io;
        for (int k = 0; k < MAX_MFMA_LAYERS; ++k) a.bias_off[k] = k < m.n_mfma_layers ? m.layer[k].bias_off : 0;
        a.sigma_off = m.sigma_off; a.rgb_off = m.rgb_off; a.sigma_act = L.desc->sigma_activation; a.app_count = L.desc->appearance_count;
        mm.is_b[i] = L.desc->xyz_dim == 4 ? 1 : 0;
        mm.wg0[i] = (int32_t)wg;
        wg += (L.io->n_rows + H2_ROWS - 1) / H2_ROWS;
        MNR_REQUIRE(wg <= 0x7fffffffL, "too many rows for one launch");
    }
    for (int i = n_segs; i <= H2_MAX_SEGS; ++i) mm.wg0[i] = (int32_t)wg;
    if (wg == 0) return MNR_OK;
    static bool lds_enabled_dev[MAX_DEVICES] = {};       // raise the dynamic-LDS cap once per device (benign if raced)
    bool &lds_enabled = lds_enabled_dev[device_slot()];
    if (!lds_enabled) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_fwd_h2<H2FG, H2BG>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * H2_CHUNK_BYTES);
        if (e != hipSuccess) return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_mlp_fwd_h2): %s", hipGetErrorString(e));
        lds_enabled = true;
    }
    hipLaunchKernelGGL((k_mlp_fwd_h2<H2FG, H2BG>), dim3((unsigned)wg), dim3(H2_THREADS), 2 * H2_CHUNK_BYTES, as_stream(stream), mm);
    return check_launch("k_mlp_fwd_h2");
}
