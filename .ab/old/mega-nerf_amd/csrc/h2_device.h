// h2_device.h -- shared pieces of the split-precision kernels (csrc/mlp_fwd_h2.hip, csrc/mlp_bwd_h2.hip, csrc/step.hip): the image
// layout of (hi, lo) f16 fragment pairs, the packers' per-element bodies, the LDS weight stream and the K-step runner.
// Design notes: mlp_fwd_h2.hip.
#pragma once
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
__device__ __forceinline__ void pack_model_h2_thread(const ModelLayout &m, uint4v *__restrict__ chunks, float *__restrict__ aux, long n_u4, long tid) {
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


struct H2Stream {
    const uint4v *g;
    uint4v *lds;
    int cur;
    float one;                       // 1.0f behind an opaque asm (h2_split8)
#ifdef H2_EXPERIMENT_NO_DMA
    int n_issued = 0;
#endif
    __device__ __forceinline__ void init(const uint4v *chunks, uint4v *ring) {
        g = chunks; lds = ring; cur = 1;
        one = 1.0f;
        asm volatile("" : "+s"(one));
        issue();
    }
    __device__ __forceinline__ void issue() {
#ifdef H2_EXPERIMENT_NO_DMA
        if (n_issued >= 2) { g += H2_CHUNK_U4; return; }      // timing experiment only: the ring keeps its first two chunks
        ++n_issued;
#endif
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

// x = hi + lo, both f16: hi = x rounded towards zero (v_cvt_pkrtz, two values per instruction), lo = x - hi (exact in fp32) by ONE
// mixed-precision FMA that reads hi as f16 straight out of its packed half (v_fma_mix_f32  x * one - hi; `one` is 1.0f the compiler
// cannot see, or it folds the product away and emits a conversion plus a subtraction): 2 VALU instructions per value (and + sub +
// 2 x half a pack: 3), and the subtraction accounts for whatever the conversion dropped (f16 subnormals included).
__device__ __forceinline__ void h2_split8(const float (&x)[8], uint4v &hi, uint4v &lo, float one) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto h = __builtin_amdgcn_cvt_pkrtz(x[2 * q], x[2 * q + 1]);
        const float l0 = __builtin_fmaf(x[2 * q], one, -(float)h[0]);
        const float l1 = __builtin_fmaf(x[2 * q + 1], one, -(float)h[1]);
        hi[q] = __builtin_bit_cast(unsigned, h);
        lo[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(l0, l1));
    }
}

template <int OFF>
__device__ __forceinline__ uint4v lds_ld4u(unsigned addr) {
    uint4v v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void pin_u(uint4v &x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ floatx4 h2_mfma(uint4v a, uint4v b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// One segment of a layer: NK K-steps whose B operands are src[0 .. NSRC) (zero beyond); K0 = index of the segment's first K-step
// inside the layer (chunk boundaries are static: a new chunk every SPC K-steps, the first at K-step 0 of the layer).
#ifndef H2_FRAG_GROUP
#define H2_FRAG_GROUP 4
#endif
struct H2NoHook {
    template <class I> __device__ __forceinline__ void operator()(I) const {}
};
// `hook(chunk index inside the layer)` runs right behind every chunk boundary (behind the DMA issue of the following chunk): the
// place for the training kernels' tape stores (a boundary drains vmcnt: stores issued just BEFORE one cost a write round trip)
// G = output blocks whose (hi, lo) fragments are read ahead of their MFMAs (32 registers at 4; the training forward takes 2)
template <int NOB, int NK, int K0, int G, int NSRC, class Hook = H2NoHook>
__device__ __forceinline__ void h2_segment_visible(floatx4 (&acc)[NOB], const float (&src)[NSRC], H2Stream &st, int lane, Hook hook = Hook()) {
    constexpr int SPC = H2_CHUNK_U4 / (NOB * 2 * 64);
    static_for<0, NK>([&](auto kc) {
        constexpr int kl = decltype(kc)::value, k = K0 + kl;
        if constexpr (k % SPC == 0) { st.next_chunk(); hook(std::integral_constant<int, k / SPC>{}); }
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (8 * kl + j < NSRC) ? src[(8 * kl + j < NSRC) ? 8 * kl + j : 0] : 0.f;
        uint4v bh, bl;
        h2_split8(x, bh, bl, st.one);
        static_assert(NOB % G == 0, "output blocks per fragment group");
        {
        const uint4v *p = st.lds + st.cur * H2_CHUNK_U4 + (k % SPC) * NOB * 2 * 64 + lane;
#pragma unroll
        for (int o0 = 0; o0 < NOB; o0 += G) {
            uint4v ah[G], al[G];
#pragma unroll
#ifdef H2_EXPERIMENT_HALF_LDS
            for (int o = 0; o < G; ++o) { ah[o] = p[((o0 + o) * 2) * 64]; al[o] = ah[o]; }        // timing experiment only
#else
            for (int o = 0; o < G; ++o) { ah[o] = p[((o0 + o) * 2) * 64]; al[o] = p[((o0 + o) * 2 + 1) * 64]; }
#endif
#pragma unroll
            for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(ah[o], bh, acc[o0 + o]);
#pragma unroll
            for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(al[o], bh, acc[o0 + o]);
#pragma unroll
            for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(ah[o], bl, acc[o0 + o]);
        }
        }
    });
}

// ---- the asm-read form: ONE software pipeline over the segment (round 5; the fp32 kernels' scheme, mlp_device.h run_segment) ------------
// batch t = (K-step t / NBATCH, output blocks (t % NBATCH) * G ..): 2G fragment reads (hi, lo), 3G MFMAs.  Three fragment buffers: batch t
// computes, t + 1 is in flight, t + 2 is requested behind the first MFMA group of t.  The accumulator pins behind every group keep the
// MFMAs above the reads that follow them in the source (left to hipcc the MFMAs sink below the asm reads, every batch gets fresh
// registers, and the forward spilled 125-196 VGPRs -- which is why rounds 3-4 kept its reads compiler-visible, with a vmcnt(0) in front of
// each chunk's first read).  A chunk boundary does not restart the pipeline: when batch t + 2 opens a new chunk the barrier is taken at
// the START of batch t, once the old chunk's reads have landed; the wavefront waits there with two batches in hand.
template <int GS, int O0, int NOB, int G>
__device__ __forceinline__ void h2_frag_load(uint4v (&ah)[G], uint4v (&al)[G], unsigned addr) {
    static_for<0, G>([&](auto oc) {
        constexpr int o = O0 + decltype(oc)::value;
        ah[decltype(oc)::value] = lds_ld4u<(GS * NOB + o) * 2048>(addr);
        al[decltype(oc)::value] = lds_ld4u<(GS * NOB + o) * 2048 + 1024>(addr);
    });
}
template <int O0, int G, int NOB>
__device__ __forceinline__ void h2_pin_acc(floatx4 (&acc)[NOB]) {
#pragma unroll
    for (int o = 0; o < G; ++o) pin(acc[O0 + o]);
}
template <int NOB, int NK, int K0, int G, int NSRC, class Hook>
__device__ __forceinline__ void h2_segment_pipe(floatx4 (&acc)[NOB], const float (&src)[NSRC], H2Stream &st, int lane, Hook hook) {
    constexpr int SPC = H2_CHUNK_U4 / (NOB * 2 * 64), NBATCH = NOB / G, T = NK * NBATCH;
    static_assert(NOB % G == 0 && NBATCH >= 2, "at least two batches per K-step");
    static_assert(SPC * NOB * 2048 <= 65536 + 2048, "fragment offsets must fit the ds_read immediate");
    auto chunk_start = [](int t) constexpr { return t % NBATCH == 0 && (K0 + t / NBATCH) % SPC == 0; };
    uint4v ah[3][G], al[3][G];
    uint4v bh = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
    if constexpr (chunk_start(0)) { st.next_chunk(); hook(std::integral_constant<int, K0 / SPC>{}); }
    unsigned addr = lds_addr(st.lds + st.cur * H2_CHUNK_U4 + lane);
    h2_frag_load<K0 % SPC, 0, NOB, G>(ah[0], al[0], addr);
    if constexpr (T > 1) h2_frag_load<(K0 + 1 / NBATCH) % SPC, (1 % NBATCH) * G, NOB, G>(ah[1], al[1], addr);
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, u = t + 2, kl = t / NBATCH, o0 = (t % NBATCH) * G;
        constexpr bool early = u < T && chunk_start(u);
        if constexpr (t % NBATCH == 0) {                       // this K-step's B operand: 8 activations -> (hi, lo) f16 octets
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (8 * kl + j < NSRC) ? src[(8 * kl + j < NSRC) ? 8 * kl + j : 0] : 0.f;
            h2_split8(x, bh, bl, st.one);
        }
        if constexpr (early) {
            wait_lgkm<0>();
            st.next_chunk();
            hook(std::integral_constant<int, (K0 + u / NBATCH) / SPC>{});
            addr = lds_addr(st.lds + st.cur * H2_CHUNK_U4 + lane);
        } else if constexpr (t + 1 < T) {
            wait_lgkm<2 * G>();
        } else {
            wait_lgkm<0>();
        }
        uint4v (&ch)[G] = ah[t % 3], (&cl)[G] = al[t % 3];
#pragma unroll
        for (int o = 0; o < G; ++o) { pin_u(ch[o]); pin_u(cl[o]); }
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(ch[o], bh, acc[o0 + o]);
        h2_pin_acc<o0, G>(acc);
        if constexpr (u < T) h2_frag_load<(K0 + u / NBATCH) % SPC, (u % NBATCH) * G, NOB, G>(ah[u % 3], al[u % 3], addr);
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(cl[o], bh, acc[o0 + o]);
#pragma unroll
        for (int o = 0; o < G; ++o) acc[o0 + o] = h2_mfma(ch[o], bl, acc[o0 + o]);
        h2_pin_acc<o0, G>(acc);
    });
}

template <int NOB, int NK, int K0, int G, bool ASM_READS, int NSRC, class Hook = H2NoHook>
__device__ __forceinline__ void h2_segment_g(floatx4 (&acc)[NOB], const float (&src)[NSRC], H2Stream &st, int lane, Hook hook = Hook()) {
#if defined(H2_EXPERIMENT_HALF_LDS)
    h2_segment_visible<NOB, NK, K0, G>(acc, src, st, lane, hook);
#else
    if constexpr (ASM_READS) h2_segment_pipe<NOB, NK, K0, G>(acc, src, st, lane, hook);
    else h2_segment_visible<NOB, NK, K0, G>(acc, src, st, lane, hook);
#endif
}
template <int NOB, int NK, int K0, int NSRC, class Hook = H2NoHook>
__device__ __forceinline__ void h2_segment(floatx4 (&acc)[NOB], const float (&src)[NSRC], H2Stream &st, int lane, Hook hook = Hook()) {
    // the data-gradient chain (its only caller) reads with inline asm: measured 0.98 -> 0.94 ms on the benchmark step; the FORWARD keeps
    // compiler-visible reads (h2_segment_g<..., false>): with two batches pinned in flight it spills 125-196 VGPRs and loses what it gains
    h2_segment_g<NOB, NK, K0, H2_FRAG_GROUP, true>(acc, src, st, lane, hook);
}


}  // namespace mnr

namespace mnr {

// ---- transposed (data-gradient) image: bwd layer order of BwdLayout, K-steps of 32 output features ----------------------------
MNR_HD int h2b_ksteps(const BwdLayerLayout &l) { return l.nsteps / 8; }
MNR_HD int h2b_spc(const BwdLayerLayout &l) { const int s = H2_CHUNK_U4 / (l.nob * 2 * 64); return s < 1 ? 1 : s; }
MNR_HD int h2b_layer_chunks(const BwdLayerLayout &l) { return (h2b_ksteps(l) + h2b_spc(l) - 1) / h2b_spc(l); }
MNR_HD int h2b_total_chunks(const BwdLayout &b) {
    int c = 0;
    for (int i = 0; i < b.n_layers; ++i) c += h2b_layer_chunks(b.layer[i]);
    return c + 1;
}
// element (row of the transposed product = input column of the nn.Linear, K-step S, lane-part p, slot j) = w[hid_src(P, 8S + j, p)][in_col(row)]
__device__ __forceinline__ void pack_bwd_h2_thread(const BwdLayout &b, uint4v *__restrict__ chunks, long tid) {
    if (tid >= (long)h2b_total_chunks(b) * H2_CHUNK_U4) return;
    const int chunk = (int)(tid / H2_CHUNK_U4), within = (int)(tid % H2_CHUNK_U4);
    uint4v v = {0u, 0u, 0u, 0u};
    int c0 = 0;
    for (int li = 0; li < b.n_layers; ++li) {
        const BwdLayerLayout &l = b.layer[li];
        const int nc = h2b_layer_chunks(l);
        if (chunk >= c0 && chunk < c0 + nc) {
            const int spc = h2b_spc(l);
            const int lane = within & 63, frag = within >> 6;
            const int hl = frag & 1, ob = (frag >> 1) % l.nob, kc = (frag >> 1) / l.nob;
            const int S = (chunk - c0) * spc + kc;
            if (kc < spc && S < h2b_ksteps(l)) {
                const int row = ob * 16 + (lane & 15), part = lane >> 4;
                if (row < l.n_rows) {
                    const int in_col = row < l.split ? l.in_off + row : l.in_off2 + (row - l.split);
                    unsigned short e[8];
                    for (int j = 0; j < 8; ++j) {
                        unsigned short hi, lo;
                        h2_split_weight(l.w[(long)hid_src(b.parts, 8 * S + j, part) * l.ld + in_col], hi, lo);
                        e[j] = hl ? lo : hi;
                    }
                    v = uint4v{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                               (unsigned)e[6] | ((unsigned)e[7] << 16)};
                }
            }
        }
        c0 += nc;
    }
    chunks[tid] = v;
}

}  // namespace mnr
