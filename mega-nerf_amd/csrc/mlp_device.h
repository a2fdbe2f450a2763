// mlp_device.h -- device-side building blocks shared by the fused MLP forward (mlp_fwd.hip) and the
// backward data-gradient chain (mlp_bwd.hip): MFMA segment runner, LDS-DMA weight stream, encoders.
#pragma once
#include <type_traits>

#include "common.h"
#include "mlp_layout.h"

#ifndef MNR_FRAG_DEPTH
#define MNR_FRAG_DEPTH 2
#endif
#ifndef MNR_FRAG_WEAVE
#define MNR_FRAG_WEAVE 1
#endif

namespace mnr {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int XYZ_, int LX_, int LD_, int APP_, int W_, int NL_, int SKIP_, int RGB_, int TILE_ = tile_for_width(W_)>
struct MlpCfg {
    static constexpr int XYZ = XYZ_, LX = LX_, LD = LD_, APP = APP_, W = W_, NL = NL_, SKIP = SKIP_, RGB = RGB_;
    static constexpr int TILE = TILE_, P = 64 / TILE;
    static constexpr int RPB = TILE * TILE / 64;                 // accumulator registers per output block
    static constexpr int H = hid_regs(W_, P);                    // hidden registers per lane
    static constexpr int NOB = W_ / TILE;
    static constexpr int EX = emb_regs(XYZ_, LX_, P);
    static constexpr bool HAS_FINAL = (LD_ > 0 || APP_ > 0);
    static constexpr int ED = emb_regs(3, LD_, P);
    static constexpr int AP = app_regs(APP_, P);
    static constexpr int NOB2 = (W_ / 2) / TILE;
    static constexpr int H2 = HAS_FINAL ? (W_ / 2) / P : H;      // inputs of the rgb head per lane
    static constexpr int GPC = CHUNK_F4 / (NOB * 64);
    static constexpr int GPC2 = HAS_FINAL ? CHUNK_F4 / (NOB2 * 64) : 1;
    static constexpr int ROWS_PER_WG = 4 * TILE;
};



// One cell's share of a multi-cell segment (training step of several submodules in one launch, csrc/step.hip): the rows of
// cell c occupy [c * cell_rows, (c + 1) * cell_rows) of the segment's input / output arrays (same architecture, private
// weights); everything here is uniform per workgroup.  Device table, one entry per cell.
struct MlpCellSeg {
    const void *packed;        // forward weight image of the cell's model (mnr_pack_model)
    const void *packed_bwd;    // transposed image (mnr_pack_model_bwd; data-gradient chain only)
    const float *emb_a;        // its appearance table
    float *d_emb_a;            // ... and that table's gradient (data-gradient chain only)
    long tape_row0;            // tape row of the cell's first row of this pass
    const int32_t *n_units;    // device-side unit count of the cell (compacted background rays) or NULL
    int32_t *zexp;             // split-precision step: per-plane exponents of the model's gradient tape (ZEXP_* below) or NULL
};
// Split-precision weight gradients scale every dZ plane by a power of two before its f16 split (gradients of 1e-6 .. 1e-12 are below
// the f16 range).  The split-precision data-gradient chain publishes, per plane, the largest row exponent it saw (row max < 2^E)
// as E + ZEXP_BIAS by atomic max; 0 = nothing seen.  Planes: trunk layer l -> l, final -> layers, dir_a -> layers + 1.
constexpr int ZEXP_BIAS = 1024, ZEXP_PLANES = 16, ZEXP_TARGET = 14;

// 16-byte store to  uniform base + 32-bit lane offset (bytes) + immediate: `global_store_dwordx4 v_off, v[data], s[base:base+1] offset:imm`.
// One address VGPR per row instead of a 64-bit pointer pair per plane (which the training kernels spilled), and a GLOBAL store where
// pointers that come out of a dynamically indexed kernel-argument struct would otherwise compile to FLAT stores (which count on
// lgkmcnt as well and so sit in every LDS wait).
typedef __attribute__((address_space(1))) char mnr_gchar;
__device__ __forceinline__ const char *uniform_ptr(const char *p);
template <int IMM>
__device__ __forceinline__ void gstore4(const float *uniform_base, unsigned byte_off, float4 v) {
    mnr_gchar *b = (mnr_gchar *)uniform_ptr(reinterpret_cast<const char *>(uniform_base));
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) f4v gf4v;
    // NON-TEMPORAL: the tapes (1.1 - 2 GB per launch) are read back launches later, by kernels that stream them once; written through, they
    // leave no dirty lines behind in the eight L2s for the next kernel to work around.  Measured (same box, alternated twice; -DMNR_PLAIN_TAPE_STORES
    // builds the other form): benchmark step 6.12 -> 6.06 ms (the launch behind the data-gradient chain, k_head_grads, 0.120 -> 0.080 ms),
    // 512-wide step 22.94 -> 22.75, split-precision step 3.10 -> 3.01 (its forward's fine launch 0.578 -> 0.519); inference unchanged.
    // (k_tgemm's output tiles are the next launch's operand: there non-temporal stores cost 18 %, csrc/tgemm.hip.)
#ifdef MNR_PLAIN_TAPE_STORES
    *(gf4v *)(b + byte_off + IMM) = f4v{v.x, v.y, v.z, v.w};
#else
    __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, (gf4v *)(b + byte_off + IMM));
#endif
}

// ---- weight stream: global -> LDS (async LDS-DMA, issued one chunk ahead) ------------------------
// `global_load_lds_dwordx4`: every lane supplies its own global address, the data lands at
// (wave-uniform LDS base) + lane*16 -- the packed image is lane-linear, so no staging registers and no
// ds_write pass are needed.  The barrier that publishes chunk c also proves every wave has finished
// reading the buffer chunk c+1 is then loaded into (2-deep ring).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// marks a pointer as wave-uniform (it is: derived from kernel arguments and block / tile indices) so that pointer + 32-bit
// lane offset selects the SGPR-base addressing form
__device__ __forceinline__ const char *uniform_ptr(const char *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ long uniform_long(long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long)(((unsigned long long)hi << 32) | lo);
}

template <int NT>        // NT = threads of the workgroup that shares the stream (256: four wavefronts; 512: eight)
struct WStreamT {
    const float4 *g;     // the next chunk to load (uniform: lives in SGPRs; the lane's 16-byte slot is added as a 32-bit offset)
    float4 *lds;         // base of the 2-chunk LDS ring
    int cur;             // buffer the MFMAs currently read
    __device__ __forceinline__ void issue() {
        // the wave number is forced into an SGPR: the LDS destination (M0) and everything else that is uniform per wave is
        // then computed on the scalar unit -- VALU instructions inside the MFMA stream cost matrix-pipe issue slots
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        float4 *dst = lds + (cur ^ 1) * CHUNK_F4 + wave * 64;      // wave-uniform base; HW adds lane*16
        // scalar base + 32-bit lane offset: the SGPR-base form of the load.  Every piece gets its own scalar base
        // (s_add_u32 / s_addc_u32): left to itself hipcc forms ONE per-lane 64-bit address and adds the piece offsets with a
        // v_lshl_add_u64 each (they exceed the 12-bit immediate) -- 8 VALU instructions per chunk inside the MFMA stream
        const unsigned lane_off = threadIdx.x * 16u;
#pragma unroll
        for (int i = 0; i < CHUNK_F4 / NT; ++i) {
            unsigned lo = lane_off;
            asm("" : "+v"(lo));          // a fresh 32-bit value per piece: otherwise its zero-extension is hoisted and the add goes 64-bit VALU again
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(reinterpret_cast<const char *>(g + i * NT)) + lo),
                                             (lds_void_t *)(dst + i * NT), 16, 0, 0);
        }
        g += CHUNK_F4;
    }
    // publish the chunk in flight (hipcc drains vmcnt before the barrier), make it current, start the next one
    __device__ __forceinline__ void next_chunk() {
        __syncthreads();
        cur ^= 1;
        issue();
    }
    // the same in pieces, for callers that weave the DMA requests into their MFMA stream: publish() = barrier + flip, then
    // issue_piece<0 .. PIECES - 1>() (any order, each once), then issued()
    static constexpr int PIECES = CHUNK_F4 / NT;
    __device__ __forceinline__ void publish() {
        __syncthreads();
        cur ^= 1;
    }
    template <int I>
    __device__ __forceinline__ void issue_piece() {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        float4 *dst = lds + (cur ^ 1) * CHUNK_F4 + wave * 64;
        unsigned lo = threadIdx.x * 16u;
        asm("" : "+v"(lo));
        __builtin_amdgcn_global_load_lds((global_cvoid_t *)(uniform_ptr(reinterpret_cast<const char *>(g + I * NT)) + lo),
                                         (lds_void_t *)(dst + I * NT), 16, 0, 0);
    }
    __device__ __forceinline__ void issued() { g += CHUNK_F4; }
};
using WStream = WStreamT<256>;

// ---- LDS reads hipcc must not see ---------------------------------------------------------------------------------------------
// Behind an LDS-DMA (global_load_lds) into an LDS array the compiler puts `s_waitcnt vmcnt(0)` in front of the next compiler-visible
// ds_read of that array.  In the chunk ring that is the first A-fragment read of chunk c, right behind the DMA burst of chunk c + 1:
// every wavefront waited for the prefetch it had just issued before it touched the chunk that was already there -- the double
// buffering overlapped nothing inside a workgroup (only the CU's other workgroup ran meanwhile; rounds 1-3).  Reading the fragments
// with inline asm and counting lgkmcnt by hand removes that wait (round 4; csrc/wgrad.hip and tgemm.hip always read this way).
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}
template <int OFF>
__device__ __forceinline__ floatx4 lds_ld4(unsigned addr) {
    floatx4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// an empty asm that "redefines" a register: MFMAs consuming it cannot be scheduled above the wait that precedes the pin
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(floatx4 &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(floatx16 &x) { asm volatile("" : "+v"(x)); }

// Static schedule of a K segment's fragment batches (TILE = 16): batch t = (group G0 + t / NBATCH, blocks (t % NBATCH) * OBB ..).  A "run" is
// a stretch of batches inside one weight chunk; its fragment reads are software-pipelined two batches deep and restart behind the
// chunk barrier that publishes the next chunk.
template <int NBATCH, int GPC, int G0, int T>
struct SegSched {
    static constexpr bool chunk_start(int t) { return t % NBATCH == 0 && (G0 + t / NBATCH) % GPC == 0 && (G0 + t / NBATCH) > 0; }
    static constexpr int run_start(int t) { int r = t; while (r > 0 && !chunk_start(r)) --r; return r; }
    static constexpr int run_end(int t) { int r = t + 1; while (r < T && !chunk_start(r)) ++r; return r; }    // one past the run's last batch
};
// the OBB A fragments of batch (group slot GS of the chunk, first block O0): asm reads at immediate offsets from the lane's chunk address
template <int GS, int O0, int NOB, int OBB>
__device__ __forceinline__ void frag_load(floatx4 (&a)[OBB], unsigned addr) {
    static_for<0, OBB>([&](auto oc) { a[decltype(oc)::value] = lds_ld4<(GS * NOB + O0 + decltype(oc)::value) * 1024>(addr); });
}
// ... and their 4 x OBB MFMAs.  The empty asm behind them "redefines" every accumulator block the batch wrote: volatile asm statements
// keep their program order, so the fragment reads that follow in the source cannot be scheduled above these MFMAs.  Without it the MFMAs
// -- pure values to the compiler -- sink below the asm reads of the following batches and groups, each of which then gets fresh
// registers: in the dir_a layer 45 ds_read_b128 (180 registers) were in flight at once and 51-63 registers went to scratch (rounds 3-4).
template <int O0, int OBB, int NOB>
__device__ __forceinline__ void frag_mfmas(floatx4 (&acc)[NOB], floatx4 (&a)[OBB], float b0, float b1, float b2, float b3) {
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) pin(a[ob]);
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) acc[O0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][0], b0, acc[O0 + ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) acc[O0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][1], b1, acc[O0 + ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) acc[O0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][2], b2, acc[O0 + ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) acc[O0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][3], b3, acc[O0 + ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) pin(acc[O0 + ob]);
}

// The same batch with the fragment reads of a LATER batch (group slot GS, first block O0N) woven in: one ds_read_b128 behind each K step's
// OBB MFMAs, into a third buffer.  A burst of four reads behind sixteen MFMAs holds the wavefront's issue port long enough to open a gap
// in the matrix pipe (measured: the strictly ordered burst form ran the forward 1.5 % and the data-gradient chain 3 % slower than the
// schedule hipcc had found on its own); one read per four MFMAs disappears in their issue shadow.  The pins behind every K step keep
// MFMAs and reads in exactly this order.
template <int O0, int GS, int O0N, int NOBF, bool LOAD, bool DMA, int OBB, int NOB, class Stream>
__device__ __forceinline__ void frag_mfmas_weave(floatx4 (&acc)[NOB], floatx4 (&a)[OBB], floatx4 (&an)[OBB], unsigned addr, Stream &st, float b0,
                                                 float b1, float b2, float b3) {
    static_assert(OBB == 4, "one read per K step: four blocks per batch");
#pragma unroll
    for (int ob = 0; ob < OBB; ++ob) pin(a[ob]);
    const float bk[4] = {b0, b1, b2, b3};
    static_for<0, 4>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int ob = 0; ob < OBB; ++ob) acc[O0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][k], bk[k], acc[O0 + ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < OBB; ++ob) pin(acc[O0 + ob]);
        if constexpr (DMA) {                      // the next-but-one chunk's DMA requests, a quarter behind each K step
            constexpr int PP = (Stream::PIECES + 3) / 4;
            static_for<0, PP>([&](auto pc) {
                constexpr int piece = k * PP + decltype(pc)::value;
                if constexpr (piece < Stream::PIECES) st.template issue_piece<piece>();
            });
            if constexpr (k == 3) st.issued();
        }
        if constexpr (LOAD) an[k] = lds_ld4<(GS * NOBF + O0N + k) * 1024>(addr);
    });
}

// does run_segment<TILE, NOB, NG, GPC, G0> take the woven one-pipeline form (and can it therefore publish the next layer's first chunk)?
template <int TILE, int NOB, int NG, int GPC, int G0>
constexpr bool seg_weaves() {
    constexpr int OBB = NOB < 4 ? NOB : 4, NBATCH = NOB / OBB, T = NG * NBATCH;
    return MNR_FRAG_WEAVE && TILE == 16 && OBB == 4 && T >= 2 && !SegSched<NBATCH, GPC, G0, T>::chunk_start(1);
}

// One K segment of a layer: NG groups of 4 steps whose B operands are b[0 .. 4*NG).
// G0 = index of the segment's first group inside the layer (chunk boundaries are static).
// PUB_END (woven form only): this is a layer's LAST segment and another layer follows -- its first chunk is published from here, two
// batches before the end, exactly like a chunk boundary inside the segment; the caller then skips the next layer's next_chunk().
// NOBF / OB0 (feature-split workgroups, mlp_fwd_split_body): the layer has NOBF output blocks per group in the weight stream and this
// wavefront computes the NOB blocks OB0 .. OB0 + NOB of them.
template <int TILE, int NOB, int NG, int GPC, int G0, bool PUB_END = false, int NOBF = NOB, int OB0 = 0, bool BOUNDARY_FIRST = false, class AccT, int NB, class Stream>
__device__ __forceinline__ void run_segment(AccT (&acc)[NOB], const float (&b)[NB], Stream &st, int lane) {
    static_assert(NB >= 4 * NG, "B register array too small");
    static_assert(!PUB_END || seg_weaves<TILE, NOB, NG, GPC, G0>(), "only the woven pipeline publishes ahead");
    static_assert(TILE == 16 || (NOBF == NOB && OB0 == 0), "feature split: 16-row tiles only");
    // Issue priority (the arbiter is oldest-first otherwise).  Forward kernels: a wavefront inside a K segment (the MFMA stream) outranks the SIMD's
    // other wavefront while that one is between segments (bias / ReLU / encodings, tape stores) -- render 1.922 -> 1.905 ms in every one of six
    // alternations, benchmark step 6.034 -> 6.019.  BOUNDARY_FIRST (the data-gradient chain, whose boundaries carry the mask arithmetic and the
    // gradient-tape stores): the other way round, the boundary outranks the stream -- chain 1.819 -> 1.800 ms (three alternations), while the
    // forward launches lose 1 % that way.  (-DMNR_NO_SETPRIO: neither.)
#ifndef MNR_NO_SETPRIO
    struct PrioGuard { __device__ ~PrioGuard() { __builtin_amdgcn_s_setprio(BOUNDARY_FIRST ? 3 : 0); } } prio_guard;
    __builtin_amdgcn_s_setprio(BOUNDARY_FIRST ? 1 : 2);
#endif
    if constexpr (TILE == 32) {
        static_for<0, NG>([&](auto gi) {
            constexpr int g = G0 + decltype(gi)::value;
            constexpr int gl = decltype(gi)::value;
            if constexpr (g % GPC == 0 && g > 0) st.next_chunk();
            const float4 *p = st.lds + st.cur * CHUNK_F4 + (g % GPC) * NOB * 64 + lane;
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) {
                const float4 a = p[ob * 64];
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * gl + 0], acc[ob], 0, 0, 0);
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * gl + 1], acc[ob], 0, 0, 0);
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * gl + 2], acc[ob], 0, 0, 0);
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * gl + 3], acc[ob], 0, 0, 0);
            }
        });
    } else {
        // 16x16x4: 32-cycle issue / 40-cycle dependent latency -> walk a batch of OBB blocks per k step.  The A fragments of MNR_FRAG_DEPTH
        // batches are in flight (asm reads at immediate offsets, hand-counted waits): batch t + 2 is requested as soon as the MFMAs of
        // batch t are issued -- across the groups of a chunk (SegSched), so only a chunk barrier restarts the pipeline.
        constexpr int OBB = NOB < 4 ? NOB : 4, NBATCH = NOB / OBB, T = NG * NBATCH;
        static_assert(NOB % OBB == 0, "NOB must be a multiple of the block batch");
        static_assert(GPC * NOBF * 1024 <= 65536, "fragment offsets must fit the ds_read immediate");
        using S = SegSched<NBATCH, GPC, G0, T>;
        unsigned addr = 0;
        // (a segment whose SECOND batch opens a chunk -- 64-wide test models only -- takes the restartable form below)
        if constexpr (seg_weaves<TILE, NOB, NG, GPC, G0>()) {
            // ONE software pipeline over the whole segment.  Three fragment buffers: batch t computes from one, batch t + 1 is in flight in
            // the second, batch t + 2 is requested -- one read behind each K step of batch t -- into the third (released by batch t - 1).
            // Chunk boundaries do not restart it: when batch t + 2 opens a new weight chunk, the chunk barrier is taken at the START of
            // batch t -- all that has to be true there is that the fragment reads of the old chunk (batches t, t + 1) have landed
            // (lgkmcnt(0)): the DMA the barrier releases may then overwrite the old buffer while the 32 MFMAs of t and t + 1 still run from
            // registers.  A wavefront therefore waits at the barrier with two batches of matrix work in hand instead of none, the DMA
            // requests of the next-but-one chunk are woven into batch t, and the first reads of the new chunk into batches t and t + 1
            // (hipcc's own schedule sank some MFMAs below the barrier the same way -- that is what round 4's kernels lived on).
            static_assert(GPC * NBATCH >= 2, "a chunk holds at least two batches");
            floatx4 a[3][OBB];
            if constexpr (S::chunk_start(0)) st.next_chunk();
            addr = lds_addr(st.lds + st.cur * CHUNK_F4 + lane);
            frag_load<G0 % GPC, OB0, NOBF>(a[0], addr);
            if constexpr (T > 1) frag_load<(G0 + 1 / NBATCH) % GPC, OB0 + (1 % NBATCH) * OBB, NOBF>(a[1], addr);
            static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, u = t + 2;
                constexpr int gl = t / NBATCH;
                constexpr bool early = (u < T && S::chunk_start(u)) || (PUB_END && u == T);
                if constexpr (early) {
                    wait_lgkm<0>();
                    st.publish();
                    addr = lds_addr(st.lds + st.cur * CHUNK_F4 + lane);
                } else if constexpr (t + 1 < T) {
                    // (behind an early barrier at t - 1 everything has landed already; the count is an upper bound either way)
                    wait_lgkm<OBB>();
                } else {
                    wait_lgkm<0>();
                }
                frag_mfmas_weave<(t % NBATCH) * OBB, (G0 + u / NBATCH) % GPC, OB0 + (u % NBATCH) * OBB, NOBF, (u < T), early>(
                    acc, a[t % 3], a[u % 3], addr, st, b[4 * gl], b[4 * gl + 1], b[4 * gl + 2], b[4 * gl + 3]);
            });
        } else {
            constexpr int D = MNR_FRAG_DEPTH;                  // fragment batches in flight
            static_assert(OBB * (D - 1) <= 15, "lgkmcnt is a 4-bit counter");
            floatx4 a[D][OBB];
            static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, t0 = S::run_start(t), t1 = S::run_end(t);
                constexpr int gl = t / NBATCH;
                if constexpr (t == t0) {
                    if constexpr (S::chunk_start(t)) st.next_chunk();
                    addr = lds_addr(st.lds + st.cur * CHUNK_F4 + lane);
                    static_for<0, D>([&](auto dc) {
                        constexpr int u = t + decltype(dc)::value;
                        if constexpr (u < t1) frag_load<(G0 + u / NBATCH) % GPC, OB0 + (u % NBATCH) * OBB, NOBF>(a[decltype(dc)::value], addr);
                    });
                }
                constexpr int newer = (t1 - 1 - t) < (D - 1) ? (t1 - 1 - t) : (D - 1);         // batches requested after this one, still in flight
                wait_lgkm<OBB * newer>();
                frag_mfmas<(t % NBATCH) * OBB>(acc, a[(t - t0) % D], b[4 * gl], b[4 * gl + 1], b[4 * gl + 2], b[4 * gl + 3]);
                if constexpr (t + D < t1) frag_load<(G0 + (t + D) / NBATCH) % GPC, OB0 + ((t + D) % NBATCH) * OBB, NOBF>(a[(t - t0) % D], addr);
            });
        }
    }
}

template <int NOB, int RPB, class AccT>
__device__ __forceinline__ void init_acc(AccT (&acc)[NOB], const float *bias_part) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int q = 0; q < RPB / 4; ++q) {
#ifdef MNR_EXPERIMENT_NO_BIAS          // timing experiment only (results invalid): what do the per-layer bias loads cost?
            const float4 v = make_float4(0.f, 0.f, 0.f, 0.f); (void)bias_part;
#else
            const float4 v = *reinterpret_cast<const float4 *>(bias_part + ob * RPB + 4 * q);
#endif
            acc[ob][4 * q + 0] = v.x; acc[ob][4 * q + 1] = v.y; acc[ob][4 * q + 2] = v.z; acc[ob][4 * q + 3] = v.w;
        }
    }
}

// max(x, 0) as one v_max_i32 on the bit pattern (negative floats are negative integers; -0 -> +0).  fmaxf costs two VALU
// instructions per element here (IEEE mode first canonicalises its operand), and VALU work inside the MFMA stream is not free.
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <int NOB, int RPB, bool RELU, class AccT, int NH>
__device__ __forceinline__ void acc_to_regs(float (&h)[NH], const AccT (&acc)[NOB]) {
    static_assert(NH >= NOB * RPB, "register array too small");
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < RPB; ++r) h[ob * RPB + r] = RELU ? relu_bits(acc[ob][r]) : acc[ob][r];
}

// Positional encoding of D coordinates into this lane's registers (layout: mlp_layout.h emb_src).
template <int D, int L, int P, int NE>
__device__ __forceinline__ void embed(float (&e)[NE], const float (&x)[D], int part) {
    constexpr int NP = emb_pairs(D, L, P);
    static_assert(NE == emb_regs(D, L, P), "embedding register count");
    float xs[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xs[d] = ldexpf(x[d], part * (L / P));     // exact: power-of-two scale
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float arg = xs[i % D] * (float)(1 << (i / D));              // == fl(2^f * x), nerf.py:22-23
        float s, c;
        sincosf(arg, &s, &c);
        e[2 * i] = s;
        e[2 * i + 1] = c;
    }
#pragma unroll
    for (int j = 2 * NP; j < NE; ++j) {
        const int dim = (j - 2 * NP) * P + part;
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) v = (dim == d && (j - 2 * NP) < cdiv(D, P)) ? x[d] : v;
        e[j] = v;
    }
}

template <int P>
__device__ __forceinline__ float reduce_parts(float v) {
    v += __shfl_xor(v, 32);
    if constexpr (P == 4) v += __shfl_xor(v, 16);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_shifted(float x) {   // F.softplus(x - 1, beta=1, threshold=20), nerf.py:38
    const float y = x - 1.f;
    return y > 20.f ? y : log1pf(expf(y));
}

}  // namespace mnr
