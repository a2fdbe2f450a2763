// chain_probe.hip -- microbenchmark of the register-chained MLP inner loop on gfx950 (diagnostics only; not part of
// libmeganerf_hip.so).  One wavefront = 16 samples x 256 features, 4 waves per workgroup share a 2 x 32 KiB LDS ring of
// MFMA-fragment-ordered weights (LDS-DMA), NL layers of 256 -> 256 with ReLU, v_mfma_f32_16x16x4_f32.
//   MODE 0: the round-1 structure (compiler-visible ds_read_b128, __syncthreads per chunk -> hipcc drains vmcnt before
//           every first read behind an LDS-DMA)
//   MODE 1: inline-asm ds_read_b128 one batch ahead, raw s_barrier, s_waitcnt vmcnt(0) only in front of the barrier
//   MODE 2: MODE 1 + activation-tape stores (16 float4 + 1 mask word pair per lane and layer), counted vmcnt
//   MODE 3: MODE 2 with the ReLU / tape work of a layer fenced off from the MFMA stream (sched_barrier)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <int OFF>
__device__ __forceinline__ floatx4 lds_ld128(unsigned addr) {
    floatx4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(floatx4 &x) { asm volatile("" : "+v"(x)); }

constexpr int CHUNK_F4 = 2048, NOB = 16, H = 64, GPC = 2;   // 32 KiB chunks; 16 output blocks; 64 hidden registers per lane

template <int MODE, int TAPE = 0>
__global__ __launch_bounds__(256, 2) void k_chain(const float4 *chunks, float *out, float *tape, long tape_rows, int nl) {
    extern __shared__ float4 ring[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane >> 4;
    const long row = ((long)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    float h[H];
#pragma unroll
    for (int i = 0; i < H; ++i) h[i] = 0.01f * ((lane + i) & 31) - 0.1f;
    const float4 *g = chunks + threadIdx.x;
    int cur = 1;
    auto issue = [&]() {
        float4 *dst = ring + (cur ^ 1) * CHUNK_F4 + wave * 64;
#pragma unroll
        for (int i = 0; i < CHUNK_F4 / 256; ++i)
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(g + i * 256), (lds_void_t *)(dst + i * 256), 16, 0, 0);
        g += CHUNK_F4;
    };
    issue();
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)ring;
    for (int l = 0; l < nl; ++l) {
        floatx4 acc[NOB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) acc[ob] = floatx4(0.01f);
        static_for<0, H / 4 / GPC>([&](auto cc) {                 // 8 chunks per layer
            constexpr int c = decltype(cc)::value;
            if constexpr (MODE == 0) {
                __syncthreads();
                cur ^= 1;
                issue();
                if constexpr (TAPE == 32 || TAPE == 33) {
                    // store-cost scaling: TAPE 32 = 16 x dwordx2 (same instruction count, half the bytes), TAPE 33 = 8 x dwordx4
                    // (half the instructions, half the bytes), both in chunk 0
                    if (c == 0 && l > 0) {
                        float *r = tape + ((long)(l & 7) * tape_rows + row) * 256 + 4 * part;
                        if constexpr (TAPE == 32) {
#pragma unroll
                            for (int q = 0; q < H / 4; ++q) *reinterpret_cast<float2 *>(r + 16 * q) = make_float2(h[4 * q], h[4 * q + 1]);
                        } else {
#pragma unroll
                            for (int q = 0; q < H / 8; ++q) *reinterpret_cast<float4 *>(r + 16 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                        }
                    }
                }
                if constexpr (TAPE == 8 || TAPE == 16) {
                    // the same 16 float4 stores, two per chunk (TAPE 8) / four in each of the first four chunks (TAPE 16)
                    if (l > 0) {
                        float *r = tape + ((long)(l & 7) * tape_rows + row) * 256 + 4 * part;
                        constexpr int per = TAPE == 8 ? 2 : 4;
                        if constexpr (c * per < H / 4) {
#pragma unroll
                            for (int q = c * per; q < c * per + per; ++q)
                                *reinterpret_cast<float4 *>(r + 16 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                        }
                    }
                }
                if constexpr (c == 0 && TAPE != 0 && TAPE < 8) {
                    // TAPE bit 0: 16 float4 stores of the previous layer's output; bit 1: packed sign-bit words (v_cmp + or per
                    // register, one 8-byte store); bit 2: sign bits as v_cmp lane masks stored with scalar stores
                    if (l > 0) {
                        float *r = tape + ((long)(l & 7) * tape_rows + row) * 256 + 4 * part;
                        if constexpr (TAPE == 7) {          // non-temporal stores: keep the tape out of the L2 the weights live in
#pragma unroll
                            for (int q = 0; q < H / 4; ++q) {
                                floatx4 v = {h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]};
                                __builtin_nontemporal_store(v, reinterpret_cast<floatx4 *>(r + 16 * q));
                            }
                        } else if constexpr (TAPE & 1) {
#pragma unroll
                            for (int q = 0; q < H / 4; ++q)
                                *reinterpret_cast<float4 *>(r + 16 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                        }
                        if constexpr ((TAPE & 2) && TAPE != 7) {
                            unsigned m0 = 0, m1 = 0;
#pragma unroll
                            for (int i = 0; i < 32; ++i) { m0 |= h[i] > 0.f ? (1u << i) : 0u; m1 |= h[32 + i] > 0.f ? (1u << i) : 0u; }
                            *reinterpret_cast<uint2 *>(tape + (8 * tape_rows) * 256 + ((long)(l & 7) * tape_rows + row) * 8 + part * 2) = make_uint2(m0, m1);
                        }
                        if constexpr ((TAPE & 4) && TAPE != 7) {
                            const size_t mpv = (size_t)(reinterpret_cast<unsigned long long *>(tape + (8 * tape_rows) * 256) +
                                                        ((long)(l & 7) * (tape_rows / 16) + ((long)blockIdx.x * 4 + wave)) * 64);
                            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mpv), hi = __builtin_amdgcn_readfirstlane((unsigned)(mpv >> 32));
                            const unsigned long long mp = ((unsigned long long)hi << 32) | lo;
#pragma unroll
                            for (int i = 0; i < H / 2; ++i) {
                                const unsigned long long b0 = __ballot(h[2 * i] > 0.f), b1 = __ballot(h[2 * i + 1] > 0.f);
                                typedef unsigned uint4v __attribute__((ext_vector_type(4)));
                                uint4v v = {(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
                                const unsigned off = (unsigned)(i * 16);
                                asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(mp), "s"(off) : "memory");
                            }
                        }
                    }
                }
                static_for<0, GPC>([&](auto gc) {
                    constexpr int gl = c * GPC + decltype(gc)::value;
                    const float4 *p = ring + cur * CHUNK_F4 + decltype(gc)::value * NOB * 64 + lane;
#pragma unroll
                    for (int o0 = 0; o0 < NOB; o0 += 4) {
                        float4 a[4];
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob) a[ob] = p[(o0 + ob) * 64];
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].x, h[4 * gl + 0], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].y, h[4 * gl + 1], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].z, h[4 * gl + 2], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                        for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].w, h[4 * gl + 3], acc[o0 + ob], 0, 0, 0);
                    }
                });
            } else {
                // tape stores of the previous layer are issued in chunk 0 (behind the DMA): 17 younger VMEM ops
                if constexpr (c == 1 && MODE >= 2) wait_vm<17>(); else wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                cur ^= 1;
                issue();
                if constexpr (c == 0 && MODE >= 2) {
                    if (l > 0) {
                        float *r = tape + ((long)(l & 7) * tape_rows + row) * 256 + 4 * part;
                        unsigned m0 = 0, m1 = 0;
#pragma unroll
                        for (int q = 0; q < H / 4; ++q) {
                            *reinterpret_cast<float4 *>(r + 16 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
                        }
#pragma unroll
                        for (int i = 0; i < 32; ++i) { m0 |= h[i] > 0.f ? (1u << i) : 0u; m1 |= h[32 + i] > 0.f ? (1u << i) : 0u; }
                        *reinterpret_cast<uint2 *>(tape + (8 * tape_rows) * 256 + ((long)(l & 7) * tape_rows + row) * 8 + part * 2) = make_uint2(m0, m1);
                    } else {
                        // keep the VMEM op count per layer constant (counted vmcnt above): 17 dummy stores of layer 0
                        float *r = tape + ((long)0 * tape_rows + row) * 256 + 4 * part;
#pragma unroll
                        for (int q = 0; q < H / 4; ++q) *reinterpret_cast<float4 *>(r + 16 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
                        *reinterpret_cast<uint2 *>(tape + (8 * tape_rows) * 256 + ((long)0 * tape_rows + row) * 8 + part * 2) = make_uint2(0u, 0u);
                    }
                    if constexpr (MODE == 3) __builtin_amdgcn_sched_barrier(0);
                }
                const unsigned base = ring_base + (unsigned)(cur * CHUNK_F4 * 16 + lane * 16);
                floatx4 a[2][4];
                auto rd = [&](auto bc, auto bufc) {
                    constexpr int b = decltype(bc)::value, buf = decltype(bufc)::value;     // batch b of the chunk: group b / 4, blocks 4 (b % 4) ..
                    static_for<0, 4>([&](auto oc) {
                        constexpr int ob = decltype(oc)::value;
                        a[buf][ob] = lds_ld128<((b / 4) * NOB * 64 + ((b % 4) * 4 + ob) * 64) * 16>(base);
                    });
                };
                rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                static_for<0, 8>([&](auto bc) {
                    constexpr int b = decltype(bc)::value, cb = b & 1, gl = c * GPC + b / 4, o0 = (b % 4) * 4;
                    if constexpr (b + 1 < 8) { rd(std::integral_constant<int, b + 1>{}, std::integral_constant<int, cb ^ 1>{}); wait_lgkm<4>(); }
                    else wait_lgkm<0>();
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) pin(a[cb][ob]);
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][ob][0], h[4 * gl + 0], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][ob][1], h[4 * gl + 1], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][ob][2], h[4 * gl + 2], acc[o0 + ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) acc[o0 + ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][ob][3], h[4 * gl + 3], acc[o0 + ob], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[ob * 4 + r] = fmaxf(acc[ob][r], 0.f) * 0.05f;
        if constexpr (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE != 0) wait_vm<0>();
    if constexpr ((TAPE & 4) && TAPE != 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < H; ++i) s += h[i];
    out[row * 4 + part] = s;
}

template <int MODE, int TAPE = 0>
static void run(const float4 *chunks, float *out, float *tape, long rows, int nl, int lds_bytes = 2 * CHUNK_F4 * 16) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain<MODE, TAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_chain<MODE, TAPE>), dim3(rows / 64), dim3(256), lds_bytes, 0, chunks, out, tape, rows, nl);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = (double)rows * nl * 256.0 * 256.0 * 2.0;
    printf("mode %d tape %d lds %3d KB: %.3f ms  %.1f TFLOP/s  (%s)\n", MODE, TAPE, lds_bytes / 1024, best, flop / best / 1e9, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char **argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 131072;
    const int nl = argc > 2 ? atoi(argv[2]) : 9;
    float4 *chunks;
    float *out, *tape;
    const size_t wbytes = (size_t)(nl * 8 + 2) * CHUNK_F4 * 16;
    (void)hipMalloc(&chunks, wbytes);
    (void)hipMalloc(&out, rows * 4 * 4);
    (void)hipMalloc(&tape, (size_t)rows * (8 * 256 + 8 * 8) * 4);
    std::vector<float> h(wbytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((float)((i * 2654435761u) & 1023) / 512.f - 1.f) * 0.06f;
    (void)hipMemcpy(chunks, h.data(), wbytes, hipMemcpyHostToDevice);
    if (argc > 3) {        // sustained-load check: many back-to-back launches, throughput per group of 20
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int grp = 0; grp < atoi(argv[3]); ++grp) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i)
                hipLaunchKernelGGL((k_chain<0, 0>), dim3(rows / 64), dim3(256), 2 * CHUNK_F4 * 16, 0, chunks, out, tape, rows, nl);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("group %2d: %.3f ms per launch  %.1f TFLOP/s\n", grp, ms / 20, (double)rows * nl * 256.0 * 256.0 * 2.0 / (ms / 20) / 1e9);
        }
        return 0;
    }
    for (int round = 0; round < 3; ++round) {
        printf("-- round %d\n", round);
        run<0>(chunks, out, tape, rows, nl);
        run<0, 1>(chunks, out, tape, rows, nl);
        run<0, 2>(chunks, out, tape, rows, nl);
        run<0, 3>(chunks, out, tape, rows, nl);
        run<0, 4>(chunks, out, tape, rows, nl);
        run<0, 5>(chunks, out, tape, rows, nl);
        run<0, 32>(chunks, out, tape, rows, nl);
        run<0, 33>(chunks, out, tape, rows, nl);
        run<1>(chunks, out, tape, rows, nl);
        run<1>(chunks, out, tape, rows, nl, 96 * 1024);
    }
    return 0;
}
