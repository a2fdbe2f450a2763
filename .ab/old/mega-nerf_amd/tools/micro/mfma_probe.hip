// mfma_probe.hip -- microbenchmark of the weight-gradient inner loop's building blocks on gfx950 (diagnostics only; not part
// of libmeganerf_hip.so).  512-thread workgroups (2 waves per SIMD), one per CU, 128 x v_mfma_f32_32x32x2_f32 per "tile":
//   bit 0: LDS fragment reads (6 ds_read_b32 per 8 MFMAs, one k-pair ahead, inline asm)
//   bit 1: raw s_barrier per tile
//   bit 2: LDS-DMA of a 64 KiB tile per tile (2-stage ring), waited at the tile boundary
//   bit 3: DMA burst at tile start instead of one piece per k-pair
// Prints ms and TFLOP/s per variant.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

template <int OFF>
__device__ __forceinline__ float lds_ld(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

constexpr int STAGE_FLOATS = 32 * 512 + 256;

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void k_probe(const float *src, float *out, int tiles) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 2) & 1, wc = wave & 3, i32 = lane & 31, kk = lane >> 5;
    floatx16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = floatx16(0.f);
    for (int i = threadIdx.x; i < 2 * STAGE_FLOATS; i += NW * 64) lds[i] = 1e-3f * (i & 15);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)lds;
    const unsigned a_off = (kk * 256 + wr * 128 + i32) * 4, b_off = (32 * 256 + kk * 256 + wc * 64 + i32) * 4;
    const float *tsrc = src + (size_t)blockIdx.x * tiles * 16384;
    float af[2][4], bf[2][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) af[0][m] = af[1][m] = 1.f + lane * 1e-3f;
#pragma unroll
    for (int n = 0; n < 2; ++n) bf[0][n] = bf[1][n] = 1.f - lane * 1e-3f;
    auto dma_piece = [&](int p, const float *t, int stage) {
        const float *s = t + (p * NW * 64 + threadIdx.x) * 4;
        float *dst = lds + stage * STAGE_FLOATS + (p * NW * 64 + wave * 64) * 4;
        __builtin_amdgcn_global_load_lds((global_cvoid_t *)s, (lds_void_t *)dst, 16, 0, 0);
    };
    constexpr int PIECES = 4096 / (NW * 64);
    if constexpr (MODE & 4) {
        for (int p = 0; p < PIECES; ++p) dma_piece(p, tsrc, 0);
    }
    int s = 0;
    for (int t = 0; t < tiles; ++t) {
        if constexpr (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (MODE & 2) __builtin_amdgcn_s_barrier();
        const float *nsrc = tsrc + (size_t)(t + 1 < tiles ? t + 1 : t) * 16384;
        const unsigned sb = base + (s ? STAGE_FLOATS * 4 : 0);
        const unsigned ab = sb + a_off, bb = sb + b_off;
        if constexpr ((MODE & 12) == 12) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) dma_piece(p, nsrc, s ^ 1);
        }
        auto frag_read = [&](auto kpc, auto bufc) {
            constexpr int kp = decltype(kpc)::value, buf = decltype(bufc)::value;
            static_for<0, 4>([&](auto mc) { constexpr int m = decltype(mc)::value; af[buf][m] = lds_ld<(2 * kp * 256 + m * 32) * 4>(ab); });
            static_for<0, 2>([&](auto nc) { constexpr int n = decltype(nc)::value; bf[buf][n] = lds_ld<(2 * kp * 256 + n * 32) * 4>(bb); });
        };
        if constexpr (MODE & 1) frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 16>([&](auto kpc) {
            constexpr int kp = decltype(kpc)::value, cur = kp & 1;
            if constexpr (MODE & 1) {
                if constexpr (kp + 1 < 16) { frag_read(std::integral_constant<int, kp + 1>{}, std::integral_constant<int, cur ^ 1>{}); wait_lgkm<6>(); }
                else wait_lgkm<0>();
#pragma unroll
                for (int m = 0; m < 4; ++m) pin(af[cur][m]);
#pragma unroll
                for (int n = 0; n < 2; ++n) pin(bf[cur][n]);
            }
            if constexpr ((MODE & 12) == 4) {
                if constexpr (kp < PIECES) dma_piece(kp, nsrc, s ^ 1);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][m], bf[cur][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        s ^= 1;
    }
    if constexpr (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) r += acc[m][n][i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = r;
}

template <int MODE, int NW>
static void run(const float *src, float *out, int tiles, int wgs) {
    const size_t ldsb = 2 * STAGE_FLOATS * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_probe<MODE, NW>), dim3(wgs), dim3(NW * 64), ldsb, 0, src, out, tiles);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    const double flop = (double)wgs * tiles * NW * 128.0 * 32 * 32 * 2 * 2;
    printf("mode %2d waves %d: %.3f ms  %.1f TFLOP/s (%s%s%s%s)\n", MODE, NW, best, flop / best / 1e9, MODE & 1 ? "lds-reads " : "",
           MODE & 2 ? "barrier " : "", MODE & 4 ? "dma " : "", MODE & 8 ? "burst" : "");
}

int main(int argc, char **argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 300, wgs = 256;
    float *src, *out;
    hipMalloc(&src, (size_t)wgs * tiles * 16384 * 4);
    hipMalloc(&out, (size_t)wgs * 512 * 4);
    hipMemset(src, 0, (size_t)wgs * tiles * 16384 * 4);
    std::vector<float> h((size_t)1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) & 1023) / 512.f - 1.f;
    for (size_t o = 0; o < (size_t)wgs * tiles * 16384; o += h.size()) hipMemcpy(src + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 8>(src, out, tiles, wgs);
    run<0, 4>(src, out, tiles, wgs);
    run<1, 8>(src, out, tiles, wgs);
    run<3, 8>(src, out, tiles, wgs);
    run<7, 8>(src, out, tiles, wgs);
    run<15, 8>(src, out, tiles, wgs);
    run<6, 8>(src, out, tiles, wgs);
    run<4, 8>(src, out, tiles, wgs);
    return 0;
}
