// calibrate.hip -- device calibration probes for bench.py (mnr_calibrate): what THIS box delivers on the resources the register-chained
// MLP kernels live on, measured in the same process right before / after a timed region, so that a slow line can be told apart from a
// slow box:
//   * fp32 MFMA rate with the chip full (512 workgroups x 4 wavefronts, two wavefronts per SIMD: the forward kernel's occupancy);
//   * the weight stream's shape: 512 workgroups pulling the SAME 2.4 MB image L2 -> LDS in 32 KiB chunks through `global_load_lds_dwordx4`,
//     free-running (two chunks in flight: bandwidth) and one chunk at a time (the round trip a chunk barrier can expose);
//   * dependent-load latency of an L2-resident, a MALL-resident and an HBM-resident line set (one wavefront, pointer chase);
//   * HBM streaming read / write bandwidth;
//   * the shader clock seen by ONE wavefront (a dependent v_fma / v_mfma chain against the constant 100 MHz counter) -- and, from the
//     MFMA probe, the clock the matrix pipes actually hold with every CU busy.
// Diagnostics only: no kernel of the hot path calls anything here.  UNLIKE every other entry point mnr_calibrate synchronises (it
// times its own launches with HIP events).
#include <algorithm>
#include <vector>

#include "common.h"

namespace mnr {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

constexpr int CAL_WGS = 512, CAL_NT = 256;
constexpr int CHUNK_BYTES = 32768, IMAGE_CHUNKS = 75;            // 2.4 MB: one 8 x 256 model's packed weight image

// ---- fp32 MFMA, chip full -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CAL_NT, 2) void k_cal_mfma(float *out, int iters, unsigned long long *wg_ticks, unsigned *wg_where) {
    const unsigned long long t_begin = wall_clock64();
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = 1.f + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[blockIdx.x * CAL_NT + threadIdx.x] = s;       // (never true: keeps the chain alive)
    __syncthreads();
    if (threadIdx.x == 0 && wg_ticks) {
        wg_ticks[2 * blockIdx.x] = t_begin;
        wg_ticks[2 * blockIdx.x + 1] = wall_clock64();
        // HW_REG_XCC_ID (20) and HW_REG_HW_ID (4), all 32 bits: which XCD / shader engine / CU the workgroup ran on
        wg_where[blockIdx.x] = ((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) | ((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu);
    }
}

// ---- the weight stream: every workgroup streams the same image global -> LDS -------------------------------------------------------
template <int DEPTH>      // chunks in flight: 2 = the kernels' ring, free-running; 1 = request, wait, barrier (round trip per chunk)
__global__ __launch_bounds__(CAL_NT, 2) void k_cal_dma(const float4 *image, float *out, int passes) {
    extern __shared__ float4 ring[];                                         // 2 x 32 KiB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane_off = threadIdx.x * 16u;
    auto issue = [&](int chunk, int buf) {
        const char *g = reinterpret_cast<const char *>(image) + (size_t)chunk * CHUNK_BYTES;
        float4 *dst = ring + buf * (CHUNK_BYTES / 16) + wave * 64;
#pragma unroll
        for (int i = 0; i < CHUNK_BYTES / 16 / CAL_NT; ++i)
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(g + (size_t)i * CAL_NT * 16 + lane_off), (lds_void_t *)(dst + i * CAL_NT), 16, 0, 0);
    };
    int c = 0;
    if (DEPTH == 2) issue(0, 0);
    for (int p = 0; p < passes; ++p) {
        for (int k = 0; k < IMAGE_CHUNKS; ++k, ++c) {
            if (DEPTH == 2) {
                issue((k + 1) % IMAGE_CHUNKS, (c + 1) & 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // chunk c has landed, chunk c + 1 (8 requests) is in flight
            } else {
                issue(k, c & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (passes < 0) out[threadIdx.x] = reinterpret_cast<float *>(ring)[threadIdx.x];
}

// ---- dependent-load latency ----------------------------------------------------------------------------------------------------------
// one 8-byte slot per 128-byte line; slot i holds the index of the next line: a full-period LCG over n_lines (a power of two)
__global__ void k_cal_chase_init(unsigned long long *buf, unsigned n_lines) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_lines) buf[(size_t)i * 16] = (i * 1664525u + 1013904223u) & (n_lines - 1);
}
__global__ void k_cal_chase(const unsigned long long *buf, unsigned warm_steps, unsigned steps, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    unsigned long long i = 0;
    for (unsigned s = 0; s < warm_steps; ++s) i = buf[i * 16];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = wall_clock64();
    for (unsigned s = 0; s < steps; ++s) i = buf[i * 16];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = i;
}

// ---- HBM streams ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cal_read(const float4 *src4, size_t n4, float *out) {
    const floatx4 *src = reinterpret_cast<const floatx4 *>(src4);
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const floatx4 v = __builtin_nontemporal_load(&src[i]);
        s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cal_write(float4 *dst4, size_t n4) {
    floatx4 *dst = reinterpret_cast<floatx4 *>(dst4);
    const floatx4 v = floatx4{1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &dst[i]);
}

// ---- shader clock seen by one wavefront ---------------------------------------------------------------------------------------------
__global__ void k_cal_clock(unsigned long long *out, int iters) {
    float x = 1.f + threadIdx.x * 1e-3f;
    const float a = 0.999f, b = 1e-4f;
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    unsigned long long t1 = wall_clock64();
    floatx16 acc = floatx16(0.f);
    float am = x * 1e-6f;
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    unsigned long long t2 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(am, b, acc, 0, 0, 0);
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" : "+v"(acc));
    unsigned long long t3 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = t3 - t2;
    }
    if (acc[0] == 12345.678f) out[2] = (unsigned long long)x;
}

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); ms = -1.f; }
    return ms;
}

}  // namespace
}  // namespace mnr

using namespace mnr;

__global__ __launch_bounds__(256) void k_cal_hog(floatx4 *buf, size_t n4, int passes, float *out) {
    float s = 0.f;
    for (int p = 0; p < passes; ++p) {
        const floatx4 v = floatx4{1.f + p, 2.f, 3.f, 4.f};
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &buf[i]);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const floatx4 r = __builtin_nontemporal_load(&buf[i]);
            s += r[0] + r[3];
        }
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" size_t mnr_calibrate_scratch_bytes(void) { return (size_t)1 << 30; }

extern "C" int mnr_calibrate_hog(void *scratch_dev, size_t bytes, int workgroups, int passes, void *stream) {
    MNR_REQUIRE(scratch_dev && bytes >= (1 << 20) && workgroups >= 1 && workgroups <= 4096 && passes >= 1, "bad arguments to mnr_calibrate_hog");
    char *base = static_cast<char *>(scratch_dev);
    hipLaunchKernelGGL(k_cal_hog, dim3(workgroups), dim3(256), 0, as_stream(stream), reinterpret_cast<floatx4 *>(base + (1 << 20)), (bytes - (1 << 20)) / 16,
                       passes, reinterpret_cast<float *>(base));
    return check_launch("k_cal_hog");
}

extern "C" int mnr_calibrate(mnr_calibration *out, void *scratch_dev, size_t scratch_bytes, void *stream) {
    MNR_REQUIRE(out && scratch_dev, "NULL argument to mnr_calibrate");
    MNR_REQUIRE(scratch_bytes >= ((size_t)64 << 20), "mnr_calibrate: at least 64 MiB of scratch");
    hipStream_t s = as_stream(stream);
    *out = mnr_calibration{};
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipGetDeviceProperties failed");
    out->cu_count = prop.multiProcessorCount;
    out->nominal_sclk_mhz = prop.clockRate / 1000.f;
    out->nominal_mclk_mhz = prop.memoryClockRate / 1000.f;
    out->l2_bytes = (int64_t)prop.l2CacheSize;
    int wall_khz = 0;
    if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) { (void)hipGetLastError(); wall_khz = 100000; }
    const double tick_ns = 1e6 / (wall_khz > 0 ? wall_khz : 100000);
    hipEvent_t ev[2];
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipEventCreate failed");
    auto timed = [&](auto &&launch, int reps) {
        launch();                                        // warm-up (code load, clocks)
        (void)hipEventRecord(ev[0], s);
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(ev[1], s);
        (void)hipEventSynchronize(ev[1]);
        return elapsed(ev[0], ev[1]) / reps;
    };
    char *base = static_cast<char *>(scratch_dev);
    float *sink = reinterpret_cast<float *>(base);                                    // 512 KiB of never-written outputs
    unsigned long long *ticks = reinterpret_cast<unsigned long long *>(base + (1 << 20));
    const float4 *image = reinterpret_cast<const float4 *>(base + (2 << 20));           // 2.4 MB "weight image" (contents irrelevant)
    char *big = base + (8 << 20);
    const size_t big_bytes = scratch_bytes - (8 << 20);

    // fp32 MFMA, chip full
    {
        const int iters = 1024;
        unsigned long long *wg_ticks = ticks + 64;                                  // [CAL_WGS][2]
        unsigned *wg_where = reinterpret_cast<unsigned *>(wg_ticks + 2 * CAL_WGS);    // [CAL_WGS]
        const float ms = timed([&] { hipLaunchKernelGGL(k_cal_mfma, dim3(CAL_WGS), dim3(CAL_NT), 0, s, sink, iters, wg_ticks, wg_where); }, 2);
        {
            std::vector<unsigned long long> ht(2 * CAL_WGS);
            std::vector<unsigned> hw(CAL_WGS);
            if (hipMemcpyAsync(ht.data(), wg_ticks, ht.size() * 8, hipMemcpyDeviceToHost, s) == hipSuccess &&
                hipMemcpyAsync(hw.data(), wg_where, hw.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
                std::vector<double> d(CAL_WGS);
                unsigned long long t0 = ~0ull, t1 = 0;
                double xs[64] = {0}; int xn[64] = {0};
                int worst = 0;
                for (int i = 0; i < CAL_WGS; ++i) {
                    d[i] = (double)(ht[2 * i + 1] - ht[2 * i]) * tick_ns * 1e-6;
                    t0 = ht[2 * i] < t0 ? ht[2 * i] : t0;
                    t1 = ht[2 * i] > t1 ? ht[2 * i] : t1;
                    const int x = (int)((hw[i] >> 16) & 63);
                    xs[x] += d[i]; xn[x]++;
                    if (d[i] > d[worst]) worst = i;
                }
                out->mfma_slowest_wg_where = (int32_t)hw[worst];
                std::vector<double> sd(d);
                std::sort(sd.begin(), sd.end());
                out->mfma_wg_ms_min = (float)sd.front(); out->mfma_wg_ms_median = (float)sd[CAL_WGS / 2]; out->mfma_wg_ms_max = (float)sd.back();
                double lo = 1e30, hi = 0;
                for (int x = 0; x < 64; ++x)
                    if (xn[x]) { const double m = xs[x] / xn[x]; lo = m < lo ? m : lo; hi = m > hi ? m : hi; }
                out->mfma_xcd_ms_fastest = (float)lo; out->mfma_xcd_ms_slowest = (float)hi;
                out->mfma_start_skew_us = (float)((double)(t1 - t0) * tick_ns * 1e-3);
            } else {
                (void)hipGetLastError();
            }
        }
        const double flop = (double)CAL_WGS * 4 * iters * 16 * 2048.0;
        out->mfma_f32_tflops = (float)(flop / (ms * 1e-3) / 1e12);
        // 64 FLOP per clock per SIMD: the clock the matrix pipes held while every CU was busy
        out->sclk_mhz_under_mfma_load = (float)(flop / (ms * 1e-3) / (64.0 * 4 * out->cu_count) / 1e6);
    }
    // weight-stream shape
    {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_cal_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CHUNK_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_cal_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CHUNK_BYTES);
        const int passes = 4;
        const double bytes = (double)CAL_WGS * passes * IMAGE_CHUNKS * CHUNK_BYTES;
        float ms = timed([&] { hipLaunchKernelGGL(k_cal_dma<2>, dim3(CAL_WGS), dim3(CAL_NT), 2 * CHUNK_BYTES, s, image, sink, passes); }, 2);
        out->dma_stream_gbps = (float)(bytes / (ms * 1e-3) / 1e9);
        ms = timed([&] { hipLaunchKernelGGL(k_cal_dma<1>, dim3(CAL_WGS), dim3(CAL_NT), 2 * CHUNK_BYTES, s, image, sink, passes); }, 2);
        out->dma_chunk_round_trip_us = (float)(ms * 1e3 / (passes * IMAGE_CHUNKS));
        // the same round trip with ONE workgroup on the chip: the uncontended latency of a 32 KiB chunk
        ms = timed([&] { hipLaunchKernelGGL(k_cal_dma<1>, dim3(1), dim3(CAL_NT), 2 * CHUNK_BYTES, s, image, sink, passes); }, 2);
        out->dma_chunk_round_trip_alone_us = (float)(ms * 1e3 / (passes * IMAGE_CHUNKS));
    }
    // dependent-load latency: 2 MiB (L2), 64 MiB (MALL), the whole scratch (HBM)
    {
        struct { unsigned lines; unsigned warm; float *dst; } sets[4] = {
            {1u << 6, 1u << 6, &out->chase_l1_ns}, {1u << 11, 1u << 11, &out->chase_l2_ns}, {1u << 19, 0, &out->chase_mall_ns}, {0, 0, &out->chase_hbm_ns}};
        // HBM set: half of the scratch; the other half is streamed over afterwards so that the memory-side cache holds none of it
        unsigned hb = 1u << 19;
        while ((size_t)hb * 2 * 128 <= big_bytes / 2 && hb < (1u << 24)) hb *= 2;
        sets[3].lines = hb;
        for (auto &st : sets) {
            if ((size_t)st.lines * 128 > big_bytes) continue;
            hipLaunchKernelGGL(k_cal_chase_init, dim3((st.lines + 255) / 256), dim3(256), 0, s, reinterpret_cast<unsigned long long *>(big), st.lines);
            if (st.dst == &out->chase_mall_ns)       // MALL set: stream it once so that it sits in the memory-side cache
                hipLaunchKernelGGL(k_cal_read, dim3(1024), dim3(256), 0, s, reinterpret_cast<const float4 *>(big), (size_t)st.lines * 8, sink);
            if (st.dst == &out->chase_hbm_ns)        // HBM set: push it out of the memory-side cache
                hipLaunchKernelGGL(k_cal_write, dim3(4096), dim3(256), 0, s, reinterpret_cast<float4 *>(big + (size_t)st.lines * 128),
                                   (big_bytes - (size_t)st.lines * 128) / 16);
            const unsigned steps = 4096;
            hipLaunchKernelGGL(k_cal_chase, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long *>(big), st.warm, steps, ticks);
            unsigned long long h[2] = {0, 0};
            if (hipMemcpyAsync(h, ticks, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            *st.dst = (float)(h[0] * tick_ns / steps);
        }
    }
    // HBM streams
    {
        const size_t n4 = big_bytes / 16;
        float ms = timed([&] { hipLaunchKernelGGL(k_cal_write, dim3(4096), dim3(256), 0, s, reinterpret_cast<float4 *>(big), n4); }, 2);
        out->hbm_write_gbps = (float)(n4 * 16.0 / (ms * 1e-3) / 1e9);
        ms = timed([&] { hipLaunchKernelGGL(k_cal_read, dim3(4096), dim3(256), 0, s, reinterpret_cast<const float4 *>(big), n4, sink); }, 2);
        out->hbm_read_gbps = (float)(n4 * 16.0 / (ms * 1e-3) / 1e9);
    }
    // one wavefront's clock
    {
        const int iters = 256;
        hipLaunchKernelGGL(k_cal_clock, dim3(1), dim3(64), 0, s, ticks, iters);        // warm-up
        hipLaunchKernelGGL(k_cal_clock, dim3(1), dim3(64), 0, s, ticks, iters);
        unsigned long long h[2] = {0, 0};
        if (hipMemcpyAsync(h, ticks, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
            // dependent v_fma_f32: 4 cycles each on a 16-lane SIMD; dependent v_mfma_f32_32x32x2_f32: 16 passes = 64 cycles each
            if (h[0]) out->sclk_mhz_fma_chain = (float)(iters * 64 * 4.0 / (h[0] * tick_ns) * 1e3);
            if (h[1]) out->sclk_mhz_mfma_chain = (float)(iters * 16 * 64.0 / (h[1] * tick_ns) * 1e3);
        } else {
            (void)hipGetLastError();
        }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return check_launch("mnr_calibrate");
}
