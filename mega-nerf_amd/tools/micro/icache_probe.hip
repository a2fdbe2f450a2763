// icache_probe.hip -- does STRAIGHT-LINE matrix code (the register-chained MLP kernels are ~150 KB of unrolled instructions per body, far
// beyond the 64 KB instruction cache a CU pair shares) run slower than the same MFMAs in a loop, and how much does a cold L2 / MALL
// (the tapes of a training step evict everything between two launches) cost a launch?  Diagnostics only.
//   ./icache_probe            prints one JSON line per variant
// Variants: MFMA loop (16 per iteration) vs `.rept`-unrolled bodies of 32 / 64 / 256 / 1024 KiB, grids of 512 (every workgroup starts at the
// same instant, as in a 1024-ray launch) and 4096 workgroups, warm (back to back) and cold (behind a 2 GiB streaming write).
// Build: hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MFMA16 \
    "v_mfma_f32_16x16x4_f32 %0, %16, %17, %0\n v_mfma_f32_16x16x4_f32 %1, %16, %17, %1\n v_mfma_f32_16x16x4_f32 %2, %16, %17, %2\n" \
    "v_mfma_f32_16x16x4_f32 %3, %16, %17, %3\n v_mfma_f32_16x16x4_f32 %4, %16, %17, %4\n v_mfma_f32_16x16x4_f32 %5, %16, %17, %5\n" \
    "v_mfma_f32_16x16x4_f32 %6, %16, %17, %6\n v_mfma_f32_16x16x4_f32 %7, %16, %17, %7\n v_mfma_f32_16x16x4_f32 %8, %16, %17, %8\n" \
    "v_mfma_f32_16x16x4_f32 %9, %16, %17, %9\n v_mfma_f32_16x16x4_f32 %10, %16, %17, %10\n v_mfma_f32_16x16x4_f32 %11, %16, %17, %11\n" \
    "v_mfma_f32_16x16x4_f32 %12, %16, %17, %12\n v_mfma_f32_16x16x4_f32 %13, %16, %17, %13\n v_mfma_f32_16x16x4_f32 %14, %16, %17, %14\n" \
    "v_mfma_f32_16x16x4_f32 %15, %16, %17, %15\n"

// REPT blocks of 16 MFMAs (128 B of code each) unrolled by the assembler; `passes` trips over the whole body
template <int REPT>
__global__ __launch_bounds__(256, 2) void k_straight(float *out, int passes) {
    extern __shared__ float lds[];
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a = 1.f + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
    // (the pass loop is written out: a body beyond 128 KiB is out of reach of s_cbranch's 16-bit offset)
    int cnt;
    unsigned long long pc;
    asm volatile("s_mov_b32 %18, %21\n s_getpc_b64 %19\n .rept %22\n" MFMA16 ".endr\n"
                 "s_sub_u32 %18, %18, 1\n s_cmp_lg_u32 %18, 0\n s_cbranch_scc0 1f\n s_setpc_b64 %19\n 1:\n"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]),
                   "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(a), "+v"(b),
                   "=&s"(cnt), "=&s"(pc)
                 : "n"(0), "s"(passes), "n"(REPT)
                 : "scc");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_flush(floatx4 *dst, size_t n4) {
    const floatx4 v = floatx4{1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, &dst[i]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int REPT>
void run(const char *name, float *out, floatx4 *big, size_t big4, int total_blocks16) {
    const int passes = total_blocks16 / REPT;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_straight<REPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int grid : {512, 1024, 4096}) {
        for (int cold = 0; cold < 2; ++cold) {
            float best = 1e30f, sum = 0.f;
            const int reps = 5;
            for (int r = 0; r < reps + 1; ++r) {
                if (cold) hipLaunchKernelGGL(k_flush, dim3(4096), dim3(256), 0, 0, big, big4);
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_straight<REPT>, dim3(grid), dim3(256), 74 * 1024, 0, out, passes);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r == 0) continue;               // first launch: code load
                best = ms < best ? ms : best;
                sum += ms;
            }
            const double flop = (double)grid * 4 * passes * REPT * 16 * 2048.0;
            printf("{\"variant\": \"%s\", \"code_kib\": %d, \"grid\": %d, \"cold\": %d, \"ms_best\": %.4f, \"ms_mean\": %.4f, \"tflops_best\": %.1f, \"tflops_mean\": %.1f}\n", name,
                   REPT * 128 / 1024, grid, cold, best, sum / reps, flop / (best * 1e-3) / 1e12, flop / (sum / reps * 1e-3) / 1e12);
        }
    }
}

int main() {
    float *out;
    floatx4 *big;
    const size_t big_bytes = (size_t)2 << 30;
    CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMalloc(&big, big_bytes));
    const int total = 8192;                       // 16-MFMA blocks per wavefront in every variant (~ one 64-row workgroup of the forward: 9 632 MFMAs = 602 blocks, x 13)
    run<1>("loop", out, big, big_bytes / 16, total);
    run<256>("straight", out, big, big_bytes / 16, total);
    run<512>("straight", out, big, big_bytes / 16, total);
    run<2048>("straight", out, big, big_bytes / 16, total);
    run<8192>("straight", out, big, big_bytes / 16, total);
    return 0;
}
