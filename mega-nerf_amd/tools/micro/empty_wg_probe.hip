// empty_wg_probe.hip -- what does a workgroup that exits at once cost?  The gather-mode launches of a routed container size their grid for
// the worst case (every row routed to every cell: rows / 64 x cells workgroups) and the surplus workgroups return on the device-side count:
// 25 600 + 51 200 workgroups per render of a 25-cell container, ~4 500 of them real.  Prints ms for grids of exiting workgroups of 256 and
// 512 threads with 74 KB / 0 KB of LDS requested.  Build: hipcc --offload-arch=gfx950 -O3 -o empty_wg_probe empty_wg_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(NT) void k_exit(const int *count, float *out) {
    extern __shared__ float lds[];
    if ((int)blockIdx.x >= *count) return;
    out[blockIdx.x * NT + threadIdx.x] = lds[threadIdx.x] + 1.f;
}

template <int NT>
void run(int lds_kb, const int *count, float *out) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_exit<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int grid : {1024, 8192, 25600, 51200, 102400}) {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_exit<NT>, dim3(grid), dim3(NT), lds_kb * 1024, 0, count, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = ms < best ? ms : best;
        }
        printf("{\"threads\": %d, \"lds_kb\": %d, \"grid\": %d, \"ms\": %.4f, \"ns_per_workgroup\": %.1f}\n", NT, lds_kb, grid, best, best * 1e6 / grid);
    }
}

int main() {
    int *count;
    float *out;
    CK(hipMalloc(&count, 4));
    CK(hipMemset(count, 0, 4));
    CK(hipMalloc(&out, 1 << 20));
    run<256>(74, count, out);
    run<256>(0, count, out);
    run<512>(74, count, out);
    run<512>(0, count, out);
    return 0;
}
