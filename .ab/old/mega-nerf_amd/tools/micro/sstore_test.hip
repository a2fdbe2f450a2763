// sstore_test.hip -- does gfx950 execute scalar stores (s_store_dwordx4 + s_dcache_wb) correctly?  (diagnostics)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long *out, const float *in, int n_regs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t mpv = (size_t)(out + ((size_t)blockIdx.x * 4 + wave) * n_regs);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mpv), hi = __builtin_amdgcn_readfirstlane((unsigned)(mpv >> 32));
    const unsigned long long mp = ((unsigned long long)hi << 32) | lo;
    for (int i = 0; i < n_regs; i += 2) {
        const float a = in[(blockIdx.x * 256 + threadIdx.x) * n_regs + i], b = in[(blockIdx.x * 256 + threadIdx.x) * n_regs + i + 1];
        const unsigned long long b0 = __ballot(a > 0.f), b1 = __ballot(b > 0.f);
        uint4v v = {(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
        const unsigned off = (unsigned)(i * 8);
        asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(mp), "s"(off) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main() {
    const int blocks = 2048, n_regs = 64;
    const size_t n = (size_t)blocks * 256 * n_regs;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 7) & 1023) - 512.f;
    float *din; unsigned long long *dout;
    (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dout, (size_t)blocks * 4 * n_regs * 8);
    (void)hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0xff, (size_t)blocks * 4 * n_regs * 8);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, din, n_regs);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> o((size_t)blocks * 4 * n_regs);
    (void)hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) for (int r = 0; r < n_regs; ++r) {
        unsigned long long want = 0;
        for (int l = 0; l < 64; ++l) if (h[((size_t)b * 256 + w * 64 + l) * n_regs + r] > 0.f) want |= 1ull << l;
        if (o[((size_t)b * 4 + w) * n_regs + r] != want) ++bad;
    }
    printf("scalar store test: %zu mismatches of %zu  (%s)\n", bad, o.size(), hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
