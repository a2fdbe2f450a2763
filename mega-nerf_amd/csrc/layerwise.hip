// layerwise.hip -- generic-width NeRF MLP evaluation for architectures the register-chained kernel does not cover
// (layer_dim > 512, e.g. configs/nerf: layer_dim 2048; odd layer counts / skip patterns; ...).
//
// Same arithmetic (exact fp32 MFMA), but one launch per nn.Linear with activations round-tripping HBM:
//   k_embed        positional encoding in the reference column order (nerf.py:8-25)
//   k_gather_rows  appearance-embedding lookup (nerf.py:149)
//   k_linear       Y = act([X1 | X2] W^T + b (+ per-row noise))      fp32 MFMA GEMM, 128 x 128 tiles
// The host side (mega_nerf/models/nerf.py::_evaluate_layerwise) sequences them exactly like nerf.py:115-160.
#include "common.h"

namespace mnr {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k_embed(float *__restrict__ out, long ldo, const float *__restrict__ x, long ldx, int D, int L, long row_div,
                        long B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const long r = i / D;
    const int d = (int)(i % D);
    const float v = x[(r / row_div) * ldx + d];
    float *o = out + r * ldo;
    o[d] = v;
    for (int f = 0; f < L; ++f) {
        float s, c;
        sincosf(ldexpf(v, f), &s, &c);
        o[D + f * 2 * D + d] = s;
        o[D + f * 2 * D + D + d] = c;
    }
}

__global__ void k_gather_rows(float *__restrict__ out, long ldo, const float *__restrict__ table, int width, int count,
                              const void *__restrict__ idx, long idx_stride, int idx_is_float, long row_div, long B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * width) return;
    const long r = i / width;
    const int c = (int)(i % width);
    const long ray = r / row_div;
    long k = idx_is_float ? (long)reinterpret_cast<const float *>(idx)[ray * idx_stride]
                          : (long)reinterpret_cast<const int32_t *>(idx)[ray * idx_stride];
    k = k < 0 ? 0 : (k >= count ? count - 1 : k);
    out[r * ldo + c] = table[k * width + c];
}

constexpr int LW_BM = 128, LW_BN = 128, LW_KT = 32, LW_LD = LW_KT + 1;   // +1: conflict-free column reads

__device__ __forceinline__ float lw_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) { const float y = v - 1.f; return y > 20.f ? y : log1pf(expf(y)); }
    return v;
}

// Y[b][n] = act( sum_k Xcat[b][k] * W[n][k] + bias[n] + row_add[b] ),  Xcat = [X1 (K1 cols) | X2 (K2 cols)]
__global__ __launch_bounds__(256) void k_linear(float *__restrict__ Y, long ldy, const float *__restrict__ X1, long ldx1, int K1,
                                                const float *__restrict__ X2, long ldx2, int K2, const float *__restrict__ W,
                                                long ldw, const float *__restrict__ bias, const float *__restrict__ row_add,
                                                long B, int N, int act) {
    __shared__ float As[LW_BM * LW_LD], Bs[LW_BN * LW_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, kk = lane >> 5;
    const long m0 = (long)blockIdx.y * LW_BM;
    const int n0 = blockIdx.x * LW_BN;
    const int K = K1 + K2;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = floatx16(0.f);

    for (int k0 = 0; k0 < K; k0 += LW_KT) {
        __syncthreads();
        // stage the two operand tiles (zero fill outside the matrix): 128 x 32 floats each, 16 per thread
        for (int e = threadIdx.x; e < LW_BM * LW_KT; e += 256) {
            const int r = e / LW_KT, k = k0 + e % LW_KT;
            const long row = m0 + r;
            float v = 0.f;
            if (row < B && k < K) v = k < K1 ? X1[row * ldx1 + k] : X2[row * ldx2 + (k - K1)];
            As[r * LW_LD + e % LW_KT] = v;
            const int n = n0 + r;
            Bs[r * LW_LD + e % LW_KT] = (n < N && k < K) ? W[(long)n * ldw + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < LW_KT; k += 2) {
            float af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[(wr * 64 + a * 32 + i32) * LW_LD + k + kk];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Bs[(wc * 64 + b * 32 + i32) * LW_LD + k + kk];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    // C layout: lane -> column (feature) n, registers -> rows
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wc * 64 + b * 32 + i32;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < B) Y[row * ldy + n] = lw_act(acc[a][b][r] + bv + (row_add ? row_add[row] : 0.f), act);
            }
        }
}

}  // namespace mnr

using namespace mnr;

extern "C" int mnr_embed(float *out, int64_t ldo, const float *x, int64_t ldx, int D, int L, int64_t rows_per_src, int64_t B,
                         void *stream) {
    MNR_REQUIRE(out && x && D > 0 && L >= 0 && B >= 0 && rows_per_src >= 1, "bad arguments to mnr_embed");
    if (B == 0) return MNR_OK;
    hipLaunchKernelGGL(k_embed, dim3((unsigned)((B * D + 255) / 256)), dim3(256), 0, as_stream(stream), out, (long)ldo, x,
                       (long)ldx, D, L, (long)rows_per_src, (long)B);
    return check_launch("k_embed");
}

extern "C" int mnr_gather_rows(float *out, int64_t ldo, const float *table, int width, int count, const void *idx,
                               int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t B, void *stream) {
    MNR_REQUIRE(out && table && idx && width > 0 && count > 0 && B >= 0 && rows_per_ray >= 1, "bad arguments to mnr_gather_rows");
    if (B == 0) return MNR_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((B * width + 255) / 256)), dim3(256), 0, as_stream(stream), out, (long)ldo,
                       table, width, count, idx, (long)idx_stride, idx_is_float, (long)rows_per_ray, (long)B);
    return check_launch("k_gather_rows");
}

extern "C" int mnr_linear(float *Y, int64_t ldy, const float *X1, int64_t ldx1, int K1, const float *X2, int64_t ldx2, int K2,
                          const float *W, int64_t ldw, const float *bias, const float *row_add, int64_t B, int N, int act,
                          void *stream) {
    MNR_REQUIRE(Y && X1 && W && K1 > 0 && K2 >= 0 && (K2 == 0 || X2) && N > 0 && B >= 0 && act >= 0 && act <= 3,
                "bad arguments to mnr_linear");
    if (B == 0) return MNR_OK;
    const dim3 grid((N + LW_BN - 1) / LW_BN, (unsigned)((B + LW_BM - 1) / LW_BM));
    MNR_REQUIRE(grid.y <= 65535, "too many rows for one mnr_linear launch (chunk the batch)");
    hipLaunchKernelGGL(k_linear, grid, dim3(256), 0, as_stream(stream), Y, (long)ldy, X1, (long)ldx1, K1, X2, (long)ldx2, K2, W,
                       (long)ldw, bias, row_add, (long)B, N, act);
    return check_launch("k_linear");
}
