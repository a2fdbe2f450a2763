// layerwise.hip -- generic-width NeRF MLP evaluation for architectures the register-chained kernel does not cover
// (layer_dim > 512, e.g. configs/nerf: layer_dim 2048; odd layer counts / skip patterns; ...).
//
// Same arithmetic (exact fp32 MFMA), but one launch per nn.Linear with activations round-tripping HBM:
//   k_embed        positional encoding in the reference column order (nerf.py:8-25)
//   k_gather_rows  appearance-embedding lookup (nerf.py:149)
//   k_linear       Y = act([X1 | X2] W^T + b (+ per-row noise))      fp32 MFMA GEMM, 128 x 128 tiles
// The host side (mega_nerf/models/nerf.py::_evaluate_layerwise) sequences them exactly like nerf.py:115-160.
//
// Training of those architectures (mega_nerf/models/layerwise_train.py) keeps every layer output in HBM and runs the
// adjoint layer by layer with
//   k_gemm          C (op)= A B^T for arbitrarily strided operands: data gradients G W and weight gradients G^T X
//                   (split over the row dimension, atomics into .grad)
//   k_act_grad      G = dY * act'(Y)          k_col_sum   bias gradients
//   k_scatter_rows  appearance-embedding gradient (per-ray pre-reduction, then atomics)
//   k_sh_apply / k_sh_backward   spherical-harmonics colour (rendering.py:300-305) outside the fused epilogue
#include "common.h"
#include "sh_device.h"

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
constexpr int LW_TILE_F = LW_BM * LW_LD;                                 // floats per staged operand tile
constexpr int LW_LDS_BYTES = 2 * 2 * LW_TILE_F * (int)sizeof(float);     // two buffers x (A, B)

__device__ __forceinline__ float lw_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) { const float y = v - 1.f; return y > 20.f ? y : log1pf(expf(y)); }
    return v;
}

// One GEMM operand: element (r, k) at p[r * sr + k * sk] for r < rows, k < klim (zero outside).  Exactly one of sr / sk
// is 1 in every use (row-major activations / weights or their transposes); `vec` = 16-byte loads along that unit stride
// are legal (pointer, the other stride and both extents are multiples of 4).
struct LwOperand {
    const float *p;
    long sr, sk, rows, klim;
};

// 128 x 32 tile -> 16 registers per thread.  Thread -> element maps keep global loads coalesced along the unit stride.
template <bool KFAST, bool VEC>
__device__ __forceinline__ void lw_fetch(float (&v)[16], const LwOperand &o, long r0, long k0) {
    const int t = threadIdx.x;
    if (KFAST) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = t + i * 256, r = e >> 3, kq = (e & 7) * 4;
                float4 x = {0.f, 0.f, 0.f, 0.f};
                if (r0 + r < o.rows && k0 + kq < o.klim) x = *reinterpret_cast<const float4 *>(o.p + (r0 + r) * o.sr + (k0 + kq));
                v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = t + i * 256, r = e >> 5, kq = e & 31;
                v[i] = (r0 + r < o.rows && k0 + kq < o.klim) ? o.p[(r0 + r) * o.sr + (k0 + kq)] : 0.f;
            }
        }
    } else {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = t + i * 256, kq = e >> 5, r = (e & 31) * 4;
                float4 x = {0.f, 0.f, 0.f, 0.f};
                if (r0 + r < o.rows && k0 + kq < o.klim) x = *reinterpret_cast<const float4 *>(o.p + (r0 + r) + (k0 + kq) * o.sk);
                v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = t + i * 256, r = e & 127, kq = e >> 7;
                v[i] = (r0 + r < o.rows && k0 + kq < o.klim) ? o.p[(r0 + r) + (k0 + kq) * o.sk] : 0.f;
            }
        }
    }
}

template <bool KFAST, bool VEC>
__device__ __forceinline__ void lw_commit(float *S, const float (&v)[16]) {
    const int t = threadIdx.x;
    if (KFAST) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = t + i * 256, r = e >> 3, kq = (e & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) S[r * LW_LD + kq + j] = v[4 * i + j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = t + i * 256;
                S[(e >> 5) * LW_LD + (e & 31)] = v[i];
            }
        }
    } else {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = t + i * 256, kq = e >> 5, r = (e & 31) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) S[(r + j) * LW_LD + kq] = v[4 * i + j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = t + i * 256;
                S[(e & 127) * LW_LD + (e >> 7)] = v[i];
            }
        }
    }
}

// acc (2 x 2 blocks of 32 x 32 per wave) += sum over up to two K phases of A_ph(m, k) B_ph(n, k); the K range of a
// phase is [kb, ke).  Software pipeline: the next tile's global loads are in flight while the MFMAs of the current one
// run from LDS (two LDS buffers, one barrier per tile).
template <bool AK, bool BK, bool VEC>
__device__ __forceinline__ void lw_mainloop(floatx16 (&acc)[2][2], float *lds, const LwOperand (&A)[2], const LwOperand (&B)[2],
                                            const long (&kb)[2], const long (&ke)[2], int n_phases, long m0, long n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, kk = lane >> 5;
    long tiles[2] = {0, 0};
    for (int p = 0; p < n_phases; ++p) tiles[p] = (ke[p] - kb[p] + LW_KT - 1) / LW_KT;
    const long total = tiles[0] + tiles[1];
    if (total == 0) return;
    float ra[16], rb[16];
    auto fetch = [&](long t) {
        const int p = t < tiles[0] ? 0 : 1;
        const long k0 = kb[p] + (t - (p ? tiles[0] : 0)) * LW_KT;
        LwOperand a = A[p], b = B[p];
        a.klim = min(a.klim, ke[p]); b.klim = min(b.klim, ke[p]);
        lw_fetch<AK, VEC>(ra, a, m0, k0);
        lw_fetch<BK, VEC>(rb, b, n0, k0);
    };
    auto commit = [&](long, int buf) {
        lw_commit<AK, VEC>(lds + buf * 2 * LW_TILE_F, ra);
        lw_commit<BK, VEC>(lds + buf * 2 * LW_TILE_F + LW_TILE_F, rb);
    };
    fetch(0);
    commit(0, 0);
    __syncthreads();
    for (long t = 0; t < total; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < total) fetch(t + 1);
        const float *As = lds + buf * 2 * LW_TILE_F, *Bs = As + LW_TILE_F;
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
        if (t + 1 < total) commit(t + 1, buf ^ 1);
        __syncthreads();
    }
}

// 16-byte loads along an operand's unit stride are legal when its base, its other stride and its extent along the unit
// stride are multiples of 4 floats
static inline bool lw_aligned(const void *p, long other_stride, long extent_unit) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (other_stride & 3) == 0 && (extent_unit & 3) == 0;
}

// Y[b][n] = act( sum_k Xcat[b][k] * W[n][k] + bias[n] + row_add[b] ),  Xcat = [X1 (K1 cols) | X2 (K2 cols)]
template <bool VEC>
__global__ __launch_bounds__(256, 2) void k_linear(float *__restrict__ Y, long ldy, const float *__restrict__ X1, long ldx1, int K1,
                                                   const float *__restrict__ X2, long ldx2, int K2, const float *__restrict__ W,
                                                   long ldw, const float *__restrict__ bias, const float *__restrict__ row_add,
                                                   long B, int N, int act) {
    extern __shared__ float lw_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, kk = lane >> 5;
    const long m0 = (long)blockIdx.y * LW_BM;
    const int n0 = blockIdx.x * LW_BN;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = floatx16(0.f);
    LwOperand A[2], Bm[2];
    A[0] = LwOperand{X1, ldx1, 1, B, K1};
    Bm[0] = LwOperand{W, ldw, 1, N, K1};
    A[1] = LwOperand{X2, ldx2, 1, B, K2};
    Bm[1] = LwOperand{W + K1, ldw, 1, N, K2};
    const long kb[2] = {0, 0}, ke[2] = {K1, K2};
    lw_mainloop<true, true, VEC>(acc, lw_lds, A, Bm, kb, ke, K2 > 0 ? 2 : 1, m0, n0);
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

// ---- generic strided GEMM:  C[m][n] (op)= sum_k A(m,k) B(n,k),  A(m,k) = A[m sam + k sak],  B(n,k) = B[n sbn + k sbk]
// mode 0: store, 1: C += (exclusive owner), 2: atomicAdd (split-K partial sums)
template <bool AK, bool BK, bool VEC>
__global__ __launch_bounds__(256, 2) void k_gemm(float *__restrict__ Cmat, long ldc, const float *__restrict__ A, long sam, long sak,
                                                 const float *__restrict__ B, long sbn, long sbk, long M, int N, long K,
                                                 long k_per_split, int mode) {
    extern __shared__ float lw_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int i32 = lane & 31, kk = lane >> 5;
    const long m0 = (long)blockIdx.y * LW_BM;
    const int n0 = blockIdx.x * LW_BN;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = floatx16(0.f);
    LwOperand Ao[2], Bo[2];
    Ao[0] = LwOperand{A, sam, sak, M, K};
    Bo[0] = LwOperand{B, sbn, sbk, N, K};
    Ao[1] = Ao[0]; Bo[1] = Bo[0];
    const long kb[2] = {(long)blockIdx.z * k_per_split, 0}, ke[2] = {min(K, kb[0] + k_per_split), 0};
    lw_mainloop<AK, BK, VEC>(acc, lw_lds, Ao, Bo, kb, ke, 1, m0, n0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wc * 64 + b * 32 + i32;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row >= M) continue;
                float *dst = Cmat + row * ldc + n;
                if (mode == 0) *dst = acc[a][b][r];
                else if (mode == 1) *dst += acc[a][b][r];
                else atomicAdd(dst, acc[a][b][r]);
            }
        }
}

// G = dY * act'(Y) expressed through the layer OUTPUT Y (what the forward kept): relu Y > 0, sigmoid Y (1 - Y),
// shifted softplus 1 - exp(-Y) (= sigmoid of the pre-activation; 1 beyond the threshold in fp32)
__global__ void k_act_grad(float *__restrict__ G, long ldg, const float *__restrict__ dY, long ldd, const float *__restrict__ Y,
                           long ldy, long R, int N, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * N) return;
    const long r = i / N;
    const int n = (int)(i % N);
    const float y = Y[r * ldy + n], d = dY[r * ldd + n];
    float g = d;
    if (act == 1) g = y > 0.f ? d : 0.f;
    else if (act == 2) g = d * (y * (1.f - y));
    else if (act == 3) g = d * (1.f - expf(-y));
    G[r * ldg + n] = g;
}

// out[n] += sum_r G[r][n]: 64 columns x 4 row phases per block over a 1024-row slab
__global__ __launch_bounds__(256) void k_col_sum(float *__restrict__ out, const float *__restrict__ G, long ldg, long R, int N) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const long r0 = (long)blockIdx.y * 1024, r1 = min(R, r0 + 1024);
    float s = 0.f;
    if (n < N)
        for (long r = r0 + ph; r < r1; r += 4) s += G[r * ldg + n];
    part[ph][c] = s;
    __syncthreads();
    if (ph == 0 && n < N) atomicAdd(out + n, (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]));
}

// table_grad[idx[ray]][c] += sum over the ray's rows of src[row][c]
__global__ void k_scatter_rows(float *__restrict__ table_grad, int width, int count, const void *__restrict__ idx, long idx_stride,
                               int idx_is_float, long rows_per_ray, const float *__restrict__ src, long lds_, long R) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_rays = (R + rows_per_ray - 1) / rows_per_ray;
    if (i >= n_rays * width) return;
    const long ray = i / width;
    const int c = (int)(i % width);
    long k = idx_is_float ? (long)reinterpret_cast<const float *>(idx)[ray * idx_stride]
                          : (long)reinterpret_cast<const int32_t *>(idx)[ray * idx_stride];
    k = k < 0 ? 0 : (k >= count ? count - 1 : k);
    float s = 0.f;
    const long r1 = min(R, (ray + 1) * rows_per_ray);
    for (long r = ray * rows_per_ray; r < r1; ++r) s += src[r * lds_ + c];
    atomicAdd(table_grad + k * width + c, s);
}

// out[r] = [sigmoid(eval_sh(coef[r][c][:], dir(r))) for c in RGB, sigma]   (rendering.py:300-305; coef is channel-major)
__global__ void k_sh_apply(float *__restrict__ out, long ldo, const float *__restrict__ coef, long ldc, const float *__restrict__ dirs,
                           long dir_stride, long rows_per_ray, int deg, long R) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int nb = (deg + 1) * (deg + 1);
    const float *d = dirs + (r / rows_per_ray) * dir_stride;
    const float *c = coef + r * ldc;
    for (int ch = 0; ch < 3; ++ch) out[r * ldo + ch] = 1.f / (1.f + expf(-eval_sh_channel(deg, c + ch * nb, d[0], d[1], d[2])));
    out[r * ldo + 3] = c[3 * nb];
}

// ---- affine appearance (nerf.py:87-89,156-158): rgb' = A[:, :3] . rgb + A[:, 3] with A = affine(embedding_a[idx]) viewed (3, 4),
// followed by the sigmoid of the rgb head.  `table` holds A for every appearance index ([count][12], one mnr_linear per
// weight version); raw = output of the rgb layer without activation.
__device__ __forceinline__ long affine_row(const void *idx, long idx_stride, int idx_is_float, long ray, int count) {
    long i = idx_is_float ? (long)reinterpret_cast<const float *>(idx)[ray * idx_stride]
                          : (long)reinterpret_cast<const int32_t *>(idx)[ray * idx_stride];
    return i < 0 ? 0 : (i >= count ? count - 1 : i);
}
__global__ void k_affine_apply(float *__restrict__ out, long ldo, const float *__restrict__ raw, long ldr, const float *__restrict__ table,
                               int count, const void *__restrict__ idx, long idx_stride, int idx_is_float, long rows_per_ray, long R) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *A = table + affine_row(idx, idx_stride, idx_is_float, r / rows_per_ray, count) * 12;
    const float x = raw[r * ldr], y = raw[r * ldr + 1], z = raw[r * ldr + 2];
    for (int c = 0; c < 3; ++c) {
        const float v = fmaf(A[4 * c + 2], z, fmaf(A[4 * c + 1], y, A[4 * c] * x)) + A[4 * c + 3];
        out[r * ldo + c] = 1.f / (1.f + expf(-v));
    }
}
// g = d_out * s (1 - s);  d_raw = A[:, :3]^T g;  d_A_row[r] = [g_c * raw_k, g_c]  (12 floats per row: the caller reduces them
// per appearance index with mnr_scatter_rows)
__global__ void k_affine_backward(float *__restrict__ d_raw, long ldr, float *__restrict__ d_arow, const float *__restrict__ d_out, long ldd,
                                  const float *__restrict__ out, long ldo, const float *__restrict__ raw, long ldri,
                                  const float *__restrict__ table, int count, const void *__restrict__ idx, long idx_stride,
                                  int idx_is_float, long rows_per_ray, long R) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float *A = table + affine_row(idx, idx_stride, idx_is_float, r / rows_per_ray, count) * 12;
    const float x[3] = {raw[r * ldri], raw[r * ldri + 1], raw[r * ldri + 2]};
    float g[3], dx[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 3; ++c) {
        const float s = out[r * ldo + c];
        g[c] = d_out[r * ldd + c] * (s * (1.f - s));
        for (int k = 0; k < 3; ++k) {
            dx[k] = fmaf(A[4 * c + k], g[c], dx[k]);
            d_arow[r * 12 + 4 * c + k] = g[c] * x[k];
        }
        d_arow[r * 12 + 4 * c + 3] = g[c];
    }
    for (int k = 0; k < 3; ++k) d_raw[r * ldr + k] = dx[k];
}

// d_coef[r][c][k] = d_out[r][c] * s (1 - s) * basis_k(dir);  d_coef[r][3 nb] = d_out[r][3]   (s = out[r][c])
__global__ void k_sh_backward(float *__restrict__ d_coef, long ldc, const float *__restrict__ d_out, long ldd,
                              const float *__restrict__ out, long ldo, const float *__restrict__ dirs, long dir_stride,
                              long rows_per_ray, int deg, long R) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int nb = (deg + 1) * (deg + 1);
    const float *d = dirs + (r / rows_per_ray) * dir_stride;
    float b[25];
    sh_basis(deg, d[0], d[1], d[2], b);
    for (int ch = 0; ch < 3; ++ch) {
        const float s = out[r * ldo + ch];
        const float g = d_out[r * ldd + ch] * (s * (1.f - s));
        for (int k = 0; k < nb; ++k) d_coef[r * ldc + ch * nb + k] = g * b[k];
    }
    d_coef[r * ldc + 3 * nb] = d_out[r * ldd + 3];
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
    const bool vec = lw_aligned(X1, ldx1, K1) && lw_aligned(W, ldw, K1) && (K2 == 0 || (lw_aligned(X2, ldx2, K2) && lw_aligned(W + K1, ldw, K2)));
    auto go = [&](auto kern) {
        static bool attr_dev[MAX_DEVICES] = {};                      // (one static per kernel instantiation; attributes are per device)
        bool &attr = attr_dev[device_slot()];
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LW_LDS_BYTES);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), LW_LDS_BYTES, as_stream(stream), Y, (long)ldy, X1, (long)ldx1, K1, X2, (long)ldx2, K2,
                           W, (long)ldw, bias, row_add, (long)B, N, act);
    };
    if (vec) go(&k_linear<true>); else go(&k_linear<false>);
    return check_launch("k_linear");
}

extern "C" int mnr_gemm(float *C, int64_t ldc, const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbn, int64_t sbk,
                        int64_t M, int N, int64_t K, int accumulate, int split_k, void *stream) {
    MNR_REQUIRE(C && A && B && M >= 0 && N > 0 && K >= 0 && accumulate >= 0 && accumulate <= 1 && split_k >= 0,
                "bad arguments to mnr_gemm");
    if (M == 0) return MNR_OK;
    const long tiles_m = (M + LW_BM - 1) / LW_BM, tiles_n = (N + LW_BN - 1) / LW_BN;
    MNR_REQUIRE(tiles_m <= 65535, "too many rows for one mnr_gemm launch (chunk the batch)");
    long splits = split_k;
    if (splits == 0) {   // auto: enough workgroups to fill 256 CUs a few times over, at least 512 reduction steps each
        splits = (2048 + tiles_m * tiles_n - 1) / (tiles_m * tiles_n);
        const long max_splits = (K + 511) / 512;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    MNR_REQUIRE(splits == 1 || accumulate == 1, "mnr_gemm: split_k > 1 accumulates atomically and needs accumulate = 1");
    long kps = (K + splits - 1) / splits;
    kps = (kps + LW_KT - 1) / LW_KT * LW_KT;
    splits = K > 0 ? (K + kps - 1) / kps : 1;
    const int mode = accumulate == 0 ? 0 : (splits > 1 ? 2 : 1);
    MNR_REQUIRE((sak == 1 || sam == 1) && (sbk == 1 || sbn == 1), "mnr_gemm operands need a unit stride");
    const bool ak = sak == 1, bk = sbk == 1;
    const bool vec = (ak ? lw_aligned(A, sam, K) : lw_aligned(A, sak, M)) && (bk ? lw_aligned(B, sbn, K) : lw_aligned(B, sbk, N));
    const dim3 grid((unsigned)tiles_n, (unsigned)tiles_m, (unsigned)splits);
    auto go = [&](auto kern) {
        static bool attr_dev[MAX_DEVICES] = {};                      // (one static per kernel instantiation; attributes are per device)
        bool &attr = attr_dev[device_slot()];
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LW_LDS_BYTES);
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), LW_LDS_BYTES, as_stream(stream), C, (long)ldc, A, (long)sam, (long)sak, B, (long)sbn,
                           (long)sbk, (long)M, N, (long)K, kps, mode);
    };
    if (ak && bk) { if (vec) go(&k_gemm<true, true, true>); else go(&k_gemm<true, true, false>); }
    else if (ak && !bk) { if (vec) go(&k_gemm<true, false, true>); else go(&k_gemm<true, false, false>); }
    else if (!ak && !bk) { if (vec) go(&k_gemm<false, false, true>); else go(&k_gemm<false, false, false>); }
    else { if (vec) go(&k_gemm<false, true, true>); else go(&k_gemm<false, true, false>); }
    return check_launch("k_gemm");
}

extern "C" int mnr_act_grad(float *G, int64_t ldg, const float *dY, int64_t ldd, const float *Y, int64_t ldy, int64_t R, int N,
                            int act, void *stream) {
    MNR_REQUIRE(G && dY && Y && R >= 0 && N > 0 && act >= 0 && act <= 3, "bad arguments to mnr_act_grad");
    if (R == 0) return MNR_OK;
    hipLaunchKernelGGL(k_act_grad, dim3((unsigned)((R * N + 255) / 256)), dim3(256), 0, as_stream(stream), G, (long)ldg, dY,
                       (long)ldd, Y, (long)ldy, (long)R, N, act);
    return check_launch("k_act_grad");
}

extern "C" int mnr_col_sum(float *out, const float *G, int64_t ldg, int64_t R, int N, void *stream) {
    MNR_REQUIRE(out && G && R >= 0 && N > 0, "bad arguments to mnr_col_sum");
    if (R == 0) return MNR_OK;
    MNR_REQUIRE((R + 1023) / 1024 <= 65535, "too many rows for one mnr_col_sum launch");
    hipLaunchKernelGGL(k_col_sum, dim3((N + 63) / 64, (unsigned)((R + 1023) / 1024)), dim3(256), 0, as_stream(stream), out, G,
                       (long)ldg, (long)R, N);
    return check_launch("k_col_sum");
}

extern "C" int mnr_scatter_rows(float *table_grad, int width, int count, const void *idx, int64_t idx_stride, int idx_is_float,
                                int64_t rows_per_ray, const float *src, int64_t ld_src, int64_t R, void *stream) {
    MNR_REQUIRE(table_grad && idx && src && width > 0 && count > 0 && rows_per_ray >= 1 && R >= 0, "bad arguments to mnr_scatter_rows");
    if (R == 0) return MNR_OK;
    const long n = (R + rows_per_ray - 1) / rows_per_ray * width;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), table_grad, width, count,
                       idx, (long)idx_stride, idx_is_float, (long)rows_per_ray, src, (long)ld_src, (long)R);
    return check_launch("k_scatter_rows");
}

extern "C" int mnr_sh_apply(float *out, int64_t ldo, const float *coef, int64_t ldc, const float *dirs, int64_t dir_stride,
                            int64_t rows_per_ray, int deg, int64_t R, void *stream) {
    MNR_REQUIRE(out && coef && dirs && deg >= 0 && deg <= 4 && rows_per_ray >= 1 && R >= 0, "bad arguments to mnr_sh_apply");
    if (R == 0) return MNR_OK;
    hipLaunchKernelGGL(k_sh_apply, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), out, (long)ldo, coef, (long)ldc,
                       dirs, (long)dir_stride, (long)rows_per_ray, deg, (long)R);
    return check_launch("k_sh_apply");
}

extern "C" int mnr_sh_backward(float *d_coef, int64_t ldc, const float *d_out, int64_t ldd, const float *out, int64_t ldo,
                               const float *dirs, int64_t dir_stride, int64_t rows_per_ray, int deg, int64_t R, void *stream) {
    MNR_REQUIRE(d_coef && d_out && out && dirs && deg >= 0 && deg <= 4 && rows_per_ray >= 1 && R >= 0, "bad arguments to mnr_sh_backward");
    if (R == 0) return MNR_OK;
    hipLaunchKernelGGL(k_sh_backward, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), d_coef, (long)ldc, d_out,
                       (long)ldd, out, (long)ldo, dirs, (long)dir_stride, (long)rows_per_ray, deg, (long)R);
    return check_launch("k_sh_backward");
}

extern "C" int mnr_affine_apply(float *out, int64_t ldo, const float *raw, int64_t ldr, const float *table, int count, const void *idx,
                                int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t R, void *stream) {
    MNR_REQUIRE(out && raw && table && idx && count >= 1 && rows_per_ray >= 1 && R >= 0, "bad arguments to mnr_affine_apply");
    if (R == 0) return MNR_OK;
    hipLaunchKernelGGL(k_affine_apply, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), out, (long)ldo, raw, (long)ldr,
                       table, count, idx, (long)idx_stride, idx_is_float, (long)rows_per_ray, (long)R);
    return check_launch("k_affine_apply");
}

extern "C" int mnr_affine_backward(float *d_raw, int64_t ldr, float *d_affine_rows, const float *d_out, int64_t ldd, const float *out,
                                   int64_t ldo, const float *raw, int64_t ldri, const float *table, int count, const void *idx,
                                   int64_t idx_stride, int idx_is_float, int64_t rows_per_ray, int64_t R, void *stream) {
    MNR_REQUIRE(d_raw && d_affine_rows && d_out && out && raw && table && idx && count >= 1 && rows_per_ray >= 1 && R >= 0,
                "bad arguments to mnr_affine_backward");
    if (R == 0) return MNR_OK;
    hipLaunchKernelGGL(k_affine_backward, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), d_raw, (long)ldr,
                       d_affine_rows, d_out, (long)ldd, out, (long)ldo, raw, (long)ldri, table, count, idx, (long)idx_stride,
                       idx_is_float, (long)rows_per_ray, (long)R);
    return check_launch("k_affine_backward");
}
