// split_probe.hip -- microbenchmark GATE for a split-precision MFMA variant of the register-chained MLP (diagnostics only; not
// part of libmeganerf_hip.so).  Question: can a chain of 256 -> 256 layers run on the 16-bit matrix
// pipe (2.5 PFLOP/s dense, 16x the fp32 pipe) with fp32 accumulation and 3 products per MAC at >= 250 TFLOP/s *effective*
// (fp32-equivalent FLOPs) and <= 2e-5 relative error per layer against fp64 -- INCLUDING the VALU cost of splitting the
// activations between layers?
//
// Scheme: x = x_hi + x_lo, w = w_hi + w_lo with both parts in a 16-bit format; acc(fp32) += w_hi x_hi + w_lo x_hi + w_hi x_lo
// (the dropped w_lo x_lo term is 2^-22 relative for f16 parts, 2^-16 for bf16 parts).
//   DT 0: f16 parts  (v_mfma_f32_16x16x32_f16):  x_hi = x with the mantissa cut to 10 bits, x_lo = x - x_hi (exact in fp32)
//   DT 1: bf16 parts (v_mfma_f32_16x16x32_bf16): x_hi = upper 16 bits of x, x_lo = upper 16 bits of (x - x_hi)
//   PROD 3: the three products above; PROD 1: w_hi x_hi only (what plain f16 / bf16 inputs would give)
// Structure = the product kernel's: one wavefront owns 16 samples x all 256 features; with the 16x16 C/D layout lane (part p =
// lane >> 4, sample n = lane & 15) holds, in accumulator block ob, features 16 ob + 4 p + {0..3}; the K order of the next layer
// is chosen so that K-step s (32 features) consumes exactly the registers of blocks 2s, 2s+1 -- activations never leave the
// register file, they are only re-split (4 VALU instructions per value) between layers.  Weights stream through a 2 x 64 KiB
// LDS ring in fragment order (LDS-DMA), 8 waves per workgroup (2 per SIMD) share it: 128 rows per pass over the weights.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o split_probe split_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

constexpr int W = 256, NOB = 16, NS = 8;          // 16 output blocks of 16 features; 8 K-steps of 32 features
constexpr int FRAG_U4 = 64;                       // one A fragment: 64 lanes x 16 bytes
constexpr int STEP_U4 = NOB * 2 * FRAG_U4;        // per K-step: (hi, lo) fragment per output block = 32 KiB
constexpr int SPC = 2;                            // K-steps per chunk
constexpr int CHUNK_U4 = SPC * STEP_U4;           // 64 KiB
constexpr int WAVES = 8, THREADS = WAVES * 64;
constexpr int H = 64;                             // activation values per lane

template <int DT>
__device__ __forceinline__ floatx4 mfma16(uint4v a, uint4v b, floatx4 c) {
    if constexpr (DT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// two fp32 values whose mantissas already fit the 16-bit format -> one packed register (exact)
template <int DT>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    if constexpr (DT == 0) {
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    } else {
        return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
    }
}

// split 8 non-negative-or-any fp32 values into the (hi, lo) B operands of one K-step
template <int DT>
__device__ __forceinline__ void split8(const float *x, uint4v &hi, uint4v &lo) {
    constexpr unsigned MASK = DT == 0 ? 0xffffe000u : 0xffff0000u;     // keep 10 / 7 explicit mantissa bits
    float h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = __uint_as_float(__float_as_uint(x[j]) & MASK);
        l[j] = x[j] - h[j];                                             // exact
        if constexpr (DT == 1) l[j] = __uint_as_float(__float_as_uint(l[j]) & MASK);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = pack2<DT>(h[2 * q], h[2 * q + 1]);
        lo[q] = pack2<DT>(l[2 * q], l[2 * q + 1]);
    }
}

// TILES: 16-row tiles a wavefront owns (1: 8 wavefronts per workgroup, 2 per SIMD; 2: 4 wavefronts, one per SIMD with the whole
// 512-register file -- every fragment read from LDS then feeds two tiles' MFMAs)
template <int DT, int PROD, int TILES, bool PIPE = false>
__global__ __launch_bounds__(THREADS / TILES, 1) void k_split(const uint4v *__restrict__ wstream, const float *__restrict__ bias,
                                                      const float *__restrict__ in, float *__restrict__ out, int nl) {
    constexpr int WAVES_T = WAVES / TILES, THREADS_T = THREADS / TILES;
    extern __shared__ uint4v ring[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = lane >> 4;
    const long row0 = ((long)blockIdx.x * WAVES_T + wave) * 16 * TILES + (lane & 15);

    const uint4v *g = wstream + threadIdx.x;
    int cur = 1;
    int n_issued = 0;
    auto issue = [&]() {
#ifdef PROBE_NO_DMA
        if (n_issued >= 2) { g += CHUNK_U4; return; }        // timing experiment: the ring keeps its first two chunks (results invalid)
#endif
        ++n_issued;
        uint4v *dst = ring + (cur ^ 1) * CHUNK_U4 + wave * 64;
#pragma unroll
        for (int i = 0; i < CHUNK_U4 / THREADS_T; ++i)
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)(g + i * THREADS_T), (lds_void_t *)(dst + i * THREADS_T), 16, 0, 0);
        g += CHUNK_U4;
    };
    issue();

    uint4v bh[TILES][NS], bl[TILES][NS];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        float x[H];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const float4 v = *reinterpret_cast<const float4 *>(in + (row0 + 16 * t) * W + 16 * ob + 4 * part);
            x[4 * ob] = v.x; x[4 * ob + 1] = v.y; x[4 * ob + 2] = v.z; x[4 * ob + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) split8<DT>(x + 8 * s, bh[t][s], bl[t][s]);
    }

    floatx4 acc[TILES][NOB];
    for (int l = 0; l < nl; ++l) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const float4 v = *reinterpret_cast<const float4 *>(bias + l * W + 16 * ob + 4 * part);
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[t][ob] = floatx4{v.x, v.y, v.z, v.w};
        }
        static_for<0, NS / SPC>([&](auto cc) {
            __syncthreads();                       // chunk landed (hipcc drains vmcnt in front of the barrier) and the other buffer is free
            cur ^= 1;
            issue();
            if constexpr (PIPE) {
                // explicit software pipeline over the 2 x 4 fragment groups of the chunk: group g + 1 is read before group g's MFMAs issue
                constexpr int NG = SPC * (NOB / 4);
                uint4v fh[2][4], fl[2][4];
                auto rd = [&](auto gc, auto bc) {
                    constexpr int gidx = decltype(gc)::value, buf = decltype(bc)::value;
                    const uint4v *p = ring + cur * CHUNK_U4 + (gidx / (NOB / 4)) * STEP_U4 + lane;
                    constexpr int o0 = (gidx % (NOB / 4)) * 4;
#pragma unroll
                    for (int o = 0; o < 4; ++o) { fh[buf][o] = p[((o0 + o) * 2) * FRAG_U4]; if constexpr (PROD == 3) fl[buf][o] = p[((o0 + o) * 2 + 1) * FRAG_U4]; }
                };
                rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                static_for<0, NG>([&](auto gc) {
                    constexpr int gidx = decltype(gc)::value, buf = gidx & 1;
                    constexpr int s = decltype(cc)::value * SPC + gidx / (NOB / 4), o0 = (gidx % (NOB / 4)) * 4;
                    if constexpr (gidx + 1 < NG) rd(std::integral_constant<int, gidx + 1>{}, std::integral_constant<int, buf ^ 1>{});
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(fh[buf][o], bh[t][s], acc[t][o0 + o]);
                        if constexpr (PROD == 3) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(fl[buf][o], bh[t][s], acc[t][o0 + o]);
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(fh[buf][o], bl[t][s], acc[t][o0 + o]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else
            static_for<0, SPC>([&](auto sc) {
                constexpr int s = decltype(cc)::value * SPC + decltype(sc)::value;
                const uint4v *p = ring + cur * CHUNK_U4 + decltype(sc)::value * STEP_U4 + lane;
#pragma unroll
                for (int o0 = 0; o0 < NOB; o0 += 4) {
                    uint4v ah[4], al[4];
#ifdef PROBE_NO_LDS
                    // timing experiment (results invalid): one fragment group per K-step is read, every output block reuses it
#pragma unroll
                    for (int o = 0; o < 4; ++o) { ah[o] = p[(o * 2) * FRAG_U4]; if constexpr (PROD == 3) al[o] = p[(o * 2 + 1) * FRAG_U4]; }
#else
#pragma unroll
                    for (int o = 0; o < 4; ++o) { ah[o] = p[((o0 + o) * 2) * FRAG_U4]; if constexpr (PROD == 3) al[o] = p[((o0 + o) * 2 + 1) * FRAG_U4]; }
#endif
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
#pragma unroll
                        for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(ah[o], bh[t][s], acc[t][o0 + o]);
                        if constexpr (PROD == 3) {
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(al[o], bh[t][s], acc[t][o0 + o]);
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[t][o0 + o] = mfma16<DT>(ah[o], bl[t][s], acc[t][o0 + o]);
                        }
                    }
                }
            });
        });
        if (l + 1 < nl) {
            // ReLU + re-split: the accumulators of blocks 2s, 2s+1 are the B operands of K-step s of the next layer
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float y[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] = __int_as_float(max(__float_as_int(acc[t][2 * s + (j >> 2)][j & 3]), 0));
                    split8<DT>(y, bh[t][s], bl[t][s]);
                }
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
            *reinterpret_cast<float4 *>(out + (row0 + 16 * t) * W + 16 * ob + 4 * part) = make_float4(acc[t][ob][0], acc[t][ob][1], acc[t][ob][2], acc[t][ob][3]);
}

// ---- host ------------------------------------------------------------------------------------------------------------------
static uint16_t f32_to_f16_rtz_bits(float f) {        // exact for values whose mantissa fits; subnormals by truncation
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xff) - 127 + 15;
    uint32_t m = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0) return (uint16_t)sign;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        return (uint16_t)(sign | (m >> (14 - e)));
    }
    return (uint16_t)(sign | (e << 10) | (m >> 13));
}
static float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    int e = (h >> 10) & 31; uint32_t m = h & 0x3ffu;
    float f;
    if (e == 0) { f = ldexpf((float)m, -24); uint32_t u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    uint32_t u = sign | ((uint32_t)(e - 15 + 127) << 23) | (m << 13);
    memcpy(&f, &u, 4); return f;
}
// (hi, lo) 16-bit parts of a weight: hi = round-to-nearest-ish (truncate after adding half an ulp), lo = rest, truncated
static void split_weight(int dt, float w, uint16_t &hi, uint16_t &lo) {
    if (dt == 0) {
        uint32_t u; memcpy(&u, &w, 4);
        u = (u + 0x1000u) & 0xffffe000u;                  // round the mantissa to 10 bits
        float h; memcpy(&h, &u, 4);
        hi = f32_to_f16_rtz_bits(h);
        const float r = w - f16_bits_to_f32(hi);
        uint32_t v; memcpy(&v, &r, 4);
        v = (v + 0x1000u) & 0xffffe000u;
        float rl; memcpy(&rl, &v, 4);
        lo = f32_to_f16_rtz_bits(rl);
    } else {
        uint32_t u; memcpy(&u, &w, 4);
        u = (u + 0x8000u) & 0xffff0000u;
        float h; memcpy(&h, &u, 4);
        hi = (uint16_t)(u >> 16);
        const float r = w - h;
        uint32_t v; memcpy(&v, &r, 4);
        v = (v + 0x8000u) & 0xffff0000u;
        lo = (uint16_t)(v >> 16);
    }
}
static int fin(int s, int p, int j) { return 32 * s + 16 * (j >> 2) + 4 * p + (j & 3); }

template <int DT, int PROD, int TILES = 1, bool PIPE = false>
static void run(const char *name, const std::vector<float> &Wt, const std::vector<float> &bias, const std::vector<float> &in, long rows,
                int nl, int reps) {
    // pack: stream[l][s][ob][hi|lo][lane] = 8 x 16 bit
    std::vector<uint16_t> st((size_t)nl * NS * NOB * 2 * 64 * 8);
    size_t o = 0;
    for (int l = 0; l < nl; ++l)
        for (int s = 0; s < NS; ++s)
            for (int ob = 0; ob < NOB; ++ob)
                for (int hl = 0; hl < 2; ++hl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            uint16_t hi, lo;
                            split_weight(DT, Wt[((size_t)l * W + 16 * ob + (lane & 15)) * W + fin(s, lane >> 4, j)], hi, lo);
                            st[o++] = hl ? lo : hi;
                        }
    uint4v *d_w; float *d_b, *d_in, *d_out;
    hipMalloc(&d_w, st.size() * 2 + CHUNK_U4 * 16);      // + one chunk: the stream prefetches one past the end
    hipMemset(d_w, 0, st.size() * 2 + CHUNK_U4 * 16);
    hipMemcpy(d_w, st.data(), st.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&d_b, bias.size() * 4); hipMemcpy(d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_in, (size_t)rows * W * 4); hipMemcpy(d_in, in.data(), (size_t)rows * W * 4, hipMemcpyHostToDevice);
    hipMalloc(&d_out, (size_t)rows * W * 4);
    auto kern = k_split<DT, PROD, TILES, PIPE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CHUNK_U4 * 16);
    const dim3 grid((unsigned)(rows / (WAVES * 16)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(THREADS / TILES), 2 * CHUNK_U4 * 16, 0, d_w, d_b, d_in, d_out, nl);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(THREADS / TILES), 2 * CHUNK_U4 * 16, 0, d_w, d_b, d_in, d_out, nl);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const hipError_t err = hipGetLastError();
    // accuracy: fp64 chain on a sample of rows
    std::vector<float> res((size_t)rows * W);
    hipMemcpy(res.data(), d_out, res.size() * 4, hipMemcpyDeviceToHost);
    double max_err = 0, max_ref = 0, se = 0, sr = 0;
    for (long r = 0; r < rows; r += rows / 64) {
        std::vector<double> a(W), b(W);
        for (int k = 0; k < W; ++k) a[k] = in[(size_t)r * W + k];
        for (int l = 0; l < nl; ++l) {
            for (int n = 0; n < W; ++n) {
                double s = bias[(size_t)l * W + n];
                for (int k = 0; k < W; ++k) s += (double)Wt[((size_t)l * W + n) * W + k] * a[k];
                b[n] = s;
            }
            if (l + 1 < nl) for (int n = 0; n < W; ++n) a[n] = b[n] > 0 ? b[n] : 0;
        }
        for (int n = 0; n < W; ++n) {
            const double e = fabs((double)res[(size_t)r * W + n] - b[n]);
            max_err = fmax(max_err, e); max_ref = fmax(max_ref, fabs(b[n])); se += e * e; sr += b[n] * b[n];
        }
    }
    const double tf = (double)rows * nl * 2.0 * W * W / (ms * 1e-3) / 1e12;
    printf("{\"variant\": \"%s\", \"layers\": %d, \"rows\": %ld, \"ms\": %.4f, \"effective_tflops\": %.1f, \"mfma_tflops_issued\": %.1f, "
           "\"max_abs_err_over_max_abs_ref\": %.3e, \"rms_err_over_rms_ref\": %.3e, \"hip_error\": \"%s\"}\n",
           name, nl, rows, ms, tf, tf * PROD, max_err / max_ref, sqrt(se / sr), hipGetErrorString(err));
    hipFree(d_w); hipFree(d_b); hipFree(d_in); hipFree(d_out);
}

int main(int argc, char **argv) {
    const long rows = argc > 1 ? atol(argv[1]) : 196608;       // multiple of 128
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int NLMAX = 8;
    std::vector<float> Wt((size_t)NLMAX * W * W), bias((size_t)NLMAX * W), in((size_t)rows * W);
    uint64_t sd = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0; };
    // nn.Linear-like weights U(-1/16, 1/16) scaled by sqrt(6) so that activations keep their magnitude through ReLU layers
    for (auto &w : Wt) w = (float)((rnd() * 2 - 1) / 16.0 * 2.449);
    for (auto &b : bias) b = (float)((rnd() * 2 - 1) / 16.0);
    // golden-like activations: post-ReLU (half of them zero), log-uniform magnitudes over 1e-4 .. 10
    for (auto &v : in) { const double u = rnd(); v = u < 0.5 ? 0.f : (float)exp(log(1e-4) + (log(10.0) - log(1e-4)) * rnd()); }
    for (int nl : {1, 8}) {
        run<0, 3>("f16 hi/lo, 3 products", Wt, bias, in, rows, nl, reps);
        run<1, 3>("bf16 hi/lo, 3 products", Wt, bias, in, rows, nl, reps);
        run<0, 3, 2>("f16 hi/lo, 3 products, 2 tiles per wavefront (4 wavefronts)", Wt, bias, in, rows, nl, reps);
        run<0, 3, 1, true>("f16 hi/lo, 3 products, fragment reads one group ahead", Wt, bias, in, rows, nl, reps);
        run<0, 3, 2, true>("f16 hi/lo, 3 products, 2 tiles per wavefront, fragment reads one group ahead", Wt, bias, in, rows, nl, reps);
        run<0, 1, 2>("plain f16 (1 product), 2 tiles per wavefront", Wt, bias, in, rows, nl, reps);
        run<0, 1>("plain f16 (1 product)", Wt, bias, in, rows, nl, reps);
        run<1, 1>("plain bf16 (1 product)", Wt, bias, in, rows, nl, reps);
    }
    return 0;
}
