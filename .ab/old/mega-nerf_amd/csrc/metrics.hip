// metrics.hip -- validation metrics on the device (reference mega_nerf/metrics.py:8-10 PSNR, :51-121 SSIM; the reference
// copies both images to the host and runs them through ATen CPU convolutions, runner.py:413-436).
//
// One pass over the two images: a workgroup owns a 16 x 32 pixel tile, stages the tile + 5-pixel halo of one colour channel
// of both images in LDS (zero outside the image = the reference's zero-padded conv2d), blurs the five moments
// (x, y, x^2, y^2, x y) along the row, then along the column (separable 11-tap Gaussian, same order as the reference:
// filt_fn1(filt_fn2(z))), forms the SSIM map value and adds the tile's SSIM sum and squared-error sum to two double
// accumulators.  HBM-bound: 24 B per pixel in, nothing out.
#include "common.h"

namespace mnr {

constexpr int MT_H = 16, MT_W = 32, MT_MAXF = 33;

__global__ __launch_bounds__(256) void k_image_metrics(const float *__restrict__ pred, const float *__restrict__ target, int H, int W,
                                                       long row_stride, const float *__restrict__ filt_dev, int fs, float c1, float c2,
                                                       double *__restrict__ acc) {
    extern __shared__ float lds[];
    const int hw = fs / 2;
    const int RH = MT_H + 2 * hw, RW = MT_W + 2 * hw;
    float *A = lds, *B = A + RH * RW, *Hb = B + RH * RW;       // Hb: [5][RH][MT_W] row-blurred moments
    __shared__ float filt[MT_MAXF];
    __shared__ double red[2][4];
    if ((int)threadIdx.x < fs) filt[threadIdx.x] = filt_dev[threadIdx.x];
    const int y0 = blockIdx.y * MT_H, x0 = blockIdx.x * MT_W;
    double ssim_sum = 0.0, se_sum = 0.0;
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
        for (int e = threadIdx.x; e < RH * RW; e += 256) {
            const int ry = e / RW, rx = e % RW;
            const int y = y0 + ry - hw, x = x0 + rx - hw;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            A[e] = in ? pred[y * row_stride + 3 * x + c] : 0.f;
            B[e] = in ? target[y * row_stride + 3 * x + c] : 0.f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < RH * MT_W; e += 256) {
            const int ry = e / MT_W, x = e % MT_W;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
            for (int k = 0; k < fs; ++k) {
                const float f = filt[k], t = A[ry * RW + x + k], u = B[ry * RW + x + k];
                s0 = fmaf(f, t, s0); s1 = fmaf(f, u, s1);
                s2 = fmaf(f, t * t, s2); s3 = fmaf(f, u * u, s3); s4 = fmaf(f, t * u, s4);
            }
            Hb[0 * RH * MT_W + e] = s0; Hb[1 * RH * MT_W + e] = s1; Hb[2 * RH * MT_W + e] = s2;
            Hb[3 * RH * MT_W + e] = s3; Hb[4 * RH * MT_W + e] = s4;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < MT_H * MT_W; e += 256) {
            const int ty = e / MT_W, x = e % MT_W;
            if (y0 + ty >= H || x0 + x >= W) continue;
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < fs; ++k) {
                const float f = filt[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] = fmaf(f, Hb[q * RH * MT_W + (ty + k) * MT_W + x], m[q]);
            }
            const float mu00 = m[0] * m[0], mu11 = m[1] * m[1], mu01 = m[0] * m[1];
            const float s00 = fmaxf(m[2] - mu00, 0.f), s11 = fmaxf(m[3] - mu11, 0.f);
            float s01 = m[4] - mu01;
            const float lim = fminf(sqrtf(s00 * s11), fabsf(s01));
            s01 = s01 > 0.f ? lim : (s01 < 0.f ? -lim : 0.f);       // sign(s01) * min(sqrt(s00 s11), |s01|)
            const float numer = (2.f * mu01 + c1) * (2.f * s01 + c2), denom = (mu00 + mu11 + c1) * (s00 + s11 + c2);
            ssim_sum += (double)(numer / denom);
            const float d = A[(ty + hw) * RW + x + hw] - B[(ty + hw) * RW + x + hw];
            se_sum += (double)(d * d);
        }
    }
    // workgroup reduction -> one atomic pair per tile
    for (int o = 32; o > 0; o >>= 1) {
        ssim_sum += __shfl_down(ssim_sum, o);
        se_sum += __shfl_down(se_sum, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = ssim_sum; red[1][wave] = se_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc + 0, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        atomicAdd(acc + 1, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    }
}

}  // namespace mnr

using namespace mnr;

extern "C" int mnr_image_metrics(const float *pred, const float *target, int H, int W, int64_t row_stride, const float *filter_dev,
                                 int filter_size, float max_val, float k1, float k2, double *acc_dev, void *stream) {
    MNR_REQUIRE(pred && target && filter_dev && acc_dev, "null pointer passed to mnr_image_metrics");
    MNR_REQUIRE(H > 0 && W > 0 && row_stride >= 3L * W, "bad image shape passed to mnr_image_metrics");
    MNR_REQUIRE(filter_size >= 1 && filter_size <= MT_MAXF && (filter_size & 1), "filter_size must be odd and <= 33");
    const int hw = filter_size / 2, RH = MT_H + 2 * hw, RW = MT_W + 2 * hw;
    const size_t lds = (size_t)(2 * RH * RW + 5 * RH * MT_W) * sizeof(float);
    const float c1 = (k1 * max_val) * (k1 * max_val), c2 = (k2 * max_val) * (k2 * max_val);
    static size_t lds_enabled_dev[MAX_DEVICES] = {};            // function attributes are per device
    size_t &lds_enabled = lds_enabled_dev[device_slot()];
    if (lds > 64 * 1024 && lds > lds_enabled) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_image_metrics), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "hipFuncSetAttribute(k_image_metrics)");
        lds_enabled = 160 * 1024;
    }
    hipLaunchKernelGGL(k_image_metrics, dim3((W + MT_W - 1) / MT_W, (H + MT_H - 1) / MT_H), dim3(256), lds, as_stream(stream), pred, target,
                       H, W, (long)row_stride, filter_dev, filter_size, c1, c2, acc_dev);
    return check_launch("k_image_metrics");
}
