// cluster_mask.hip -- the inner loop of scripts/create_cluster_masks.py (reference :157-187) as one kernel.
//
// For every ray: S samples z = near (1 - t_s) + far t_s, the distance of every sample to every cell centroid,
// ratio = dist_j / (min_j dist_j + 1e-8) per sample, and per cell the minimum ratio over the samples of the ray.
// The reference materialises (rays x S x cells) distance tensors; here a thread owns a ray, walks its samples and keeps
// one running minimum per cell in registers, so HBM sees 32 B in and 4 * cells B out per ray.  The kernel is VALU-bound:
// the reference's bits need an IEEE sqrt and an IEEE divide per (sample, cell) pair, ~34 VALU instructions; a
// conservative squared-distance filter plus a low-discrepancy sample order (see the loop) dismisses most pairs before
// either, ~14 instructions per pair (MI355X, 4608 x 3456 x 1000 x 8: 110 ms -> 47 ms per image, same bits).
//
// Distances follow torch.cdist's matmul formulation (it is taken whenever either side has more than 25 rows):
//   row  = [-2 x, |x|^2, 1],  col = [c, 1, |c|^2],  d = sqrt(max(row . col, 0))
// accumulated as an fma chain in column order (what MKL sgemm does for K = 5; verified against torch on CPU),
// |.|^2 summed left to right.  Compiled with -ffp-contract=off so that only the explicit fmaf calls fuse.
#include "common.h"

namespace mnr {

template <int MAXC>
__global__ __launch_bounds__(256) void k_cluster_ratios(float *__restrict__ ratios, uint8_t *__restrict__ masks,
                                                        const float *__restrict__ rays, long n_rays,
                                                        const float *__restrict__ z_steps, int S,
                                                        const float *__restrict__ centroids, int n_c, int dim0, float margin) {
    __shared__ float4 cs[MAXC];   // (c_a, c_b, c_c, |c|^2); 2-D clustering leaves c_a unused
    __shared__ float zs[1024];
    if (threadIdx.x < MAXC) {
        float4 c = {0.f, 0.f, 0.f, 0.f};
        if ((int)threadIdx.x < n_c) {
            const float *p = centroids + 3 * threadIdx.x;
            c.x = p[0]; c.y = p[1]; c.z = p[2];
            c.w = dim0 == 0 ? (c.x * c.x + c.y * c.y) + c.z * c.z : c.y * c.y + c.z * c.z;
        }
        cs[threadIdx.x] = c;
    }
    const long ray = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = ray < n_rays;
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f}, near = 0.f, far = 0.f;
    if (live) {
        const float4 a = reinterpret_cast<const float4 *>(rays + ray * 8)[0], b = reinterpret_cast<const float4 *>(rays + ray * 8)[1];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; d[0] = a.w; d[1] = b.x; d[2] = b.y; near = b.z; far = b.w;
    }
    float best[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) best[j] = INFINITY;

    for (int s0 = 0; s0 < S; s0 += 1024) {
        __syncthreads();
        for (int i = threadIdx.x; i < 1024 && s0 + i < S; i += blockDim.x) zs[i] = z_steps[s0 + i];
        __syncthreads();
        const int ns = min(1024, S - s0);
        int pow2 = 1;
        while (pow2 < ns) pow2 <<= 1;
        // Samples are visited in a low-discrepancy order (odd stride modulo a power of two = a bijection): after a few
        // dozen samples every running minimum is close to final, and a pair whose squared distance cannot beat it is
        // dismissed before its IEEE sqrt and divide (most pairs, coherently across the wave).  min is order-independent,
        // so the result is bit-identical to the in-order scan.
        const int stride = (int)(0.6180339887f * pow2) | 1;
        for (int it = 0; it < pow2; ++it) {
            const int s = (it * stride) & (pow2 - 1);
            if (s >= ns) continue;
            const float t = zs[s];
            const float z = near * (1.f - t) + far * t;
            const float p0 = o[0] + d[0] * z, p1 = o[1] + d[1] * z, p2 = o[2] + d[2] * z;
            const float xn = dim0 == 0 ? (p0 * p0 + p1 * p1) + p2 * p2 : p1 * p1 + p2 * p2;
            const float a0 = -2.f * p0, a1 = -2.f * p1, a2 = -2.f * p2;
            float sq[MAXC];
            float sqmin = INFINITY;
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                // (eight centroid reads in flight at most: hipcc hoisted all MAXC float4 reads in front of the loop -- 256 registers at 64
                // cells, 96 VGPRs spilled out of a 512-register kernel)
                if (j % 8 == 0 && j > 0) asm volatile("" ::: "memory");
                const float4 c = cs[j];
                float acc = dim0 == 0 ? fmaf(a1, c.y, a0 * c.x) : a1 * c.y;
                acc = fmaf(a2, c.z, acc);
                acc = acc + xn;            // fma(|x|^2, 1, acc)
                acc = acc + c.w;           // fma(1, |c|^2, acc)
                sq[j] = j < n_c ? fmaxf(acc, 0.f) : INFINITY;
                sqmin = fminf(sqmin, sq[j]);
            }
            const float den = sqrtf(sqmin) + 1e-8f;   // sqrt is monotonic: min of the roots = root of the min
#pragma unroll
            for (int j = 0; j < MAXC; ++j) {
                // dist/den < best  needs  sq < (best den)^2 up to rounding; 1e-6 relative slack keeps the filter conservative
                const float lim = best[j] * den;
                // (j < n_c: a padded slot has sq = best = inf, and inf <= inf sent it through the sqrt and the divide for every sample -- 36 cells
                // in the 64-slot instantiation ran slower than 64)
                if (j < n_c && sq[j] <= lim * lim * 1.000001f) best[j] = fminf(best[j], sqrtf(sq[j]) / den);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
        if (j < n_c) {
            if (ratios) ratios[ray * n_c + j] = best[j];
            if (masks) masks[(long)j * n_rays + ray] = best[j] <= margin ? 1 : 0;
        }
}

template <int MAXC>
static int launch(float *ratios, uint8_t *masks, const float *rays, int64_t n_rays, const float *z_steps, int S,
                  const float *centroids, int n_c, int dim0, float margin, hipStream_t st) {
    hipLaunchKernelGGL(k_cluster_ratios<MAXC>, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, ratios, masks, rays,
                       (long)n_rays, z_steps, S, centroids, n_c, dim0, margin);
    return check_launch("k_cluster_ratios");
}

}  // namespace mnr

using namespace mnr;

extern "C" int mnr_cluster_min_ratios(float *ratios_out, uint8_t *masks_out, const float *rays, int64_t n_rays,
                                      const float *z_steps, int n_samples, const float *centroids, int n_centroids,
                                      int cluster_2d, float boundary_margin, void *stream) {
    MNR_REQUIRE(rays && z_steps && centroids && (ratios_out || masks_out), "null pointer passed to mnr_cluster_min_ratios");
    MNR_REQUIRE(n_rays >= 0 && n_samples > 0 && n_centroids > 0, "bad sizes passed to mnr_cluster_min_ratios");
    if (n_centroids > 64) return set_err(MNR_E_UNSUPPORTED, "mnr_cluster_min_ratios supports at most 64 centroids (got %d)", n_centroids);
    if (n_rays == 0) return MNR_OK;
    MNR_REQUIRE((n_rays + 255) / 256 < (1LL << 31), "too many rays for one mnr_cluster_min_ratios launch");
    const int dim0 = cluster_2d ? 1 : 0;
    hipStream_t st = as_stream(stream);
    if (n_centroids <= 8) return launch<8>(ratios_out, masks_out, rays, n_rays, z_steps, n_samples, centroids, n_centroids, dim0, boundary_margin, st);
    if (n_centroids <= 16) return launch<16>(ratios_out, masks_out, rays, n_rays, z_steps, n_samples, centroids, n_centroids, dim0, boundary_margin, st);
    if (n_centroids <= 32) return launch<32>(ratios_out, masks_out, rays, n_rays, z_steps, n_samples, centroids, n_centroids, dim0, boundary_margin, st);
    return launch<64>(ratios_out, masks_out, rays, n_rays, z_steps, n_samples, centroids, n_centroids, dim0, boundary_margin, st);
}
