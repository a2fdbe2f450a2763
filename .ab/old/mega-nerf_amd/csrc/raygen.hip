// raygen.hip -- ray generation (mega_nerf/ray_utils.py) as elementwise gfx950 kernels.
// Compiled with -ffp-contract=off: every mul/add rounds separately, like the torch CPU kernels.
#include "common.h"

namespace mnr {

// ray_utils.py:6-18
__global__ void k_ray_dirs(float *__restrict__ out, int W, int H, float fx, float fy, float cx, float cy, float c) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)W * H) return;
    const float i = (float)(p % W) + c, j = (float)(p / W) + c;
    const float dx = (i - cx) / fx, dy = -(j - cy) / fy, dz = -1.f;
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    out[3 * p + 0] = dx / n;
    out[3 * p + 1] = dy / n;
    out[3 * p + 2] = dz / n;
}

// ray_utils.py:65-84: distance from the origin to the intersection with the plane x = altitude
__device__ __forceinline__ float plane_bound(float ox, float oy, float oz, float dx, float dy, float dz, float alt) {
    const float ndotu = -dx;                               // d . (-1,0,0)
    const float wx = ox - alt, wy = oy, wz = oz;           // w = o - plane_point
    const float si = wx / ndotu;                           // -(w . n) / ndotu,  w . n = -wx
    const float ix = wx + si * dx + alt, iy = wy + si * dy, iz = wz + si * dz;
    const float ex = ox - ix, ey = oy - iy, ez = oz - iz;
    return sqrtf(ex * ex + ey * ey + ez * ez);
}

// ray_utils.py:21-62 for one (direction, pose) pair
__device__ __forceinline__ void ray_from_pose(float *__restrict__ out8, const float *__restrict__ d, const float *__restrict__ m,
                                              float near, float far, int has_alt, float alt0, float alt1) {
    const float a = d[0], b = d[1], c = d[2];
    float rx = a * m[0] + b * m[1] + c * m[2];
    float ry = a * m[4] + b * m[5] + c * m[6];
    float rz = a * m[8] + b * m[9] + c * m[10];
    const float n = sqrtf(rx * rx + ry * ry + rz * rz);
    rx /= n; ry /= n; rz /= n;
    const float ox = m[3], oy = m[7], oz = m[11];
    float nb = near, fb = far;
    if (has_alt) {
        if (ox < alt0 && rx > 0.f) nb = plane_bound(ox, oy, oz, rx, ry, rz, alt0);
        nb = fmaxf(nb, near);
        if (ox < alt1 && rx > 0.f) fb = plane_bound(ox, oy, oz, rx, ry, rz, alt1);
        fb = fminf(fb, far);
        fb = fmaxf(nb, fb);
    }
    float4 *o4 = reinterpret_cast<float4 *>(out8);
    o4[0] = make_float4(ox, oy, oz, rx);
    o4[1] = make_float4(ry, rz, nb, fb);
}

__global__ void k_get_rays(float *__restrict__ out, const float *__restrict__ dirs, long P, int dirs_shared,
                           const float *__restrict__ c2w, int n_poses, float near, float far, int has_alt, float alt0,
                           float alt1) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P * n_poses) return;
    const long pose = t / P, p = t % P;
    ray_from_pose(out + t * 8, dirs + (dirs_shared ? p : t) * 3, c2w + pose * 12, near, far, has_alt, alt0, alt1);
}

// filesystem_dataset.py:96-124: the ray of every (image, pixel) pair of a shuffled training chunk, straight from the two
// index columns (the reference builds a (#unique images x #unique pixels) ray table per 64 K rows and gathers from it)
__global__ void k_get_rays_indexed(float *__restrict__ out, const float *__restrict__ dirs, long n_dirs,
                                   const int32_t *__restrict__ pixel_idx, const float *__restrict__ c2w, int n_poses,
                                   const int32_t *__restrict__ img_idx, long M, float near, float far, int has_alt, float alt0,
                                   float alt1, int32_t *__restrict__ err) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    long p = pixel_idx[t], im = img_idx[t];
    if (p < 0 || p >= n_dirs || im < 0 || im >= n_poses) {      // corrupt chunk: flag it, stay in bounds
        if (err) atomicOr(err, 1);
        p = p < 0 ? 0 : (p >= n_dirs ? n_dirs - 1 : p);
        im = im < 0 ? 0 : (im >= n_poses ? n_poses - 1 : im);
    }
    ray_from_pose(out + t * 8, dirs + p * 3, c2w + im * 12, near, far, has_alt, alt0, alt1);
}

}  // namespace mnr

using namespace mnr;

extern "C" int mnr_ray_directions(float *out_dev, int W, int H, float fx, float fy, float cx, float cy,
                                  int center_pixels, void *stream) {
    MNR_REQUIRE(out_dev && W > 0 && H > 0, "bad arguments to mnr_ray_directions");
    const long n = (long)W * H;
    hipLaunchKernelGGL(k_ray_dirs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), out_dev, W, H, fx,
                       fy, cx, cy, center_pixels ? 0.5f : 0.f);
    return check_launch("k_ray_dirs");
}

extern "C" int mnr_get_rays(float *out_dev, const float *dirs_dev, int64_t P, int n_dirs_sets, const float *c2w_dev,
                            int n_poses, float near, float far, const float *alt, void *stream) {
    MNR_REQUIRE(P >= 0 && n_poses >= 0, "bad arguments to mnr_get_rays");
    const long n = (long)P * n_poses;
    if (n == 0) return MNR_OK;                       // empty input: nothing to enqueue
    MNR_REQUIRE(out_dev && dirs_dev && c2w_dev, "NULL pointer passed to mnr_get_rays");
    MNR_REQUIRE(n_dirs_sets == 1 || n_dirs_sets == n_poses, "n_dirs_sets must be 1 or n_poses");
    MNR_REQUIRE((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0, "out_dev must be 16-byte aligned");
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), out_dev, dirs_dev,
                       (long)P, n_dirs_sets == 1 ? 1 : 0, c2w_dev, n_poses, near, far,
                       alt ? 1 : 0, alt ? alt[0] : 0.f, alt ? alt[1] : 0.f);
    return check_launch("k_get_rays");
}

extern "C" int mnr_get_rays_indexed(float *out_dev, const float *dirs_dev, int64_t n_dirs, const int32_t *pixel_idx_dev,
                                    const float *c2w_dev, int n_poses, const int32_t *img_idx_dev, int64_t M, float near, float far,
                                    const float *alt, int32_t *err_flag_dev, void *stream) {
    MNR_REQUIRE(M >= 0 && n_dirs > 0 && n_poses > 0, "bad arguments to mnr_get_rays_indexed");
    if (M == 0) return MNR_OK;
    MNR_REQUIRE(out_dev && dirs_dev && pixel_idx_dev && c2w_dev && img_idx_dev, "NULL pointer passed to mnr_get_rays_indexed");
    MNR_REQUIRE((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0, "out_dev must be 16-byte aligned");
    hipLaunchKernelGGL(k_get_rays_indexed, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), out_dev, dirs_dev,
                       (long)n_dirs, pixel_idx_dev, c2w_dev, n_poses, img_idx_dev, (long)M, near, far, alt ? 1 : 0,
                       alt ? alt[0] : 0.f, alt ? alt[1] : 0.f, err_flag_dev);
    return check_launch("k_get_rays_indexed");
}
