// sh_device.h -- real spherical-harmonics colour basis (reference mega_nerf/spherical_harmonics.py:55-107), deg <= 4.
#pragma once
#include <hip/hip_runtime.h>

namespace mnr {

// spherical_harmonics.py:55-107 (deg <= 4), coefficients c[k] for one colour channel
__device__ __forceinline__ float eval_sh_channel(int deg, const float *c, float x, float y, float z) {
    float r = 0.28209479177387814f * c[0];
    if (deg > 0) {
        r = r - 0.4886025119029199f * y * c[1] + 0.4886025119029199f * z * c[2] - 0.4886025119029199f * x * c[3];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + 1.0925484305920792f * xy * c[4] + -1.0925484305920792f * yz * c[5] +
                0.31539156525252005f * (2.0f * zz - xx - yy) * c[6] + -1.0925484305920792f * xz * c[7] +
                0.5462742152960396f * (xx - yy) * c[8];
            if (deg > 2) {
                r = r + -0.5900435899266435f * y * (3 * xx - yy) * c[9] + 2.890611442640554f * xy * z * c[10] +
                    -0.4570457994644658f * y * (4 * zz - xx - yy) * c[11] +
                    0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy) * c[12] +
                    -0.4570457994644658f * x * (4 * zz - xx - yy) * c[13] + 1.445305721320277f * z * (xx - yy) * c[14] +
                    -0.5900435899266435f * x * (xx - 3 * yy) * c[15];
                if (deg > 3) {
                    r = r + 2.5033429417967046f * xy * (xx - yy) * c[16] + -1.7701307697799304f * yz * (3 * xx - yy) * c[17] +
                        0.9461746957575601f * xy * (7 * zz - 1) * c[18] + -0.6690465435572892f * yz * (7 * zz - 3) * c[19] +
                        0.10578554691520431f * (zz * (35 * zz - 30) + 3) * c[20] +
                        -0.6690465435572892f * xz * (7 * zz - 3) * c[21] + 0.47308734787878004f * (xx - yy) * (7 * zz - 1) * c[22] +
                        -1.7701307697799304f * xz * (xx - 3 * yy) * c[23] +
                        0.6258357354491761f * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * c[24];
                }
            }
        }
    }
    return r;
}

// the (deg + 1)^2 basis values themselves (eval_sh_channel is linear in c): gradient of the colour w.r.t. c[k]
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = 0.28209479177387814f;
    if (deg < 1) return;
    b[1] = -0.4886025119029199f * y; b[2] = 0.4886025119029199f * z; b[3] = -0.4886025119029199f * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = 1.0925484305920792f * xy; b[5] = -1.0925484305920792f * yz; b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    b[7] = -1.0925484305920792f * xz; b[8] = 0.5462742152960396f * (xx - yy);
    if (deg < 3) return;
    b[9] = -0.5900435899266435f * y * (3 * xx - yy); b[10] = 2.890611442640554f * xy * z;
    b[11] = -0.4570457994644658f * y * (4 * zz - xx - yy); b[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
    b[13] = -0.4570457994644658f * x * (4 * zz - xx - yy); b[14] = 1.445305721320277f * z * (xx - yy);
    b[15] = -0.5900435899266435f * x * (xx - 3 * yy);
    if (deg < 4) return;
    b[16] = 2.5033429417967046f * xy * (xx - yy); b[17] = -1.7701307697799304f * yz * (3 * xx - yy);
    b[18] = 0.9461746957575601f * xy * (7 * zz - 1); b[19] = -0.6690465435572892f * yz * (7 * zz - 3);
    b[20] = 0.10578554691520431f * (zz * (35 * zz - 30) + 3); b[21] = -0.6690465435572892f * xz * (7 * zz - 3);
    b[22] = 0.47308734787878004f * (xx - yy) * (7 * zz - 1); b[23] = -1.7701307697799304f * xz * (xx - 3 * yy);
    b[24] = 0.6258357354491761f * (xx * (xx - 3 * yy) - yy * (3 * xx - yy));
}

}  // namespace mnr
