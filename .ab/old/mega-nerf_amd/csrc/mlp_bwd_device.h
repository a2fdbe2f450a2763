// mlp_bwd_device.h -- argument block and per-lane helpers of the data-gradient chain kernels (csrc/mlp_bwd.hip: fp32 matrix pipe;
// csrc/mlp_bwd_h2.hip: split-precision).  Design notes: mlp_bwd.hip.
#pragma once
#include "mlp_device.h"

namespace mnr {

struct MlpBwdArgs {
    const float4 *chunks;          // backward chunk stream
    const float *aux;              // forward aux image (sigma / rgb head weights in lane order)
    const float *tape;             // forward activations
    float *gtape;                  // gradient tape (same plane layout): dZ of every layer
    long tape_rows;
    TapeLayout tl;
    const float *d_out;  long d_out_stride;     // dL/d(out) [rows][>=4]
    const float *out;    long out_stride;       // forward output (rgb after sigmoid, sigma after activation)
    float *dheads;                 // [rows][4]: dL/d(rgb pre-sigmoid) x3, dL/d(sigma pre-activation)
    float *d_emb_a;                // [appearance_count][APP] gradient, atomically accumulated (may be NULL)
    const void *idx;  long idx_stride;  int idx_is_float;
    int rows_per_ray, app_count, sigma_act, sigma_off, rgb_off;
    long n_rows;
    const int32_t *n_units_dev;  int rows_per_unit;
    long tape_row0;                // tape / gradient-tape row of this launch's row 0
    const float *dd_in;            // rgb_dim != 3: dL/d(dir_a output) [n_rows][W/2] supplied by the caller (see mnr_mlp_grad_io)
    const MlpCellSeg *dcells;      // several cells' rows side by side in one segment (device table; csrc/step.hip), else NULL:
    long cell_rows;                // cell c owns rows [c * cell_rows, ...) of d_out / out / idx space and tape rows from its tape_row0
    long aux_byte_off;             // offset of the aux block inside a forward image (dcells)
};

// ReLU masks: the forward pass left the sign bits of every activation packed per lane (TapeLayout mask planes), so a
// layer's mask is NH/32 words per lane -- loaded before the layer's MFMA loop, consumed after it.
template <int NH>
struct MaskBits { uint32_t w[(NH + 31) / 32]; };
template <int NH>
__device__ __forceinline__ MaskBits<NH> mask_load(const float *plane, long row, int width, int part) {
    constexpr int NW = (NH + 31) / 32;
    const uint32_t *r = reinterpret_cast<const uint32_t *>(plane) + row * width + part * NW;
    MaskBits<NH> m;
    if constexpr (NW == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(r); m.w[0] = v.x; m.w[1] = v.y; }
    else {
#pragma unroll
        for (int i = 0; i < NW; ++i) m.w[i] = r[i];
    }
    return m;
}
// dZ = dH masked by ReLU'
template <int NH>
__device__ __forceinline__ void mask_apply(float (&g)[NH], const MaskBits<NH> &m) {
#pragma unroll
    // bit -> 0 / ~0 by a sign-extending 1-bit field extract, then AND: 2 VALU instructions per value (select form: 3).
    // Only the extract is inline asm (LLVM canonicalises `x & sext(bit)` back into compare + select; the asm result is opaque to it).
    // The AND must stay compiler-visible: g[] arrives straight from MFMA accumulators, and the hazard recognizer inserts the
    // MFMA-result -> VALU-read wait states only for instructions it knows -- an inline-asm v_and on an accumulator register read
    // half-finished sums (16x16x32 f16 MFMAs, round 3).
    for (int i = 0; i < NH; ++i) {
        int t;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(m.w[i / 32]), "n"(i % 32));
        g[i] = __int_as_float(__float_as_int(g[i]) & t);
    }
}
// write dZ to the gradient tape (flat register i <-> feature 4P*(i/4) + 4*part + i%4).  Callers issue this right
// AFTER a chunk barrier of the next layer: a barrier drains vmcnt, so a store issued just before one would stall the
// wave for a full HBM write round trip; issued after it, the store has a whole chunk of MFMAs to complete.
// Addressing: uniform plane base (SGPR pair) + this lane's 32-bit byte offset (row * width + 4 part floats; < 2^32, checked by the
// host) + immediate: `global_store_dwordx4 v_off, v[data], s[plane] offset:imm` (mlp_device.h gstore4).
template <int P, int NH>
__device__ __forceinline__ void gtape_store(const float (&g)[NH], const float *plane, unsigned row_byte_off, bool valid) {
    if (!valid) return;
    static_for<0, NH / 4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        gstore4<16 * P * q>(plane, row_byte_off, make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]));
    });
}

template <int NOB, class AccT>
__device__ __forceinline__ void zero_acc(AccT (&acc)[NOB]) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc[ob] = AccT(0.f);
}

// host side of one data-gradient segment: validates the io and fills the argument block (mlp_bwd.hip)
int fill_bwd_args(MlpBwdArgs &a, const ModelLayout &m, const void *packed_fwd_dev, const void *packed_bwd_dev, const mnr_model_desc *d,
                  const mnr_mlp_grad_io *io);

}  // namespace mnr
