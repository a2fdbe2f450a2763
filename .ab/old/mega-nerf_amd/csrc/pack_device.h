// pack_device.h -- nn.Module parameters -> packed MFMA-fragment images, one output element per thread: the bodies of
// k_pack_model (forward image, mlp_pack.hip), k_pack_bwd (transposed image of the data-gradient chain, mlp_bwd.hip) and of
// the step's all-models-in-one-launch form (step.hip).
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace mnr {

__device__ __forceinline__ void pack_model_aux_thread(const ModelLayout &m, float *__restrict__ aux, long a);

// One thread per float4 of the chunk stream, then one thread per float of the aux image.
__device__ __forceinline__ void pack_model_thread(const ModelLayout &m, float4 *__restrict__ chunks, float *__restrict__ aux, long tid) {
    const long n_f4 = (long)m.total_chunks * CHUNK_F4;
    const int P = m.parts, tile = m.tile;
    if (tid < n_f4) {
        const int chunk = (int)(tid / CHUNK_F4), within = (int)(tid % CHUNK_F4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        int li = -1;
        for (int i = 0; i < m.n_mfma_layers; ++i)
            if (chunk >= m.layer[i].chunk0 && chunk < m.layer[i].chunk0 + m.layer[i].nchunks) li = i;
        if (li >= 0) {
            const LayerLayout &l = m.layer[li];
            const int lane = within & 63, blk = within >> 6;           // blk = gic * nob + ob
            const int gic = blk / l.nob, ob = blk % l.nob;
            const int g = (chunk - l.chunk0) * l.gpc + gic;
            if (gic < l.gpc && g < l.ngroups) {
                const int row = ob * tile + lane % tile, part = lane / tile;
                float t[4];
                for (int c = 0; c < 4; ++c) {
                    const int col = layer_src_col(l, P, 4 * g + c, part);
                    t[c] = (col >= 0 && row < l.n_out) ? l.w[(long)row * l.ld + col] : 0.f;
                }
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
        chunks[tid] = v;
        return;
    }
    pack_model_aux_thread(m, aux, tid - n_f4);
}

// element `a` of the aux block (biases in lane order, sigma / rgb head weights): shared by the fp32 and the split-precision image
__device__ __forceinline__ void pack_model_aux_thread(const ModelLayout &m, float *__restrict__ aux, long a) {
    const int P = m.parts;
    if (a < 0 || a >= m.aux_floats) return;
    float v = 0.f;
    // biases: [P][n_out/P] per layer, flat register i <-> feature 4P*(i/4) + 4*part + i%4
    for (int i = 0; i < m.n_mfma_layers; ++i) {
        const LayerLayout &l = m.layer[i];
        const long o = a - l.bias_off;
        if (o >= 0 && o < l.n_out) {
            const int regs = l.n_out / P, part = (int)(o / regs), r = (int)(o % regs);
            v = l.b[hid_src(P, r, part)];
        }
    }
    {
        const long o = a - m.sigma_off;
        const int H = m.sigma_in_regs;
        if (o >= 0 && o < P * H) v = m.sigma_w[hid_src(P, (int)(o % H), (int)(o / H))];
        else if (o == P * H) v = m.sigma_b[0];
    }
    {
        const long o = a - m.rgb_off;
        const int H = m.rgb_in_regs, per = P * H;
        if (o >= 0 && o < (long)m.rgb_dim * per) {
            const int c = (int)(o / per), q = (int)(o % per);
            v = m.rgb_w[(long)c * (P * H) + hid_src(P, q % H, q / H)];
        } else if (o >= (long)m.rgb_dim * per && o < (long)m.rgb_dim * per + m.rgb_dim) {
            v = m.rgb_b[o - (long)m.rgb_dim * per];
        }
    }
    aux[a] = v;
}


__device__ __forceinline__ void pack_bwd_thread(const BwdLayout &b, float4 *__restrict__ chunks, long tid) {
    if (tid >= (long)b.total_chunks * CHUNK_F4) return;
    const int chunk = (int)(tid / CHUNK_F4), within = (int)(tid % CHUNK_F4);
    const int P = b.parts, tile = b.tile;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int li = -1;
    for (int i = 0; i < b.n_layers; ++i)
        if (chunk >= b.layer[i].chunk0 && chunk < b.layer[i].chunk0 + b.layer[i].nchunks) li = i;
    if (li >= 0) {
        const BwdLayerLayout &l = b.layer[li];
        const int lane = within & 63, blk = within >> 6;
        const int gic = blk / l.nob, ob = blk % l.nob;
        const int g = (chunk - l.chunk0) * l.gpc + gic;
        if (gic < l.gpc && g < l.ngroups) {
            const int row = ob * tile + lane % tile, part = lane / tile;
            if (row < l.n_rows) {
                const int in_col = row < l.split ? l.in_off + row : l.in_off2 + (row - l.split);
                float t[4];
                for (int c = 0; c < 4; ++c) t[c] = l.w[(long)hid_src(P, 4 * g + c, part) * l.ld + in_col];
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
    chunks[tid] = v;
}


}  // namespace mnr
