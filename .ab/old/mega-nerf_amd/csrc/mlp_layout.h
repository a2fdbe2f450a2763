// mlp_layout.h -- the packed weight image shared by the pack kernel, the fused MLP kernels and
// the host-side layout queries (mnr_layout_*).  Everything here is integer index math.
//
// Design ("register-chained MLP"): one wavefront owns TILE samples (TILE = 32 with
// v_mfma_f32_32x32x2_f32, 16 with v_mfma_f32_16x16x4_f32) and ALL W features of them.  The MFMA is
// issued as  D[feature][sample] += Wgt[feature][k] * Act[k][sample]:  weights are the A operand
// (streamed global -> LDS -> VGPR), activations the B operand.  The C/D register layout of one layer
//   lane l: sample = l % TILE, part = l / TILE   (PARTS = 64 / TILE lane-parts per sample)
//   flat accumulator register i  <->  feature 4*PARTS*(i/4) + 4*part + (i%4)
// is, by construction of the K ordering below, exactly the B-operand layout of the next layer, so
// activations never leave the register file.  K "step" s of a layer = one B register per lane; an
// MFMA consumes step s of every part at once (K = PARTS per MFMA).  Steps are grouped by 4 (one
// float4 A-fragment load feeds 4 MFMAs).
//
// Packed image = sequence of fixed-size CHUNKs (32 KiB) in consumption order; a layer starts on a
// chunk boundary; chunk = up to GPC groups x NOB output blocks x 64 lanes x float4.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MNR_HD __host__ __device__ inline
#else
#define MNR_HD inline
#endif

namespace mnr {

constexpr int CHUNK_BYTES = 32768;
constexpr int CHUNK_F4 = CHUNK_BYTES / 16;
constexpr int MAX_MFMA_LAYERS = 18;   // trunk (<=16) + final + dir_a

MNR_HD constexpr int pad4(int x) { return (x + 3) & ~3; }
MNR_HD constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- positional-embedding segment (nerf.py:8-25) -------------------------------------------------
// part p evaluates frequencies [p*L/P, (p+1)*L/P) of every dimension: pair i = (fi, d) = (i / D, i % D),
// register 2i = sin, 2i+1 = cos; then ceil(D/P) raw-coordinate registers (dim = j*P + p); zero pad to 4.
MNR_HD constexpr int emb_cols(int D, int L) { return L > 0 ? D + 2 * D * L : 0; }
MNR_HD constexpr int emb_pairs(int D, int L, int P) { return D * (L / P); }
MNR_HD constexpr int emb_regs(int D, int L, int P) { return L > 0 ? pad4(2 * emb_pairs(D, L, P) + cdiv(D, P)) : 0; }
MNR_HD constexpr int emb_src(int D, int L, int P, int s, int p) {
    const int np = emb_pairs(D, L, P);
    if (s < 2 * np) {
        const int i = s >> 1, is_cos = s & 1;
        const int f = p * (L / P) + i / D, d = i % D;
        return D + f * 2 * D + (is_cos ? D : 0) + d;
    }
    const int j = s - 2 * np, dim = j * P + p;
    return (j < cdiv(D, P) && dim < D) ? dim : -1;
}
// ---- hidden segment: the previous layer's accumulator registers ---------------------------------
MNR_HD constexpr int hid_regs(int W, int P) { return W / P; }
MNR_HD constexpr int hid_src(int P, int s, int p) { return 4 * P * (s / 4) + 4 * p + (s % 4); }
// ---- appearance-embedding segment: part p holds a[p*A/P .. (p+1)*A/P) ----------------------------
MNR_HD constexpr int app_regs(int A, int P) { return A > 0 ? pad4(A / P) : 0; }
MNR_HD constexpr int app_src(int A, int P, int s, int p) { return s < A / P ? p * (A / P) + s : -1; }

enum SegType : int32_t { SEG_NONE = 0, SEG_EMB = 1, SEG_HID = 2, SEG_APP = 3 };
struct Seg {
    int32_t type, nsteps, col0, D, L;   // D,L for SEG_EMB; D = A for SEG_APP
};
struct LayerLayout {
    Seg seg[3];
    int32_t nseg, nsteps, ngroups;
    int32_t n_out, ld;          // rows / leading dimension of the nn.Linear weight
    int32_t nob, gpc;           // output blocks of TILE rows; groups per chunk
    int32_t chunk0, nchunks;    // position in the chunk stream
    int32_t bias_off;           // float offset in the aux image: [P][n_out_regs]
    const float *w, *b;
};
struct ModelLayout {
    int32_t tile, parts, W;
    int32_t n_mfma_layers;      // trunk + (final, dir_a if present)
    int32_t has_final;
    int32_t total_chunks;       // incl. one trailing dummy chunk (over-prefetch target)
    int32_t sigma_off, sigma_in_regs;           // aux: [P][H] weights, then 4 floats (bias, pad)
    int32_t rgb_off, rgb_in_regs, rgb_dim;      // aux: [rgb_dim][P][Hin] weights, then pad4(rgb_dim) biases
    int32_t aux_floats;
    LayerLayout layer[MAX_MFMA_LAYERS];
    const float *sigma_w, *sigma_b, *rgb_w, *rgb_b;
};

MNR_HD int layer_src_col(const LayerLayout &l, int P, int s, int p) {
    int s0 = 0;
    for (int i = 0; i < l.nseg; ++i) {
        const Seg &g = l.seg[i];
        if (s < s0 + g.nsteps) {
            const int r = s - s0;
            int c = -1;
            if (g.type == SEG_EMB) c = emb_src(g.D, g.L, P, r, p);
            else if (g.type == SEG_HID) c = hid_src(P, r, p);
            else if (g.type == SEG_APP) c = app_src(g.D, P, r, p);
            return c < 0 ? -1 : g.col0 + c;
        }
        s0 += g.nsteps;
    }
    return -1;
}

// Default MFMA tile.  16x16x4 (16 samples per wave) keeps W/4 input + W/4 accumulator registers per lane:
// two workgroups fit per CU for W <= 256 (better latency hiding, finer scheduling quantum; measured
// 133 vs 127 TFLOP/s, round 1) and W = 512 fits at one.  32x32x2 (mfma_tile = 32) stays selectable for W <= 256.
MNR_HD constexpr int tile_for_width(int W) { return (void)W, 16; }

struct ArchDims {   // the subset of mnr_model_desc the layout depends on
    int xyz_dim, pos_xyz_dim, pos_dir_dim, layers, skip_mask, W, app_dim, rgb_dim, tile;
};

// Returns 0 on success, else a static error string is stored in *err.
inline int build_layout(const ArchDims &a, ModelLayout &m, const char **err) {
    *err = nullptr;
    const int tile = a.tile ? a.tile : tile_for_width(a.W), P = 64 / tile;
    if (tile != 16 && tile != 32) { *err = "mfma_tile must be 0, 16 or 32"; return -1; }
    if (tile == 32 && a.W > 256) { *err = "mfma_tile 32 needs layer_dim <= 256 (register budget)"; return -1; }
    if (a.W % tile || a.W < tile || a.W > 512) { *err = "layer_dim must be a multiple of the MFMA tile and <= 512"; return -1; }
    if ((a.W / 2) % tile && (a.pos_dir_dim > 0 || a.app_dim > 0)) { *err = "layer_dim/2 must be a multiple of the MFMA tile"; return -1; }
    if (a.layers < 1 || a.layers > 16) { *err = "layers must be in 1..16"; return -1; }
    if (a.pos_xyz_dim % P || a.pos_dir_dim % P) { *err = "frequency counts must be multiples of the lane-part count"; return -1; }
    if (a.app_dim % P) { *err = "appearance_dim must be a multiple of the lane-part count"; return -1; }
    if (a.skip_mask & 1) { *err = "layer 0 cannot be a skip layer"; return -1; }
    m = ModelLayout{};
    m.tile = tile; m.parts = P; m.W = a.W;
    const int E = emb_regs(a.xyz_dim, a.pos_xyz_dim, P), Ecols = emb_cols(a.xyz_dim, a.pos_xyz_dim);
    const int H = hid_regs(a.W, P);
    if (E == 0) { *err = "pos_xyz_dim == 0 is not supported"; return -1; }
    int chunk = 0, aux = 0, n = 0;
    auto finish = [&](LayerLayout &l, int n_out) {
        l.nsteps = 0;
        for (int i = 0; i < l.nseg; ++i) l.nsteps += l.seg[i].nsteps;
        l.ngroups = l.nsteps / 4;
        l.n_out = n_out;
        l.nob = n_out / tile;
        l.gpc = CHUNK_F4 / (l.nob * 64);
        l.chunk0 = chunk;
        l.nchunks = cdiv(l.ngroups, l.gpc);
        chunk += l.nchunks;
        l.bias_off = aux;
        aux += P * (n_out / P);
    };
    for (int i = 0; i < a.layers; ++i) {
        LayerLayout &l = m.layer[n++];
        if (i == 0) {
            l.nseg = 1; l.seg[0] = Seg{SEG_EMB, E, 0, a.xyz_dim, a.pos_xyz_dim}; l.ld = Ecols;
        } else if ((a.skip_mask >> i) & 1) {
            l.nseg = 2; l.seg[0] = Seg{SEG_EMB, E, 0, a.xyz_dim, a.pos_xyz_dim};
            l.seg[1] = Seg{SEG_HID, H, Ecols, 0, 0}; l.ld = Ecols + a.W;
        } else {
            l.nseg = 1; l.seg[0] = Seg{SEG_HID, H, 0, 0, 0}; l.ld = a.W;
        }
        finish(l, a.W);
    }
    m.has_final = (a.pos_dir_dim > 0 || a.app_dim > 0) ? 1 : 0;
    if (m.has_final) {
        LayerLayout &f = m.layer[n++];
        f.nseg = 1; f.seg[0] = Seg{SEG_HID, H, 0, 0, 0}; f.ld = a.W;
        finish(f, a.W);
        LayerLayout &d = m.layer[n++];
        const int ED = emb_regs(3, a.pos_dir_dim, P), EDcols = emb_cols(3, a.pos_dir_dim), AP = app_regs(a.app_dim, P);
        d.nseg = 0;
        d.seg[d.nseg++] = Seg{SEG_HID, H, 0, 0, 0};
        if (ED) d.seg[d.nseg++] = Seg{SEG_EMB, ED, a.W, 3, a.pos_dir_dim};
        if (AP) d.seg[d.nseg++] = Seg{SEG_APP, AP, a.W + EDcols, a.app_dim, 0};
        d.ld = a.W + EDcols + a.app_dim;
        finish(d, a.W / 2);
    }
    m.n_mfma_layers = n;
    m.total_chunks = chunk + 1;
    m.sigma_in_regs = H;
    m.sigma_off = aux; aux += P * H + 4;
    m.rgb_dim = a.rgb_dim;
    m.rgb_in_regs = m.has_final ? (a.W / 2) / P : H;
    m.rgb_off = aux; aux += a.rgb_dim * P * m.rgb_in_regs + pad4(a.rgb_dim);
    m.aux_floats = pad4(aux);
    return 0;
}

inline size_t packed_bytes(const ModelLayout &m) {
    return (size_t)m.total_chunks * CHUNK_BYTES + (size_t)m.aux_floats * 4;
}

// ---- training tape: what the forward pass keeps for the backward pass ---------------------------
// Row-major planes [rows_cap][width]; plane p starts at float offset off_p * rows_cap.
//   act[l]  post-ReLU output of trunk layer l (width W)      -> ReLU masks + wgrad inputs
//   fin     xyz_encoding_final output (W), dact: dir_a post-ReLU (W/2)
//   embx / embd: positional encodings in REFERENCE column order (nerf.py:20-25), app: gathered embedding_a rows
//   mask[l] / dmask: the ReLU sign bits of act[l] / dact, one bit per feature packed per lane in C-layout register
//   order (word (part, w), bit b <-> flat register 32 w + b of that lane-part): what the data-gradient chain reads
//   instead of the fp32 activations (32x less traffic, 2 instead of 64 mask registers)
struct TapeLayout {
    int32_t act_off[16];
    int32_t fin_off, dact_off, embx_off, embd_off, app_off;
    int32_t embx_w, embd_w, app_w;
    int32_t mask_off[16], dmask_off;
    int32_t mask_w, dmask_w;                 // 32-bit words per row
    int32_t floats_per_row;
};
inline TapeLayout tape_layout(const ArchDims &a) {
    TapeLayout t{};
    int off = 0;
    for (int l = 0; l < a.layers; ++l) { t.act_off[l] = off; off += a.W; }
    const bool has_final = a.pos_dir_dim > 0 || a.app_dim > 0;
    t.fin_off = off; off += has_final ? a.W : 0;
    t.dact_off = off; off += has_final ? a.W / 2 : 0;
    t.embx_w = pad4(emb_cols(a.xyz_dim, a.pos_xyz_dim));
    t.embx_off = off; off += t.embx_w;
    t.embd_w = pad4(emb_cols(3, a.pos_dir_dim));
    t.embd_off = off; off += t.embd_w;
    t.app_w = pad4(a.app_dim);
    t.app_off = off; off += t.app_w;
    const int tile = a.tile ? a.tile : tile_for_width(a.W), P = 64 / tile;
    t.mask_w = pad4(P * ((a.W / P + 31) / 32));
    t.dmask_w = pad4(P * ((a.W / 2 / P + 31) / 32));
    for (int l = 0; l < a.layers; ++l) { t.mask_off[l] = off; off += t.mask_w; }
    t.dmask_off = off; off += has_final ? t.dmask_w : 0;
    t.floats_per_row = off;
    return t;
}

// ---- backward (data-gradient) weight stream: transposed layers in reverse order -------------------
//   bwd layer 0: dir_a^T   rows = [final features (W) | appearance inputs (A)], K = dir_a outputs (W/2)
//   bwd layer 1: final^T   rows = W, K = W
//   bwd layer 2+j: trunk layer (L-1-j)^T for j = 0 .. L-2 (hidden-input columns only), rows = W, K = W
// Packed A operand element (row r, step s, part p) = weight[out = hid_src(P, s, p)][in = in_col(r)].
struct BwdLayerLayout {
    int32_t n_rows, n_rows_pad, nsteps, ngroups, nob, gpc, chunk0, nchunks;
    int32_t ld, in_off, in_off2, split;     // in_col(r) = r < split ? in_off + r : in_off2 + (r - split)
    const float *w;
};
struct BwdLayout {
    int32_t tile, parts, W, n_layers, has_final, total_chunks;
    int32_t app_rows;                        // appearance-gradient rows appended to dir_a^T (0 if none)
    BwdLayerLayout layer[MAX_MFMA_LAYERS];
};
inline int build_bwd_layout(const ArchDims &a, BwdLayout &b, const char **err) {
    ModelLayout m;
    if (build_layout(a, m, err)) return -1;
    b = BwdLayout{};
    const int tile = m.tile, P = m.parts;
    b.tile = tile; b.parts = P; b.W = a.W; b.has_final = m.has_final;
    const int Ecols = emb_cols(a.xyz_dim, a.pos_xyz_dim), EDcols = emb_cols(3, a.pos_dir_dim);
    int chunk = 0, n = 0;
    auto finish = [&](BwdLayerLayout &l, int n_rows, int k_feats) {
        l.n_rows = n_rows;
        l.n_rows_pad = cdiv(n_rows, 4 * tile) * 4 * tile;   // whole batches of 4 output blocks (run_segment)
        l.nob = l.n_rows_pad / tile;
        l.nsteps = k_feats / P;
        l.ngroups = l.nsteps / 4;
        l.gpc = CHUNK_F4 / (l.nob * 64);
        if (l.gpc < 1) l.gpc = 1;
        l.chunk0 = chunk;
        l.nchunks = cdiv(l.ngroups, l.gpc);
        chunk += l.nchunks;
    };
    if (m.has_final) {
        BwdLayerLayout &d = b.layer[n++];
        b.app_rows = a.app_dim;
        d.ld = a.W + EDcols + a.app_dim; d.in_off = 0; d.split = a.W; d.in_off2 = a.W + EDcols;
        finish(d, a.W + a.app_dim, a.W / 2);
        BwdLayerLayout &f = b.layer[n++];
        f.ld = a.W; f.in_off = 0; f.split = a.W; f.in_off2 = 0;
        finish(f, a.W, a.W);
    }
    for (int l = a.layers - 1; l >= 1; --l) {
        BwdLayerLayout &t = b.layer[n++];
        const bool skip = (a.skip_mask >> l) & 1;
        t.ld = skip ? Ecols + a.W : a.W; t.in_off = skip ? Ecols : 0; t.split = a.W; t.in_off2 = 0;
        finish(t, a.W, a.W);
    }
    b.n_layers = n;
    b.total_chunks = chunk + 1;
    for (int i = 0; i < n; ++i)
        if ((long)b.layer[i].nob * 64 * 16 > CHUNK_BYTES) { *err = "backward layer too wide for one chunk group"; return -1; }
    return 0;
}
inline size_t packed_bwd_bytes(const BwdLayout &b) { return (size_t)b.total_chunks * CHUNK_BYTES; }

}  // namespace mnr
