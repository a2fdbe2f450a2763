// mlp_pack.hip -- nn.Module parameters -> packed MFMA-fragment image (see mlp_layout.h).
// One launch re-packs a whole model (device -> device); called after every optimiser step.
#include <string.h>

#include "common.h"
#include "mlp_layout.h"

namespace mnr {

static thread_local char g_err[512];
char *err_buf() { return g_err; }
int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int layout_from_desc(const mnr_model_desc *d, ModelLayout &m) {
    if (!d) return set_err(MNR_E_INVALID, "model desc is NULL");
    ArchDims a{d->xyz_dim, d->pos_xyz_dim, d->pos_dir_dim, d->layers, d->skip_mask, d->layer_dim, d->appearance_dim,
               d->rgb_dim, d->mfma_tile};
    if (d->xyz_dim != 3 && d->xyz_dim != 4) return set_err(MNR_E_UNSUPPORTED, "xyz_dim must be 3 or 4 (got %d)", d->xyz_dim);
    if (d->rgb_dim < 1 || d->rgb_dim > 75) return set_err(MNR_E_UNSUPPORTED, "rgb_dim out of range: %d", d->rgb_dim);
    const char *err = nullptr;
    if (build_layout(a, m, &err)) return set_err(MNR_E_UNSUPPORTED, "unsupported architecture: %s", err);
    int n = 0;
    for (int i = 0; i < d->layers; ++i, ++n) { m.layer[n].w = d->layer_w[i]; m.layer[n].b = d->layer_b[i]; }
    if (m.has_final) {
        m.layer[n].w = d->final_w; m.layer[n].b = d->final_b; ++n;
        m.layer[n].w = d->dir_a_w; m.layer[n].b = d->dir_a_b; ++n;
    }
    m.sigma_w = d->sigma_w; m.sigma_b = d->sigma_b; m.rgb_w = d->rgb_w; m.rgb_b = d->rgb_b;
    return MNR_OK;
}

// One thread per float4 of the chunk stream, then one thread per float of the aux image.
__global__ void k_pack_model(ModelLayout m, float4 *__restrict__ chunks, float *__restrict__ aux) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
    const long a = tid - n_f4;
    if (a >= m.aux_floats) return;
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

}  // namespace mnr

using namespace mnr;

extern "C" {

int mnr_version(void) { return MNR_VERSION; }
const char *mnr_last_error(void) { return err_buf(); }

int mnr_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n > 0 ? 1 : 0;
}

size_t mnr_packed_model_bytes(const mnr_model_desc *desc) {
    ModelLayout m;
    if (layout_from_desc(desc, m) != MNR_OK) return 0;
    return packed_bytes(m);
}

int mnr_pack_model(void *packed_dev, size_t bytes, const mnr_model_desc *desc, void *stream) {
    ModelLayout m;
    int rc = layout_from_desc(desc, m);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(packed_dev != nullptr, "packed_dev is NULL");
    MNR_REQUIRE(bytes >= packed_bytes(m), "packed buffer too small: %zu < %zu", bytes, packed_bytes(m));
    for (int i = 0; i < m.n_mfma_layers; ++i)
        MNR_REQUIRE(m.layer[i].w && m.layer[i].b, "missing weight/bias pointer for MFMA layer %d", i);
    MNR_REQUIRE(m.sigma_w && m.sigma_b && m.rgb_w && m.rgb_b, "missing sigma/rgb head pointers");
    float4 *chunks = reinterpret_cast<float4 *>(packed_dev);
    float *aux = reinterpret_cast<float *>(reinterpret_cast<char *>(packed_dev) + (size_t)m.total_chunks * CHUNK_BYTES);
    const long total = (long)m.total_chunks * CHUNK_F4 + m.aux_floats;
    const int bs = 256;
    hipLaunchKernelGGL(k_pack_model, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, as_stream(stream), m, chunks, aux);
    return check_launch("k_pack_model");
}

int mnr_layout_parts(const mnr_model_desc *desc) {
    ModelLayout m;
    if (layout_from_desc(desc, m) != MNR_OK) return -1;
    return m.parts;
}

static int mfma_layer_index(const ModelLayout &m, const mnr_model_desc *d, int layer) {
    if (layer < 0 || layer >= m.n_mfma_layers) return -1;
    (void)d;
    return layer;
}

int mnr_layout_num_steps(const mnr_model_desc *desc, int layer) {
    ModelLayout m;
    if (layout_from_desc(desc, m) != MNR_OK) return -1;
    int li = mfma_layer_index(m, desc, layer);
    if (li < 0) return set_err(MNR_E_INVALID, "layer %d out of range", layer);
    return m.layer[li].nsteps;
}

int mnr_layout_src_col(const mnr_model_desc *desc, int layer, int step, int part) {
    ModelLayout m;
    if (layout_from_desc(desc, m) != MNR_OK) return -2;
    int li = mfma_layer_index(m, desc, layer);
    if (li < 0 || step < 0 || step >= m.layer[li].nsteps || part < 0 || part >= m.parts) return -2;
    return layer_src_col(m.layer[li], m.parts, step, part);
}

}  // extern "C"
