// mlp_pack.hip -- nn.Module parameters -> packed MFMA-fragment image (see mlp_layout.h).
// One launch re-packs a whole model (device -> device); called after every optimiser step.
#include <string.h>

#include "common.h"
#include "mlp_layout.h"
#include "pack_device.h"

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

__global__ void k_pack_model(ModelLayout m, float4 *__restrict__ chunks, float *__restrict__ aux) {
    pack_model_thread(m, chunks, aux, (long)blockIdx.x * blockDim.x + threadIdx.x);
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
