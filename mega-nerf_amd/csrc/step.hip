// step.hip -- one whole training step per call (mnr_train_step): runner.py:244-277 over rendering.py:15-173 for one or
// several independent submodules ("cells"), as a fixed sequence of 12 kernel launches + one memset on one stream (+ 2 per further cell).
//
//   memset            gradients, background-ray counts, error flags, loss, weight-gradient queue heads (one region)
//   k_step_begin      one 1024-thread workgroup per cell: batch -> workspace, _intersect_sphere + near/far (rendering.py:33-45,
//                     396-417), stable compaction of the rays with a background segment (+ their rays / image indices)
//   k_step_samples    coarse samples of both branches (rendering.py:47-56, 82-87, 420-483; the background ones stored in the
//                     flipped order the MLP sees, quirk Q1/Q2), every random number of the step (counter-based Philox)
//   MLP coarse        fg + bg rows of ALL cells: k_mlp_fwd_multi<fg, bg, true>, grid = (workgroups per cell, cells)
//   k_step_mid        one wavefront per ray: coarse compositing weights -> _sample_pdf -> fine points (rendering.py:212-225)
//   MLP fine
//   k_step_tail       one wavefront per ray: coarse/fine merge, compositing of both branches, fg/bg blend, MSE, and the adjoints
//                     of all of these down to dL/d(raw MLP outputs) (rendering.py:102-131, 336-393; runner.py:370)
//   k_mlp_bwd_multi   data-gradient chains of all four (branch, pass) segments of all cells
//   k_head_grads_jobs sigma / rgb head gradients of every (cell, branch, pass)
//   k_wgrad2 (+ reduce) per cell
//   k_step_adam       torch.optim.Adam's update of every parameter of every cell (independent optimisers: parscripts/run_8.txt)
//   k_step_pack       every forward / transposed weight image
//
// The ray-parallel kernels restate csrc/render.hip's stage kernels (same operations in the same order, so the step's
// gradients equal the stage-by-stage path's: tests/test_gpu_step.py) with the intermediate arrays of a ray kept in LDS /
// registers.  Compiled with -ffp-contract=off like render.hip (bit-exact sample positions).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "mlp_layout.h"
#include "h2_device.h"
#include "pack_device.h"
#include "step_internal.h"

namespace mnr {

static constexpr int WPB = 4;                 // wavefronts per block in the ray-parallel kernels
static constexpr int MAXC = MNR_STEP_MAX_CELLS;

struct SSphere { float cx, cy, cz, rx, ry, rz; };

__device__ __forceinline__ void s_norm_ray(const SSphere &sp, const float *ray, float (&o)[3], float (&d)[3]) {
    o[0] = (ray[0] - sp.cx) / sp.rx; o[1] = (ray[1] - sp.cy) / sp.ry; o[2] = (ray[2] - sp.cz) / sp.rz;
    d[0] = ray[3] / sp.rx; d[1] = ray[4] / sp.ry; d[2] = ray[5] / sp.rz;
}
__device__ __forceinline__ float s_dot3(const float (&a)[3], const float (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// rendering.py:472-483 (as render.hip::perturb_z)
__device__ __forceinline__ float s_perturb_z(float zc, float zl, float zr, bool first, bool last, float perturb, float rnd) {
    const float upper = last ? zc : 0.5f * (zc + zr);
    const float lower = first ? zc : 0.5f * (zl + zc);
    return lower + (upper - lower) * (perturb * rnd);
}

// _depth2pts_outside (rendering.py:420-469) for one sample: q[0..4) = (point on the unit sphere, inverse depth)
__device__ __forceinline__ void s_bg_point(const SSphere &sp, const float *ray, float depth, float *q, float &depth_real) {
    float o[3], d[3];
    s_norm_ray(sp, ray, o, d);
    const float dd = s_dot3(d, d);
    const float d1 = -s_dot3(d, o) / dd;
    const float pm[3] = {o[0] + d1 * d[0], o[1] + d1 * d[1], o[2] + d1 * d[2]};
    const float pm_norm = sqrtf(s_dot3(pm, pm));
    const float ray_d_cos = 1.f / sqrtf(dd);
    const float d2 = sqrtf(1.f - pm_norm * pm_norm) * ray_d_cos;
    const float dsum = d1 + d2;
    const float ps[3] = {o[0] + dsum * d[0], o[1] + dsum * d[1], o[2] + dsum * d[2]};
    float ax[3] = {o[1] * ps[2] - o[2] * ps[1], o[2] * ps[0] - o[0] * ps[2], o[0] * ps[1] - o[1] * ps[0]};
    const float an = sqrtf(s_dot3(ax, ax)) + 1e-8f;
    ax[0] /= an; ax[1] /= an; ax[2] /= an;
    const float phi = asinf(pm_norm);
    const float theta = asinf(pm_norm * depth);
    const float ang = phi - theta;
    const float ca = cosf(ang), sa = sinf(ang);
    const float cr[3] = {ax[1] * ps[2] - ax[2] * ps[1], ax[2] * ps[0] - ax[0] * ps[2], ax[0] * ps[1] - ax[1] * ps[0]};
    const float adp = s_dot3(ax, ps);
    float pn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pn[c] = ps[c] * ca + cr[c] * sa + ax[c] * adp * (1.f - ca);
    const float nn = sqrtf(s_dot3(pn, pn));
    depth_real = 1.f / (depth + 1e-8f) * cosf(theta) + d1;
    q[0] = pn[0] / nn; q[1] = pn[1] / nn; q[2] = pn[2] / nn; q[3] = depth;
}

// ---- counter-based random numbers (Philox4x32-10) ----------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }     // [0, 1), 24 bits

// ---- workspace layout ------------------------------------------------------------------------------------------------------
struct StepWs {
    size_t zero_begin, grads, scal, loss, wcount, zero_end;
    size_t rays, idx, target, far, last_delta, bg_slot, bg_list, rays_bg, idx_bg;
    size_t z_c, xyz_c, z_f, xyz_f, raw_c, raw_f, draw_c, draw_f;
    size_t zb_asc, zb_c, pts_c, dr_c, zb_f, pts_f, dr_f, braw_c, braw_f, bdraw_c, bdraw_f;
    size_t noise_fc, noise_ff, noise_bc, noise_bf, u_f, u_b;
    size_t rgb, depth_var, bg_lambda;
    size_t tape_f, gtape_f, dheads_f, tape_b, gtape_b, dheads_b;
    size_t ep_job, slab;
    size_t tab_cells, tab_pack, tab_adam, t_c, t_bc, t_f, t_bf;
    size_t dd_fc, dd_ff, dd_bc, dd_bf;        // spherical-harmonics models: dL/d(dir_a output) of the four (branch, pass) row sets [rows][W/2]
    size_t sticky;                            // int32 [MAXC]: health bits that survive the per-step memset (cleared by mnr_step_create)
    // 512-wide foreground (Building, README "Larger models"): buffers of its tiled backward, one (cell, pass) at a time -- zero-padded
    // [position embedding] / [direction | appearance] input planes, head / layer gradients, the zero-padded copies of the two weight
    // matrices whose hidden-input columns do not start 16-byte aligned (skip layer, dir_a layer), the weight-gradient job workspace
    size_t w_emb, w_side, w_grgb, w_dsrc, w_dapp, w_gsig, w_df, w_dh[8], w_wskip, w_wdir, w_wgws;
    size_t grad_stride, total;
};
struct StepDims {
    long C, N, Nc, Nf, Sb, Sfb, cap_f, cap_b, fpr_f, fpr_b;
    int sh_deg;        // >= 0: spherical-harmonics colour head (rgb_dim = 3 (sh_deg + 1)^2, no direction encoding); -1: the plain rgb head
    int wide;          // foreground 8 x 512 (forward: the wavefront-pair kernel; backward: tiled GEMMs + weight-gradient jobs, csrc/tgemm.hip / wgrad.hip)
};
constexpr int WIDE_EP = 96, WIDE_SP = 96;       // pitches of the zero-padded [75 embedding] and [27 direction + 48 appearance] input planes

static int step_dims(const mnr_step_cfg *cfg, const mnr_model_desc *fg, const mnr_model_desc *bg, StepDims &D) {
    MNR_REQUIRE(cfg && fg && bg, "NULL argument");
    MNR_REQUIRE(cfg->n_cells >= 1 && cfg->n_cells <= MAXC, "n_cells must be 1..%d", MAXC);
    MNR_REQUIRE(cfg->n_rays >= 1 && cfg->coarse_samples >= 4 && cfg->fine_samples >= 2 && cfg->coarse_samples % 2 == 0 &&
                cfg->fine_samples % 2 == 0, "bad ray / sample counts");
    D.C = cfg->n_cells; D.N = cfg->n_rays; D.Nc = cfg->coarse_samples; D.Nf = cfg->fine_samples; D.Sb = D.Nc / 2; D.Sfb = D.Nf / 2;
    if ((D.N * D.Sb) % 64 || (D.N * D.Sfb) % 64)
        return set_err(MNR_E_UNSUPPORTED, "the fused step needs n_rays * samples / 2 to be a multiple of 64 (one workgroup tile)");
    const bool shapes = (D.Nc == 64 && D.Nf == 128) || (D.Nc == 256 && D.Nf == 512);
    if (!shapes) return set_err(MNR_E_UNSUPPORTED, "the fused step is instantiated for 64 + 128 and 256 + 512 samples per ray");
    if (cfg->split_precision && ((D.N * D.Sb) % 128 || (D.N * D.Sfb) % 128))
        return set_err(MNR_E_UNSUPPORTED, "the split-precision step needs n_rays * samples / 2 to be a multiple of 128 (one workgroup tile)");
    D.cap_f = D.N * (D.Nc + D.Nf); D.cap_b = D.N * (D.Sb + D.Sfb);
    D.fpr_f = mnr_tape_floats_per_row(fg); D.fpr_b = mnr_tape_floats_per_row(bg);
    if (D.fpr_f <= 0 || D.fpr_b <= 0) return set_err(MNR_E_UNSUPPORTED, "no training kernels for this architecture");
    D.wide = fg->layer_dim == 512 ? 1 : 0;
    const bool trunk = fg->xyz_dim == 3 && bg->xyz_dim == 4 && fg->pos_xyz_dim == 12 && bg->pos_xyz_dim == 12 && fg->appearance_dim == 48 &&
                       bg->appearance_dim == 48 && (fg->layer_dim == 256 || fg->layer_dim == 512) && bg->layer_dim == 256 && fg->layers == 8 && bg->layers == 8 &&
                       fg->skip_mask == 16 && bg->skip_mask == 16 && (fg->mfma_tile == 0 || fg->mfma_tile == 16) &&
                       (bg->mfma_tile == 0 || bg->mfma_tile == 16);
    const bool plain = fg->pos_dir_dim == 4 && bg->pos_dir_dim == 4 && fg->rgb_dim == 3 && bg->rgb_dim == 3;
    // configs/mega-nerf-sh-3/*.yaml: sh_deg 2, pos_dir_dim 0 -> 27 colour coefficients, dir_a_encoding over [features | appearance]
    // (48 coefficients: sh_deg 3, the degree BASELINE.json's configs[4] words)
    const bool sh = fg->pos_dir_dim == 0 && bg->pos_dir_dim == 0 && fg->rgb_dim == bg->rgb_dim && (fg->rgb_dim == 27 || fg->rgb_dim == 48);
    if (!trunk || !(plain || sh))
        return set_err(MNR_E_UNSUPPORTED, "the fused step covers the default 8x256 foreground / background models and their sh_deg 2 / 3 forms");
    if (sh && cfg->split_precision) return set_err(MNR_E_UNSUPPORTED, "no split-precision kernels for the spherical-harmonics colour head");
    if (D.wide && (sh || cfg->split_precision)) return set_err(MNR_E_UNSUPPORTED, "the 512-wide foreground runs on the fp32 kernels with the plain rgb head");
    D.sh_deg = !sh ? -1 : (fg->rgb_dim == 27 ? 2 : 3);
    return MNR_OK;
}

static void step_layout(const mnr_step_cfg *cfg, const StepDims &D, StepWs &L) {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const long C = D.C, N = D.N, CN = C * N;
    L.zero_begin = off;
    L.grad_stride = ((size_t)cfg->grad_floats_per_cell * 4 + 255) / 256 * 256;
    L.grads = take(L.grad_stride * C);
    L.scal = take(2 * MAXC * 4);             // n_bg[MAXC], then err[MAXC]
    L.loss = take(C * 4);
    L.wcount = take(C * 256);
    L.zero_end = off;
    L.rays = take(CN * 32); L.idx = take(CN * 4); L.target = take(CN * 12);
    L.far = take(CN * 4); L.last_delta = take(CN * 4); L.bg_slot = take(CN * 4); L.bg_list = take(CN * 4);
    L.rays_bg = take(CN * 32); L.idx_bg = take(CN * 4);
    L.z_c = take(CN * D.Nc * 4); L.xyz_c = take(CN * D.Nc * 12); L.z_f = take(CN * D.Nf * 4); L.xyz_f = take(CN * D.Nf * 12);
    L.raw_c = take(CN * D.Nc * 16); L.raw_f = take(CN * D.Nf * 16); L.draw_c = take(CN * D.Nc * 16); L.draw_f = take(CN * D.Nf * 16);
    L.zb_asc = take(CN * D.Sb * 4); L.zb_c = take(CN * D.Sb * 4); L.pts_c = take(CN * D.Sb * 16); L.dr_c = take(CN * D.Sb * 4);
    L.zb_f = take(CN * D.Sfb * 4); L.pts_f = take(CN * D.Sfb * 16); L.dr_f = take(CN * D.Sfb * 4);
    L.braw_c = take(CN * D.Sb * 16); L.braw_f = take(CN * D.Sfb * 16); L.bdraw_c = take(CN * D.Sb * 16); L.bdraw_f = take(CN * D.Sfb * 16);
    L.noise_fc = take(CN * D.Nc * 4); L.noise_ff = take(CN * D.Nf * 4); L.noise_bc = take(CN * D.Sb * 4); L.noise_bf = take(CN * D.Sfb * 4);
    L.u_f = take(CN * D.Nf * 4); L.u_b = take(CN * D.Sfb * 4);
    L.rgb = take(CN * 12); L.depth_var = take(CN * 4); L.bg_lambda = take(CN * 4);
    L.tape_f = take((size_t)D.fpr_f * C * D.cap_f * 4); L.gtape_f = take((size_t)D.fpr_f * C * D.cap_f * 4); L.dheads_f = take((size_t)C * D.cap_f * 16);
    L.tape_b = take((size_t)D.fpr_b * C * D.cap_b * 4); L.gtape_b = take((size_t)D.fpr_b * C * D.cap_b * 4); L.dheads_b = take((size_t)C * D.cap_b * 16);
    L.ep_job = take(C * wgrad_ep_job_bytes()); L.slab = take(wgrad_slab_bytes());
    L.tab_cells = take(4 * C * sizeof(MlpCellSeg));
    L.tab_pack = take(0);       // sized below (depends on the plan's tables); placeholder keeps the order explicit
    L.tab_adam = take(0);
    L.t_c = take(D.Nc * 4); L.t_bc = take(D.Sb * 4); L.t_f = take(D.Nf * 4); L.t_bf = take(D.Sfb * 4);
    L.sticky = take(MAXC * 4);
    L.dd_fc = L.dd_ff = L.dd_bc = L.dd_bf = 0;
    if (D.sh_deg >= 0) {
        L.dd_fc = take(CN * D.Nc * 128 * 4); L.dd_ff = take(CN * D.Nf * 128 * 4); L.dd_bc = take(CN * D.Sb * 128 * 4); L.dd_bf = take(CN * D.Sfb * 128 * 4);
    }
    L.w_emb = L.w_side = L.w_grgb = L.w_dsrc = L.w_dapp = L.w_gsig = L.w_df = L.w_wskip = L.w_wdir = L.w_wgws = 0;
    for (size_t &o : L.w_dh) o = 0;
    if (D.wide) {
        const size_t B = (size_t)N * (D.Nc > D.Nf ? D.Nc : D.Nf);
        L.w_emb = take(B * WIDE_EP * 4); L.w_side = take(B * WIDE_SP * 4); L.w_grgb = take(B * 4 * 4); L.w_dsrc = take(B * 256 * 4);
        L.w_dapp = take(B * 48 * 4); L.w_gsig = take(B * 4); L.w_df = take(B * 512 * 4);
        for (size_t &o : L.w_dh) o = take(B * 512 * 4);
        L.w_wskip = take((size_t)C * 512 * (WIDE_EP + 512) * 4); L.w_wdir = take((size_t)C * 256 * (512 + WIDE_SP) * 4);
        L.w_wgws = take(mnr_wgrad_workspace_bytes());
    }
    L.total = off;
}

// ---- k_step_begin ------------------------------------------------------------------------------------------------------------
struct BeginArgs { mnr_step_batch b[MAXC]; };

__global__ __launch_bounds__(1024) void k_step_begin(BeginArgs ba, long N, SSphere sp, float *__restrict__ rays_o, uint32_t *__restrict__ idx_o,
                                                     float *__restrict__ target_o, float *__restrict__ far_o, float *__restrict__ last_delta_o,
                                                     int32_t *__restrict__ slot_o, int32_t *__restrict__ list_o, float *__restrict__ rays_bg_o,
                                                     uint32_t *__restrict__ idx_bg_o, int32_t *__restrict__ n_bg_o, int32_t *__restrict__ err_o) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int cell = blockIdx.x;
    const mnr_step_batch &b = ba.b[cell];
    const long base = (long)cell * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { base_s = 0; err_o[cell] = 0; }          // (the cell's error flag belongs to this block: cleared here, no memset launch in front)
    __syncthreads();
    for (long start = 0; start < N; start += 1024) {
        const long i = start + threadIdx.x;
        int f = 0;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
        uint32_t ix = 0;
        if (i < N) {
            // the batch itself, or rows select[i] of a device-resident training set (mnr_step_batch::select: the gathers of
            // memory_dataset.py:47-53 / a DataLoader's collation folded into this copy)
            const long src = b.select ? (long)b.select[i] : i;
            r0 = reinterpret_cast<const float4 *>(b.rays)[2 * src];
            r1 = reinterpret_cast<const float4 *>(b.rays)[2 * src + 1];
            ix = reinterpret_cast<const uint32_t *>(b.idx)[src];
            reinterpret_cast<float4 *>(rays_o)[2 * (base + i)] = r0;
            reinterpret_cast<float4 *>(rays_o)[2 * (base + i) + 1] = r1;
            idx_o[base + i] = ix;
            if (b.target_u8) {
                // uint8 colours -> fp32 through the caller's 256-entry table of the CPU's i / 255. values (dataset_utils.py:30)
                target_o[3 * (base + i)] = b.u8_table[b.target_u8[3 * src]]; target_o[3 * (base + i) + 1] = b.u8_table[b.target_u8[3 * src + 1]];
                target_o[3 * (base + i) + 2] = b.u8_table[b.target_u8[3 * src + 2]];
            } else if (b.target) {
                target_o[3 * (base + i)] = b.target[3 * src]; target_o[3 * (base + i) + 1] = b.target[3 * src + 1];
                target_o[3 * (base + i) + 2] = b.target[3 * src + 2];
            }
            // rendering.py:33-45, 396-417 (as render.hip::k_ray_setup)
            const float ray[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            float o[3], d[3];
            s_norm_ray(sp, ray, o, d);
            const float dd = s_dot3(d, d);
            const float d1 = -s_dot3(d, o) / dd;
            const float p[3] = {o[0] + d1 * d[0], o[1] + d1 * d[1], o[2] + d1 * d[2]};
            const float ray_d_cos = 1.f / sqrtf(dd);
            const float pn = s_dot3(p, p);
            if (pn >= 1.f) atomicOr(err_o + cell, 1);
            const float d2 = sqrtf(1.f - pn) * ray_d_cos;
            const float near = ray[6], far = ray[7];
            const float fg_far = fmaxf(d1 + d2, near);
            f = far > fg_far;
            far_o[base + i] = fminf(far, fg_far);
            last_delta_o[base + i] = f ? fg_far : 1e10f;
        }
        // stable compaction (ascending ray order, like the boolean-mask indexing of rendering.py:37)
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (i < N) {
            const int k = f ? off + before : -1;
            slot_o[base + i] = k;
            if (f) {
                list_o[base + k] = (int32_t)i;
                reinterpret_cast<float4 *>(rays_bg_o)[2 * (base + k)] = r0;
                reinterpret_cast<float4 *>(rays_bg_o)[2 * (base + k) + 1] = r1;
                idx_bg_o[base + k] = ix;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_bg_o[cell] = base_s;
}

// ---- k_step_samples ----------------------------------------------------------------------------------------------------------
struct SamplesArgs {
    mnr_step_randoms inj[MAXC];
    long C, N;
    int Nc, Nf, Sb, Sfb;
    float perturb;
    int noise, has_inj;
    unsigned seed_lo, seed_hi, step_lo, step_hi;
    unsigned long long cell_key[MAXC];       // added to the seed: the cell's position in the plan, or mnr_step_batch::rng_cell_plus1 - 1
    SSphere sp;
    const float *rays, *far, *rays_bg, *t_c, *t_bc;
    const int32_t *scal;
    float *z_c, *xyz_c, *zb_asc, *zb_c, *pts_c, *dr_c;
    float *noise_fc, *noise_ff, *noise_bc, *noise_bf, *u_f, *u_b;
};

__global__ __launch_bounds__(256) void k_step_samples(SamplesArgs a) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long CN = a.C * a.N;
    if (t >= CN * (a.Nc + a.Nf)) return;
    // element t of a random stream laid out [cell][per_cell]: the injected value if the caller supplied that stream for the cell
    // (parity tests), else Philox with key (seed + cell) and counter (element inside the cell, stream, step) -- a cell draws the
    // same numbers whether it shares the launch with other cells or runs alone with seed + cell.  `unit` > 0: the stream belongs
    // to the compacted background rays -- rows past the cell's count do not exist in an injected array (and are never read)
    auto pick = [&](const float *mnr_step_randoms::*member, long per_cell, unsigned stream, int unit) -> float {
        const long cell = t / per_cell, local = t - cell * per_cell;
        const float *p = a.has_inj ? a.inj[cell].*member : nullptr;
        if (!p) {
            const unsigned long long sd = (((unsigned long long)a.seed_hi << 32) | a.seed_lo) + a.cell_key[cell];
            return u01(philox4x32(make_uint4((unsigned)local, (unsigned)((unsigned long long)local >> 32), stream, a.step_lo),
                                  make_uint2((unsigned)sd, (unsigned)(sd >> 32) ^ a.step_hi)).x);
        }
        if (unit > 0 && local / unit >= (long)a.scal[cell]) return 0.f;
        return p[local];
    };
    // sigma noise of the MLP rows (rendering.py:294, 321) -- row order = the MLP's (cell-major arrays, one per pass)
    if (a.noise) {
        if (t < CN * a.Nc) a.noise_fc[t] = pick(&mnr_step_randoms::fg_noise_coarse, a.N * a.Nc, 0u, 0);
        if (t < CN * a.Nf) a.noise_ff[t] = pick(&mnr_step_randoms::fg_noise_fine, a.N * a.Nf, 1u, 0);
        if (t < CN * a.Sb) a.noise_bc[t] = pick(&mnr_step_randoms::bg_noise_coarse, a.N * a.Sb, 2u, a.Sb);
        if (t < CN * a.Sfb) a.noise_bf[t] = pick(&mnr_step_randoms::bg_noise_fine, a.N * a.Sfb, 3u, a.Sfb);
    }
    if (a.perturb > 0.f) {
        if (t < CN * a.Nf) a.u_f[t] = pick(&mnr_step_randoms::fg_u, a.N * a.Nf, 4u, 0);
        if (t < CN * a.Sfb) a.u_b[t] = pick(&mnr_step_randoms::bg_u, a.N * a.Sfb, 5u, a.Sfb);
    }
    // foreground coarse sample (rendering.py:82-87; as render.hip::k_fg_samples)
    if (t < CN * a.Nc) {
        const long r = t / a.Nc;
        const int s = (int)(t - r * a.Nc), S = a.Nc;
        const float *ray = a.rays + r * 8;
        const float near = ray[6], far = a.far[r];
        const float *tt = a.t_c;
        float z = near * (1.f - tt[s]) + far * tt[s];
        if (a.perturb > 0.f) {
            const float zl = s > 0 ? near * (1.f - tt[s - 1]) + far * tt[s - 1] : z;
            const float zr = s < S - 1 ? near * (1.f - tt[s + 1]) + far * tt[s + 1] : z;
            z = s_perturb_z(z, zl, zr, s == 0, s == S - 1, a.perturb, pick(&mnr_step_randoms::fg_perturb, a.N * a.Nc, 6u, 0));
        }
        a.z_c[t] = z;
        a.xyz_c[3 * t + 0] = ray[0] + ray[3] * z;
        a.xyz_c[3 * t + 1] = ray[1] + ray[4] * z;
        a.xyz_c[3 * t + 2] = ray[2] + ray[5] * z;
    }
    // background coarse sample (rendering.py:47-56; as render.hip::k_bg_samples), stored ascending for the sampler's bins and in
    // the flipped order of rendering.py:271-273 for the MLP / compositing (depth_real is NOT flipped: quirk Q2)
    if (t < CN * a.Sb) {
        const long g = t / a.Sb;                        // compacted background unit: cell * N + k
        const long cell = g / a.N, k = g - cell * a.N;
        if (k < (long)a.scal[cell]) {
            const int s = (int)(t - g * a.Sb), S = a.Sb;
            const float *tt = a.t_bc;
            float depth = tt[s];
            if (a.perturb > 0.f)
                depth = s_perturb_z(depth, s > 0 ? tt[s - 1] : depth, s < S - 1 ? tt[s + 1] : depth, s == 0, s == S - 1, a.perturb,
                                    pick(&mnr_step_randoms::bg_perturb, a.N * a.Sb, 7u, a.Sb));
            float q[4], dr;
            s_bg_point(a.sp, a.rays_bg + g * 8, depth, q, dr);
            a.zb_asc[t] = depth;
            a.dr_c[t] = dr;
            const long tf = g * a.Sb + (S - 1 - s);
            a.zb_c[tf] = depth;
            *reinterpret_cast<float4 *>(a.pts_c + 4 * tf) = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
}

// ---- wave-level pieces of the ray kernels (restated from render.hip) -------------------------------------------------------
template <class Tv>
__device__ __forceinline__ Tv s_wave_sum(Tv v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void s_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): LDS writes visible to the wave
}
__device__ __forceinline__ float s_wave_max(const float *p, int n, int lane) {
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, p[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}

// per-lane state of one composited ray (lane owns the contiguous samples k = lane * E + e); forward as k_composite, kept for
// the adjoint (k_composite_bwd)
template <int E>
struct Comp {
    float alpha[E], ex[E], delta[E], tt[E], T[E], w[E], z[E];
    float4 c[E];
    float lambda;
};

// z(k) from an LDS array, raw(k) through a loader; computes weights (and lambda); render.hip::k_composite / k_composite_bwd
template <int E, class LoadRaw>
__device__ __forceinline__ void comp_forward(Comp<E> &st, const float *zl, int S, int lane, float last, int flip, LoadRaw load_raw) {
    double prod = 1.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        st.alpha[e] = 0.f; st.ex[e] = 1.f; st.delta[e] = 0.f; st.tt[e] = 1.f; st.z[e] = 0.f; st.c[e] = make_float4(0, 0, 0, 0);
        if (k < S) {
            const float zk = zl[k];
            st.z[e] = zk;
            st.c[e] = load_raw(k);
            st.delta[e] = (k == S - 1) ? last : (flip ? zk - zl[k + 1] : zl[k + 1] - zk);
            st.ex[e] = expf(-st.delta[e] * st.c[e].w);
            st.alpha[e] = 1.f - st.ex[e];
            st.tt[e] = 1.f - st.alpha[e] + 1e-8f;
            prod *= (double)st.tt[e];
        }
    }
    double incl = prod;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o);
        if (lane >= o) incl *= up;
    }
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0;
    st.lambda = (float)__shfl(incl, 63);
    double run = excl;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        st.T[e] = (float)run; st.w[e] = 0.f;
        if (k < S) {
            st.w[e] = st.alpha[e] * st.T[e];
            run *= (double)st.tt[e];
        }
    }
}

// dL/d(raw) of one ray from dL/d(rgb) (gr, gg, gb) and dL/d(lambda); store(e, k, float4) receives the gradient of the lane's e-th sample k
template <int E, class Store>
__device__ __forceinline__ void comp_backward(const Comp<E> &st, int S, int lane, float gr, float gg, float gb, float dlam, Store store) {
    float gk[E];
    float gw_lane = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        gk[e] = 0.f;
        if (k < S) {
            gk[e] = gr * st.c[e].x + gg * st.c[e].y + gb * st.c[e].z;
            gw_lane += gk[e] * st.w[e];
        }
    }
    float suf = gw_lane;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float dn = __shfl_down(suf, o);
        if (lane + o < 64) suf += dn;
    }
    float after = suf - gw_lane;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const int k = lane * E + e;
        if (k < S) {
            const float dalpha = gk[e] * st.T[e] - (after + dlam * st.lambda) / st.tt[e];
            const float dsigma = dalpha * st.delta[e] * st.ex[e];
            store(e, k, make_float4(st.w[e] * gr, st.w[e] * gg, st.w[e] * gb, dsigma));
            after += gk[e] * st.w[e];
        }
    }
}

// _sample_pdf / _sample_cdf (rendering.py:486-536) for one ray; bins / w / cdf are LDS arrays of nb + 1 / nb / nb + 1 floats,
// already filled with the mid-points and the weights + 1e-8 (render.hip::k_sample_pdf<true>); emit(f, z) receives sample f
template <class Emit>
__device__ __forceinline__ void sample_pdf_wave(float *bins, float *w, float *cdf, int nb, int nf, int det, const float *u, int lane, Emit emit) {
    const int V = 8, ILP = 4;
    const int nv = nb / V, q = nv / ILP;
    float p0 = 0.f;
    if (lane < V) {
        float part[ILP] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < q; ++i)
#pragma unroll
            for (int k = 0; k < ILP; ++k) part[k] += w[(i * ILP + k) * V + lane];
        for (int j = q * ILP; j < nv; ++j) part[0] += w[j * V + lane];
        part[0] += part[1];
        part[0] += part[2];
        part[0] += part[3];
        p0 = part[0];
    }
    float total = 0.f;
    for (int k = nv * V; k < nb; ++k) total += w[k];
#pragma unroll
    for (int l = 0; l < V; ++l) total += __shfl(p0, l);
    for (int i = lane; i < nb; i += 64) w[i] = w[i] / total;
    s_lds_fence();
    if (lane == 0) {
        double acc = 0.0;
        cdf[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            acc += (double)w[i];
            cdf[i + 1] = (float)acc;
        }
    }
    s_lds_fence();
    for (int f = lane; f < nf; f += 64) {
        const float uu = u[f];
        int lo = 0, hi = nb + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, nb);
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-8f) denom = 1.f;
        const float bb = bins[below], ba = bins[above];
        emit(f, bb + (uu - cb) / denom * (ba - bb));
    }
    (void)det;
}

// ---- k_step_mid --------------------------------------------------------------------------------------------------------------
struct MidArgs {
    long C, N;
    int Nc, Nf, Sb, Sfb, det;
    SSphere sp;
    const float *rays, *rays_bg, *last_delta, *z_c, *raw_c, *zb_asc, *zb_c, *braw_c, *u_f, *u_b, *t_f, *t_bf;
    const int32_t *scal;
    float *z_f, *xyz_f, *zb_f, *pts_f, *dr_f;
    long unit0, unit1;               // units [unit0, unit1) of the 2 C N (foreground rays, then background slots)
};

template <int EC, int EB>
__global__ __launch_bounds__(64 * WPB) void k_step_mid(MidArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long unit = a.unit0 + (long)blockIdx.x * WPB + wave;
    const long CN = a.C * a.N;
    if (unit >= a.unit1) return;
    const int per_wave = 4 * a.Nc + 8;
    float *zl = smem + wave * per_wave;          // Nc      z of the ray as the compositing sees it
    float *bins = zl + a.Nc;                     // <= Nc - 1
    float *w = bins + a.Nc;                      // <= Nc - 2
    float *cdf = w + a.Nc;                       // <= Nc - 1
    if (unit < CN) {
        // ---- foreground ray: rendering.py:195-225 ----
        const long r = unit;
        const int S = a.Nc, nb = S - 2;
        for (int i = lane; i < S; i += 64) zl[i] = a.z_c[r * S + i];
        s_lds_fence();
        float last = a.last_delta[r];
        if (last < 1e10f) last = last - s_wave_max(zl, S, lane);              // rendering.py:192-193
        Comp<EC> st;
        const float4 *raw = reinterpret_cast<const float4 *>(a.raw_c) + r * S;
        comp_forward<EC>(st, zl, S, lane, last, 0, [&](int k) { return raw[k]; });
        for (int i = lane; i <= nb; i += 64) bins[i] = 0.5f * (zl[i] + zl[i + 1]);           // rendering.py:213
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const int k = lane * EC + e;
            if (k >= 1 && k <= nb) w[k - 1] = st.w[e] + 1e-8f;                                 // :215 [:, 1:-1], :497
        }
        s_lds_fence();
        const float *ray = a.rays + r * 8;
        const float o0 = ray[0], o1 = ray[1], o2 = ray[2], d0 = ray[3], d1 = ray[4], d2 = ray[5];
        float *zf = a.z_f + r * a.Nf;
        float *xf = a.xyz_f + r * a.Nf * 3;
        sample_pdf_wave(bins, w, cdf, nb, a.Nf, a.det, a.det ? a.t_f : a.u_f + r * a.Nf, lane, [&](int f, float z) {
            zf[f] = z;
            xf[3 * f] = o0 + d0 * z; xf[3 * f + 1] = o1 + d1 * z; xf[3 * f + 2] = o2 + d2 * z;
        });
    } else {
        // ---- compacted background ray (flip: rendering.py:271-273; weights in flipped order against ascending bins: quirk Q1) ----
        const long g = unit - CN;
        const long cell = g / a.N, k0 = g - cell * a.N;
        if (k0 >= (long)a.scal[cell]) return;
        const int S = a.Sb, nb = S - 2;
        for (int i = lane; i < S; i += 64) zl[i] = a.zb_c[g * S + i];
        s_lds_fence();
        Comp<EB> st;
        const float4 *raw = reinterpret_cast<const float4 *>(a.braw_c) + g * S;
        comp_forward<EB>(st, zl, S, lane, 1e10f, 1, [&](int k) { return raw[k]; });
        const float *za = a.zb_asc + g * S;
        for (int i = lane; i <= nb; i += 64) bins[i] = 0.5f * (za[i] + za[i + 1]);
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int k = lane * EB + e;
            if (k >= 1 && k <= nb && k < S) w[k - 1] = st.w[e] + 1e-8f;
        }
        s_lds_fence();
        const float *ray = a.rays_bg + g * 8;
        float *zf = a.zb_f + g * a.Sfb;
        sample_pdf_wave(bins, w, cdf, nb, a.Sfb, a.det, a.det ? a.t_bf : a.u_b + g * a.Sfb, lane, [&](int f, float z) {
            zf[f] = z;
            float q[4], dr;
            s_bg_point(a.sp, ray, z, q, dr);
            *reinterpret_cast<float4 *>(a.pts_f + 4 * (g * a.Sfb + f)) = make_float4(q[0], q[1], q[2], q[3]);
            a.dr_f[g * a.Sfb + f] = dr;
        });
    }
}

// ---- k_step_tail -------------------------------------------------------------------------------------------------------------
struct TailArgs {
    long C, N;
    int Nc, Nf, Sb, Sfb;
    const float *z_c, *z_f, *raw_c, *raw_f, *zb_c, *zb_f, *braw_c, *braw_f, *last_delta, *target;
    const int32_t *slot;
    float *draw_c, *draw_f, *bdraw_c, *bdraw_f;
    float *rgb, *depth_var, *bg_lambda, *loss;
};

// stable rank sort of cat([fine, coarse]) (rendering.py:336-350; render.hip::k_merge_sorted): zm[rank] = key, src[rank] = element
__device__ __forceinline__ void merge_wave(float *key, float *zm, int *src, const float *zfine, int Sa, const float *zcoarse, int Sb, int flip,
                                           int lane) {
    const int St = Sa + Sb;
    // (zm / src start as the identity: with NaN depths -- a diverged model -- every rank below collapses to 0 and the entries not
    // written there must still be valid sample numbers, not whatever the LDS held)
    for (int i = lane; i < St; i += 64) { const float k = i < Sa ? zfine[i] : zcoarse[i - Sa]; key[i] = k; zm[i] = k; src[i] = i; }
    s_lds_fence();
    for (int e = lane; e < St; e += 64) {
        const float ke = key[e];
        int rank = 0;
        if (flip) {
            for (int j = 0; j < St; ++j) { const float kj = key[j]; rank += (kj > ke) || (kj == ke && j < e); }
        } else {
            for (int j = 0; j < St; ++j) { const float kj = key[j]; rank += (kj < ke) || (kj == ke && j < e); }
        }
        zm[rank] = ke;
        src[rank] = e;
    }
    s_lds_fence();
}

template <int EF, int EB>
__global__ __launch_bounds__(64 * WPB) void k_step_tail(TailArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * WPB + wave;
    if (r >= a.C * a.N) return;
    const long cell = r / a.N;
    const int Sm = a.Nc + a.Nf, Smb = a.Sb + a.Sfb;
    float *key = smem + wave * 3 * Sm;
    float *zm = key + Sm;
    int *src = reinterpret_cast<int *>(zm + Sm);

    // ---- background branch of this ray (the wave of the ray does both branches: the blend couples them) ----
    const int slot = a.slot[r];
    Comp<EB> sb;
    int srcb[EB];
    float bgr = 0.f, bgg = 0.f, bgb = 0.f;
    const long g = cell * a.N + (slot >= 0 ? slot : 0);
    if (slot >= 0) {
        merge_wave(key, zm, src, a.zb_f + g * a.Sfb, a.Sfb, a.zb_c + g * a.Sb, a.Sb, 1, lane);
#pragma unroll
        for (int e = 0; e < EB; ++e) { const int k = lane * EB + e; srcb[e] = k < Smb ? src[k] : 0; }
        const float4 *rf = reinterpret_cast<const float4 *>(a.braw_f) + g * a.Sfb, *rc = reinterpret_cast<const float4 *>(a.braw_c) + g * a.Sb;
        comp_forward<EB>(sb, zm, Smb, lane, 1e10f, 1, [&](int k) { const int s = src[k]; return s < a.Sfb ? rf[s] : rc[s - a.Sfb]; });
        float rr = 0.f, gg = 0.f, bb = 0.f;
#pragma unroll
        for (int e = 0; e < EB; ++e) { rr += sb.w[e] * sb.c[e].x; gg += sb.w[e] * sb.c[e].y; bb += sb.w[e] * sb.c[e].z; }
        bgr = s_wave_sum(rr); bgg = s_wave_sum(gg); bgb = s_wave_sum(bb);
        s_lds_fence();
    }

    // ---- foreground branch ----
    const float *zf = a.z_f + r * a.Nf;
    merge_wave(key, zm, src, zf, a.Nf, a.z_c + r * a.Nc, a.Nc, 0, lane);
    int srcf[EF];
#pragma unroll
    for (int e = 0; e < EF; ++e) { const int k = lane * EF + e; srcf[e] = k < Sm ? src[k] : 0; }
    float last = a.last_delta[r];
    if (last < 1e10f) last = last - s_wave_max(zf, a.Nf, lane);                 // rendering.py:224-225 (fine-only max: quirk Q4)
    Comp<EF> sf;
    {
        const float4 *rf = reinterpret_cast<const float4 *>(a.raw_f) + r * a.Nf, *rc = reinterpret_cast<const float4 *>(a.raw_c) + r * a.Nc;
        comp_forward<EF>(sf, zm, Sm, lane, last, 0, [&](int k) { const int s = src[k]; return s < a.Nf ? rf[s] : rc[s - a.Nf]; });
    }
    float rr = 0.f, gg = 0.f, bb = 0.f, dsum = 0.f;
#pragma unroll
    for (int e = 0; e < EF; ++e) {
        rr += sf.w[e] * sf.c[e].x; gg += sf.w[e] * sf.c[e].y; bb += sf.w[e] * sf.c[e].z;
        dsum += sf.w[e] * sf.z[e];
    }
    rr = s_wave_sum(rr); gg = s_wave_sum(gg); bb = s_wave_sum(bb);
    dsum = s_wave_sum(dsum);
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < EF; ++e) { const float df = sf.z[e] - dsum; var += sf.w[e] * (df * df); }
    var = s_wave_sum(var);
    const float lam = sf.lambda;

    // ---- blend (rendering.py:102-131), loss (runner.py:370 mse_loss, mean over n_rays x 3) and their adjoints ----
    float rgb[3] = {rr, gg, bb};
    if (slot >= 0) { rgb[0] = rr + bgr * lam; rgb[1] = gg + bgg * lam; rgb[2] = bb + bgb * lam; }
    const float *tg = a.target + r * 3;
    const float inv = 1.f / (float)(3 * a.N);
    float d[3], sq = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float df = rgb[c] - tg[c]; sq += df * df; d[c] = (2.f * inv) * df; }
    if (lane == 0) {
        a.rgb[3 * r] = rgb[0]; a.rgb[3 * r + 1] = rgb[1]; a.rgb[3 * r + 2] = rgb[2];
        a.depth_var[r] = var;
        a.bg_lambda[r] = lam;
        atomicAdd(a.loss + cell, sq * inv);
    }
    float dlam = 0.f;
    if (slot >= 0) { dlam += d[0] * bgr; dlam += d[1] * bgg; dlam += d[2] * bgb; }
    {
        float4 *df = reinterpret_cast<float4 *>(a.draw_f) + r * a.Nf, *dc = reinterpret_cast<float4 *>(a.draw_c) + r * a.Nc;
        comp_backward<EF>(sf, Sm, lane, d[0], d[1], d[2], dlam, [&](int e, int, float4 v) {
            const int s = srcf[e];
            if (s < a.Nf) df[s] = v; else dc[s - a.Nf] = v;
        });
    }
    if (slot >= 0) {
        float4 *df = reinterpret_cast<float4 *>(a.bdraw_f) + g * a.Sfb, *dc = reinterpret_cast<float4 *>(a.bdraw_c) + g * a.Sb;
        comp_backward<EB>(sb, Smb, lane, lam * d[0], lam * d[1], lam * d[2], 0.f, [&](int e, int, float4 v) {
            const int s = srcb[e];
            if (s < a.Sfb) df[s] = v; else dc[s - a.Sfb] = v;
        });
    }
}

// ---- k_render_tail: the inference form of k_step_tail (rendering.py:102-139, 336-393 with get_depth / get_bg_fg_rgb) --------------
struct RTailArgs {
    long N;
    int Nc, Nf, Sb, Sfb;
    const float *z_c, *z_f, *raw_c, *raw_f, *zb_c, *zb_f, *braw_c, *braw_f, *dr_c, *dr_f, *last_delta;
    const int32_t *slot;
    float *rgb, *depth, *fg_rgb, *bg_rgb, *fg_depth, *bg_depth, *bg_lambda;
};

template <int EF, int EB>
__global__ __launch_bounds__(64 * WPB) void k_render_tail(RTailArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * WPB + wave;
    if (r >= a.N) return;
    const int Sm = a.Nc + a.Nf, Smb = a.Sb + a.Sfb;
    float *key = smem + wave * 3 * Sm;
    float *zm = key + Sm;
    int *src = reinterpret_cast<int *>(zm + Sm);
    const int slot = a.slot[r];
    float bgr = 0.f, bgg = 0.f, bgb = 0.f, bgd = 0.f;
    if (slot >= 0) {
        const long g = slot;
        merge_wave(key, zm, src, a.zb_f + g * a.Sfb, a.Sfb, a.zb_c + g * a.Sb, a.Sb, 1, lane);
        Comp<EB> sb;
        const float4 *rf = reinterpret_cast<const float4 *>(a.braw_f) + g * a.Sfb, *rc = reinterpret_cast<const float4 *>(a.braw_c) + g * a.Sb;
        comp_forward<EB>(sb, zm, Smb, lane, 1e10f, 1, [&](int k) { const int s = src[k]; return s < a.Sfb ? rf[s] : rc[s - a.Sfb]; });
        float rr = 0.f, gg = 0.f, bb = 0.f, dd = 0.f;
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int k = lane * EB + e;
            rr += sb.w[e] * sb.c[e].x; gg += sb.w[e] * sb.c[e].y; bb += sb.w[e] * sb.c[e].z;
            if (k < Smb) {
                // depth_real of the merged sample: the fine pass's own, or the coarse pass's UNFLIPPED array (quirk Q2)
                const int s = src[k];
                dd += sb.w[e] * (s < a.Sfb ? a.dr_f[g * a.Sfb + s] : a.dr_c[g * a.Sb + (s - a.Sfb)]);
            }
        }
        bgr = s_wave_sum(rr); bgg = s_wave_sum(gg); bgb = s_wave_sum(bb); bgd = s_wave_sum(dd);
        s_lds_fence();
    }
    const float *zf = a.z_f + r * a.Nf;
    merge_wave(key, zm, src, zf, a.Nf, a.z_c + r * a.Nc, a.Nc, 0, lane);
    float last = a.last_delta[r];
    if (last < 1e10f) last = last - s_wave_max(zf, a.Nf, lane);
    Comp<EF> sf;
    {
        const float4 *rf = reinterpret_cast<const float4 *>(a.raw_f) + r * a.Nf, *rc = reinterpret_cast<const float4 *>(a.raw_c) + r * a.Nc;
        comp_forward<EF>(sf, zm, Sm, lane, last, 0, [&](int k) { const int s = src[k]; return s < a.Nf ? rf[s] : rc[s - a.Nf]; });
    }
    float rr = 0.f, gg = 0.f, bb = 0.f, dsum = 0.f;
#pragma unroll
    for (int e = 0; e < EF; ++e) {
        rr += sf.w[e] * sf.c[e].x; gg += sf.w[e] * sf.c[e].y; bb += sf.w[e] * sf.c[e].z;
        dsum += sf.w[e] * sf.z[e];
    }
    rr = s_wave_sum(rr); gg = s_wave_sum(gg); bb = s_wave_sum(bb); dsum = s_wave_sum(dsum);
    if (lane != 0) return;
    const float lam = sf.lambda;
    const float br = slot >= 0 ? bgr * lam : 0.f, bgv = slot >= 0 ? bgg * lam : 0.f, bbv = slot >= 0 ? bgb * lam : 0.f;
    const float bd = slot >= 0 ? bgd * lam : 0.f;
    a.rgb[3 * r] = rr + br; a.rgb[3 * r + 1] = gg + bgv; a.rgb[3 * r + 2] = bb + bbv;
    a.bg_lambda[r] = lam;
    if (a.fg_rgb) { a.fg_rgb[3 * r] = rr; a.fg_rgb[3 * r + 1] = gg; a.fg_rgb[3 * r + 2] = bb; }
    if (a.bg_rgb) { a.bg_rgb[3 * r] = br; a.bg_rgb[3 * r + 1] = bgv; a.bg_rgb[3 * r + 2] = bbv; }
    if (a.depth) a.depth[r] = dsum + bd;
    if (a.fg_depth) a.fg_depth[r] = dsum;
    if (a.bg_depth) a.bg_depth[r] = bd;
}

// ---- optimiser + re-pack -----------------------------------------------------------------------------------------------------
// gate: NULL, or the device-side count of rays with a background segment of the tensor's cell -- the reference steps the background
// optimiser only when the batch had such rays (runner.py:268-272); steps: how many updates the tensor's optimiser has applied so far
// (torch.optim.Adam's per-parameter `step`, the exponent of the bias corrections), advanced by k_step_pack behind this kernel.
struct AdamTensor { float *p; const float *g; float *m, *v; long n, block0; const int32_t *gate; const int32_t *steps; };

__global__ __launch_bounds__(256) void k_step_adam(const AdamTensor *__restrict__ tab, int n_tensors, double beta1, double beta2, double eps,
                                                   double lr) {
    // binary search: the tensor whose block range holds this block
    int lo = 0, hi = n_tensors - 1;
    const long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const AdamTensor t = tab[lo];
    if (t.gate && *t.gate <= 0) return;                                     // (uniform per block)
    // torch/optim/adam.py _single_tensor_adam: the scalars are Python doubles, rounded to fp32 where they meet a tensor
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
        const double step = (double)(*t.steps + 1);
        const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
        sh[0] = (float)(lr / bc1);                                           // step_size
        sh[1] = (float)sqrt(bc2);                                            // bias_correction2_sqrt
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1];
    const float w1 = (float)(1.0 - beta1), b2f = (float)beta2, w2 = (float)(1.0 - beta2), epsf = (float)eps;
    const long i0 = ((b - t.block0) * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + j;
        if (i < t.n) {
            // no weight decay, no amsgrad
            const float g = t.g[i];
            const float m = t.m[i] + (g - t.m[i]) * w1;                     // exp_avg.lerp_(grad, 1 - beta1)
            const float v = b2f * t.v[i] + w2 * g * g;                      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = sqrtf(v) / bc2_sqrt + epsf;
            t.p[i] = t.p[i] - step_size * (m / denom);
            t.m[i] = m; t.v[i] = v;
        }
    }
}

struct PackJob {
    int kind;                  // 0: forward image (ModelLayout), 1: transposed image (BwdLayout), 2 / 3: their split-precision forms
    long block0, nblocks, n_u4;
    float4 *chunks;
    float *aux;
    // end-of-step bookkeeping, done by thread 0 of a model's forward-image job when the kernel runs behind k_step_adam (after_adam):
    const int32_t *gate;       // as AdamTensor::gate: the model was not updated this step -> its images are current, its step count stands
    int32_t *steps;            // the model's optimiser step counter (forward-image jobs only)
    const float *loss;         // foreground forward-image jobs only: the cell's loss / error flag of this step -> sticky health bits
    const int32_t *err;
    int32_t *sticky;
    ModelLayout m;
    BwdLayout b;
};
__global__ __launch_bounds__(256) void k_step_pack(const PackJob *__restrict__ jobs, int n_jobs, int after_adam) {
    int j = 0;
    const long blk = blockIdx.x;
    for (int i = 1; i < n_jobs; ++i) j += blk >= jobs[i].block0;
    const PackJob &job = jobs[j];
    const long tid = (blk - job.block0) * 256 + threadIdx.x;
    if (after_adam) {
        const bool skipped = job.gate && *job.gate <= 0;
        if (tid == 0) {
            if (job.steps && !skipped) job.steps[0] += 1;
            if (job.sticky) {
                const float l = job.loss[0];
                const int bits = ((l - l) != 0.f ? MNR_STEP_STICKY_NONFINITE : 0) | (job.err[0] ? MNR_STEP_STICKY_OUTSIDE : 0);
                if (bits) job.sticky[0] |= bits;
            }
        }
        if (skipped) return;
    }
    if (job.kind == 0) pack_model_thread(job.m, job.chunks, job.aux, tid);
    else if (job.kind == 1) pack_bwd_thread(job.b, job.chunks, tid);
    else if (job.kind == 2) pack_model_h2_thread(job.m, reinterpret_cast<uint4v *>(job.chunks), job.aux, job.n_u4, tid);
    else pack_bwd_h2_thread(job.b, reinterpret_cast<uint4v *>(job.chunks), tid);
}


// ---- 512-wide foreground: the head adjoints of one (cell, pass) in ONE pass over its rows -------------------------------------------
// per row r:  g_rgb = d_rgb * s (1 - s)  (sigmoid),  g_sig = d_sigma * act'(sigma)  (ReLU, or 1 - exp(-sigma) for the shifted softplus);
//   d_src[r][j]  = dact[r][j] > 0 ? sum_c g_rgb[c] W_rgb[c][j] : 0        (data gradient through the rgb layer + ReLU adjoint of dir_a)
//   dW_rgb[c][j] += g_rgb[c] dact[r][j],  db_rgb[c] += g_rgb[c],  dW_sigma[k] += g_sig hs7[r][k],  db_sigma += g_sig
// (what k_act_grad x 3, k_gemm x 3 and k_col_sum x 2 did in eight launches, each re-reading a [rows][256 | 512] plane: 0.45 ms per fine
// pass; here dact and hs7 are read once and d_src written once).  Thread (g, j) of a block's four row groups owns column j of dact / d_src and
// columns j, j + 256 of hs7 for the rows of group g; few long blocks, the groups' sums meet in LDS: one set of atomics per block (as k_head_grads).
constexpr int WH_U = 4;              // rows in flight per iteration
constexpr int WH_G = 4;              // row groups per block (256 threads each): one set of atomics per block for four times the wavefronts
__global__ __launch_bounds__(256 * WH_G) void k_wide_head_adjoint(const float *__restrict__ d_out, const float *__restrict__ out, const float *__restrict__ dact,
                                                           const float *__restrict__ hs7, const float *__restrict__ rgb_w, long B, int sigma_softplus,
                                                           float *__restrict__ d_src, float *__restrict__ g_sig_out, float *__restrict__ d_rgb_w,
                                                           float *__restrict__ d_rgb_b, float *__restrict__ d_sigma_w, float *__restrict__ d_sigma_b) {
    __shared__ float red[WH_G - 1][5][256];
    __shared__ float redb[WH_G - 1][4];
    const int j = threadIdx.x & 255, g = threadIdx.x >> 8;
    const long per = ((B + gridDim.x - 1) / gridDim.x + WH_U * WH_G - 1) / (WH_U * WH_G) * (WH_U * WH_G);
    const long rb = (long)blockIdx.x * per, re = min(B, rb + per);
    const float w0 = rgb_w[j], w1 = rgb_w[256 + j], w2 = rgb_w[512 + j];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, s0 = 0.f, s1 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, bs = 0.f;
    for (long r0 = rb + g * WH_U; r0 < re; r0 += WH_U * WH_G) {
        float4 go[WH_U], o[WH_U];
        float x[WH_U], h0[WH_U], h1[WH_U];
#pragma unroll
        for (int u = 0; u < WH_U; ++u) {
            const long r = min(r0 + u, re - 1);
            go[u] = *reinterpret_cast<const float4 *>(d_out + r * 4);
            o[u] = *reinterpret_cast<const float4 *>(out + r * 4);
            x[u] = dact[r * 256 + j]; h0[u] = hs7[r * 512 + j]; h1[u] = hs7[r * 512 + 256 + j];
        }
#pragma unroll
        for (int u = 0; u < WH_U; ++u) {
            if (r0 + u >= re) break;
            const float g0 = go[u].x * (o[u].x * (1.f - o[u].x)), g1 = go[u].y * (o[u].y * (1.f - o[u].y)), g2 = go[u].z * (o[u].z * (1.f - o[u].z));
            const float gs = sigma_softplus ? go[u].w * (1.f - expf(-o[u].w)) : (o[u].w > 0.f ? go[u].w : 0.f);
            float d = g0 * w0;
            d = fmaf(g1, w1, d);
            d = fmaf(g2, w2, d);
            d_src[(r0 + u) * 256 + j] = x[u] > 0.f ? d : 0.f;
            a0 = fmaf(g0, x[u], a0); a1 = fmaf(g1, x[u], a1); a2 = fmaf(g2, x[u], a2);
            s0 = fmaf(gs, h0[u], s0); s1 = fmaf(gs, h1[u], s1);
            if (j == 0) { b0 += g0; b1 += g1; b2 += g2; bs += gs; g_sig_out[r0 + u] = gs; }
        }
    }
    // the block's row groups meet in LDS; group 0 adds the block's sums with one set of atomics
    if (g > 0) {
        red[g - 1][0][j] = a0; red[g - 1][1][j] = a1; red[g - 1][2][j] = a2; red[g - 1][3][j] = s0; red[g - 1][4][j] = s1;
        if (j == 0) { redb[g - 1][0] = b0; redb[g - 1][1] = b1; redb[g - 1][2] = b2; redb[g - 1][3] = bs; }
    }
    __syncthreads();
    if (g > 0 || rb >= re) return;
#pragma unroll
    for (int q = 0; q < WH_G - 1; ++q) {
        a0 += red[q][0][j]; a1 += red[q][1][j]; a2 += red[q][2][j]; s0 += red[q][3][j]; s1 += red[q][4][j];
        if (j == 0) { b0 += redb[q][0]; b1 += redb[q][1]; b2 += redb[q][2]; bs += redb[q][3]; }
    }
    atomicAdd(d_rgb_w + j, a0); atomicAdd(d_rgb_w + 256 + j, a1); atomicAdd(d_rgb_w + 512 + j, a2);
    atomicAdd(d_sigma_w + j, s0); atomicAdd(d_sigma_w + 256 + j, s1);
    if (j == 0) { atomicAdd(d_rgb_b, b0); atomicAdd(d_rgb_b + 1, b1); atomicAdd(d_rgb_b + 2, b2); atomicAdd(d_sigma_b, bs); }
}

}  // namespace mnr

using namespace mnr;

// =====================================================================================================================
struct mnr_step_plan {
    mnr_step_cfg cfg;
    std::vector<mnr_step_model> models;
    StepDims D;
    StepWs L;
    char *ws;
    std::vector<float> tables;        // host copies of the four linspace tables (cfg's pointers are not kept)
    int n_pack_jobs, n_adam_tensors;
    long pack_blocks, adam_blocks;
    SSphere sp;
    std::vector<hipEvent_t> events;   // profiling: n_slots x MNR_STEP_SPANS x (start, stop)
    std::vector<uint8_t> ev_alias;    // ... which event of the slot holds boundary (span, end): adjacent spans share one record (mark2)
    int prof_slots = 0;
    long prof_step = 0;
    // single-cell split-precision plans run the background branch of the forward (coarse pass -> fine samples -> fine pass) on a stream
    // of their own, forked / joined by two events: the foreground passes are whole rounds of workgroups, the background's would be a
    // partial round behind each of them (section 3e of DESIGN.md)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_after_coarse = false;   // side-stream schedule: the background branch beside the foreground's fine pass only (see mnr_train_step)
    ~mnr_step_plan() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
};

// the shortest decimal that round-trips the float, read back as a double: 0.9f -> 0.9, the value torch.optim.Adam computes with
static double as_typed(float f) {
    char buf[32];
    snprintf(buf, sizeof buf, "%.7g", (double)f);
    return strtod(buf, nullptr);
}

static size_t pack_table_bytes(int C) { return (size_t)4 * C * sizeof(PackJob); }
static size_t adam_table_bytes(int C) { return (size_t)2 * C * 32 * sizeof(AdamTensor); }

static void finish_layout(const mnr_step_cfg *cfg, const StepDims &D, StepWs &L) {
    step_layout(cfg, D, L);
    // the two variable-size tables go behind everything else
    size_t off = L.total;
    L.tab_pack = off; off += (pack_table_bytes((int)D.C) + 255) / 256 * 256;
    L.tab_adam = off; off += (adam_table_bytes((int)D.C) + 255) / 256 * 256;
    L.total = off;
}

extern "C" int mnr_step_query(const mnr_step_cfg *cfg, const mnr_model_desc *fg, const mnr_model_desc *bg, mnr_step_layout *out) {
    MNR_REQUIRE(out, "NULL argument");
    StepDims D;
    int rc = step_dims(cfg, fg, bg, D);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(cfg->grad_floats_per_cell > 0, "grad_floats_per_cell must be positive");
    StepWs L;
    finish_layout(cfg, D, L);
    out->workspace_bytes = L.total;
    out->grad_offset = L.grads; out->grad_stride = L.grad_stride;
    out->loss_offset = L.loss; out->rgb_offset = L.rgb; out->depth_var_offset = L.depth_var; out->bg_lambda_offset = L.bg_lambda;
    out->n_bg_offset = L.scal; out->err_offset = L.scal + MAXC * 4;
    out->tape_fg_offset = L.tape_f; out->tape_bg_offset = L.tape_b; out->tape_fg_rows = D.C * D.cap_f; out->tape_bg_rows = D.C * D.cap_b;
    out->gtape_fg_offset = L.gtape_f; out->gtape_bg_offset = L.gtape_b;
    out->sticky_offset = L.sticky;
    return MNR_OK;
}

// tensors of one model in (param, grad, m, v) form
static int adam_tensors_of(const mnr_step_model &M, const int32_t *gate, std::vector<AdamTensor> &out) {
    const mnr_model_desc &d = M.desc;
    const int W = d.layer_dim, E = emb_cols(d.xyz_dim, d.pos_xyz_dim), ED = emb_cols(3, d.pos_dir_dim);
    int err = 0;
    auto add = [&](const float *p, float *g, float *m, float *v, long n) {
        if (!p || !g || !m || !v) { err = 1; return; }
        out.push_back(AdamTensor{const_cast<float *>(p), g, m, v, n, 0, gate, M.adam_steps_dev});
    };
    for (int l = 0; l < d.layers; ++l) {
        const long in = l == 0 ? E : (((d.skip_mask >> l) & 1) ? E + W : W);
        add(d.layer_w[l], M.grad.layer_w[l], M.adam_m.layer_w[l], M.adam_v.layer_w[l], (long)W * in);
        add(d.layer_b[l], M.grad.layer_b[l], M.adam_m.layer_b[l], M.adam_v.layer_b[l], W);
    }
    add(d.embedding_a, M.grad.embedding_a, M.adam_m.embedding_a, M.adam_v.embedding_a, (long)d.appearance_count * d.appearance_dim);
    add(d.final_w, M.grad.final_w, M.adam_m.final_w, M.adam_v.final_w, (long)W * W);
    add(d.final_b, M.grad.final_b, M.adam_m.final_b, M.adam_v.final_b, W);
    add(d.dir_a_w, M.grad.dir_a_w, M.adam_m.dir_a_w, M.adam_v.dir_a_w, (long)(W / 2) * (W + ED + d.appearance_dim));
    add(d.dir_a_b, M.grad.dir_a_b, M.adam_m.dir_a_b, M.adam_v.dir_a_b, W / 2);
    add(d.sigma_w, M.grad.sigma_w, M.adam_m.sigma_w, M.adam_v.sigma_w, W);
    add(d.sigma_b, M.grad.sigma_b, M.adam_m.sigma_b, M.adam_v.sigma_b, 1);
    add(d.rgb_w, M.grad.rgb_w, M.adam_m.rgb_w, M.adam_v.rgb_w, (long)d.rgb_dim * (W / 2));
    add(d.rgb_b, M.grad.rgb_b, M.adam_m.rgb_b, M.adam_v.rgb_b, d.rgb_dim);
    MNR_REQUIRE(!err && M.adam_steps_dev, "mnr_step_create: a parameter / gradient / Adam-moment / step-counter pointer is missing");
    return MNR_OK;
}

// 512-wide foreground: the zero-padded copies of the two weight matrices whose hidden-input block does not start on a 16-byte boundary in
// nn.Linear's layout -- skip layer [512][75 + 512] -> [512][96 + 512], dir_a layer [256][512 + 75] -> [256][512 + 96] -- which the tiled
// data-gradient GEMMs address (models/layerwise.py keeps the same copies); refreshed behind every optimiser step
static int wide_refresh_weights(mnr_step_plan *p, hipStream_t s) {
    if (!p->D.wide) return MNR_OK;
    const int C = (int)p->D.C;
    bool ok = true;
    for (int c = 0; c < C; ++c) {
        const mnr_model_desc &d = p->models[2 * c].desc;
        float *wskip = reinterpret_cast<float *>(p->ws + p->L.w_wskip) + (size_t)c * 512 * (WIDE_EP + 512);
        float *wdir = reinterpret_cast<float *>(p->ws + p->L.w_wdir) + (size_t)c * 256 * (512 + WIDE_SP);
        const float *w4 = d.layer_w[4];
        const size_t dp = (size_t)(WIDE_EP + 512) * 4, sp = (size_t)(75 + 512) * 4;
        ok = ok && hipMemcpy2DAsync(wskip, dp, w4, sp, 75 * 4, 512, hipMemcpyDeviceToDevice, s) == hipSuccess;
        ok = ok && hipMemcpy2DAsync(wskip + WIDE_EP, dp, w4 + 75, sp, 512 * 4, 512, hipMemcpyDeviceToDevice, s) == hipSuccess;
        ok = ok && hipMemcpy2DAsync(wdir, (size_t)(512 + WIDE_SP) * 4, d.dir_a_w, (size_t)(512 + 75) * 4, (512 + 75) * 4, 256, hipMemcpyDeviceToDevice, s) == hipSuccess;
    }
    if (!ok) return set_err(MNR_E_LAUNCH, "wide_refresh_weights: %s", hipGetErrorString(hipGetLastError()));
    return MNR_OK;
}

extern "C" int mnr_step_repack(mnr_step_plan *p, void *stream) {
    MNR_REQUIRE(p, "NULL plan");
    hipLaunchKernelGGL(k_step_pack, dim3((unsigned)p->pack_blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const PackJob *>(p->ws + p->L.tab_pack), p->n_pack_jobs, 0);
    int rc = check_launch("k_step_pack");
    if (rc) return rc;
    return wide_refresh_weights(p, as_stream(stream));
}

extern "C" int mnr_step_create(mnr_step_plan **out, const mnr_step_cfg *cfg, const mnr_step_model *models, void *workspace_dev,
                               size_t workspace_bytes, void *stream) {
    MNR_REQUIRE(out && cfg && models && workspace_dev, "NULL argument");
    StepDims D;
    int rc = step_dims(cfg, &models[0].desc, &models[1].desc, D);
    if (rc != MNR_OK) return rc;
    MNR_REQUIRE(cfg->t_coarse && cfg->t_bg_coarse && cfg->t_fine && cfg->t_bg_fine, "linspace tables missing");
    MNR_REQUIRE(cfg->sphere_radius[0] > 0 && cfg->sphere_radius[1] > 0 && cfg->sphere_radius[2] > 0, "sphere_radius must be positive");
    StepWs L;
    finish_layout(cfg, D, L);
    MNR_REQUIRE(workspace_bytes >= L.total, "workspace too small: %zu < %zu", workspace_bytes, L.total);
    const int C = (int)D.C;
    hipStream_t s = as_stream(stream);
    char *ws = reinterpret_cast<char *>(workspace_dev);
    auto *plan = new mnr_step_plan();
    plan->cfg = *cfg; plan->D = D; plan->L = L; plan->ws = ws;
    plan->models.assign(models, models + 2 * C);
    plan->cfg.t_coarse = plan->cfg.t_bg_coarse = plan->cfg.t_fine = plan->cfg.t_bg_fine = nullptr;
    plan->sp = SSphere{cfg->sphere_center[0], cfg->sphere_center[1], cfg->sphere_center[2], cfg->sphere_radius[0], cfg->sphere_radius[1],
                       cfg->sphere_radius[2]};
    auto fail = [&](int code) { delete plan; return code; };
    // every cell: same architectures, all pointers present, gradient views inside the workspace's gradient area
    std::vector<MlpCellSeg> cells(4 * C);
    std::vector<PackJob> jobs;
    std::vector<AdamTensor> adam;
    long pack_blocks = 0;
    for (int c = 0; c < C; ++c) {
        for (int k = 0; k < 2; ++k) {
            const mnr_step_model &M = models[2 * c + k];
            const mnr_model_desc &d0 = models[k].desc, &d = M.desc;
            if (d.xyz_dim != d0.xyz_dim || d.pos_xyz_dim != d0.pos_xyz_dim || d.pos_dir_dim != d0.pos_dir_dim || d.layers != d0.layers ||
                d.skip_mask != d0.skip_mask || d.layer_dim != d0.layer_dim || d.appearance_dim != d0.appearance_dim ||
                d.appearance_count != d0.appearance_count || d.rgb_dim != d0.rgb_dim || d.sigma_activation != d0.sigma_activation)
                return fail(set_err(MNR_E_INVALID, "mnr_step_create: every cell must have the architecture of cell 0"));
            const bool split = cfg->split_precision != 0;
            void *img_f = split ? M.packed_h2_dev : M.packed_dev, *img_b = split ? M.packed_bwd_h2_dev : M.packed_bwd_dev;
            const bool wide_fg = D.wide && k == 0;          // (its backward reads the nn.Linear weights themselves: no transposed image)
            if (!img_f || (!img_b && !wide_fg) || !d.embedding_a || !M.grad.embedding_a)
                return fail(set_err(MNR_E_INVALID, "mnr_step_create: cell %d: packed image / embedding pointers missing", c));
            const char *g0 = ws + L.grads + (size_t)c * L.grad_stride;
            const char *ge = g0 + (size_t)cfg->grad_floats_per_cell * 4;
            const char *gp = reinterpret_cast<const char *>(M.grad.sigma_b);
            if (gp < g0 || gp >= ge) return fail(set_err(MNR_E_INVALID, "mnr_step_create: cell %d: gradients must live in the workspace's gradient area", c));
            ModelLayout ml;
            BwdLayout bl;
            if ((rc = layout_from_desc(&d, ml)) != MNR_OK || (!wide_fg && (rc = bwd_layout_from_desc(&d, bl)) != MNR_OK)) return fail(rc);
            PackJob jf{};
            jf.m = ml; jf.chunks = reinterpret_cast<float4 *>(img_f);
            if (split) {
                jf.kind = 2; jf.n_u4 = (long)h2_total_chunks(ml) * H2_CHUNK_U4;
                jf.aux = reinterpret_cast<float *>(reinterpret_cast<char *>(img_f) + (size_t)jf.n_u4 * 16);
                jf.nblocks = (jf.n_u4 + ml.aux_floats + 255) / 256;
            } else {
                jf.kind = 0;
                jf.aux = reinterpret_cast<float *>(reinterpret_cast<char *>(img_f) + (size_t)ml.total_chunks * CHUNK_BYTES);
                jf.nblocks = ((long)ml.total_chunks * CHUNK_F4 + ml.aux_floats + 255) / 256;
            }
            jf.block0 = pack_blocks;
            pack_blocks += jf.nblocks;
            jobs.push_back(jf);
            const size_t jfi = jobs.size() - 1;
            // background models are stepped only on batches with background rays (runner.py:268-272): gated on the cell's device-side count
            const int32_t *gate = k == 1 ? reinterpret_cast<const int32_t *>(ws + L.scal) + c : nullptr;
            if (!wide_fg) {
                PackJob jb{};
                jb.kind = split ? 3 : 1; jb.b = bl; jb.chunks = reinterpret_cast<float4 *>(img_b);
                jb.block0 = pack_blocks;
                jb.nblocks = split ? ((long)h2b_total_chunks(bl) * H2_CHUNK_U4 + 255) / 256 : ((long)bl.total_chunks * CHUNK_F4 + 255) / 256;
                pack_blocks += jb.nblocks;
                jb.gate = gate;
                jobs.push_back(jb);
            }
            jobs[jfi].gate = gate;
            jobs[jfi].steps = M.adam_steps_dev;
            if (k == 0) {
                PackJob &jf0 = jobs[jfi];
                jf0.loss = reinterpret_cast<const float *>(ws + L.loss) + c;
                jf0.err = reinterpret_cast<const int32_t *>(ws + L.scal) + MAXC + c;
                jf0.sticky = reinterpret_cast<int32_t *>(ws + L.sticky) + c;
            }
            if ((rc = adam_tensors_of(M, gate, adam)) != MNR_OK) return fail(rc);
            // the four (branch, pass) cell tables: fg coarse, fg fine, bg coarse, bg fine
            for (int pass = 0; pass < 2; ++pass) {
                MlpCellSeg &e = cells[(2 * k + pass) * C + c];
                e.packed = img_f; e.packed_bwd = img_b; e.emb_a = d.embedding_a; e.d_emb_a = M.grad.embedding_a;
                const long cap = k == 0 ? D.cap_f : D.cap_b, first = k == 0 ? D.N * D.Nc : D.N * D.Sb;
                e.tape_row0 = (long)c * cap + (pass ? first : 0);
                e.n_units = k == 0 ? nullptr : reinterpret_cast<const int32_t *>(ws + L.scal) + c;
                // exponent words of the model's gradient-tape planes: ints 32.. (fg) / 48.. (bg) of the cell's zeroed control block
                e.zexp = split ? reinterpret_cast<int32_t *>(ws + L.wcount + (size_t)c * 256) + 32 + 16 * k : nullptr;
            }
        }
    }
    MNR_REQUIRE((int)adam.size() <= 2 * C * 32, "internal: Adam table overflow");
    long ab = 0;
    for (AdamTensor &t : adam) { t.block0 = ab; ab += (t.n + 1023) / 1024; }
    plan->n_pack_jobs = (int)jobs.size(); plan->pack_blocks = pack_blocks;
    plan->n_adam_tensors = (int)adam.size(); plan->adam_blocks = ab;
    plan->tables.assign(cfg->t_coarse, cfg->t_coarse + D.Nc);
    plan->tables.insert(plan->tables.end(), cfg->t_bg_coarse, cfg->t_bg_coarse + D.Sb);
    plan->tables.insert(plan->tables.end(), cfg->t_fine, cfg->t_fine + D.Nf);
    plan->tables.insert(plan->tables.end(), cfg->t_bg_fine, cfg->t_bg_fine + D.Sfb);
    bool ok = true;
    auto up = [&](size_t off, const void *src, size_t bytes) { ok = ok && hipMemcpyAsync(ws + off, src, bytes, hipMemcpyHostToDevice, s) == hipSuccess; };
    up(L.tab_cells, cells.data(), cells.size() * sizeof(MlpCellSeg));
    up(L.tab_pack, jobs.data(), jobs.size() * sizeof(PackJob));
    up(L.tab_adam, adam.data(), adam.size() * sizeof(AdamTensor));
    up(L.t_c, plan->tables.data(), D.Nc * 4);
    up(L.t_bc, plan->tables.data() + D.Nc, D.Sb * 4);
    up(L.t_f, plan->tables.data() + D.Nc + D.Sb, D.Nf * 4);
    up(L.t_bf, plan->tables.data() + D.Nc + D.Sb + D.Nf, D.Sfb * 4);
    ok = ok && hipMemsetAsync(ws + L.sticky, 0, MAXC * 4, s) == hipSuccess;
    if (D.wide)      // the pad columns of the weight copies and of the two input planes stay zero from here on (only valid columns are rewritten)
        ok = ok && hipMemsetAsync(ws + L.w_emb, 0, L.w_grgb - L.w_emb, s) == hipSuccess &&
             hipMemsetAsync(ws + L.w_wskip, 0, L.w_wgws - L.w_wskip, s) == hipSuccess;
    // the host vectors above die with this scope: the copies must have left them
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) return fail(set_err(MNR_E_LAUNCH, "mnr_step_create: table upload failed: %s", hipGetErrorString(hipGetLastError())));
    // Measured on the benchmark step: split-precision step 3.36 -> 3.21 ms; fp32 step 6.39 -> 6.35 ms (its foreground passes are longer,
    // the partial rounds weigh less) -- within the box-to-box spread, and it would make every per-launch duration of the forward
    // kernel an overlapped one, so the fp32 step keeps one stream unless MNR_STEP_TWO_STREAMS is set.
    if (C == 1 && !D.wide && !getenv("MNR_STEP_ONE_STREAM") && (cfg->split_precision || getenv("MNR_STEP_TWO_STREAMS"))) {
        const char *mode = getenv("MNR_STEP_TWO_STREAMS");
        plan->fork_after_coarse = mode && mode[0] == '2';
        // (failure to get the side stream is not an error: the step then runs its two branches in one launch each, as multi-cell plans do)
        if (hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking) != hipSuccess) plan->side = nullptr;
        if (plan->side && (hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming) != hipSuccess)) {
            (void)hipStreamDestroy(plan->side);
            plan->side = nullptr;
        }
        (void)hipGetLastError();
    }
    rc = mnr_step_repack(plan, stream);
    if (rc != MNR_OK) return fail(rc);
    *out = plan;
    return MNR_OK;
}

extern "C" void mnr_step_destroy(mnr_step_plan *p) { delete p; }

extern "C" int mnr_step_profile(mnr_step_plan *p, int n_slots) {
    MNR_REQUIRE(p && n_slots >= 0 && n_slots <= 4096, "bad arguments to mnr_step_profile");
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    p->events.clear();
    p->ev_alias.clear();
    p->prof_slots = 0;
    p->prof_step = 0;
    for (int i = 0; i < n_slots * MNR_STEP_SPANS * 2; ++i) {
        hipEvent_t e;
        // (timing-only events: a default event performs a system-scope fence when it is recorded -- a cache write-back and invalidation,
        // 18 times per step, measured at 0.064 ms of a 6.1 ms step; these events are only ever read through hipEventElapsedTime)
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) {
            (void)hipGetLastError();
            if (hipEventCreate(&e) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipEventCreate failed");
        }
        p->events.push_back(e);
    }
    p->ev_alias.assign((size_t)n_slots * MNR_STEP_SPANS * 2, 0);
    p->prof_slots = n_slots;
    return MNR_OK;
}

extern "C" int mnr_step_kernel_times(mnr_step_plan *p, int slot, float *ms_out) {
    MNR_REQUIRE(p && ms_out && slot >= 0 && slot < p->prof_slots, "bad arguments to mnr_step_kernel_times");
    for (int i = 0; i < MNR_STEP_SPANS; ++i) {
        const size_t base = (size_t)slot * MNR_STEP_SPANS * 2;
        const hipEvent_t a = p->events[base + p->ev_alias[base + 2 * i]], b = p->events[base + p->ev_alias[base + 2 * i + 1]];
        if (hipEventElapsedTime(&ms_out[i], a, b) != hipSuccess) { (void)hipGetLastError(); ms_out[i] = -1.f; }
    }
    return MNR_OK;
}

// Backward of the 512-wide foreground model over the rows of one (cell, pass): the adjoint of nerf.py:115-160 layer by layer, as
// models/layerwise.py::LayerwiseTape.backward sequences it from Python -- head adjoints (k_act_grad / k_gemm / k_col_sum), the data
// gradients as tiled GEMMs with the ReLU gate and the sigma head's rank-1 term fused (k_tgemm), every layer's weight gradient as jobs of
// the batched kernel (k_wgrad2<1>, two launches per call), the appearance-embedding gradient scattered per ray -- all enqueued from here,
// on the step's own workspace, with no host read.  Gradients ACCUMULATE into the cell's gradient area (zeroed by the step's memset).
static int wide_fg_backward(mnr_step_plan *p, int c, int pass, int idx_is_float, hipStream_t s) {
    const StepDims &D = p->D;
    const StepWs &L = p->L;
    char *ws = p->ws;
    void *st = reinterpret_cast<void *>(s);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    const mnr_step_model &M = p->models[2 * c];
    const mnr_model_desc &d = M.desc;
    const mnr_model_grads &G = M.grad;
    const long S = pass ? D.Nf : D.Nc, B = D.N * S, capT = D.C * D.cap_f, t0 = (long)c * D.cap_f + (pass ? D.N * D.Nc : 0);
    constexpr int W = 512, H2 = 256, E = 75, ED = 27, A = 48, Ep = WIDE_EP, Sp = WIDE_SP;
    const TapeLayout tl = tape_layout(ArchDims{d.xyz_dim, d.pos_xyz_dim, d.pos_dir_dim, d.layers, d.skip_mask, d.layer_dim, d.appearance_dim, d.rgb_dim, d.mfma_tile});
    const float *tape = F(L.tape_f);
    auto plane = [&](int off, int width) { return tape + (long)off * capT + t0 * width; };
    const float *hs[8];
    for (int l = 0; l < 8; ++l) hs[l] = plane(tl.act_off[l], W);
    const float *fin = plane(tl.fin_off, W), *dact = plane(tl.dact_off, H2);
    const float *out = F(pass ? L.raw_f : L.raw_c) + (long)c * B * 4, *d_out = F(pass ? L.draw_f : L.draw_c) + (long)c * B * 4;
    float *emb = F(L.w_emb), *side = F(L.w_side), *g_rgb = F(L.w_grgb), *d_src = F(L.w_dsrc), *d_app = F(L.w_dapp), *g_sig = F(L.w_gsig), *d_f = F(L.w_df);
    const float *xyz = F(pass ? L.xyz_f : L.xyz_c) + (long)c * B * 3, *dirs = F(L.rays) + (long)c * D.N * 8 + 3;
    const void *idx = ws + L.idx + (size_t)c * D.N * 4;
    const float *wskip = F(L.w_wskip) + (size_t)c * W * (Ep + W), *wdir = F(L.w_wdir) + (size_t)c * H2 * (W + Sp);
    int rc;
#define WIDE_OK(call) do { if ((rc = (call)) != MNR_OK) return rc; } while (0)
    // the two zero-padded input planes the weight-gradient jobs read (pad columns were zeroed when the plan was made)
    WIDE_OK(mnr_embed(emb, Ep, xyz, 3, 3, d.pos_xyz_dim, 1, B, st));
    WIDE_OK(mnr_embed(side, Sp, dirs, 8, 3, d.pos_dir_dim, S, B, st));
    WIDE_OK(mnr_gather_rows(side + ED, Sp, d.embedding_a, A, d.appearance_count, idx, 1, idx_is_float, S, B, st));
    // weight-gradient jobs: dW[256 m .. ][col0 + 256 n ..] += dZ[:, 256 m ..]^T . X[:, 256 n ..] (+ db from the first job of an output half)
    mnr_wgrad_job jobs[MNR_WGRAD_MAX_JOBS];
    int nj = 0;
    auto flush = [&]() -> int {
        if (nj == 0) return MNR_OK;
        const int r2 = mnr_wgrad_jobs(jobs, nj, B, ws + L.w_wgws, mnr_wgrad_workspace_bytes(), st);
        nj = 0;
        return r2;
    };
    struct Part { const float *x; long ldx; int cols, col0; };
    auto wgrad = [&](float *gw, long ldw, float *gb, const float *dz, long ldz, int n_out, const Part *parts, int n_parts) -> int {
        for (int mh = 0; mh < n_out / 256; ++mh) {
            float *db = gb + 256 * mh;
            for (int q = 0; q < n_parts; ++q) {
                const Part &P = parts[q];
                const bool wide_in = P.ldx > 128;
                for (int nh = 0; nh < (wide_in ? P.cols / 256 : 1); ++nh) {
                    if (nj == MNR_WGRAD_MAX_JOBS) { const int r2 = flush(); if (r2) return r2; }
                    mnr_wgrad_job &j = jobs[nj++];
                    j.dz = dz + 256 * mh; j.ldz = ldz;
                    j.in = P.x + 256 * nh; j.ldin = P.ldx;
                    j.in_cols = wide_in ? 256 : P.cols; j.in_block = wide_in ? 256 : (int)P.ldx;
                    j.dw = gw + (long)256 * mh * ldw + P.col0 + 256 * nh; j.ldw = ldw;
                    j.db = db; db = nullptr;
                }
            }
        }
        return MNR_OK;
    };
    auto dgrad_t = [&](float *dX, const float *Gz, long ldg, int n_out, const float *wt, long ldw, int k_in, const float *gate, const float *r1_row, const float *r1_col) -> int {
        mnr_tgemm g{};
        g.a[0] = Gz; g.lda[0] = ldg; g.b[0] = wt; g.ldb[0] = ldw; g.k[0] = n_out;
        g.n_phases = 1; g.b_kslow = 1;
        g.c = dX; g.ldc = k_in; g.m = B; g.n = k_in;
        if (gate) { g.gate = gate; g.ldgate = k_in; }
        if (r1_row) { g.r1_row = r1_row; g.r1_stride = 1; g.r1_col = r1_col; }
        return mnr_tgemm_run(&g, st);
    };
    // ---- both heads in one pass over the rows: sigmoid / sigma-activation adjoints, their weight and bias gradients, the data gradient
    // through rgb.weight with the ReLU adjoint of the dir_a output (k_wide_head_adjoint; MNR_WIDE_SEPARATE_HEADS=1: the eight launches
    // of models/layerwise.py instead) ----
    static const bool separate_heads = getenv("MNR_WIDE_SEPARATE_HEADS") != nullptr;
    if (!separate_heads) {
        static const long cap = getenv("MNR_WIDE_HEAD_BLOCKS") ? atol(getenv("MNR_WIDE_HEAD_BLOCKS")) : 512;
        const long nb = (B + 255) / 256;
        hipLaunchKernelGGL(k_wide_head_adjoint, dim3((unsigned)(nb > cap ? cap : (nb < 1 ? 1 : nb))), dim3(256 * WH_G), 0, s, d_out, out, dact, hs[7], d.rgb_w, B,
                           d.sigma_activation ? 1 : 0, d_src, g_sig, G.rgb_w, G.rgb_b, G.sigma_w, G.sigma_b);
        WIDE_OK(check_launch("k_wide_head_adjoint"));
    } else {
        WIDE_OK(mnr_act_grad(g_rgb, 3, d_out, 4, out, 4, B, 3, 2, st));
        WIDE_OK(mnr_gemm(G.rgb_w, H2, g_rgb, 1, 3, dact, 1, H2, 3, H2, B, 1, 0, st));
        WIDE_OK(mnr_col_sum(G.rgb_b, g_rgb, 3, B, 3, st));
        WIDE_OK(mnr_gemm(d_src, H2, g_rgb, 3, 1, d.rgb_w, 1, H2, B, H2, 3, 0, 1, st));
        WIDE_OK(mnr_act_grad(d_src, H2, d_src, H2, dact, H2, B, H2, 1, st));
    }
    // ---- dir_a layer ----
    {
        const Part parts[2] = {{fin, W, W, 0}, {side, Sp, ED + A, W}};
        WIDE_OK(wgrad(G.dir_a_w, W + ED + A, G.dir_a_b, d_src, H2, H2, parts, 2));
    }
    WIDE_OK(mnr_gemm(d_app, A, d_src, H2, 1, d.dir_a_w + W + ED, 1, W + ED + A, B, A, H2, 0, 1, st));
    WIDE_OK(mnr_scatter_rows(G.embedding_a, A, d.appearance_count, idx, 1, idx_is_float, S, d_app, A, B, st));
    WIDE_OK(dgrad_t(d_f, d_src, H2, H2, wdir, W + Sp, W, nullptr, nullptr, nullptr));
    // ---- xyz_encoding_final + sigma head ----
    {
        const Part parts[1] = {{hs[7], W, W, 0}};
        WIDE_OK(wgrad(G.final_w, W, G.final_b, d_f, W, W, parts, 1));
    }
    if (separate_heads) {
        WIDE_OK(mnr_act_grad(g_sig, 1, d_out + 3, 4, out + 3, 4, B, 1, d.sigma_activation ? 3 : 1, st));
        WIDE_OK(mnr_gemm(G.sigma_w, W, g_sig, 1, 1, hs[7], 1, W, 1, W, B, 1, 0, st));
        WIDE_OK(mnr_col_sum(G.sigma_b, g_sig, 1, B, 1, st));
    }
    float *d_h = F(L.w_dh[0]);
    WIDE_OK(dgrad_t(d_h, d_f, W, W, d.final_w, W, W, hs[7], g_sig, d.sigma_w));
    // ---- trunk, last layer first: d_h already carries the ReLU adjoint of layer i's output ----
    for (int i = 7; i >= 0; --i) {
        const bool has_emb = i == 0 || ((d.skip_mask >> i) & 1);
        Part parts[2];
        int np = 0;
        if (has_emb) parts[np++] = Part{emb, Ep, E, 0};
        if (i > 0) parts[np++] = Part{hs[i - 1], W, W, has_emb ? E : 0};
        WIDE_OK(wgrad(G.layer_w[i], (has_emb ? E : 0) + (i > 0 ? W : 0), G.layer_b[i], d_h, W, W, parts, np));
        if (i > 0) {
            float *nxt = F(L.w_dh[8 - i]);
            const float *wt = has_emb ? wskip + Ep : d.layer_w[i];
            WIDE_OK(dgrad_t(nxt, d_h, W, W, wt, has_emb ? Ep + W : W, W, hs[i - 1], nullptr, nullptr));
            d_h = nxt;
        }
    }
    WIDE_OK(flush());
#undef WIDE_OK
    return MNR_OK;
}

extern "C" int mnr_train_step(mnr_step_plan *p, const mnr_step_batch *batches, const mnr_step_randoms *randoms, double lr, int64_t adam_step,
                              uint64_t seed, int flags, void *stream) {
    MNR_REQUIRE(p && batches && adam_step >= 1, "bad arguments to mnr_train_step");
    const StepDims &D = p->D;
    const StepWs &L = p->L;
    const int C = (int)D.C;
    char *ws = p->ws;
    hipStream_t s = as_stream(stream);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    for (int c = 0; c < C; ++c) {
        MNR_REQUIRE(batches[c].rays && batches[c].idx && (batches[c].target || (batches[c].target_u8 && batches[c].u8_table)), "cell %d: NULL batch pointer", c);
        MNR_REQUIRE(batches[c].idx_is_float == batches[0].idx_is_float, "all cells must pass image indices of the same type");
    }
    if (hipMemsetAsync(ws + L.zero_begin, 0, L.zero_end - L.zero_begin, s) != hipSuccess) return set_err(MNR_E_LAUNCH, "hipMemsetAsync(step)");
    int32_t *scal = I(L.scal);
    // profiling (mnr_step_profile): span i of this step's slot
    const long slot = p->prof_slots ? p->prof_step++ % p->prof_slots : -1;
    const size_t ev0 = slot >= 0 ? (size_t)slot * MNR_STEP_SPANS * 2 : 0;
    auto mark = [&](int span, int end) {
        if (slot < 0) return;
        p->ev_alias[ev0 + 2 * span + end] = (uint8_t)(2 * span + end);
        (void)hipEventRecord(p->events[ev0 + 2 * span + end], s);
    };
    // the end of span `a` and the beginning of span `b` with nothing enqueued between them: ONE record (every record is a packet the GPU's
    // command processor works through between two kernels)
    auto mark2 = [&](int a, int b) {
        if (slot < 0) return;
        p->ev_alias[ev0 + 2 * a + 1] = p->ev_alias[ev0 + 2 * b] = (uint8_t)(2 * b);
        (void)hipEventRecord(p->events[ev0 + 2 * b], s);
    };
    mark(0, 0);
    // ---- begin ----
    {
        BeginArgs ba{};
        for (int c = 0; c < C; ++c) ba.b[c] = batches[c];
        hipLaunchKernelGGL(k_step_begin, dim3(C), dim3(1024), 0, s, ba, D.N, p->sp, F(L.rays), reinterpret_cast<uint32_t *>(ws + L.idx), F(L.target),
                           F(L.far), F(L.last_delta), I(L.bg_slot), I(L.bg_list), F(L.rays_bg), reinterpret_cast<uint32_t *>(ws + L.idx_bg), scal, scal + MAXC);
        int rc = check_launch("k_step_begin");
        if (rc) return rc;
    }
    const bool noise = p->cfg.sigma_noise != 0, rnd_u = p->cfg.perturb > 0.f;
    // ---- coarse samples + random numbers ----
    {
        SamplesArgs a{};
        if (randoms) { for (int c = 0; c < C; ++c) a.inj[c] = randoms[c]; a.has_inj = 1; }
        a.C = D.C; a.N = D.N; a.Nc = (int)D.Nc; a.Nf = (int)D.Nf; a.Sb = (int)D.Sb; a.Sfb = (int)D.Sfb;
        a.perturb = p->cfg.perturb; a.noise = noise ? 1 : 0;
        for (int c = 0; c < C; ++c) a.cell_key[c] = batches[c].rng_cell_plus1 > 0 ? (unsigned long long)(batches[c].rng_cell_plus1 - 1) : (unsigned long long)c;
        a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.step_lo = (unsigned)adam_step; a.step_hi = (unsigned)((uint64_t)adam_step >> 32);
        a.sp = p->sp;
        a.rays = F(L.rays); a.far = F(L.far); a.rays_bg = F(L.rays_bg); a.t_c = F(L.t_c); a.t_bc = F(L.t_bc); a.scal = scal;
        a.z_c = F(L.z_c); a.xyz_c = F(L.xyz_c); a.zb_asc = F(L.zb_asc); a.zb_c = F(L.zb_c); a.pts_c = F(L.pts_c); a.dr_c = F(L.dr_c);
        a.noise_fc = F(L.noise_fc); a.noise_ff = F(L.noise_ff); a.noise_bc = F(L.noise_bc); a.noise_bf = F(L.noise_bf);
        a.u_f = F(L.u_f); a.u_b = F(L.u_b);
        const long total = D.C * D.N * (D.Nc + D.Nf);
        hipLaunchKernelGGL(k_step_samples, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        int rc = check_launch("k_step_samples");
        if (rc) return rc;
    }
    if (p->side) mark(0, 1);          // (one stream: shared with the coarse pass's opening record below)
    // ---- MLP passes: segment descriptions over the cell-major arrays ----
    const mnr_step_model &M0f = p->models[0], &M0b = p->models[1];
    const bool split = p->cfg.split_precision != 0;
    const MlpCellSeg *tabs = reinterpret_cast<const MlpCellSeg *>(ws + L.tab_cells);
    const long capT_f = D.C * D.cap_f, capT_b = D.C * D.cap_b;
    // branch: 0 = both models in one launch, 1 = foreground only, 2 = background only
    auto fwd_pass = [&](int pass, int branch, hipStream_t st) -> int {
        mnr_mlp_io io[2] = {};
        const long Sf = pass ? D.Nf : D.Nc, Sbb = pass ? D.Sfb : D.Sb;
        io[0].xyz = F(pass ? L.xyz_f : L.xyz_c); io[0].xyz_stride = 3;
        io[0].dir = F(L.rays) + 3; io[0].dir_stride = 8;
        io[0].idx = ws + L.idx; io[0].idx_stride = 1; io[0].idx_is_float = batches[0].idx_is_float;
        io[0].rows_per_ray = (int32_t)Sf;
        io[0].sigma_noise = noise ? F(pass ? L.noise_ff : L.noise_fc) : nullptr;
        io[0].out = F(pass ? L.raw_f : L.raw_c); io[0].out_stride = 4;
        io[0].n_rows = D.C * D.N * Sf; io[0].rows_per_unit = (int32_t)Sf; io[0].apply_sh_deg = D.sh_deg;
        io[1].xyz = F(pass ? L.pts_f : L.pts_c); io[1].xyz_stride = 4;
        io[1].dir = F(L.rays_bg) + 3; io[1].dir_stride = 8;
        io[1].idx = ws + L.idx_bg; io[1].idx_stride = 1; io[1].idx_is_float = batches[0].idx_is_float;
        io[1].rows_per_ray = (int32_t)Sbb;
        io[1].sigma_noise = noise ? F(pass ? L.noise_bf : L.noise_bc) : nullptr;
        io[1].out = F(pass ? L.braw_f : L.braw_c); io[1].out_stride = 4;
        io[1].n_rows = D.C * D.N * Sbb; io[1].rows_per_unit = (int32_t)Sbb; io[1].apply_sh_deg = D.sh_deg;
        mnr_mlp_launch seg[2] = {};
        seg[0].packed_dev = split ? M0f.packed_h2_dev : M0f.packed_dev; seg[0].desc = &M0f.desc; seg[0].io = &io[0];
        seg[0].tape_dev = F(L.tape_f); seg[0].tape_rows = capT_f; seg[0].tape_row0 = 0;
        seg[1].packed_dev = split ? M0b.packed_h2_dev : M0b.packed_dev; seg[1].desc = &M0b.desc; seg[1].io = &io[1];
        seg[1].tape_dev = F(L.tape_b); seg[1].tape_rows = capT_b; seg[1].tape_row0 = 0;
        const CellTable ct[2] = {{tabs + (0 + pass) * C, D.N * Sf}, {tabs + (2 + pass) * C, D.N * Sbb}};
        const int first = branch == 2 ? 1 : 0, n = branch == 0 ? 2 : 1;
        if (D.wide) {
            // 512-wide foreground: the tape-writing wavefront-pair kernel, one launch per cell (csrc/mlp_fwd_pair.hip); the 256-wide
            // background cells as the two-model kernel's background segment alone
            int rc2 = mlp_forward_multi_impl(seg + 1, 1, ct + 1, st);
            for (int c = 0; c < C && rc2 == MNR_OK; ++c) {
                const mnr_step_model &Mc = p->models[2 * c];
                mnr_mlp_io ic = io[0];
                ic.xyz = io[0].xyz + (long)c * D.N * Sf * 3; ic.dir = io[0].dir + (long)c * D.N * 8;
                ic.idx = reinterpret_cast<const char *>(io[0].idx) + (long)c * D.N * 4;
                if (ic.sigma_noise) ic.sigma_noise = io[0].sigma_noise + (long)c * D.N * Sf;
                ic.out = io[0].out + (long)c * D.N * Sf * 4; ic.n_rows = D.N * Sf;
                ModelLayout ml;
                if ((rc2 = layout_from_desc(&Mc.desc, ml)) != MNR_OK) break;
                rc2 = mlp_forward_pair_dispatch(ml, Mc.packed_dev, &Mc.desc, &ic, st, nullptr, 0, F(L.tape_f), capT_f, (long)c * D.cap_f + (pass ? D.N * D.Nc : 0));
            }
            return rc2;
        }
        return split ? mlp_forward_multi_h2_impl(seg + first, n, ct + first, st) : mlp_forward_multi_impl(seg + first, n, ct + first, st);
    };
    auto mid = [&](long unit0, long unit1, hipStream_t st) -> int {
        MidArgs a{};
        a.C = D.C; a.N = D.N; a.Nc = (int)D.Nc; a.Nf = (int)D.Nf; a.Sb = (int)D.Sb; a.Sfb = (int)D.Sfb; a.det = rnd_u ? 0 : 1;
        a.sp = p->sp;
        a.rays = F(L.rays); a.rays_bg = F(L.rays_bg); a.last_delta = F(L.last_delta); a.z_c = F(L.z_c); a.raw_c = F(L.raw_c);
        a.zb_asc = F(L.zb_asc); a.zb_c = F(L.zb_c); a.braw_c = F(L.braw_c); a.u_f = F(L.u_f); a.u_b = F(L.u_b); a.t_f = F(L.t_f); a.t_bf = F(L.t_bf);
        a.scal = scal;
        a.z_f = F(L.z_f); a.xyz_f = F(L.xyz_f); a.zb_f = F(L.zb_f); a.pts_f = F(L.pts_f); a.dr_f = F(L.dr_f);
        a.unit0 = unit0; a.unit1 = unit1;
        const size_t sh = (size_t)WPB * (4 * D.Nc + 8) * sizeof(float);
        const dim3 grid((unsigned)((unit1 - unit0 + WPB - 1) / WPB)), block(64 * WPB);
        if (D.Nc == 64) hipLaunchKernelGGL((k_step_mid<1, 1>), grid, block, sh, st, a);
        else hipLaunchKernelGGL((k_step_mid<4, 2>), grid, block, sh, st, a);
        return check_launch("k_step_mid");
    };
    int rc = MNR_OK;
    const long CN = D.C * D.N;
    if (p->side && p->fork_after_coarse) {
        // The background branch (coarse pass -> fine samples -> fine pass: 69 + 138 workgroups at the benchmark shape) beside the
        // foreground's FINE pass only: the foreground's coarse launch (1024 workgroups = two whole rounds of the 512 resident slots)
        // runs alone and ends without a partial round; the background's workgroups then share the 4.4 rounds of the fine phase
        // instead of adding a partial round to each of the two passes.
        hipStream_t s2 = p->side;
        mark(1, 0);
        if ((rc = fwd_pass(0, 1, s))) return rc;
        mark(1, 1);
        if (hipEventRecord(p->ev_fork, s) != hipSuccess || hipStreamWaitEvent(s2, p->ev_fork, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        if ((rc = fwd_pass(0, 2, s2)) || (rc = mid(CN, 2 * CN, s2)) || (rc = fwd_pass(1, 2, s2))) return rc;
        if (hipEventRecord(p->ev_join, s2) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream join failed: %s", hipGetErrorString(hipGetLastError()));
        mark(2, 0);
        if ((rc = mid(0, CN, s))) return rc;
        mark(2, 1);
        mark(3, 0);
        if ((rc = fwd_pass(1, 1, s))) return rc;
        if (hipStreamWaitEvent(s, p->ev_join, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream join failed: %s", hipGetErrorString(hipGetLastError()));
        mark(3, 1);
    } else if (p->side) {
        // two branches side by side: the background's coarse pass -> fine samples -> fine pass on the plan's own stream, forked behind the
        // sample kernel and joined in front of the ray tail; the spans below then time the FOREGROUND launches (the background runs inside them)
        hipStream_t s2 = p->side;
        if (hipEventRecord(p->ev_fork, s) != hipSuccess || hipStreamWaitEvent(s2, p->ev_fork, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        if ((rc = fwd_pass(0, 2, s2)) || (rc = mid(CN, 2 * CN, s2)) || (rc = fwd_pass(1, 2, s2))) return rc;
        if (hipEventRecord(p->ev_join, s2) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream join failed: %s", hipGetErrorString(hipGetLastError()));
        mark(1, 0);
        if ((rc = fwd_pass(0, 1, s))) return rc;
        mark(1, 1);
        mark(2, 0);
        if ((rc = mid(0, CN, s))) return rc;
        mark(2, 1);
        mark(3, 0);
        if ((rc = fwd_pass(1, 1, s))) return rc;
        mark(3, 1);
        if (hipStreamWaitEvent(s, p->ev_join, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_train_step: stream join failed: %s", hipGetErrorString(hipGetLastError()));
    } else {
        mark2(0, 1);
        if ((rc = fwd_pass(0, 0, s))) return rc;
        // ---- coarse weights -> fine samples ----
        mark2(1, 2);
        if ((rc = mid(0, 2 * CN, s))) return rc;
        mark2(2, 3);
        if ((rc = fwd_pass(1, 0, s))) return rc;
        mark(3, 1);
    }
    // ---- merge, compositing, blend, loss and their adjoints ----
    if (slot >= 0) {               // (one stream: shares the fine pass's closing record; the two-stream schedules joined in between)
        if (p->side) mark(4, 0);
        else p->ev_alias[ev0 + 2 * 4] = p->ev_alias[ev0 + 2 * 3 + 1];
    }
    {
        TailArgs a{};
        a.C = D.C; a.N = D.N; a.Nc = (int)D.Nc; a.Nf = (int)D.Nf; a.Sb = (int)D.Sb; a.Sfb = (int)D.Sfb;
        a.z_c = F(L.z_c); a.z_f = F(L.z_f); a.raw_c = F(L.raw_c); a.raw_f = F(L.raw_f); a.zb_c = F(L.zb_c); a.zb_f = F(L.zb_f);
        a.braw_c = F(L.braw_c); a.braw_f = F(L.braw_f); a.last_delta = F(L.last_delta); a.target = F(L.target); a.slot = I(L.bg_slot);
        a.draw_c = F(L.draw_c); a.draw_f = F(L.draw_f); a.bdraw_c = F(L.bdraw_c); a.bdraw_f = F(L.bdraw_f);
        a.rgb = F(L.rgb); a.depth_var = F(L.depth_var); a.bg_lambda = F(L.bg_lambda); a.loss = F(L.loss);
        const long rays = D.C * D.N;
        const size_t sh = (size_t)WPB * 3 * (D.Nc + D.Nf) * sizeof(float);
        const dim3 grid((unsigned)((rays + WPB - 1) / WPB)), block(64 * WPB);
        if (D.Nc == 64) hipLaunchKernelGGL((k_step_tail<3, 2>), grid, block, sh, s, a);
        else hipLaunchKernelGGL((k_step_tail<12, 6>), grid, block, sh, s, a);
        rc = check_launch("k_step_tail");
        if (rc) return rc;
    }
    // ---- spherical-harmonics colour head: dL/d(raw rgb) -> dL/d(dir_a output) + rgb layer gradients, every (cell, branch, pass) ----
    if (D.sh_deg >= 0) {
        const TapeLayout tlf = tape_layout(ArchDims{M0f.desc.xyz_dim, M0f.desc.pos_xyz_dim, M0f.desc.pos_dir_dim, M0f.desc.layers, M0f.desc.skip_mask,
                                                    M0f.desc.layer_dim, M0f.desc.appearance_dim, M0f.desc.rgb_dim, M0f.desc.mfma_tile});
        const TapeLayout tlb = tape_layout(ArchDims{M0b.desc.xyz_dim, M0b.desc.pos_xyz_dim, M0b.desc.pos_dir_dim, M0b.desc.layers, M0b.desc.skip_mask,
                                                    M0b.desc.layer_dim, M0b.desc.appearance_dim, M0b.desc.rgb_dim, M0b.desc.mfma_tile});
        std::vector<ShHeadJob> jobs;
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < 2; ++k)
                for (int pass = 0; pass < 2; ++pass) {
                    const mnr_step_model &M = p->models[2 * c + k];
                    const long S = k == 0 ? (pass ? D.Nf : D.Nc) : (pass ? D.Sfb : D.Sb);
                    ShHeadJob j{};
                    j.d_out = F(k ? (pass ? L.bdraw_f : L.bdraw_c) : (pass ? L.draw_f : L.draw_c));
                    j.out = F(k ? (pass ? L.braw_f : L.braw_c) : (pass ? L.raw_f : L.raw_c));
                    j.dirs = F(k ? L.rays_bg : L.rays) + 3; j.dir_stride = 8; j.rows_per_ray = (int)S; j.sh_deg = D.sh_deg;
                    j.dact = F(k ? L.tape_b : L.tape_f) + (long)(k ? tlb : tlf).dact_off * (k ? capT_b : capT_f);
                    j.dd = F(k ? (pass ? L.dd_bf : L.dd_bc) : (pass ? L.dd_ff : L.dd_fc));
                    j.rgb_w = M.desc.rgb_w; j.d_rgb_w = M.grad.rgb_w; j.d_rgb_b = M.grad.rgb_b;
                    j.out_row0 = (long)c * D.N * S;
                    j.tape_row0 = (long)c * (k ? D.cap_b : D.cap_f) + (pass ? D.N * (k ? D.Sb : D.Nc) : 0);
                    j.n_rows = D.N * S;
                    j.n_units_dev = k ? scal + c : nullptr; j.rows_per_unit = (int)S;
                    const long nb = (D.N * S + 255) / 256;           // 256 rows per block: 16 iterations of 4 wavefronts x 4 rows
                    j.n_blocks = (int)(k ? 64 : (nb > 512 ? 512 : (nb < 1 ? 1 : nb)));
                    jobs.push_back(j);
                }
        for (size_t i = 0; i < jobs.size() && rc == MNR_OK; i += SH_HEAD_MAX_JOBS)
            rc = sh_head_bwd_jobs(jobs.data() + i, (int)std::min<size_t>(SH_HEAD_MAX_JOBS, jobs.size() - i), s);
        if (rc) return rc;
    }
    mark2(4, 5);
    // ---- data-gradient chains: fg coarse, fg fine, bg coarse, bg fine (all cells each) ----
    {
        mnr_mlp_grad_io g[4] = {};
        mnr_mlp_grad_launch seg[4] = {};
        CellTable ct[4];
        for (int k = 0; k < 2; ++k)
            for (int pass = 0; pass < 2; ++pass) {
                const int i = 2 * k + pass;
                const long S = k == 0 ? (pass ? D.Nf : D.Nc) : (pass ? D.Sfb : D.Sb);
                const mnr_step_model &M = p->models[k];
                g[i].tape = F(k ? L.tape_b : L.tape_f); g[i].gtape = F(k ? L.gtape_b : L.gtape_f);
                g[i].tape_rows = k ? capT_b : capT_f; g[i].tape_row0 = 0;
                g[i].d_out = F(k ? (pass ? L.bdraw_f : L.bdraw_c) : (pass ? L.draw_f : L.draw_c)); g[i].d_out_stride = 4;
                g[i].out = F(k ? (pass ? L.braw_f : L.braw_c) : (pass ? L.raw_f : L.raw_c)); g[i].out_stride = 4;
                g[i].dheads = F(k ? L.dheads_b : L.dheads_f);
                g[i].idx = ws + (k ? L.idx_bg : L.idx); g[i].idx_stride = 1; g[i].idx_is_float = batches[0].idx_is_float;
                g[i].rows_per_ray = (int32_t)S; g[i].n_rows = D.C * D.N * S; g[i].rows_per_unit = (int32_t)S;
                g[i].grad = M.grad;
                if (D.sh_deg >= 0) g[i].dd_in = F(k ? (pass ? L.dd_bf : L.dd_bc) : (pass ? L.dd_ff : L.dd_fc));
                seg[i].packed_fwd_dev = split ? M.packed_h2_dev : M.packed_dev; seg[i].packed_bwd_dev = split ? M.packed_bwd_h2_dev : M.packed_bwd_dev;
                seg[i].desc = &M.desc; seg[i].io = &g[i];
                ct[i] = CellTable{tabs + i * C, D.N * S};
            }
        if (D.wide) {
            if ((rc = mlp_backward_chain_multi_impl(seg + 2, 2, ct + 2, s))) return rc;
            for (int c = 0; c < C; ++c)
                for (int pass = 0; pass < 2; ++pass)
                    if ((rc = wide_fg_backward(p, c, pass, batches[0].idx_is_float, s))) return rc;
        } else {
            rc = split ? mlp_backward_chain_multi_h2_impl(seg, 4, ct, s) : mlp_backward_chain_multi_impl(seg, 4, ct, s);
            if (rc) return rc;
        }
    }
    mark2(5, 6);
    // ---- head gradients: per cell one dense foreground job + two device-counted background jobs ----
    {
        const TapeLayout tlf = tape_layout(ArchDims{M0f.desc.xyz_dim, M0f.desc.pos_xyz_dim, M0f.desc.pos_dir_dim, M0f.desc.layers, M0f.desc.skip_mask,
                                                    M0f.desc.layer_dim, M0f.desc.appearance_dim, M0f.desc.rgb_dim, M0f.desc.mfma_tile});
        const TapeLayout tlb = tape_layout(ArchDims{M0b.desc.xyz_dim, M0b.desc.pos_xyz_dim, M0b.desc.pos_dir_dim, M0b.desc.layers, M0b.desc.skip_mask,
                                                    M0b.desc.layer_dim, M0b.desc.appearance_dim, M0b.desc.rgb_dim, M0b.desc.mfma_tile});
        std::vector<HeadJob> jobs;
        for (int c = 0; c < C; ++c) {
            const mnr_model_grads &Gf = p->models[2 * c].grad, &Gb = p->models[2 * c + 1].grad;
            const long bf = (D.cap_f + 767) / 768;
            if (!D.wide)      // (wide foreground: its head gradients are part of wide_fg_backward)
            jobs.push_back(HeadJob{F(L.dheads_f), F(L.tape_f) + (long)tlf.act_off[M0f.desc.layers - 1] * capT_f, F(L.tape_f) + (long)tlf.dact_off * capT_f,
                                   c * D.cap_f, D.cap_f, nullptr, 0, (int)(bf > 256 ? 256 : bf), Gf.sigma_w, Gf.sigma_b, D.sh_deg >= 0 ? nullptr : Gf.rgb_w,
                                   D.sh_deg >= 0 ? nullptr : Gf.rgb_b});
            for (int pass = 0; pass < 2; ++pass)
                jobs.push_back(HeadJob{F(L.dheads_b), F(L.tape_b) + (long)tlb.act_off[M0b.desc.layers - 1] * capT_b, F(L.tape_b) + (long)tlb.dact_off * capT_b,
                                       c * D.cap_b + (pass ? D.N * D.Sb : 0), D.N * (pass ? D.Sfb : D.Sb), scal + c, (int)(pass ? D.Sfb : D.Sb), 48,
                                       Gb.sigma_w, Gb.sigma_b, D.sh_deg >= 0 ? nullptr : Gb.rgb_w, D.sh_deg >= 0 ? nullptr : Gb.rgb_b});
        }
        for (size_t i = 0; i < jobs.size() && rc == MNR_OK; i += HEAD_MAX_JOBS)
            rc = head_grads_jobs(jobs.data() + i, (int)std::min<size_t>(HEAD_MAX_JOBS, jobs.size() - i), 256, s);
        if (rc) return rc;
    }
    mark2(6, 7);
    // ---- weight gradients, cell by cell (persistent launches: no tail to share between cells) ----
    for (int c = 0; c < C; ++c) {
        mnr_wgrad_region rg[2] = {};
        const mnr_step_model &Mf = p->models[2 * c], &Mb = p->models[2 * c + 1];
        rg[0].desc = &Mf.desc; rg[0].tape = F(L.tape_f); rg[0].gtape = F(L.gtape_f); rg[0].tape_rows = capT_f;
        rg[0].n_ranges = 1; rg[0].row0[0] = c * D.cap_f; rg[0].n_rows[0] = D.cap_f; rg[0].grad = Mf.grad;
        rg[1].desc = &Mb.desc; rg[1].tape = F(L.tape_b); rg[1].gtape = F(L.gtape_b); rg[1].tape_rows = capT_b;
        rg[1].n_ranges = 2;
        rg[1].row0[0] = c * D.cap_b; rg[1].n_rows[0] = D.N * D.Sb; rg[1].n_units_dev[0] = scal + c; rg[1].rows_per_unit[0] = (int32_t)D.Sb;
        rg[1].row0[1] = c * D.cap_b + D.N * D.Sb; rg[1].n_rows[1] = D.N * D.Sfb; rg[1].n_units_dev[1] = scal + c; rg[1].rows_per_unit[1] = (int32_t)D.Sfb;
        rg[1].grad = Mb.grad;
        int32_t *ctl = reinterpret_cast<int32_t *>(ws + L.wcount + (size_t)c * 256);
        const int32_t *zexp[2] = {ctl + 32, ctl + 48};
        rc = wgrad_regions_launch(rg + (D.wide ? 1 : 0), D.wide ? 1 : 2, ctl, reinterpret_cast<int32_t *>(ws + L.ep_job + (size_t)c * wgrad_ep_job_bytes()), F(L.slab), s,
                                  split && !getenv("MNR_STEP_F32_WGRAD") ? zexp : nullptr);
        if (rc) return rc;
    }
    if (flags & MNR_STEP_NO_OPTIMIZER) { mark(7, 1); return MNR_OK; }
    // ---- Adam (torch.optim.Adam defaults of runner.py:169-171) + re-pack ----
    mark2(7, 8);
    {
        // the hyper-parameters as the Python doubles the caller typed (0.9, 0.999, 1e-8), not as the widened floats of the cfg struct
        hipLaunchKernelGGL(k_step_adam, dim3((unsigned)p->adam_blocks), dim3(256), 0, s, reinterpret_cast<const AdamTensor *>(ws + L.tab_adam),
                           p->n_adam_tensors, as_typed(p->cfg.adam_beta1), as_typed(p->cfg.adam_beta2), as_typed(p->cfg.adam_eps), lr);
        rc = check_launch("k_step_adam");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_step_pack, dim3((unsigned)p->pack_blocks), dim3(256), 0, s, reinterpret_cast<const PackJob *>(ws + L.tab_pack),
                       p->n_pack_jobs, 1);
    rc = check_launch("k_step_pack");
    if (rc == MNR_OK) rc = wide_refresh_weights(p, s);
    mark(8, 1);
    return rc;
}

// =====================================================================================================================
// mnr_render_fwd: one inference render (rendering.py:15-173 with the evaluation flags) as six launches, stateless.
struct RenderWs {
    size_t far, last_delta, bg_slot, bg_list, rays_bg, idx_bg, rays, idx;
    size_t z_c, xyz_c, z_f, xyz_f, raw_c, raw_f, zb_asc, zb_c, pts_c, dr_c, zb_f, pts_f, dr_f, braw_c, braw_f, total;
};
static void render_layout(long N, long Nc, long Nf, RenderWs &L) {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const long Sb = Nc / 2, Sfb = Nf / 2;
    L.rays = take(N * 32); L.idx = take(N * 4);
    L.far = take(N * 4); L.last_delta = take(N * 4); L.bg_slot = take(N * 4); L.bg_list = take(N * 4); L.rays_bg = take(N * 32); L.idx_bg = take(N * 4);
    L.z_c = take(N * Nc * 4); L.xyz_c = take(N * Nc * 12); L.z_f = take(N * Nf * 4); L.xyz_f = take(N * Nf * 12);
    L.raw_c = take(N * Nc * 16); L.raw_f = take(N * Nf * 16);
    L.zb_asc = take(N * Sb * 4); L.zb_c = take(N * Sb * 4); L.pts_c = take(N * Sb * 16); L.dr_c = take(N * Sb * 4);
    L.zb_f = take(N * Sfb * 4); L.pts_f = take(N * Sfb * 16); L.dr_f = take(N * Sfb * 4); L.braw_c = take(N * Sb * 16); L.braw_f = take(N * Sfb * 16);
    L.total = off;
}
static int render_dims_ok(long N, long Nc, long Nf) {
    if (N < 1 || !((Nc == 64 && Nf == 128) || (Nc == 256 && Nf == 512)))
        return set_err(MNR_E_UNSUPPORTED, "mnr_render_fwd is instantiated for 64 + 128 and 256 + 512 samples per ray");
    return MNR_OK;
}

extern "C" size_t mnr_render_workspace_bytes(int64_t n_rays, int coarse_samples, int fine_samples) {
    if (render_dims_ok(n_rays, coarse_samples, fine_samples) != MNR_OK) return 0;
    RenderWs L;
    render_layout(n_rays, coarse_samples, fine_samples, L);
    return L.total;
}

extern "C" int mnr_mlp_forward_multi_h2(const mnr_mlp_launch *segs, int n_segs, void *stream);

// routing buffers of a routed render (mnr_render_io::route_workspace): per container the blend weights, row lists, inverse lists and the
// cells' compact outputs of ONE pass (the coarse and the fine pass follow each other and share them), the row counts, the device cell table
struct RouteWs {
    struct Part { size_t weights, lists, inverse, sub_out, blend, counts, table; long cap; } fg, bg;
    size_t exit_pts, total;
};
// ncol: columns the cells write per row (4; rgb_dim + 1 when spherical-harmonics cells are blended on their raw coefficients)
static void route_layout(long N, long Nc, long Nf, int n_cells, int ncol, RouteWs &L) {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    auto part = [&](RouteWs::Part &P, long cap) {
        P.cap = cap;
        P.weights = take((size_t)n_cells * cap * 4); P.lists = take((size_t)n_cells * cap * 4); P.inverse = take((size_t)n_cells * cap * 4);
        P.sub_out = take((size_t)n_cells * cap * ncol * 4); P.blend = take(ncol == 4 ? 0 : (size_t)cap * ncol * 4);
        P.counts = take(ROUTE_PREP_MAX * 4); P.table = take(ROUTE_PREP_MAX * sizeof(mnr_mlp_cell));
    };
    part(L.fg, N * (Nc > Nf ? Nc : Nf));
    part(L.bg, N * ((Nc > Nf ? Nc : Nf) / 2));
    L.exit_pts = take((size_t)N * 12);
    L.total = off;
}
extern "C" size_t mnr_render_route_workspace_bytes(int64_t n_rays, int coarse_samples, int fine_samples, int n_cells, int out_cols) {
    if (render_dims_ok(n_rays, coarse_samples, fine_samples) != MNR_OK || n_cells < 1 || n_cells > ROUTE_PREP_MAX || out_cols < 4 || out_cols > 64) return 0;
    RouteWs L;
    route_layout(n_rays, coarse_samples, fine_samples, n_cells, out_cols, L);
    return L.total;
}
extern "C" int mnr_sh_apply(float *out_dev, int64_t ldo, const float *coef_dev, int64_t ldc, const float *dirs_dev, int64_t dir_stride,
                            int64_t rows_per_ray, int deg, int64_t R, void *stream);
extern "C" int mnr_mlp_forward_cells_multi(const mnr_mlp_cells_launch *segs, int n_segs, void *stream);

// a side stream + fork / join events a caller may lend to mnr_render_fwd (mnr_render_io::side): the background branch then runs beside
// the foreground's passes (the step plans own theirs)
struct mnr_side {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
extern "C" int mnr_side_create(mnr_side **out) {
    MNR_REQUIRE(out, "NULL argument");
    mnr_side *sd = new mnr_side();
    if (hipStreamCreateWithFlags(&sd->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sd->join, hipEventDisableTiming) != hipSuccess) {
        if (sd->fork) (void)hipEventDestroy(sd->fork);
        if (sd->stream) (void)hipStreamDestroy(sd->stream);
        delete sd;
        return set_err(MNR_E_LAUNCH, "mnr_side_create: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = sd;
    return MNR_OK;
}
extern "C" void mnr_side_destroy(mnr_side *sd) {
    if (!sd) return;
    (void)hipEventDestroy(sd->fork);
    (void)hipEventDestroy(sd->join);
    (void)hipStreamDestroy(sd->stream);
    delete sd;
}

extern "C" int mnr_render_fwd(const mnr_render_io *r, void *stream) {
    MNR_REQUIRE(r && r->fg && r->bg && r->rays && r->idx && r->rgb && r->bg_lambda && r->n_bg && r->err &&
                r->workspace && r->t_coarse_dev && r->t_bg_coarse_dev && r->t_fine_dev && r->t_bg_fine_dev, "NULL argument to mnr_render_fwd");
    const int n_cells = r->n_cells;
    MNR_REQUIRE(n_cells >= 0 && n_cells <= ROUTE_PREP_MAX, "n_cells must be in 0..%d", ROUTE_PREP_MAX);
    if (n_cells == 0) MNR_REQUIRE(r->fg_packed && r->bg_packed, "NULL weight image");
    else MNR_REQUIRE(r->fg_cell_packed && r->bg_cell_packed && r->fg_cell_emb && r->bg_cell_emb && r->centroids_host && r->route_workspace &&
                     r->boundary_margin >= 1.f && !r->side, "routed render: cell arrays, centroids, margin >= 1, routing workspace required (one stream)");
    const long N = r->n_rays, Nc = r->coarse_samples, Nf = r->fine_samples, Sb = Nc / 2, Sfb = Nf / 2;
    int rc = render_dims_ok(N, Nc, Nf);
    if (rc != MNR_OK) return rc;
    // spherical-harmonics models (rgb_dim = 3 (deg + 1)^2 coefficients, configs/mega-nerf-sh-3: deg 2): colour epilogue inside the MLP launches
    const int sh_deg = r->fg->rgb_dim == 27 && r->bg->rgb_dim == 27 ? 2 : (r->fg->rgb_dim == 48 && r->bg->rgb_dim == 48 ? 3 : -1);
    MNR_REQUIRE(sh_deg < 0 || !r->split_precision, "no split-precision kernels for the spherical-harmonics colour head");
    RenderWs L;
    render_layout(N, Nc, Nf, L);
    MNR_REQUIRE(r->workspace_bytes >= L.total, "workspace too small: %zu < %zu", r->workspace_bytes, L.total);
    MNR_REQUIRE(r->sphere_radius[0] > 0 && r->sphere_radius[1] > 0 && r->sphere_radius[2] > 0, "sphere_radius must be positive");
    char *ws = reinterpret_cast<char *>(r->workspace);
    hipStream_t s = as_stream(stream);
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    const SSphere sp{r->sphere_center[0], r->sphere_center[1], r->sphere_center[2], r->sphere_radius[0], r->sphere_radius[1], r->sphere_radius[2]};
    {
        BeginArgs ba{};
        ba.b[0].rays = r->rays; ba.b[0].idx = r->idx; ba.b[0].idx_is_float = r->idx_is_float; ba.b[0].target = nullptr;
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1024), 0, s, ba, N, sp, F(L.rays), reinterpret_cast<uint32_t *>(ws + L.idx), (float *)nullptr,
                           F(L.far), F(L.last_delta), I(L.bg_slot), I(L.bg_list), F(L.rays_bg), reinterpret_cast<uint32_t *>(ws + L.idx_bg), r->n_bg, r->err);
        if ((rc = check_launch("k_step_begin"))) return rc;
    }
    {
        SamplesArgs a{};
        a.C = 1; a.N = N; a.Nc = (int)Nc; a.Nf = 0; a.Sb = (int)Sb; a.Sfb = 0;       // (Nf = 0: no fine-pass random streams to fill)
        a.perturb = 0.f; a.noise = 0; a.sp = sp;
        a.rays = F(L.rays); a.far = F(L.far); a.rays_bg = F(L.rays_bg); a.t_c = r->t_coarse_dev; a.t_bc = r->t_bg_coarse_dev; a.scal = r->n_bg;
        a.z_c = F(L.z_c); a.xyz_c = F(L.xyz_c); a.zb_asc = F(L.zb_asc); a.zb_c = F(L.zb_c); a.pts_c = F(L.pts_c); a.dr_c = F(L.dr_c);
        const long total = N * Nc;
        hipLaunchKernelGGL(k_step_samples, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        if ((rc = check_launch("k_step_samples"))) return rc;
    }
    RouteWs RL{};
    // spherical-harmonics cells under a soft blend write their raw rgb_dim + 1 outputs: the blend runs on coefficients (as the reference's)
    const int route_ncol = (n_cells > 0 && sh_deg >= 0 && r->boundary_margin > 1.f) ? r->fg->rgb_dim + 1 : 4;
    if (n_cells > 0) {
        MNR_REQUIRE(!r->split_precision, "routed render: fp32 kernels");
        route_layout(N, Nc, Nf, n_cells, route_ncol, RL);
        MNR_REQUIRE(r->route_workspace_bytes >= RL.total, "routing workspace too small: %zu < %zu", r->route_workspace_bytes, RL.total);
        if (!r->cluster_2d)            // (cluster_2d: the background routes per sample on o + d * depth_real, see fwd_pass)
        if ((rc = bg_exit_points_launch(F(L.rays_bg), r->n_bg, N, r->sphere_center, r->sphere_radius,
                                        reinterpret_cast<float *>(reinterpret_cast<char *>(r->route_workspace) + RL.exit_pts), s))) return rc;
    }
    auto fwd_pass = [&](int pass, int branch, hipStream_t st) -> int {
        mnr_mlp_io io[2] = {};
        const long Sf = pass ? Nf : Nc, Sbb = pass ? Sfb : Sb;
        io[0].xyz = F(pass ? L.xyz_f : L.xyz_c); io[0].xyz_stride = 3; io[0].dir = F(L.rays) + 3; io[0].dir_stride = 8;
        io[0].idx = ws + L.idx; io[0].idx_stride = 1; io[0].idx_is_float = r->idx_is_float; io[0].rows_per_ray = (int32_t)Sf;
        io[0].out = F(pass ? L.raw_f : L.raw_c); io[0].out_stride = 4; io[0].n_rows = N * Sf; io[0].rows_per_unit = (int32_t)Sf; io[0].apply_sh_deg = sh_deg;
        io[1].xyz = F(pass ? L.pts_f : L.pts_c); io[1].xyz_stride = 4; io[1].dir = F(L.rays_bg) + 3; io[1].dir_stride = 8;
        io[1].idx = ws + L.idx_bg; io[1].idx_stride = 1; io[1].idx_is_float = r->idx_is_float; io[1].rows_per_ray = (int32_t)Sbb;
        io[1].out = F(pass ? L.braw_f : L.braw_c); io[1].out_stride = 4; io[1].n_rows = N * Sbb; io[1].n_units_dev = r->n_bg;
        io[1].rows_per_unit = (int32_t)Sbb; io[1].apply_sh_deg = sh_deg;
        mnr_mlp_launch seg[2] = {};
        seg[0].packed_dev = r->fg_packed; seg[0].desc = r->fg; seg[0].io = &io[0];
        seg[1].packed_dev = r->bg_packed; seg[1].desc = r->bg; seg[1].io = &io[1];
        const int first = branch == 2 ? 1 : 0, n = branch == 0 ? 2 : 1;
        if (n_cells > 0) {
            // ---- merged containers: route -> all cells of both containers in one gather-mode launch -> blend (mega_nerf.py:19-61) ----
            MNR_REQUIRE(branch == 0, "routed render: both branches on one stream");
            char *rw = reinterpret_cast<char *>(r->route_workspace);
            auto RF = [&](size_t off) { return reinterpret_cast<float *>(rw + off); };
            auto RI = [&](size_t off) { return reinterpret_cast<int32_t *>(rw + off); };
            const long B[2] = {N * Sf, N * Sbb};
            const RouteWs::Part *P[2] = {&RL.fg, &RL.bg};
            const int ncol = route_ncol;            // 4, or rgb_dim + 1: SH cells blended on their raw coefficients
            RoutePrep prep{};
            for (int q = 0; q < 2; ++q) {
                RoutePrepSeg &g = prep.s[q];
                g.table = reinterpret_cast<mnr_mlp_cell *>(rw + P[q]->table); g.lists = RI(P[q]->lists); g.counts = RI(P[q]->counts);
                g.sub_out = RF(P[q]->sub_out); g.B = B[q]; g.n = n_cells; g.out_stride = ncol;
                for (int c = 0; c < n_cells; ++c) {
                    g.packed[c] = (q ? r->bg_cell_packed : r->fg_cell_packed)[c];
                    g.emb[c] = (q ? r->bg_cell_emb : r->fg_cell_emb)[c];
                }
            }
            int rc2 = route_prepare_launch(prep, st);
            if (rc2) return rc2;
            // foreground rows route on their own position; a background ray's rows all on its sphere-exit point -- both in one launch
            {
                RouteProblem pa{io[0].xyz, 3, B[0], nullptr, 0, 1, RF(RL.fg.weights), RI(RL.fg.lists), RI(RL.fg.counts), RI(RL.fg.inverse), nullptr, 0};
                RouteProblem pb{RF(RL.exit_pts), 3, B[1], r->n_bg, (int)Sbb, (int)Sbb, RF(RL.bg.weights), RI(RL.bg.lists), RI(RL.bg.counts), RI(RL.bg.inverse), nullptr, 0};
                if (r->cluster_2d) {
                    // `cluster_2d` containers (mega_nerf.py:16,22: distances over y, z only): a background row routes on its sample's TRUE position
                    // o + d * depth_real (rendering.py:458-461), not on the ray's sphere-exit point
                    pb.pos = F(L.rays_bg); pb.pos_stride = 8;
                    pb.ray_depth = F(pass ? L.dr_f : L.dr_c); pb.depth_flip = pass ? 0 : 1;
                }
                if ((rc2 = route2_launch(pa, pb, r->centroids_host, n_cells, r->cluster_2d ? 1 : 0, r->boundary_margin, st))) return rc2;
            }
            float *outs[2] = {io[0].out, io[1].out};
            mnr_mlp_cells_launch cl[2] = {};
            for (int q = 0; q < 2; ++q) {
                io[q].n_rows = B[q]; io[q].out = nullptr; io[q].n_units_dev = nullptr; io[q].out_stride = ncol;
                if (ncol != 4) io[q].apply_sh_deg = -1;
                cl[q].desc = q ? r->bg : r->fg; cl[q].cells_dev = reinterpret_cast<const mnr_mlp_cell *>(rw + P[q]->table); cl[q].n_cells = n_cells; cl[q].io = &io[q];
            }
            // (the smaller segment -- the background's -- first: its workgroups start at once and the foreground's fill the chip behind them)
            const mnr_mlp_cells_launch ordered[2] = {cl[1], cl[0]};
            rc2 = mnr_mlp_forward_cells_multi(ordered, 2, st);
            if (rc2 == MNR_E_UNSUPPORTED)          // (other architectures -- 512-wide cells, spherical-harmonics heads: one launch per container;
                                                   // the background's on a side stream beside the foreground's was measured: 11.84 against 11.79 ms)
                for (int q = 0; q < 2 && (q == 0 || rc2 == MNR_OK); ++q) rc2 = mnr_mlp_forward_cells(ordered[q].desc, ordered[q].cells_dev, n_cells, ordered[q].io, st);
            if (rc2) return rc2;
            const float *wts[2] = {r->boundary_margin > 1.f ? RF(RL.fg.weights) : nullptr, r->boundary_margin > 1.f ? RF(RL.bg.weights) : nullptr};
            float *dst[2] = {ncol == 4 ? outs[0] : RF(RL.fg.blend), ncol == 4 ? outs[1] : RF(RL.bg.blend)};
            {
                CombineProblem ca{dst[0], ncol, RF(RL.fg.sub_out), B[0] * ncol, ncol, RI(RL.fg.inverse), wts[0], B[0], nullptr, 0};
                CombineProblem cb{dst[1], ncol, RF(RL.bg.sub_out), B[1] * ncol, ncol, RI(RL.bg.inverse), wts[1], B[1], r->n_bg, (int)Sbb};
                if ((rc2 = combine2_launch(ca, cb, ncol, n_cells, st))) return rc2;
            }
            if (ncol != 4) {
                // eval_sh + sigmoid on the BLENDED coefficients (rendering.py:300-306 behind mega_nerf.py:45-49)
                if ((rc2 = mnr_sh_apply(outs[0], 4, dst[0], ncol, io[0].dir, io[0].dir_stride, Sf, sh_deg, B[0], st))) return rc2;
                if ((rc2 = mnr_sh_apply(outs[1], 4, dst[1], ncol, io[1].dir, io[1].dir_stride, Sbb, sh_deg, B[1], st))) return rc2;
            }
            return MNR_OK;
        }
        if (r->split_precision) return mnr_mlp_forward_multi_h2(seg + first, n, st);
        if (r->fg->layer_dim == 512 || r->bg->layer_dim == 512) {
            // 512-wide models (Building): one launch per model -- the wavefront-pair kernel for a 512-wide one, the two-model kernel
            // (with one segment) for a 256-wide background
            for (int i = first; i < first + n; ++i) {
                int rc2;
                if (seg[i].desc->layer_dim == 512) {
                    ModelLayout ml;
                    if ((rc2 = layout_from_desc(seg[i].desc, ml)) != MNR_OK) return rc2;
                    rc2 = mlp_forward_pair_dispatch(ml, seg[i].packed_dev, seg[i].desc, seg[i].io, st, nullptr, 0, nullptr, 0, 0);
                } else {
                    rc2 = mlp_forward_multi_impl(seg + i, 1, nullptr, st);
                }
                if (rc2 != MNR_OK) return rc2;
            }
            return MNR_OK;
        }
        return mlp_forward_multi_impl(seg + first, n, nullptr, st);
    };
    auto mid = [&](long unit0, long unit1, hipStream_t st) -> int {
        MidArgs a{};
        a.C = 1; a.N = N; a.Nc = (int)Nc; a.Nf = (int)Nf; a.Sb = (int)Sb; a.Sfb = (int)Sfb; a.det = 1; a.sp = sp;
        a.rays = F(L.rays); a.rays_bg = F(L.rays_bg); a.last_delta = F(L.last_delta); a.z_c = F(L.z_c); a.raw_c = F(L.raw_c);
        a.zb_asc = F(L.zb_asc); a.zb_c = F(L.zb_c); a.braw_c = F(L.braw_c); a.t_f = r->t_fine_dev; a.t_bf = r->t_bg_fine_dev; a.scal = r->n_bg;
        a.z_f = F(L.z_f); a.xyz_f = F(L.xyz_f); a.zb_f = F(L.zb_f); a.pts_f = F(L.pts_f); a.dr_f = F(L.dr_f);
        a.unit0 = unit0; a.unit1 = unit1;
        const size_t sh = (size_t)WPB * (4 * Nc + 8) * sizeof(float);
        const dim3 grid((unsigned)((unit1 - unit0 + WPB - 1) / WPB)), block(64 * WPB);
        if (Nc == 64) hipLaunchKernelGGL((k_step_mid<1, 1>), grid, block, sh, st, a);
        else hipLaunchKernelGGL((k_step_mid<4, 2>), grid, block, sh, st, a);
        return check_launch("k_step_mid");
    };
    if (r->side) {
        // the background branch (coarse pass -> fine samples -> fine pass) beside the foreground's, on the lent stream
        const mnr_side *sd = reinterpret_cast<const mnr_side *>(r->side);
        if (hipEventRecord(sd->fork, s) != hipSuccess || hipStreamWaitEvent(sd->stream, sd->fork, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_render_fwd: stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        if ((rc = fwd_pass(0, 2, sd->stream)) || (rc = mid(N, 2 * N, sd->stream)) || (rc = fwd_pass(1, 2, sd->stream))) return rc;
        if (hipEventRecord(sd->join, sd->stream) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_render_fwd: stream join failed: %s", hipGetErrorString(hipGetLastError()));
        if ((rc = fwd_pass(0, 1, s)) || (rc = mid(0, N, s)) || (rc = fwd_pass(1, 1, s))) return rc;
        if (hipStreamWaitEvent(s, sd->join, 0) != hipSuccess)
            return set_err(MNR_E_LAUNCH, "mnr_render_fwd: stream join failed: %s", hipGetErrorString(hipGetLastError()));
    } else {
        if ((rc = fwd_pass(0, 0, s)) || (rc = mid(0, 2 * N, s)) || (rc = fwd_pass(1, 0, s))) return rc;
    }
    {
        RTailArgs a{};
        a.N = N; a.Nc = (int)Nc; a.Nf = (int)Nf; a.Sb = (int)Sb; a.Sfb = (int)Sfb;
        a.z_c = F(L.z_c); a.z_f = F(L.z_f); a.raw_c = F(L.raw_c); a.raw_f = F(L.raw_f); a.zb_c = F(L.zb_c); a.zb_f = F(L.zb_f);
        a.braw_c = F(L.braw_c); a.braw_f = F(L.braw_f); a.dr_c = F(L.dr_c); a.dr_f = F(L.dr_f); a.last_delta = F(L.last_delta); a.slot = I(L.bg_slot);
        a.rgb = r->rgb; a.depth = r->depth; a.fg_rgb = r->fg_rgb; a.bg_rgb = r->bg_rgb; a.fg_depth = r->fg_depth; a.bg_depth = r->bg_depth;
        a.bg_lambda = r->bg_lambda;
        const size_t sh = (size_t)WPB * 3 * (Nc + Nf) * sizeof(float);
        const dim3 grid((unsigned)((N + WPB - 1) / WPB)), block(64 * WPB);
        if (Nc == 64) hipLaunchKernelGGL((k_render_tail<3, 2>), grid, block, sh, s, a);
        else hipLaunchKernelGGL((k_render_tail<12, 6>), grid, block, sh, s, a);
        if ((rc = check_launch("k_render_tail"))) return rc;
    }
    return MNR_OK;
}
