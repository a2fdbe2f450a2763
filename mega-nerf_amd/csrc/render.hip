// render.hip -- the non-MLP stages of rendering.render_rays (mega_nerf/rendering.py) on gfx950.
//
// Compiled with -ffp-contract=off so that sample positions / cdf values are formed with the same
// separate fp32 roundings as the torch CPU kernels of the reference (bit-exact sample indices for
// identical inputs).  Ray-parallel stages run one wavefront (64 lanes) per ray: the transmittance
// product is a wave-level inclusive scan (double precision, like the reference's CPU cumprod).
// Every stage takes an optional device-side unit count (`n_units_dev`): background-ray lists are
// compacted on the device and never synchronise with the host.
#include "common.h"
#include "route_internal.h"

namespace mnr {

static constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ long unit_limit(long n_max, const int32_t *n_dev) {
    return n_dev ? (long)(*n_dev) : n_max;
}

struct Sphere {
    float cx, cy, cz, rx, ry, rz;
    int has_radius;
};

static Sphere make_sphere(const float *c, const float *r) {
    Sphere s{0, 0, 0, 1, 1, 1, 0};
    if (r) {
        s.has_radius = 1;
        s.rx = r[0]; s.ry = r[1]; s.rz = r[2];
        if (c) { s.cx = c[0]; s.cy = c[1]; s.cz = c[2]; }
    }
    return s;
}

// Normalised ray of rendering.py:398-400 / :428-430 (centre/radius only applied when a radius is given)
__device__ __forceinline__ void normalise_ray(const Sphere &sp, const float *ray, float (&o)[3], float (&d)[3]) {
    if (sp.has_radius) {
        o[0] = (ray[0] - sp.cx) / sp.rx; o[1] = (ray[1] - sp.cy) / sp.ry; o[2] = (ray[2] - sp.cz) / sp.rz;
        d[0] = ray[3] / sp.rx; d[1] = ray[4] / sp.ry; d[2] = ray[5] / sp.rz;
    } else {
        o[0] = ray[0]; o[1] = ray[1]; o[2] = ray[2];
        d[0] = ray[3]; d[1] = ray[4]; d[2] = ray[5];
    }
}
__device__ __forceinline__ float dot3(const float (&a)[3], const float (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// rendering.py:33-45, 396-417
__global__ void k_ray_setup(const float *__restrict__ rays, long N, Sphere sp, float *__restrict__ far_out,
                            float *__restrict__ last_delta, int32_t *__restrict__ flag, int32_t *__restrict__ err) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float *ray = rays + i * 8;
    float o[3], d[3];
    normalise_ray(sp, ray, o, d);
    const float dd = dot3(d, d);
    const float d1 = -dot3(d, o) / dd;
    const float p[3] = {o[0] + d1 * d[0], o[1] + d1 * d[1], o[2] + d1 * d[2]};
    const float ray_d_cos = 1.f / sqrtf(dd);
    const float pn = dot3(p, p);
    if (pn >= 1.f) atomicOr(err, 1);
    const float d2 = sqrtf(1.f - pn) * ray_d_cos;
    const float near = ray[6], far = ray[7];
    const float fg_far = fmaxf(d1 + d2, near);
    const int has_bg = far > fg_far;
    far_out[i] = fminf(far, fg_far);
    last_delta[i] = has_bg ? fg_far : 1e10f;
    flag[i] = has_bg;
}

// Stable compaction of flagged rays by one 1024-thread workgroup (ascending ray order, like the
// boolean-mask indexing of rendering.py:37).  slot[] holds the flags on entry.
__global__ __launch_bounds__(1024) void k_compact(long N, int32_t *__restrict__ slot, int32_t *__restrict__ list,
                                                 int32_t *__restrict__ n_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (long start = 0; start < N; start += 1024) {
        const long i = start + threadIdx.x;
        const int f = (i < N) ? slot[i] : 0;
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (i < N) {
            const int k = f ? off + before : -1;
            slot[i] = k;
            if (f) list[k] = (int32_t)i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            base_s += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base_s;
}

// z of sample s with optional stratified jitter (rendering.py:472-483); zc/zl/zr = this, left, right value
__device__ __forceinline__ float perturb_z(float zc, float zl, float zr, bool first, bool last, float perturb, float rnd) {
    const float upper = last ? zc : 0.5f * (zc + zr);
    const float lower = first ? zc : 0.5f * (zl + zc);
    return lower + (upper - lower) * (perturb * rnd);
}

// rendering.py:82-87
__global__ void k_fg_samples(const float *__restrict__ rays, const float *__restrict__ far_in, long N, int S,
                             const float *__restrict__ t, float perturb, const float *__restrict__ rnd,
                             float *__restrict__ z_out, float *__restrict__ xyz_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const long r = i / S;
    const int s = (int)(i % S);
    const float *ray = rays + r * 8;
    const float near = ray[6], far = far_in ? far_in[r] : ray[7];
    float z = near * (1.f - t[s]) + far * t[s];
    if (perturb > 0.f) {
        const float zl = s > 0 ? near * (1.f - t[s - 1]) + far * t[s - 1] : z;
        const float zr = s < S - 1 ? near * (1.f - t[s + 1]) + far * t[s + 1] : z;
        z = perturb_z(z, zl, zr, s == 0, s == S - 1, perturb, rnd[i]);
    }
    z_out[i] = z;
    if (xyz_out) {
        xyz_out[3 * i + 0] = ray[0] + ray[3] * z;
        xyz_out[3 * i + 1] = ray[1] + ray[4] * z;
        xyz_out[3 * i + 2] = ray[2] + ray[5] * z;
    }
}

__global__ void k_fg_points(const float *__restrict__ rays, long N, int S, const float *__restrict__ z,
                            float *__restrict__ xyz_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const float *ray = rays + (i / S) * 8;
    const float zz = z[i];
    xyz_out[3 * i + 0] = ray[0] + ray[3] * zz;
    xyz_out[3 * i + 1] = ray[1] + ray[4] * zz;
    xyz_out[3 * i + 2] = ray[2] + ray[5] * zz;
}

// rendering.py:47-56 + _depth2pts_outside :420-469 (NeRF++ inverted sphere parametrisation)
__global__ void k_bg_samples(const float *__restrict__ rays, const int32_t *__restrict__ bg_list,
                             const int32_t *__restrict__ n_bg, long N_max, int S, const float *__restrict__ t,
                             float perturb, const float *__restrict__ rnd, const float *__restrict__ z_in, Sphere sp,
                             int include_xyz_real, int cluster_2d, float *__restrict__ z_out, float *__restrict__ pts,
                             float *__restrict__ depth_real_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nb = unit_limit(N_max, n_bg);
    if (i >= nb * S) return;
    const long k = i / S;
    const int s = (int)(i % S);
    const float *ray = rays + (long)(bg_list ? bg_list[k] : k) * 8;
    float depth;
    if (z_in) {
        depth = z_in[i];
    } else {
        depth = t[s];
        if (perturb > 0.f)
            depth = perturb_z(depth, s > 0 ? t[s - 1] : depth, s < S - 1 ? t[s + 1] : depth, s == 0, s == S - 1, perturb,
                              rnd[i]);
        if (z_out) z_out[i] = depth;
    }
    float o[3], d[3];
    normalise_ray(sp, ray, o, d);
    const float dd = dot3(d, d);
    const float d1 = -dot3(d, o) / dd;
    const float pm[3] = {o[0] + d1 * d[0], o[1] + d1 * d[1], o[2] + d1 * d[2]};
    const float pm_norm = sqrtf(dot3(pm, pm));
    const float ray_d_cos = 1.f / sqrtf(dd);
    const float d2 = sqrtf(1.f - pm_norm * pm_norm) * ray_d_cos;
    const float dsum = d1 + d2;
    const float ps[3] = {o[0] + dsum * d[0], o[1] + dsum * d[1], o[2] + dsum * d[2]};
    float ax[3] = {o[1] * ps[2] - o[2] * ps[1], o[2] * ps[0] - o[0] * ps[2], o[0] * ps[1] - o[1] * ps[0]};
    const float an = sqrtf(dot3(ax, ax)) + 1e-8f;
    ax[0] /= an; ax[1] /= an; ax[2] /= an;
    const float phi = asinf(pm_norm);
    const float theta = asinf(pm_norm * depth);
    const float ang = phi - theta;
    const float ca = cosf(ang), sa = sinf(ang);
    const float cr[3] = {ax[1] * ps[2] - ax[2] * ps[1], ax[2] * ps[0] - ax[0] * ps[2], ax[0] * ps[1] - ax[1] * ps[0]};
    const float adp = dot3(ax, ps);
    float pn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pn[c] = ps[c] * ca + cr[c] * sa + ax[c] * adp * (1.f - ca);
    const float nn = sqrtf(dot3(pn, pn));
    const float depth_real = 1.f / (depth + 1e-8f) * cosf(theta) + d1;
    depth_real_out[i] = depth_real;
    const int ncol = include_xyz_real ? 7 : 4;
    float *q = pts + i * ncol;
    if (include_xyz_real) {
        const float m = cluster_2d ? depth_real : dsum;     // rendering.py:459-464
        q[0] = ray[0] + ray[3] * m; q[1] = ray[1] + ray[4] * m; q[2] = ray[2] + ray[5] * m;
        q += 3;
    }
    q[0] = pn[0] / nn; q[1] = pn[1] / nn; q[2] = pn[2] / nn; q[3] = depth;
}

// ------------------------------------------------------------------------------------------------
// _sample_pdf / _sample_cdf (rendering.py:486-536): one wavefront per ray.
// LDS per wave: bins[nb+1], w[nb] (-> pdf), cdf[nb+1].
// The normaliser reproduces torch-CPU sum(-1) association (8 vector lanes x 4 ILP rows, scalar
// tail first, then lane partials) and the cdf its double-accumulated sequential cumsum, so that the
// searchsorted indices are bit-exact for identical inputs (DESIGN.md "bit-exact indices").
template <bool FROM_Z>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_sample_pdf(
    const float *__restrict__ bins_or_z, long bins_stride, const float *__restrict__ weights, long w_stride, long N,
    const int32_t *__restrict__ n_dev, int nb, int nf, int det, const float *__restrict__ u, float *__restrict__ samples,
    int32_t *__restrict__ inds_out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (ray >= unit_limit(N, n_dev)) return;            // whole wave exits; no block-level barriers below
    const int per_wave = 3 * nb + 8;
    float *bins = smem + wave * per_wave;               // nb + 1
    float *w = bins + nb + 1;                           // nb
    float *cdf = w + nb;                                // nb + 1
    const float *brow = bins_or_z + ray * bins_stride;
    const float *wrow = weights + ray * w_stride;
    for (int i = lane; i <= nb; i += 64)
        bins[i] = FROM_Z ? 0.5f * (brow[i] + brow[i + 1]) : brow[i];          // rendering.py:213
    for (int i = lane; i < nb; i += 64) w[i] = (FROM_Z ? wrow[i + 1] : wrow[i]) + 1e-8f;   // :215 [:,1:-1], :497
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): LDS writes visible to the wave
    // ---- normaliser in torch association order ----
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
    // ---- pdf, cdf ----
    for (int i = lane; i < nb; i += 64) w[i] = w[i] / total;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) {
        double acc = 0.0;
        cdf[0] = 0.f;
        for (int i = 0; i < nb; ++i) {
            acc += (double)w[i];
            cdf[i + 1] = (float)acc;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- inverse-cdf sampling ----
    for (int f = lane; f < nf; f += 64) {
        const float uu = det ? u[f] : u[ray * nf + f];
        int lo = 0, hi = nb + 1;                        // first index with cdf[idx] > u  (searchsorted right=True)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, nb);
        const float cb = cdf[below], ca = cdf[above];
        float denom = ca - cb;
        if (denom < 1e-8f) denom = 1.f;
        const float bb = bins[below], ba = bins[above];
        samples[ray * nf + f] = bb + (uu - cb) / denom * (ba - bb);
        if (inds_out) inds_out[ray * nf + f] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// Merge of coarse + fine samples (rendering.py:336-350): stable rank sort, one wavefront per ray.
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_merge_sorted(
    const float *__restrict__ za, const float4 *__restrict__ rawa, const float *__restrict__ dra, int Sa,
    const float *__restrict__ zb, const float4 *__restrict__ rawb, const float *__restrict__ drb, int Sb, long N,
    const int32_t *__restrict__ n_dev, int flip, float *__restrict__ z_out, float4 *__restrict__ raw_out,
    float *__restrict__ dr_out, int32_t *__restrict__ order_out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (ray >= unit_limit(N, n_dev)) return;
    const int St = Sa + Sb;
    float *key = smem + wave * St;
    for (int i = lane; i < St; i += 64) key[i] = i < Sa ? za[ray * Sa + i] : zb[ray * Sb + (i - Sa)];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int e = lane; e < St; e += 64) {
        const float ke = key[e];
        int rank = 0;
        if (flip) {
            for (int j = 0; j < St; ++j) { const float kj = key[j]; rank += (kj > ke) || (kj == ke && j < e); }
        } else {
            for (int j = 0; j < St; ++j) { const float kj = key[j]; rank += (kj < ke) || (kj == ke && j < e); }
        }
        const long o = ray * St + rank;
        z_out[o] = ke;
        if (raw_out) raw_out[o] = e < Sa ? rawa[ray * Sa + e] : rawb[ray * Sb + (e - Sa)];
        if (dr_out) dr_out[o] = e < Sa ? dra[ray * Sa + e] : drb[ray * Sb + (e - Sa)];
        if (order_out) order_out[o] = e;
    }
}

// ------------------------------------------------------------------------------------------------
// Volume compositing (rendering.py:353-393): one wavefront per ray, lane l owns the contiguous
// samples [l*E, (l+1)*E); transmittance = wave-level inclusive product scan in double.
template <class Tv>
__device__ __forceinline__ Tv wave_sum(Tv v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int E>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_composite(mnr_composite_io io) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (ray >= unit_limit(io.N, io.n_units_dev)) return;
    const int S = io.S;
    const float *z = io.z + ray * S;
    const float4 *raw = reinterpret_cast<const float4 *>(io.raw) + ray * S;
    // last delta (rendering.py:192-193 / 224-225, 203 / 235)
    float last = io.last_delta ? io.last_delta[ray] : 1e10f;
    if (io.zmax_src && last < 1e10f) {
        float m = -INFINITY;
        for (int i = lane; i < io.zmax_S; i += 64) m = fmaxf(m, io.zmax_src[ray * io.zmax_S + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        last = last - m;
    }
    float alpha[E], zz[E];
    float4 c[E];
    double prod = 1.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        alpha[e] = 0.f; zz[e] = 0.f; c[e] = make_float4(0, 0, 0, 0);
        if (k < S) {
            zz[e] = z[k];
            c[e] = raw[k];
            float delta;
            if (k == S - 1) delta = last;
            else delta = io.flip ? zz[e] - z[k + 1] : z[k + 1] - zz[e];
            alpha[e] = 1.f - expf(-delta * c[e].w);
            prod *= (double)(1.f - alpha[e] + 1e-8f);
        }
    }
    // inclusive scan of the per-lane products
    double incl = prod;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o);
        if (lane >= o) incl *= up;
    }
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0;
    const double total = __shfl(incl, 63);
    float wgt[E];
    float r = 0.f, g = 0.f, b = 0.f, dsum = 0.f;
    double run = excl;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        wgt[e] = 0.f;
        if (k < S) {
            // T_{k-1} as the reference sees it: the double running product rounded to fp32 (T_{-1} = 1)
            const float T = (float)run;
            wgt[e] = alpha[e] * T;
            run *= (double)(1.f - alpha[e] + 1e-8f);
            r += wgt[e] * c[e].x; g += wgt[e] * c[e].y; b += wgt[e] * c[e].z;
            const float dv = io.depth_real ? io.depth_real[ray * S + k] : zz[e];
            dsum += wgt[e] * dv;
            if (io.weights) io.weights[ray * S + k] = wgt[e];
        }
    }
    if (io.rgb) {
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b);
        if (lane == 0) { io.rgb[ray * 3 + 0] = r; io.rgb[ray * 3 + 1] = g; io.rgb[ray * 3 + 2] = b; }
    }
    if (io.bg_lambda && lane == 0) io.bg_lambda[ray] = (float)total;
    if (io.depth || io.depth_var) {
        dsum = wave_sum(dsum);
        if (io.depth && lane == 0) io.depth[ray] = dsum;
        if (io.depth_var) {
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float df = zz[e] - dsum;
                v += wgt[e] * (df * df);
            }
            v = wave_sum(v);
            if (lane == 0) io.depth_var[ray] = v;
        }
    }
}

// rendering.py:102-139
__global__ void k_bg_blend(float *__restrict__ rgb, float *__restrict__ depth, const float *__restrict__ lam,
                           const int32_t *__restrict__ slot, const float *__restrict__ bg_rgb,
                           const float *__restrict__ bg_depth, long N, float *__restrict__ fg_rgb_o,
                           float *__restrict__ bg_rgb_o, float *__restrict__ fg_depth_o, float *__restrict__ bg_depth_o) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = slot[i];
    const float l = lam[i];
    if (rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float f = rgb[3 * i + c];
            const float bv = k >= 0 ? bg_rgb[3 * (long)k + c] * l : 0.f;
            if (fg_rgb_o) fg_rgb_o[3 * i + c] = f;
            if (bg_rgb_o) bg_rgb_o[3 * i + c] = bv;
            rgb[3 * i + c] = f + bv;
        }
    }
    if (depth) {
        const float f = depth[i];
        const float bv = k >= 0 ? bg_depth[k] * l : 0.f;
        if (fg_depth_o) fg_depth_o[i] = f;
        if (bg_depth_o) bg_depth_o[i] = bv;
        depth[i] = f + bv;
    }
}

}  // namespace mnr

using namespace mnr;

static inline unsigned nblk(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" int mnr_ray_setup(const float *rays, int64_t N, const float *c, const float *r, float *far_out,
                             float *last_delta, int32_t *bg_list, int32_t *bg_slot, int32_t *n_bg, int32_t *err,
                             void *stream) {
    MNR_REQUIRE(rays && far_out && last_delta && bg_list && bg_slot && n_bg && err && N >= 0, "bad arguments to mnr_ray_setup");
    hipStream_t s = as_stream(stream);
    if (N > 0) {
        hipLaunchKernelGGL(k_ray_setup, dim3(nblk(N, 256)), dim3(256), 0, s, rays, (long)N, make_sphere(c, r), far_out,
                           last_delta, bg_slot, err);
        int rc = check_launch("k_ray_setup");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, (long)N, bg_slot, bg_list, n_bg);
    return check_launch("k_compact");
}

extern "C" int mnr_fg_samples(const float *rays, const float *far, int64_t N, int S, const float *t, float perturb,
                              const float *rnd, float *z_out, float *xyz_out, void *stream) {
    MNR_REQUIRE(rays && t && z_out && S > 0 && N >= 0, "bad arguments to mnr_fg_samples");
    MNR_REQUIRE(!(perturb > 0.f) || rnd, "perturb > 0 needs rand_dev");
    if (N == 0) return MNR_OK;
    hipLaunchKernelGGL(k_fg_samples, dim3(nblk((long)N * S, 256)), dim3(256), 0, as_stream(stream), rays, far, (long)N, S, t,
                       perturb, rnd, z_out, xyz_out);
    return check_launch("k_fg_samples");
}

extern "C" int mnr_fg_points(const float *rays, int64_t N, int S, const float *z, float *xyz_out, void *stream) {
    MNR_REQUIRE(rays && z && xyz_out && S > 0 && N >= 0, "bad arguments to mnr_fg_points");
    if (N == 0) return MNR_OK;
    hipLaunchKernelGGL(k_fg_points, dim3(nblk((long)N * S, 256)), dim3(256), 0, as_stream(stream), rays, (long)N, S, z, xyz_out);
    return check_launch("k_fg_points");
}

extern "C" int mnr_bg_samples(const float *rays, const int32_t *bg_list, const int32_t *n_bg, int64_t N_max, int S,
                              const float *t, float perturb, const float *rnd, const float *z_in, const float *c,
                              const float *r, int include_xyz_real, int cluster_2d, float *z_out, float *pts,
                              float *depth_real, void *stream) {
    MNR_REQUIRE(rays && pts && depth_real && S > 0 && N_max >= 0, "bad arguments to mnr_bg_samples");
    MNR_REQUIRE(z_in || t, "need either z_in_dev or t_dev");
    MNR_REQUIRE(z_in || !(perturb > 0.f) || rnd, "perturb > 0 needs rand_dev");
    if (N_max == 0) return MNR_OK;
    hipLaunchKernelGGL(k_bg_samples, dim3(nblk((long)N_max * S, 256)), dim3(256), 0, as_stream(stream), rays, bg_list, n_bg,
                       (long)N_max, S, t, perturb, rnd, z_in, make_sphere(c, r), include_xyz_real, cluster_2d, z_out, pts,
                       depth_real);
    return check_launch("k_bg_samples");
}

extern "C" int mnr_sample_pdf(const float *bins, int64_t bins_stride, const float *weights, int64_t w_stride, int64_t N,
                              const int32_t *n_dev, int nb, int nf, int det, const float *u, float *samples,
                              int32_t *inds, void *stream) {
    MNR_REQUIRE(bins && weights && u && samples && nb >= 1 && nf >= 1 && N >= 0, "bad arguments to mnr_sample_pdf");
    MNR_REQUIRE(nb < 512 * 8, "nb too large");
    if (N == 0) return MNR_OK;
    const size_t sh = (size_t)WAVES_PER_BLOCK * (3 * nb + 8) * sizeof(float);
    hipLaunchKernelGGL(k_sample_pdf<false>, dim3(nblk(N, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), sh,
                       as_stream(stream), bins, (long)bins_stride, weights, (long)w_stride, (long)N, n_dev, nb, nf, det, u,
                       samples, inds);
    return check_launch("k_sample_pdf");
}

extern "C" int mnr_sample_fine(const float *z, const float *weights, int64_t N, const int32_t *n_dev, int S, int nf,
                               int det, const float *u, float *samples, int32_t *inds, void *stream) {
    MNR_REQUIRE(z && weights && u && samples && S >= 3 && nf >= 1 && N >= 0, "bad arguments to mnr_sample_fine");
    if (N == 0) return MNR_OK;
    const int nb = S - 2;
    const size_t sh = (size_t)WAVES_PER_BLOCK * (3 * nb + 8) * sizeof(float);
    hipLaunchKernelGGL(k_sample_pdf<true>, dim3(nblk(N, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), sh, as_stream(stream),
                       z, (long)S, weights, (long)S, (long)N, n_dev, nb, nf, det, u, samples, inds);
    return check_launch("k_sample_pdf<z>");
}

extern "C" int mnr_merge_sorted(const float *za, const float *rawa, const float *dra, int Sa, const float *zb,
                                const float *rawb, const float *drb, int Sb, int64_t N, const int32_t *n_dev, int flip,
                                float *z_out, float *raw_out, float *dr_out, int32_t *order_out, void *stream) {
    MNR_REQUIRE(za && zb && z_out && Sa > 0 && Sb > 0 && N >= 0, "bad arguments to mnr_merge_sorted");
    MNR_REQUIRE(!raw_out || (rawa && rawb), "raw inputs required");
    MNR_REQUIRE(!dr_out || (dra && drb), "depth_real inputs required");
    if (N == 0) return MNR_OK;
    const size_t sh = (size_t)WAVES_PER_BLOCK * (Sa + Sb) * sizeof(float);
    hipLaunchKernelGGL(k_merge_sorted, dim3(nblk(N, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), sh, as_stream(stream), za,
                       reinterpret_cast<const float4 *>(rawa), dra, Sa, zb, reinterpret_cast<const float4 *>(rawb), drb, Sb,
                       (long)N, n_dev, flip, z_out, reinterpret_cast<float4 *>(raw_out), dr_out, order_out);
    return check_launch("k_merge_sorted");
}

extern "C" int mnr_sort_rows(const float *a, int Sa, const float *b, int Sb, int64_t N, const int32_t *n_dev, float *out,
                             void *stream) {
    return mnr_merge_sorted(a, nullptr, nullptr, Sa, b, nullptr, nullptr, Sb, N, n_dev, 0, out, nullptr, nullptr, nullptr,
                            stream);
}

extern "C" int mnr_composite(const mnr_composite_io *io, void *stream) {
    MNR_REQUIRE(io && io->z && io->raw && io->S > 0 && io->N >= 0, "bad arguments to mnr_composite");
    MNR_REQUIRE(io->S <= 64 * 16, "at most 1024 samples per ray");
    if (io->N == 0) return MNR_OK;
    const int E = (io->S + 63) / 64;
    const dim3 grid(nblk(io->N, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    hipStream_t s = as_stream(stream);
#define MNR_COMP(EE) hipLaunchKernelGGL(k_composite<EE>, grid, block, 0, s, *io)
    if (E <= 1) MNR_COMP(1);
    else if (E <= 2) MNR_COMP(2);
    else if (E <= 3) MNR_COMP(3);
    else if (E <= 4) MNR_COMP(4);
    else if (E <= 6) MNR_COMP(6);
    else if (E <= 8) MNR_COMP(8);
    else if (E <= 12) MNR_COMP(12);
    else MNR_COMP(16);
#undef MNR_COMP
    return check_launch("k_composite");
}

extern "C" int mnr_bg_blend(float *rgb, float *depth, const float *lam, const int32_t *slot, const float *bg_rgb,
                            const float *bg_depth, int64_t N, float *fg_rgb_o, float *bg_rgb_o, float *fg_depth_o,
                            float *bg_depth_o, void *stream) {
    MNR_REQUIRE(lam && slot && N >= 0, "bad arguments to mnr_bg_blend");
    MNR_REQUIRE(!rgb || bg_rgb, "bg_rgb required");
    MNR_REQUIRE(!depth || bg_depth, "bg_depth required");
    if (N == 0) return MNR_OK;
    hipLaunchKernelGGL(k_bg_blend, dim3(nblk(N, 256)), dim3(256), 0, as_stream(stream), rgb, depth, lam, slot, bg_rgb,
                       bg_depth, (long)N, fg_rgb_o, bg_rgb_o, fg_depth_o, bg_depth_o);
    return check_launch("k_bg_blend");
}

// =================================================================================================
// Backward of the rendering stages (training).  The reference gets these from autograd over
// rendering.py:353-393 (compositing), :336-350 (sort/gather) and :102-131 (fg/bg blend).
// =================================================================================================
namespace mnr {

// Compositing backward w.r.t. the raw MLP outputs, one wavefront per ray.
//   rgb = sum_k w_k c_k,  w_k = alpha_k T_{k-1},  T_k = prod_{i<=k}(1 - alpha_i + 1e-8),  lambda = T_{S-1}
//   dL/dc_k     = w_k * dL/drgb
//   dL/dalpha_k = g_k T_{k-1} - (sum_{m>k} g_m w_m + dL/dlambda * lambda) / (1 - alpha_k + 1e-8),  g_k = dL/drgb . c_k
//   dL/dsigma_k = dL/dalpha_k * delta_k * exp(-delta_k sigma_k)
template <int E>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_composite_bwd(mnr_composite_grad_io io) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ray = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (ray >= unit_limit(io.N, io.n_units_dev)) return;
    const int S = io.S;
    const float *z = io.z + ray * S;
    const float4 *raw = reinterpret_cast<const float4 *>(io.raw) + ray * S;
    float last = io.last_delta ? io.last_delta[ray] : 1e10f;
    if (io.zmax_src && last < 1e10f) {
        float m = -INFINITY;
        for (int i = lane; i < io.zmax_S; i += 64) m = fmaxf(m, io.zmax_src[ray * io.zmax_S + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        last = last - m;
    }
    const float gr = io.d_rgb[ray * 3 + 0], gg = io.d_rgb[ray * 3 + 1], gb = io.d_rgb[ray * 3 + 2];
    const float dlam = io.d_bg_lambda ? io.d_bg_lambda[ray] : 0.f;
    float alpha[E], ex[E], delta[E], tt[E];
    float4 c[E];
    double prod = 1.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        alpha[e] = 0.f; ex[e] = 1.f; delta[e] = 0.f; tt[e] = 1.f; c[e] = make_float4(0, 0, 0, 0);
        if (k < S) {
            const float zk = z[k];
            c[e] = raw[k];
            delta[e] = (k == S - 1) ? last : (io.flip ? zk - z[k + 1] : z[k + 1] - zk);
            ex[e] = expf(-delta[e] * c[e].w);
            alpha[e] = 1.f - ex[e];
            tt[e] = 1.f - alpha[e] + 1e-8f;
            prod *= (double)tt[e];
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
    const float lambda = (float)__shfl(incl, 63);
    // forward weights and per-lane sums of g_k w_k
    float T[E], w[E], gk[E];
    double run = excl;
    float gw_lane = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = lane * E + e;
        T[e] = (float)run; w[e] = 0.f; gk[e] = 0.f;
        if (k < S) {
            w[e] = alpha[e] * T[e];
            run *= (double)tt[e];
            gk[e] = gr * c[e].x + gg * c[e].y + gb * c[e].z;
            gw_lane += gk[e] * w[e];
        }
    }
    // suffix sums: sum over lanes > this lane (exclusive), then within the lane from the back
    float suf = gw_lane;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float dn = __shfl_down(suf, o);
        if (lane + o < 64) suf += dn;
    }
    float after = suf - gw_lane;                 // sum of g_m w_m over all samples in higher lanes
    float4 *dr = reinterpret_cast<float4 *>(io.d_raw) + ray * S;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const int k = lane * E + e;
        if (k < S) {
            const float dalpha = gk[e] * T[e] - (after + dlam * lambda) / tt[e];
            const float dsigma = dalpha * delta[e] * ex[e];
            dr[k] = make_float4(w[e] * gr, w[e] * gg, w[e] * gb, dsigma);
            after += gk[e] * w[e];
        }
    }
}

// scatter the merged-order gradient back to the fine / coarse arrays (inverse of k_merge_sorted)
__global__ void k_merge_bwd(const float4 *__restrict__ d_merged, const int32_t *__restrict__ order, int Sa, int Sb, long N,
                            const int32_t *__restrict__ n_dev, float4 *__restrict__ d_a, float4 *__restrict__ d_b) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int St = Sa + Sb;
    if (i >= unit_limit(N, n_dev) * St) return;
    const long ray = i / St;
    const int e = order[i];
    if (e < Sa) d_a[ray * Sa + e] = d_merged[i];
    else d_b[ray * Sb + (e - Sa)] = d_merged[i];
}

// rgb = fg + lambda * bg[slot]:  d_lambda[ray] = d_rgb . bg[slot],  d_bg[slot] = lambda * d_rgb
__global__ void k_bg_blend_bwd(const float *__restrict__ d_rgb, const float *__restrict__ lam, const int32_t *__restrict__ slot,
                               const float *__restrict__ bg_rgb, long N, float *__restrict__ d_lambda,
                               float *__restrict__ d_bg_rgb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = slot[i];
    float dl = 0.f;
    if (k >= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dl += d_rgb[3 * i + c] * bg_rgb[3 * (long)k + c];
            d_bg_rgb[3 * (long)k + c] = lam[i] * d_rgb[3 * i + c];
        }
    }
    d_lambda[i] = dl;
}

}  // namespace mnr

extern "C" int mnr_composite_backward(const mnr_composite_grad_io *io, void *stream) {
    MNR_REQUIRE(io && io->z && io->raw && io->d_rgb && io->d_raw && io->S > 0 && io->N >= 0, "bad arguments to mnr_composite_backward");
    MNR_REQUIRE(io->S <= 64 * 16, "at most 1024 samples per ray");
    if (io->N == 0) return MNR_OK;
    const int E = (io->S + 63) / 64;
    const dim3 grid(nblk(io->N, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    hipStream_t s = as_stream(stream);
#define MNR_COMPB(EE) hipLaunchKernelGGL(k_composite_bwd<EE>, grid, block, 0, s, *io)
    if (E <= 1) MNR_COMPB(1);
    else if (E <= 2) MNR_COMPB(2);
    else if (E <= 3) MNR_COMPB(3);
    else if (E <= 4) MNR_COMPB(4);
    else if (E <= 6) MNR_COMPB(6);
    else if (E <= 8) MNR_COMPB(8);
    else if (E <= 12) MNR_COMPB(12);
    else MNR_COMPB(16);
#undef MNR_COMPB
    return check_launch("k_composite_bwd");
}

extern "C" int mnr_merge_backward(const float *d_merged, const int32_t *order, int Sa, int Sb, int64_t N, const int32_t *n_dev,
                                  float *d_a, float *d_b, void *stream) {
    MNR_REQUIRE(d_merged && order && d_a && d_b && Sa > 0 && Sb > 0 && N >= 0, "bad arguments to mnr_merge_backward");
    if (N == 0) return MNR_OK;
    hipLaunchKernelGGL(k_merge_bwd, dim3(nblk((long)N * (Sa + Sb), 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(d_merged), order, Sa, Sb, (long)N, n_dev,
                       reinterpret_cast<float4 *>(d_a), reinterpret_cast<float4 *>(d_b));
    return check_launch("k_merge_bwd");
}

extern "C" int mnr_bg_blend_backward(const float *d_rgb, const float *lam, const int32_t *slot, const float *bg_rgb, int64_t N,
                                     float *d_lambda, float *d_bg_rgb, void *stream) {
    MNR_REQUIRE(d_rgb && lam && slot && bg_rgb && d_lambda && d_bg_rgb && N >= 0, "bad arguments to mnr_bg_blend_backward");
    if (N == 0) return MNR_OK;
    hipLaunchKernelGGL(k_bg_blend_bwd, dim3(nblk(N, 256)), dim3(256), 0, as_stream(stream), d_rgb, lam, slot, bg_rgb, (long)N,
                       d_lambda, d_bg_rgb);
    return check_launch("k_bg_blend_bwd");
}

// =================================================================================================
// MegaNeRF router (mega_nerf/models/mega_nerf.py:19-49): blend weights + per-cell row lists on the device.
// =================================================================================================
namespace mnr {

constexpr int ROUTE_MAX_SUB = 64;
struct Centroids {
    float c[ROUTE_MAX_SUB][3];
    int n;
};

// Block = 1024 rows (16 wavefronts).  The append to the cells' row lists is aggregated per BLOCK: pass 1 leaves every wavefront's count of
// rows routed to cell i in LDS, one thread per cell turns them into offsets and takes the block's range with ONE atomic on counts[i]
// (round 4 took one per wavefront: 1 024 same-address atomics per cell for a coarse pass -- 48-71 us of a routed render), pass 2
// writes the row ids.  `inverse` (optional) [n_sub][B]: the position of `row` in cell i's list, -1 where it was not routed there -- what
// k_route_combine needs (round 4 built it with a fill + an inversion launch per evaluation).
constexpr int ROUTE_BLOCK = 1024;
// (a device function taking the block index: k_route runs one routing problem per launch, k_route2 two side by side)
__device__ __forceinline__ void route_block(const float *__restrict__ pos, long pos_stride, long B,
                                            const int32_t *__restrict__ n_dev, int rows_per_unit, const Centroids &cen, int d0,
                                            float margin, float *__restrict__ weights, int32_t *__restrict__ lists,
                                            int32_t *__restrict__ counts, int32_t *__restrict__ inverse, int pos_rows, long blk,
                                            const float *__restrict__ ray_depth = nullptr, int depth_flip = 0) {
    __shared__ int wcnt[ROUTE_MAX_SUB][ROUTE_BLOCK / 64];
    __shared__ int base[ROUTE_MAX_SUB];
    __shared__ float4 sc[ROUTE_MAX_SUB];                                   // (c_x, c_y, c_z, |c|^2 over the clustered axes)
    const long n = n_dev ? (long)(*n_dev) * rows_per_unit : B;
    if (blk * ROUTE_BLOCK >= n) return;                                    // (uniform: the whole block is past the device-side count)
    // The centroids go through LDS once: read from the kernel-argument block inside the three loops below, every cell cost a chain of
    // dependent scalar loads -- ~15 us per launch whatever the row count (round 5 trace: 20 us for 4 416 rows and for 65 536).
    if ((int)threadIdx.x < cen.n) {
        const int i = threadIdx.x;
        float cn = 0.f;
        for (int k = d0; k < 3; ++k) cn = k == d0 ? cen.c[i][k] * cen.c[i][k] : cn + cen.c[i][k] * cen.c[i][k];
        sc[i] = make_float4(cen.c[i][0], cen.c[i][1], cen.c[i][2], cn);
    }
    __syncthreads();
    const long row = blk * ROUTE_BLOCK + threadIdx.x;
    const bool valid = row < n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float p[3] = {0.f, 0.f, 0.f};
    // (pos_rows > 1: one position per pos_rows consecutive rows -- the background rows of a ray under 3-D clustering all carry the ray's
    // sphere-exit point, rendering.py:463-464)
    const long prow = pos_rows > 1 ? row / pos_rows : row;
    if (valid) {
        p[0] = pos[prow * pos_stride]; p[1] = pos[prow * pos_stride + 1]; p[2] = pos[prow * pos_stride + 2];
        if (ray_depth) {
            // `cluster_2d` background rows: `pos` holds rays (origin, direction), the routing point is the sample's true position
            // o + d * depth_real (rendering.py:458-461; the same separately rounded product and sum as k_bg_samples' 7-column form)
            // (depth_flip: the coarse background rows reach the MLP in the flipped order of rendering.py:271-273, their depth_real is stored ascending)
            const float m = ray_depth[depth_flip ? prow * pos_rows + (pos_rows - 1 - (row - prow * pos_rows)) : row];
            p[0] = p[0] + pos[prow * pos_stride + 3] * m; p[1] = p[1] + pos[prow * pos_stride + 4] * m; p[2] = p[2] + pos[prow * pos_stride + 5] * m;
        }
    }
    // distances: torch.cdist(x[:, d0:3], centroids[:, d0:]) (mega_nerf.py:22,31).  ATen takes its MATMUL formulation whenever either side
    // has more than 25 rows (cdist mode "use_mm_for_euclid_dist_if_necessary"; _euclidean_dist): [-2x, |x|^2, 1] . [c, 1, |c|^2] as one
    // sgemm -- an fma chain in column order (checked against torch / MKL: oracle/nerf_oracle.py cdist_mm) --, clamp_min(0), sqrt.  The two
    // formulations agree to rounding near the scene, but not for the background's routing points under cluster_2d (rendering.py:459-461:
    // o + d * depth_real with depth_real up to 1e8, quirk Q2): there |x|^2 swallows the centroid terms, every cell is equally far, hard
    // routing picks cell 0 and the blend weighs all cells alike.  That is what the reference computes, so it is what is computed here.
    // (ASSUMPTION: `n` is this launch's whole row count.  The reference calls cdist once per model_chunk_size chunk (rendering.py:283-331), so a
    // ragged LAST chunk of <= 25 rows with <= 25 centroids takes the direct formula there; a render's row counts are multiples of the samples
    // per ray (>= 32), so the case cannot arise through render_rays -- only through MegaNeRF.forward on a hand-made batch of <= 25 rows,
    // where this kernel takes the direct formula too.  The fma order of the matmul form is pinned against torch CPU / MKL, the parity
    // target; cuBLAS sgemm on the reference's CUDA path may round differently.)
    const bool mm = n > 25 || cen.n > 25;
    float xn = 0.f;
    for (int k = d0; k < 3; ++k) xn = k == d0 ? p[k] * p[k] : xn + p[k] * p[k];
    auto dist = [&](int i) {
        const float4 c4 = sc[i];
        const float c[3] = {c4.x, c4.y, c4.z};
        if (!mm) {
            float s = 0.f;
            for (int k = d0; k < 3; ++k) { const float t = p[k] - c[k]; s += t * t; }
            return sqrtf(s);
        }
        float acc = 0.f;
        for (int k = d0; k < 3; ++k) acc = k == d0 ? (-2.f * p[k]) * c[k] : fmaf(-2.f * p[k], c[k], acc);
        acc = fmaf(xn, 1.f, acc);
        acc = fmaf(1.f, c4.w, acc);
        return sqrtf(fmaxf(acc, 0.f));
    };
    float dmin = INFINITY;
    int amin = 0;
    for (int i = 0; i < cen.n; ++i) {
        const float d = dist(i);
        if (d < dmin) { dmin = d; amin = i; }
    }
    float wsum = 0.f;
    if (margin > 1.f) {
        for (int i = 0; i < cen.n; ++i) {
            const float d = dist(i);
            wsum += d > margin * dmin ? 0.f : 1.f / (d + 1e-8f);
        }
    }
    // pass 1: weights out; which cells take this row (bit i of rmask); per-wavefront counts
    unsigned long long rmask = 0ull;
    for (int i = 0; i < cen.n; ++i) {
        float w;
        if (margin > 1.f) {
            const float d = dist(i);
            w = d > margin * dmin ? 0.f : (1.f / (d + 1e-8f)) / wsum;
        } else {
            w = i == amin ? 1.f : 0.f;
        }
        const bool routed = valid && w > 0.f;
        if (valid) weights[(long)i * B + row] = w;
        if (routed) rmask |= 1ull << i;
        const unsigned long long m = __ballot(routed);
        if (lane == 0) wcnt[i][wave] = __popcll(m);
    }
    __syncthreads();
    if ((int)threadIdx.x < cen.n) {
        int total = 0;
        for (int w = 0; w < ROUTE_BLOCK / 64; ++w) { const int c = wcnt[threadIdx.x][w]; wcnt[threadIdx.x][w] = total; total += c; }
        base[threadIdx.x] = total ? atomicAdd(counts + threadIdx.x, total) : 0;
    }
    __syncthreads();
    // pass 2: row ids into the block's range of every list, positions into the inverse map
    for (int i = 0; i < cen.n; ++i) {
        const bool routed = (rmask >> i) & 1ull;
        const unsigned long long m = __ballot(routed);
        const int slot = base[i] + wcnt[i][wave] + __popcll(m & ((1ull << lane) - 1ull));
        if (routed) lists[(long)i * B + slot] = (int32_t)row;
        if (inverse && valid) inverse[(long)i * B + row] = routed ? slot : -1;
    }
}
__global__ __launch_bounds__(ROUTE_BLOCK) void k_route(const float *__restrict__ pos, long pos_stride, long B,
                                                       const int32_t *__restrict__ n_dev, int rows_per_unit, Centroids cen, int d0,
                                                       float margin, float *__restrict__ weights, int32_t *__restrict__ lists,
                                                       int32_t *__restrict__ counts, int32_t *__restrict__ inverse, int pos_rows) {
    route_block(pos, pos_stride, B, n_dev, rows_per_unit, cen, d0, margin, weights, lists, counts, inverse, pos_rows, (long)blockIdx.x);
}
// two routing problems over the same centroids in one launch (the foreground and the background container of a render pass): blocks
// [0, nb_a) belong to the first
__global__ __launch_bounds__(ROUTE_BLOCK) void k_route2(RouteProblem a, RouteProblem b, Centroids cen, int d0, float margin, int nb_a) {
    if ((int)blockIdx.x < nb_a)
        route_block(a.pos, a.pos_stride, a.B, a.n_dev, a.rows_per_unit, cen, d0, margin, a.weights, a.lists, a.counts, a.inverse, a.pos_rows, (long)blockIdx.x,
                    a.ray_depth, a.depth_flip);
    else
        route_block(b.pos, b.pos_stride, b.B, b.n_dev, b.rows_per_unit, cen, d0, margin, b.weights, b.lists, b.counts, b.inverse, b.pos_rows,
                    (long)blockIdx.x - nb_a, b.ray_depth, b.depth_flip);
}

__global__ void k_route_accumulate(float *__restrict__ out, long out_stride, const float *__restrict__ sub, long sub_stride,
                                   int n_cols, const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                   const float *__restrict__ weights, int assign) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *count) return;
    const long dst = list[r];
    const float w = weights ? weights[dst] : 1.f;
    for (int c = 0; c < n_cols; ++c) {
        const float v = sub[r * sub_stride + c] * w;
        if (assign) out[dst * out_stride + c] = v;
        else out[dst * out_stride + c] += v;
    }
}

// ---- one-pass blend of all cells (replaces n_sub k_route_accumulate launches) ----------------------------------------
// pos[i][row] = index of `row` in cell i's compact list (or -1): k_route's inverse map
// out[row] = sum over the cells in index order of w_i[row] * sub_i[pos_i[row]]  (same order and roundings as applying
// k_route_accumulate cell after cell to a zeroed output: mega_nerf.py:43-49)
__device__ __forceinline__ void combine_row(float *__restrict__ out, long out_stride, const float *__restrict__ sub, long cell_stride,
                                            long sub_stride, int n_cols, const int32_t *__restrict__ pos, const float *__restrict__ weights,
                                            int n_sub, long B, const int32_t *__restrict__ n_dev, int rows_per_unit, int zero_rest, long row) {
    const long n = n_dev ? (long)(*n_dev) * rows_per_unit : B;
    if (row >= n) {
        if (zero_rest && row < B)
            for (int c = 0; c < n_cols; ++c) out[row * out_stride + c] = 0.f;
        return;
    }
    // (32 columns at a time: 49 = the raw outputs of an sh_deg 3 cell, blended on their coefficients)
    for (int c0 = 0; c0 < n_cols; c0 += 32) {
        const int nc = n_cols - c0 < 32 ? n_cols - c0 : 32;
        float acc[32];
        for (int c = 0; c < nc; ++c) acc[c] = 0.f;
        for (int i = 0; i < n_sub; ++i) {
            const int32_t p = pos[(long)i * B + row];
            if (p < 0) continue;
            const float w = weights ? weights[(long)i * B + row] : 1.f;
            const float *src = sub + i * cell_stride + p * sub_stride + c0;
            for (int c = 0; c < nc; ++c) acc[c] = acc[c] + src[c] * w;
        }
        for (int c = 0; c < nc; ++c) out[row * out_stride + c0 + c] = acc[c];
    }
}
__global__ void k_route_combine(float *__restrict__ out, long out_stride, const float *__restrict__ sub, long cell_stride,
                                long sub_stride, int n_cols, const int32_t *__restrict__ pos, const float *__restrict__ weights,
                                int n_sub, long B, const int32_t *__restrict__ n_dev, int rows_per_unit, int zero_rest) {
    combine_row(out, out_stride, sub, cell_stride, sub_stride, n_cols, pos, weights, n_sub, B, n_dev, rows_per_unit, zero_rest,
                (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// the blends of two containers in one launch: blocks [0, nb_a) belong to the first
__global__ void k_route_combine2(CombineProblem a, CombineProblem b, int n_cols, int n_sub, int nb_a) {
    if ((int)blockIdx.x < nb_a)
        combine_row(a.out, a.out_stride, a.sub, a.cell_stride, a.sub_stride, n_cols, a.pos, a.weights, n_sub, a.B, a.n_dev, a.rows_per_unit, 1,
                    (long)blockIdx.x * blockDim.x + threadIdx.x);
    else
        combine_row(b.out, b.out_stride, b.sub, b.cell_stride, b.sub_stride, n_cols, b.pos, b.weights, n_sub, b.B, b.n_dev, b.rows_per_unit, 1,
                    (long)(blockIdx.x - nb_a) * blockDim.x + threadIdx.x);
}

}  // namespace mnr

namespace mnr {
__global__ void k_zero_i32(int32_t *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
}  // namespace mnr

static int route_impl(const float *pos, int64_t pos_stride, int64_t B, const int32_t *n_dev, int rows_per_unit, const float *centroids, int n_sub,
                      int d0, float margin, float *weights, int32_t *lists, int32_t *counts, int32_t *inverse, void *stream, int pos_rows = 1,
                      bool counts_are_zero = false) {
    MNR_REQUIRE(pos && centroids && weights && lists && counts && B >= 0, "bad arguments to mnr_route");
    MNR_REQUIRE(n_sub >= 1 && n_sub <= ROUTE_MAX_SUB, "n_sub must be in 1..%d", ROUTE_MAX_SUB);
    MNR_REQUIRE(d0 == 0 || d0 == 1, "cluster_dim_start must be 0 or 1");
    MNR_REQUIRE(margin >= 1.f, "boundary_margin must be >= 1");
    hipStream_t s = as_stream(stream);
    // (a kernel, not hipMemsetAsync: the runtime's fill leaves a ~6 us bubble in front of it on the stream, four times per routed render)
    if (!counts_are_zero) {
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, s, counts, n_sub);
        int rc = check_launch("k_zero_i32");
        if (rc) return rc;
    }
    if (B == 0) return MNR_OK;
    Centroids cen;
    cen.n = n_sub;
    for (int i = 0; i < n_sub; ++i)
        for (int k = 0; k < 3; ++k) cen.c[i][k] = centroids[3 * i + k];
    hipLaunchKernelGGL(k_route, dim3(nblk(B, ROUTE_BLOCK)), dim3(ROUTE_BLOCK), 0, s, pos, (long)pos_stride, (long)B, n_dev, rows_per_unit, cen,
                       d0, margin, weights, lists, counts, inverse, pos_rows);
    return check_launch("k_route");
}

// ---- the routed render as part of mnr_render_fwd (csrc/step.hip): internal entry points ----------------------------------------------
namespace mnr {
// sphere-exit point of every compacted background ray: the routing position of ALL its rows under 3-D clustering (rendering.py:463-464:
// ray_o + ray_d * (d1 + d2), the `include_xyz_real` columns k_bg_samples writes per sample; same operations in the same order)
__global__ void k_bg_exit_points(const float *__restrict__ rays_bg, const int32_t *__restrict__ n_bg, long N_max, Sphere sp, float *__restrict__ out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= unit_limit(N_max, n_bg)) return;
    const float *ray = rays_bg + k * 8;
    float o[3], d[3];
    normalise_ray(sp, ray, o, d);
    const float dd = dot3(d, d);
    const float d1 = -dot3(d, o) / dd;
    const float pm[3] = {o[0] + d1 * d[0], o[1] + d1 * d[1], o[2] + d1 * d[2]};
    const float pm_norm = sqrtf(dot3(pm, pm));
    const float ray_d_cos = 1.f / sqrtf(dd);
    const float d2 = sqrtf(1.f - pm_norm * pm_norm) * ray_d_cos;
    const float dsum = d1 + d2;
    out[3 * k] = ray[0] + ray[3] * dsum; out[3 * k + 1] = ray[1] + ray[4] * dsum; out[3 * k + 2] = ray[2] + ray[5] * dsum;
}
int bg_exit_points_launch(const float *rays_bg, const int32_t *n_bg, long N_max, const float *c, const float *r, float *out, hipStream_t s) {
    if (N_max == 0) return MNR_OK;
    hipLaunchKernelGGL(k_bg_exit_points, dim3(nblk(N_max, 256)), dim3(256), 0, s, rays_bg, n_bg, N_max, make_sphere(c, r), out);
    return check_launch("k_bg_exit_points");
}
// zero the row counts and write the device cell tables of up to two containers' routed evaluations (one launch per render pass)
__global__ void k_route_prepare(RoutePrep a) {
    const int t = threadIdx.x;
    for (int q = 0; q < 2; ++q) {
        const RoutePrepSeg &g = a.s[q];
        if (t < g.n) {
            g.counts[t] = 0;
            mnr_mlp_cell c;
            c.packed_dev = g.packed[t]; c.embedding_a = g.emb[t]; c.row_index = g.lists + (long)t * g.B; c.count = g.counts + t;
            c.out = g.sub_out + (long)t * g.B * g.out_stride;
            g.table[t] = c;
        }
    }
}
int route_prepare_launch(const RoutePrep &a, hipStream_t s) {
    hipLaunchKernelGGL(k_route_prepare, dim3(1), dim3(64), 0, s, a);
    return check_launch("k_route_prepare");
}
int route2_launch(const RouteProblem &a, const RouteProblem &b, const float *centroids_host, int n_sub, int d0, float margin, hipStream_t s) {
    MNR_REQUIRE(n_sub >= 1 && n_sub <= ROUTE_MAX_SUB && (d0 == 0 || d0 == 1) && margin >= 1.f, "bad arguments to route2_launch");
    Centroids cen;
    cen.n = n_sub;
    for (int i = 0; i < n_sub; ++i)
        for (int k = 0; k < 3; ++k) cen.c[i][k] = centroids_host[3 * i + k];
    const int nb_a = (int)nblk(a.B, ROUTE_BLOCK), nb_b = (int)nblk(b.B, ROUTE_BLOCK);
    if (nb_a + nb_b == 0) return MNR_OK;
    hipLaunchKernelGGL(k_route2, dim3(nb_a + nb_b), dim3(ROUTE_BLOCK), 0, s, a, b, cen, d0, margin, nb_a);
    return check_launch("k_route2");
}
int combine2_launch(const CombineProblem &a, const CombineProblem &b, int n_cols, int n_sub, hipStream_t s) {
    MNR_REQUIRE(n_cols > 0 && n_cols <= 64 && n_sub >= 1 && n_sub <= ROUTE_MAX_SUB, "bad arguments to combine2_launch");
    const int nb_a = (int)nblk(a.B, 256), nb_b = (int)nblk(b.B, 256);
    if (nb_a + nb_b == 0) return MNR_OK;
    hipLaunchKernelGGL(k_route_combine2, dim3(nb_a + nb_b), dim3(256), 0, s, a, b, n_cols, n_sub, nb_a);
    return check_launch("k_route_combine2");
}
int route_launch(const float *pos, long pos_stride, int pos_rows, long B, const int32_t *n_dev, int rows_per_unit, const float *centroids_host, int n_sub,
                 int d0, float margin, float *weights, int32_t *lists, int32_t *counts, int32_t *inverse, hipStream_t s) {
    return route_impl(pos, pos_stride, B, n_dev, rows_per_unit, centroids_host, n_sub, d0, margin, weights, lists, counts, inverse, s, pos_rows, true);
}
}  // namespace mnr

extern "C" int mnr_route(const float *pos, int64_t pos_stride, int64_t B, const int32_t *n_dev, int rows_per_unit,
                         const float *centroids, int n_sub, int d0, float margin, float *weights, int32_t *lists,
                         int32_t *counts, void *stream) {
    return route_impl(pos, pos_stride, B, n_dev, rows_per_unit, centroids, n_sub, d0, margin, weights, lists, counts, nullptr, stream);
}

extern "C" int mnr_route_indexed(const float *pos, int64_t pos_stride, int64_t B, const int32_t *n_dev, int rows_per_unit,
                                 const float *centroids, int n_sub, int d0, float margin, float *weights, int32_t *lists,
                                 int32_t *counts, int32_t *inverse, void *stream) {
    MNR_REQUIRE(inverse, "bad arguments to mnr_route_indexed");
    return route_impl(pos, pos_stride, B, n_dev, rows_per_unit, centroids, n_sub, d0, margin, weights, lists, counts, inverse, stream);
}

extern "C" int mnr_route_accumulate(float *out, int64_t out_stride, const float *sub, int64_t sub_stride, int n_cols,
                                    const int32_t *list, const int32_t *count, int64_t B_max, const float *weights,
                                    int assign, void *stream) {
    MNR_REQUIRE(out && sub && list && count && n_cols > 0 && B_max >= 0, "bad arguments to mnr_route_accumulate");
    if (B_max == 0) return MNR_OK;
    hipLaunchKernelGGL(k_route_accumulate, dim3(nblk(B_max, 256)), dim3(256), 0, as_stream(stream), out, (long)out_stride, sub,
                       (long)sub_stride, n_cols, list, count, weights, assign);
    return check_launch("k_route_accumulate");
}

extern "C" int mnr_route_combine_indexed(float *out, int64_t out_stride, const float *sub_all, int64_t cell_stride, int64_t sub_stride,
                                         int n_cols, const int32_t *inverse, const float *weights, int n_sub, int64_t B,
                                         const int32_t *n_dev, int rows_per_unit, void *stream) {
    MNR_REQUIRE(out && sub_all && inverse && n_cols > 0 && n_cols <= 64 && B >= 0, "bad arguments to mnr_route_combine_indexed");
    MNR_REQUIRE(n_sub >= 1 && n_sub <= ROUTE_MAX_SUB, "n_sub must be in 1..%d", ROUTE_MAX_SUB);
    if (B == 0) return MNR_OK;
    hipLaunchKernelGGL(k_route_combine, dim3(nblk(B, 256)), dim3(256), 0, as_stream(stream), out, (long)out_stride, sub_all, (long)cell_stride,
                       (long)sub_stride, n_cols, inverse, weights, n_sub, (long)B, n_dev, rows_per_unit, 1);
    return check_launch("k_route_combine");
}
