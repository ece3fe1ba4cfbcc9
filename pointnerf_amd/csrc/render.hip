// render.hip -- ray-dist + alpha-composite volume renderer (forward and backward) and the C-ABI entry
// points of the fused render path.
//
// Replaces the ray_dist block of NeuralPointsRayMarching.forward
// (models/neural_points_volumetric_model.py:271-279) and ray_march
// (models/rendering/diff_ray_marching.py:508-554; radiance_render / alpha_blend / no_tone_map,
// models/rendering/diff_render_func.py:36-62).  One 64-lane wavefront per ray: the cummax over the
// perspective depth and the exclusive transmittance product are wave scans (shuffle), no intermediate
// [R,SR] tensors of the reference (sigma, opacity, cumprod, blend weight, ...) are materialised except
// the ones the caller asks for.
#include "mlp_common.h"

int pn_agg_forward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                          const float *d_raydir, const float *d_sample_loc, const float *d_xyz_pers, const float *d_loc_pers,
                          const int32_t *d_sample_pidx,
                          const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                          float *d_decoded, float *d_weight, const PnSaved &sv, long long cap_samples, bool train, bool save_x0,
                          hipStream_t s);
int pn_agg_backward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                           const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                           const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                           const float *d_decoded, const float *d_weight, const float *d_grad_decoded,
                           const PnSaved &sv, long long n_valid, float *d_grad_params, const pnerf_point_grads *pg,
                           float *d_partials, bool x0_saved, hipStream_t s);
size_t pn_wgrad_partials_bytes();

namespace {
constexpr int TPB = 256;

struct RmArgs {
    pnerf_camera cam;
    const float *sample_loc, *decoded;
    const int *nn;
    const float *ray_dist;            // stand-alone ray_march(): distances and validity supplied by the caller
    const unsigned char *valid8;
    int R, SR;
};

__device__ __forceinline__ float wave_incl_max(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { float y = __shfl_up(v, off, 64); if (lane >= off) v = fmaxf(v, y); }
    return v;
}
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { float y = __shfl_up(v, off, 64); if (lane >= off) v *= y; }
    return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { float y = __shfl_up(v, off, 64); if (lane >= off) v += y; }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// perspective depth of a world position: ((p - campos) @ camrot)[2]   (point_query.py:101-108)
__device__ __forceinline__ float pers_z(const pnerf_camera &c, const float *p) {
    return (p[0] - c.campos[0]) * c.camrot[2] + (p[1] - c.campos[1]) * c.camrot[5] + (p[2] - c.campos[2]) * c.camrot[8];
}

// Per-sample quantities of one 64-sample chunk of a ray.  `cm_prev` is the running cummax before the
// chunk.  Returns delta (ray_dist), valid flag, and updates cm_prev.
__device__ __forceinline__ void chunk_raydist(const RmArgs &a, int r, int s, int lane, float &cm_prev, float &delta, bool &valid) {
    const int SR = a.SR;
    if (a.ray_dist) {
        valid = s < SR && a.valid8[(long long)r * SR + s] != 0;
        delta = s < SR ? a.ray_dist[(long long)r * SR + s] : 0.f;
        return;
    }
    float z = -INFINITY, znext = -INFINITY;
    if (s < SR) z = pers_z(a.cam, a.sample_loc + ((long long)r * SR + s) * 3);
    if (s + 1 < SR) znext = pers_z(a.cam, a.sample_loc + ((long long)r * SR + s + 1) * 3);
    float cm = fmaxf(wave_incl_max(z, lane), cm_prev);             // torch.cummax(sample_loc[...,2])  :271
    const float vs = a.cam.vsize_z;
    float d = (s + 1 < SR) ? fmaxf(cm, znext) - cm : vs;            // cm[s+1] - cm[s]; last = vsize[2]   :272
    bool m = d < 1e-8f;
    if (a.cam.raydist_mode_unit > 0) m = m || (d > 2.f * vs);       // :274-276
    if (m) d = vs;                                                  // :278
    valid = s < SR && a.nn[(long long)r * SR + s] > 0;              // ray_valid = any_K(mask)  point_aggregators.py:741
    delta = valid ? d : 0.f;                                        // :279
    cm_prev = __shfl(cm, 63, 64);
}

__global__ __launch_bounds__(TPB) void k_raymarch_forward(RmArgs a, float *__restrict__ ray_color, float *__restrict__ opacity_out,
                                                          float *__restrict__ bg_trans, float *__restrict__ blend_w,
                                                          float *__restrict__ acc_trans) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int SR = a.SR;
    float cm_prev = -INFINITY, T_carry = 1.f;
    float cr = 0.f, cg = 0.f, cb = 0.f;
    for (int base = 0; base < SR; base += 64) {
        const int s = base + lane;
        float delta; bool valid;
        chunk_raydist(a, r, s, lane, cm_prev, delta, valid);
        float sigma = 0.f, rr = 0.f, gg = 0.f, bb = 0.f;
        if (s < SR) {
            const float4 f = *reinterpret_cast<const float4 *>(a.decoded + ((long long)r * SR + s) * 4);
            sigma = valid ? f.x : 0.f; rr = f.y; gg = f.z; bb = f.w;
        }
        const float op = 1.f - expf(-sigma * delta);                    // diff_ray_marching.py:530
        const float u = s < SR ? (1.f - op + 1e-10f) : 1.f;             // :533
        const float incl = wave_incl_prod(u, lane);
        float Tx = __shfl_up(incl, 1, 64);                              // exclusive product by lane shift   :533-539
        if (lane == 0) Tx = 1.f;
        Tx *= T_carry;
        const float bw = op * Tx;                                       // alpha_blend
        if (s < SR) {
            opacity_out[(long long)r * SR + s] = op;
            blend_w[(long long)r * SR + s] = bw;
            if (acc_trans) acc_trans[(long long)r * SR + s] = Tx;
        }
        cr += bw * rr; cg += bw * gg; cb += bw * bb;
        T_carry *= __shfl(incl, 63, 64);
    }
    cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb);
    if (lane == 0) {
        if (a.cam.has_bg) { cr += a.cam.bg[0] * T_carry; cg += a.cam.bg[1] * T_carry; cb += a.cam.bg[2] * T_carry; }   // :543-545
        ray_color[3 * r] = cr; ray_color[3 * r + 1] = cg; ray_color[3 * r + 2] = cb;
        bg_trans[r] = T_carry;
    }
}

// Backward: dL/d(ray_color) -> dL/d(decoded) = (d sigma, d r, d g, d b) per sample.
//   color = sum_s bw_s rgb_s + bg T_end ;  bw_s = op_s T_s ; T_s = prod_{j<s} u_j ; u = 1 - op + eps ; op = 1 - exp(-sigma delta)
//   d color / d op_s = T_s rgb_s - (sum_{j>s} bw_j rgb_j + T_end bg) / u_s
__global__ __launch_bounds__(TPB) void k_raymarch_backward(RmArgs a, const float *__restrict__ grad_color, float *__restrict__ grad_decoded) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int SR = a.SR;
    const float g0 = grad_color[3 * r], g1 = grad_color[3 * r + 1], g2 = grad_color[3 * r + 2];
    // pass 1: T_end and total = sum_j bw_j (g . rgb_j)
    float cm_prev = -INFINITY, T_carry = 1.f, tot = 0.f;
    for (int base = 0; base < SR; base += 64) {
        const int s = base + lane;
        float delta; bool valid;
        chunk_raydist(a, r, s, lane, cm_prev, delta, valid);
        float sigma = 0.f, c = 0.f;
        if (s < SR) {
            const float4 f = *reinterpret_cast<const float4 *>(a.decoded + ((long long)r * SR + s) * 4);
            sigma = valid ? f.x : 0.f; c = g0 * f.y + g1 * f.z + g2 * f.w;
        }
        const float op = 1.f - expf(-sigma * delta);
        const float u = s < SR ? (1.f - op + 1e-10f) : 1.f;
        const float incl = wave_incl_prod(u, lane);
        float Tx = __shfl_up(incl, 1, 64);
        if (lane == 0) Tx = 1.f;
        Tx *= T_carry;
        tot += op * Tx * c;
        T_carry *= __shfl(incl, 63, 64);
    }
    tot = wave_sum(tot);
    const float T_end = T_carry;
    const float gbg = a.cam.has_bg ? (g0 * a.cam.bg[0] + g1 * a.cam.bg[1] + g2 * a.cam.bg[2]) * T_end : 0.f;
    // pass 2: per-sample gradients with the running prefix of bw_j c_j
    cm_prev = -INFINITY; T_carry = 1.f;
    float pre_carry = 0.f;
    for (int base = 0; base < SR; base += 64) {
        const int s = base + lane;
        float delta; bool valid;
        chunk_raydist(a, r, s, lane, cm_prev, delta, valid);
        float sigma = 0.f, c = 0.f;
        if (s < SR) {
            const float4 f = *reinterpret_cast<const float4 *>(a.decoded + ((long long)r * SR + s) * 4);
            sigma = valid ? f.x : 0.f; c = g0 * f.y + g1 * f.z + g2 * f.w;
        }
        const float e = expf(-sigma * delta);
        const float op = 1.f - e;
        const float u = s < SR ? (1.f - op + 1e-10f) : 1.f;
        const float incl = wave_incl_prod(u, lane);
        float Tx = __shfl_up(incl, 1, 64);
        if (lane == 0) Tx = 1.f;
        Tx *= T_carry;
        const float bwc = op * Tx * c;
        const float incl_sum = wave_incl_sum(bwc, lane) + pre_carry;      // sum_{j<=s}
        const float suffix = tot - incl_sum + gbg;                        // sum_{j>s} bw_j c_j + T_end (g.bg)
        const float dop = Tx * c - suffix / u;
        if (s < SR) {
            const float bw = op * Tx;
            float4 o;
            o.x = valid ? dop * delta * e : 0.f;                          // d op / d sigma = delta exp(-sigma delta)
            o.y = bw * g0; o.z = bw * g1; o.w = bw * g2;
            *reinterpret_cast<float4 *>(grad_decoded + ((long long)r * SR + s) * 4) = o;
        }
        pre_carry = __shfl(incl_sum, 63, 64);
        T_carry *= __shfl(incl, 63, 64);
    }
}

// ---- row gather / scatter-add (NeuralPoints.forward's index_select and its backward) -----------
__global__ __launch_bounds__(TPB) void k_gather_rows(const float *__restrict__ src, int n_src, int width, const int *__restrict__ idx,
                                                     long long n_idx, float *__restrict__ dst) {
    const long long total = n_idx * width;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long i = e / width;
        const int c = (int)(e - i * width);
        int p = idx[i];
        p = p < 0 ? 0 : (p >= n_src ? n_src - 1 : p);         // torch.clamp(sample_pidx, min=0)  neural_points.py:708
        dst[e] = src[(long long)p * width + c];
    }
}
// width a multiple of 4 with width / 4 a power of two <= 64 (the embedding: 32 floats = 8 lanes x 16 bytes per row): a group of W4 lanes copies one
// row with 16-byte accesses, no per-element division.  (Round 3's 4.2 TB/s for this kernel came from the loop vectoriser, which the library has
// had switched off since round 4 -- the packed-fp32 fault, csrc/Makefile --: the scalar form above fell to 3.1 TB/s; profiles/r0[345]_microbench.json.)
template <int W4>
__global__ __launch_bounds__(TPB) void k_gather_rows_v4(const float4 *__restrict__ src, int n_src, const int *__restrict__ idx, long long n_idx, float4 *__restrict__ dst) {
    const int c = threadIdx.x & (W4 - 1);
    for (long long i = ((long long)blockIdx.x * TPB + threadIdx.x) / W4; i < n_idx; i += (long long)gridDim.x * (TPB / W4)) {
        int p = idx[i];
        p = p < 0 ? 0 : (p >= n_src ? n_src - 1 : p);
        const float4 v = src[(long long)p * W4 + c];
        pn_f4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<pn_f4 *>(dst + i * W4 + c));
    }
}
// -1 slots were gathered from point 0 (A.9 of SURVEY.md: the reference lets point 0 collect gradient from every
// empty slot).  Those are the vast majority of the slots, all hitting ONE row: they are summed per block in LDS
// first and leave as one atomic per column per block; real indices go straight to global atomics.
__global__ __launch_bounds__(TPB) void k_scatter_add_rows(const float *__restrict__ grad_rows, const int *__restrict__ idx, long long n_idx,
                                                          int width, float *__restrict__ grad_src, int n_src) {
    __shared__ float row0[64];
    const bool use_lds = width <= 64;
    if (use_lds && threadIdx.x < 64) row0[threadIdx.x] = 0.f;
    __syncthreads();
    const long long total = n_idx * width;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < total; e += (long long)gridDim.x * TPB) {
        const long long i = e / width;
        const int c = (int)(e - i * width);
        int p = idx[i];
        const float g = grad_rows[e];
        if (p <= 0 && use_lds) { atomicAdd(&row0[c], g); continue; }
        p = p < 0 ? 0 : (p >= n_src ? n_src - 1 : p);
        atomicAdd(&grad_src[(long long)p * width + c], g);
    }
    __syncthreads();
    if (use_lds && threadIdx.x < width && row0[threadIdx.x] != 0.f) atomicAdd(&grad_src[threadIdx.x], row0[threadIdx.x]);
}

// ---- the zero-one regulariser on the confidences of the hit rays' neighbor slots, fused --------------------------------------------
// Reference: conf_coefficient = gradient_clamp(points_conf[clamp(sample_pidx, min=0)], 1e-4, 1) (point_aggregators.py:722-724, 812) and
// loss_zero_one = mean(log(v) + log(1 - v)), v = clamp(conf_coefficient, eps, 1 - eps) (base_rendering_model.py:630-641): a gather, two
// clamps, two logs, a reduction and their autograd mirror over [R'', SR, K] elements (84 M at configs[3]) -- ~20 element-wise passes
// in ATen.  Here: one pass that sums the terms per block (the caller adds the per-block partials: deterministic), one pass that
// adds  g * (1 / v - 1 / (1 - v)) [eps <= c' <= 1 - eps]  into the confidence gradient, with the point-0 flood of the empty slots
//  reduced per block first.
__device__ __forceinline__ float pn_zero_one_value(const float *__restrict__ conf, int n, int p, float eps, bool &inside) {
    p = p < 0 ? 0 : (p >= n ? n - 1 : p);
    const float c = fminf(fmaxf(conf[p], 1e-4f), 1.0f);         // gradient_clamp: clamp forward, identity backward
    inside = c >= eps && c <= 1.f - eps;                        // torch.clamp's backward mask (bounds included)
    return fminf(fmaxf(c, eps), 1.f - eps);
}
__global__ __launch_bounds__(TPB) void k_zero_one_forward(const float *__restrict__ conf, int n, const int *__restrict__ idx, long long n_idx, float eps,
                                                          float *__restrict__ partial) {
    __shared__ float red[TPB / 64];
    float acc = 0.f;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < n_idx; e += (long long)gridDim.x * TPB) {
        bool inside;
        const float v = pn_zero_one_value(conf, n, idx[e], eps, inside);
        acc += logf(v) + logf(1.f - v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(TPB) void k_zero_one_backward(const float *__restrict__ conf, int n, const int *__restrict__ idx, long long n_idx, float eps,
                                                           const float *__restrict__ gscale, float *__restrict__ grad_conf) {
    __shared__ float row0;
    if (threadIdx.x == 0) row0 = 0.f;
    __syncthreads();
    const float gs = gscale[0];
    float mine0 = 0.f;
    for (long long e = (long long)blockIdx.x * TPB + threadIdx.x; e < n_idx; e += (long long)gridDim.x * TPB) {
        const int p = idx[e];
        bool inside;
        const float v = pn_zero_one_value(conf, n, p, eps, inside);
        if (!inside) continue;
        const float g = gs * (1.f / v - 1.f / (1.f - v));
        if (p <= 0) mine0 += g;
        else atomicAdd(&grad_conf[p >= n ? n - 1 : p], g);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine0 += __shfl_xor(mine0, off, 64);
    if ((threadIdx.x & 63) == 0 && mine0 != 0.f) atomicAdd(&row0, mine0);
    __syncthreads();
    if (threadIdx.x == 0 && row0 != 0.f) atomicAdd(&grad_conf[0], row0);
}
// the same two passes over the DENSE neighbor table [R][slots] with the per-ray hit flags: only rays that hit the cloud count (the reference
// forms conf_coefficient for the R'' hit rays only).  One workgroup per ray at a time: no [R'', SR, K] copy of the table is ever made.
__global__ __launch_bounds__(TPB) void k_zero_one_forward_rays(const float *__restrict__ conf, int n, const int *__restrict__ idx, const int *__restrict__ hit, int R, int slots,
                                                               float eps, float *__restrict__ partial) {
    __shared__ float red[TPB / 64];
    float acc = 0.f;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        if (hit[r] <= 0) continue;
        const int *row = idx + (long long)r * slots;
        for (int e = threadIdx.x; e < slots; e += TPB) {
            bool inside;
            const float v = pn_zero_one_value(conf, n, row[e], eps, inside);
            acc += logf(v) + logf(1.f - v);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        partial[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(TPB) void k_zero_one_backward_rays(const float *__restrict__ conf, int n, const int *__restrict__ idx, const int *__restrict__ hit, int R, int slots,
                                                                float eps, const float *__restrict__ gscale, float *__restrict__ grad_conf) {
    __shared__ float row0;
    if (threadIdx.x == 0) row0 = 0.f;
    __syncthreads();
    const float gs = gscale[0];
    float mine0 = 0.f;
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        if (hit[r] <= 0) continue;
        const int *row = idx + (long long)r * slots;
        for (int e = threadIdx.x; e < slots; e += TPB) {
            const int p = row[e];
            bool inside;
            const float v = pn_zero_one_value(conf, n, p, eps, inside);
            if (!inside) continue;
            const float g = gs * (1.f / v - 1.f / (1.f - v));
            if (p <= 0) mine0 += g;
            else atomicAdd(&grad_conf[p >= n ? n - 1 : p], g);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine0 += __shfl_xor(mine0, off, 64);
    if ((threadIdx.x & 63) == 0 && mine0 != 0.f) atomicAdd(&row0, mine0);
    __syncthreads();
    if (threadIdx.x == 0 && row0 != 0.f) atomicAdd(&grad_conf[0], row0);
}
}  // namespace

namespace {
// colour term of the training loss over the DENSE ray colours with the rays' hit flags (models/base_rendering_model.py:543-551:
// ray_masked_coarse_raycolor = sum over the hit rays' (colour - gt)^2, the caller divides by its global element count): no compaction of the
// hit rays (argsort + index_selects) on the way to a scalar.  partial[b] = block sums; backward writes d colour for EVERY ray (0 for a miss),
// which is what the renderer's backward reads.
__global__ __launch_bounds__(256) void k_color_loss_forward_rays(const float *__restrict__ col, const float *__restrict__ gt, const int *__restrict__ hit, int R,
                                                                 float *__restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256)
        if (hit[r] > 0) {
            const float dx = col[3 * r] - gt[3 * r], dy = col[3 * r + 1] - gt[3 * r + 1], dz = col[3 * r + 2] - gt[3 * r + 2];
            acc += (dx * dx + dy * dy) + dz * dz;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_color_loss_backward_rays(const float *__restrict__ col, const float *__restrict__ gt, const int *__restrict__ hit, int R,
                                                                  const float *__restrict__ gscale, float *__restrict__ gcol) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float g = hit[r] > 0 ? 2.f * gscale[0] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) gcol[3 * r + c] = g == 0.f ? 0.f : g * (col[3 * r + c] - gt[3 * r + c]);
}
}  // namespace
extern "C" int pnerf_color_loss_blocks(int R) {
    const int b = (R + 255) / 256;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}
extern "C" int pnerf_color_loss_forward_rays(const float *d_ray_color, const float *d_gt, const int32_t *d_ray_hit, int R, float *d_partial, void *stream) {
    if (!d_ray_color || !d_gt || !d_ray_hit || !d_partial || R < 0) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_color_loss_forward_rays, dim3(pnerf_color_loss_blocks(R)), dim3(256), 0, (hipStream_t)stream, d_ray_color, d_gt, d_ray_hit, R, d_partial);
    PN_CHECK_LAUNCH();
    return 0;
}
extern "C" int pnerf_color_loss_backward_rays(const float *d_ray_color, const float *d_gt, const int32_t *d_ray_hit, int R, const float *d_gscale,
                                              float *d_grad_ray_color, void *stream) {
    if (R == 0) return 0;
    if (!d_ray_color || !d_gt || !d_ray_hit || !d_gscale || !d_grad_ray_color || R < 0) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_color_loss_backward_rays, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_ray_color, d_gt, d_ray_hit, R, d_gscale, d_grad_ray_color);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_zero_one_forward_rays(const float *d_conf, int n_points, const int32_t *d_idx, const int32_t *d_ray_hit, int R, int slots_per_ray, float eps,
                                           float *d_partial, void *stream) {
    if (!d_conf || !d_partial || !d_idx || !d_ray_hit || n_points <= 0 || R < 0 || slots_per_ray <= 0) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_zero_one_forward_rays, dim3(pnerf_zero_one_blocks((int64_t)R * TPB)), dim3(TPB), 0, (hipStream_t)stream, d_conf, n_points, d_idx, d_ray_hit, R, slots_per_ray, eps, d_partial);
    PN_CHECK_LAUNCH();
    return 0;
}
extern "C" int pnerf_zero_one_backward_rays(const float *d_conf, int n_points, const int32_t *d_idx, const int32_t *d_ray_hit, int R, int slots_per_ray, float eps,
                                            const float *d_gscale, float *d_grad_conf, void *stream) {
    if (R == 0) return 0;
    if (!d_conf || !d_idx || !d_ray_hit || !d_gscale || !d_grad_conf || n_points <= 0 || R < 0 || slots_per_ray <= 0) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_zero_one_backward_rays, dim3(pnerf_zero_one_blocks((int64_t)R * TPB)), dim3(TPB), 0, (hipStream_t)stream, d_conf, n_points, d_idx, d_ray_hit, R, slots_per_ray, eps, d_gscale, d_grad_conf);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_zero_one_blocks(int64_t n_idx) {
    const long long b = (n_idx + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
extern "C" int pnerf_zero_one_forward(const float *d_conf, int n_points, const int32_t *d_idx, int64_t n_idx, float eps, float *d_partial, void *stream) {
    if (!d_conf || !d_partial || n_points <= 0 || n_idx < 0 || (n_idx > 0 && !d_idx)) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_zero_one_forward, dim3(pnerf_zero_one_blocks(n_idx)), dim3(TPB), 0, (hipStream_t)stream, d_conf, n_points, d_idx, (long long)n_idx, eps, d_partial);
    PN_CHECK_LAUNCH();
    return 0;
}
extern "C" int pnerf_zero_one_backward(const float *d_conf, int n_points, const int32_t *d_idx, int64_t n_idx, float eps, const float *d_gscale,
                                       float *d_grad_conf, void *stream) {
    if (n_idx == 0) return 0;
    if (!d_conf || !d_idx || !d_gscale || !d_grad_conf || n_points <= 0 || n_idx < 0) return PNERF_E_INVAL;
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_zero_one_backward, dim3(pnerf_zero_one_blocks(n_idx)), dim3(TPB), 0, (hipStream_t)stream, d_conf, n_points, d_idx, (long long)n_idx, eps, d_gscale, d_grad_conf);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_gather_rows(const float *d_src, int n_src, int width, const int32_t *d_idx, int64_t n_idx, float *d_dst, void *stream) {
    if (n_idx == 0) return 0;                      // nothing to gather (a chunk of rays that hit nothing): pointers may be null
    if (!d_src || !d_idx || !d_dst || n_src <= 0 || width <= 0 || n_idx < 0) return PNERF_E_INVAL;
    long long total = n_idx * width;
    int grid = (int)((total + TPB - 1) / TPB < 16384 ? (total + TPB - 1) / TPB : 16384);
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    const bool al = ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0);
    const long long rows_per_block = TPB / (width / 4 > 0 ? width / 4 : 1);
    const int gridv = (int)((n_idx + rows_per_block - 1) / rows_per_block < 16384 ? (n_idx + rows_per_block - 1) / rows_per_block : 16384);
    if (al && width == 32) hipLaunchKernelGGL(k_gather_rows_v4<8>, dim3(gridv), dim3(TPB), 0, (hipStream_t)stream, (const float4 *)d_src, n_src, d_idx, (long long)n_idx, (float4 *)d_dst);
    else if (al && width == 4) hipLaunchKernelGGL(k_gather_rows_v4<1>, dim3(gridv), dim3(TPB), 0, (hipStream_t)stream, (const float4 *)d_src, n_src, d_idx, (long long)n_idx, (float4 *)d_dst);
    else hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, d_src, n_src, width, d_idx, (long long)n_idx, d_dst);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_scatter_add_rows(const float *d_grad_rows, const int32_t *d_idx, int64_t n_idx, int width, float *d_grad_src, int n_src, void *stream) {
    if (n_idx == 0) return 0;
    if (!d_grad_rows || !d_idx || !d_grad_src || n_src <= 0 || width <= 0 || n_idx < 0) return PNERF_E_INVAL;
    long long total = n_idx * width;
    int grid = (int)((total + TPB - 1) / TPB < 4096 ? (total + TPB - 1) / TPB : 4096);
    PnProfScope prof(PNK_GATHER, (hipStream_t)stream);
    hipLaunchKernelGGL(k_scatter_add_rows, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, d_grad_rows, d_idx, (long long)n_idx, width, d_grad_src, n_src);
    PN_CHECK_LAUNCH();
    return 0;
}

// flags[p] = 1 for every point p that occurs in the neighbor table, flags[0] = 1 if any slot is empty (empty slots read point 0 like the
// reference, neural_points.py:709, and the zero-one regulariser differentiates through that read): the rows a rank's point gradients can be
// non-zero in -- what a data-parallel caller exchanges instead of the dense [N, 39] gradient (pointnerf_amd/dist.py).  Plain stores of the
// same value race benignly; the table is read as int4 where it can be.
namespace {
__global__ __launch_bounds__(256) void k_touched_flags(const int *__restrict__ pidx, long long n, int n_points, int *__restrict__ flags) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int p = pidx[i];
        if (p >= 0 && p < n_points) flags[p] = 1;
        else if (p < 0) flags[0] = 1;
    }
}
}  // namespace
extern "C" int pnerf_touched_flags(const int32_t *d_pidx, int64_t n, int32_t n_points, int32_t *d_flags, void *stream) {
    if (n < 0 || n_points < 0 || (n > 0 && !d_pidx) || (n_points > 0 && !d_flags)) return PNERF_E_INVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n_points == 0) return 0;
    if (hipMemsetAsync(d_flags, 0, (size_t)n_points * sizeof(int32_t), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n == 0) return 0;
    const long long blocks = (n + 255) / 256;
    PnProfScope prof(PNK_GATHER, s);
    hipLaunchKernelGGL(k_touched_flags, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, d_pidx, (long long)n, n_points, d_flags);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t pnerf_agg_workspace_bytes(int64_t n_valid_max, int K) {
    if (K <= 0 || K > PNERF_MAX_K || n_valid_max < 0) return 0;
    // inference: only fs lives here; training: the caller passes a pnerf_agg_saved_bytes() area instead.
    long long rows, samples;
    pn_saved_bytes(n_valid_max, K, &rows, &samples);
    return pn_align((size_t)samples * PN_H * sizeof(float)) + pn_cls_bytes(samples) + pn_wgrad_partials_bytes();
}

static int check_common(const pnerf_camera *cam, const pnerf_points *pts, int R, int SR, int K) {
    if (!cam || !pts || R < 0 || SR <= 0 || K <= 0 || K > PNERF_MAX_K) return PNERF_E_INVAL;
    if (pts->feat_dim != PN_F) return PNERF_E_UNSUP;
    if (!pts->xyz || !pts->embedding || !pts->conf || !pts->dir || !pts->color) return PNERF_E_UNSUP;
    return 0;
}

extern "C" int pnerf_render_forward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                                    const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                                    const int32_t *d_sample_nn, const int32_t *d_valid_list, const int32_t *d_counters,
                                    int R, int SR, int K,
                                    float *d_decoded, float *d_weight, float *d_ray_color, float *d_opacity,
                                    float *d_bg_trans, float *d_blend_w,
                                    void *d_saved, int64_t n_valid_max, void *d_ws, size_t ws_bytes, void *stream) {
    int rc = check_common(cam, pts, R, SR, K);
    if (rc) return rc;
    if (!d_packed_mlp || !d_params || !d_raydir || !d_sample_loc || !d_sample_pidx || !d_sample_nn || !d_valid_list || !d_counters) return PNERF_E_INVAL;
    if (!d_decoded || !d_weight || !d_ray_color || !d_opacity || !d_bg_trans || !d_blend_w) return PNERF_E_INVAL;
    if (R == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PnSaved sv;
    const bool train = d_saved != nullptr;
    if (train) sv = pn_saved_carve(d_saved, n_valid_max, K);
    else {
        if (!d_ws || ws_bytes < pnerf_agg_workspace_bytes(n_valid_max, K)) return PNERF_E_WS;
        sv = PnSaved();
        pn_saved_bytes(n_valid_max, K, &sv.rows, &sv.samples);
        sv.fs = (float *)d_ws;
        pn_cls_carve((char *)d_ws + pn_align((size_t)sv.samples * PN_H * sizeof(float)), sv.samples, sv);
    }
    if (hipMemsetAsync(d_decoded, 0, (size_t)R * SR * 4 * sizeof(float), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(d_weight, 0, (size_t)R * SR * K * sizeof(float), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n_valid_max > 0) {
        rc = pn_agg_forward_launch(cam, pts, d_params, d_packed_mlp, d_raydir, d_sample_loc, nullptr, nullptr, d_sample_pidx, d_valid_list, d_counters,
                                   R, SR, K, d_decoded, d_weight, sv, n_valid_max, train, /*save_x0=*/false, s);
        if (rc) return rc;
    }
    RmArgs ra;
    ra.cam = *cam; ra.sample_loc = d_sample_loc; ra.decoded = d_decoded; ra.nn = d_sample_nn; ra.ray_dist = nullptr; ra.valid8 = nullptr; ra.R = R; ra.SR = SR;
    { PnProfScope prof(PNK_RAYMARCH_FWD, s);
    hipLaunchKernelGGL(k_raymarch_forward, dim3(pn_cdiv(R, TPB / 64)), dim3(TPB), 0, s, ra, d_ray_color, d_opacity, d_bg_trans, d_blend_w, (float *)nullptr); }
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_render_backward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                                     const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                                     const int32_t *d_sample_nn, const int32_t *d_valid_list, const int32_t *d_counters,
                                     int R, int SR, int K, int64_t n_valid,
                                     const float *d_decoded, const float *d_weight, const float *d_opacity,
                                     const float *d_grad_ray_color,
                                     void *d_saved, float *d_grad_params, const pnerf_point_grads *pg,
                                     void *d_ws, size_t ws_bytes, void *stream) {
    (void)d_opacity;
    int rc = check_common(cam, pts, R, SR, K);
    if (rc) return rc;
    if (!d_packed_mlp || !d_params || !d_raydir || !d_sample_loc || !d_sample_pidx || !d_sample_nn || !d_valid_list || !d_counters) return PNERF_E_INVAL;
    if (!d_decoded || !d_weight || !d_grad_ray_color || !d_saved || !d_grad_params || !pg || !d_ws) return PNERF_E_INVAL;
    const size_t gd_bytes = pn_align((size_t)R * SR * 4 * sizeof(float));
    if (ws_bytes < gd_bytes + pn_wgrad_partials_bytes()) return PNERF_E_WS;
    if (R == 0 || n_valid == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    float *grad_decoded = (float *)d_ws;
    float *partials = (float *)((char *)d_ws + gd_bytes);
    RmArgs ra;
    ra.cam = *cam; ra.sample_loc = d_sample_loc; ra.decoded = d_decoded; ra.nn = d_sample_nn; ra.ray_dist = nullptr; ra.valid8 = nullptr; ra.R = R; ra.SR = SR;
    { PnProfScope prof(PNK_RAYMARCH_BWD, s);
    hipLaunchKernelGGL(k_raymarch_backward, dim3(pn_cdiv(R, TPB / 64)), dim3(TPB), 0, s, ra, d_grad_ray_color, grad_decoded); }
    PN_CHECK_LAUNCH();
    PnSaved sv = pn_saved_carve(d_saved, n_valid, K);
    return pn_agg_backward_launch(cam, pts, d_params, d_packed_mlp, d_raydir, d_sample_loc, d_sample_pidx, d_valid_list, d_counters,
                                  R, SR, K, d_decoded, d_weight, grad_decoded, sv, n_valid, d_grad_params, pg, partials, /*x0_saved=*/false, s);
}

extern "C" size_t pnerf_render_backward_workspace_bytes(int R, int SR) {
    return pn_align((size_t)R * SR * 4 * sizeof(float)) + pn_wgrad_partials_bytes();
}

// ================================ stand-alone (level-1) entry points ================================

extern "C" size_t pnerf_compact_workspace_bytes(int64_t n) { return pn_align(pn_scan_scratch_ints(n) * sizeof(int)); }

// d_list = ascending indices i with d_nn[i] > 0, d_counters[0] = their number (builds the aggregator's work list from
// a caller-supplied validity array: PointAggregator.forward's ray_valid = any_K(sample_pnt_mask), point_aggregators.py:741)
extern "C" int pnerf_compact_valid(const int32_t *d_nn, int64_t n, int32_t *d_list, int32_t *d_counters, void *d_ws, size_t ws_bytes, void *stream) {
    if (!d_nn || !d_list || !d_counters || !d_ws || n < 0) return PNERF_E_INVAL;
    if (ws_bytes < pnerf_compact_workspace_bytes(n)) return PNERF_E_WS;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_counters, 0, 8 * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n == 0) return 0;
    PnProfScope prof(PNK_COMPACT, s);
    return pn_compact_gt0_i32(d_nn, n, d_list, d_counters, (int *)d_ws, s);
}

extern "C" int pnerf_agg_forward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                                 const float *d_raydir, const float *d_sample_loc, const float *d_xyz_pers, const float *d_loc_pers,
                                 const int32_t *d_sample_pidx, const int32_t *d_valid_list, const int32_t *d_counters,
                                 int R, int SR, int K, float *d_decoded, float *d_weight,
                                 void *d_saved, int64_t n_valid_max, void *d_ws, size_t ws_bytes, void *stream) {
    int rc = check_common(cam, pts, R, SR, K);
    if (rc) return rc;
    if (!d_packed_mlp || !d_params || !d_raydir || !d_sample_loc || !d_sample_pidx || !d_valid_list || !d_counters || !d_decoded || !d_weight) return PNERF_E_INVAL;
    if ((d_xyz_pers == nullptr) != (d_loc_pers == nullptr)) return PNERF_E_INVAL;
    if (R == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PnSaved sv;
    const bool train = d_saved != nullptr;
    if (train) sv = pn_saved_carve(d_saved, n_valid_max, K);
    else {
        if (!d_ws || ws_bytes < pnerf_agg_workspace_bytes(n_valid_max, K)) return PNERF_E_WS;
        sv = PnSaved();
        pn_saved_bytes(n_valid_max, K, &sv.rows, &sv.samples);
        sv.fs = (float *)d_ws;
        pn_cls_carve((char *)d_ws + pn_align((size_t)sv.samples * PN_H * sizeof(float)), sv.samples, sv);
    }
    if (hipMemsetAsync(d_decoded, 0, (size_t)R * SR * 4 * sizeof(float), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(d_weight, 0, (size_t)R * SR * K * sizeof(float), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n_valid_max == 0) return 0;
    return pn_agg_forward_launch(cam, pts, d_params, d_packed_mlp, d_raydir, d_sample_loc, d_xyz_pers, d_loc_pers, d_sample_pidx,
                                 d_valid_list, d_counters, R, SR, K, d_decoded, d_weight, sv, n_valid_max, train, /*save_x0=*/train, s);
}

extern "C" int pnerf_agg_backward(const pnerf_camera *cam, const pnerf_points *pts, const void *d_packed_mlp, const float *d_params,
                                  const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                                  const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K, int64_t n_valid,
                                  const float *d_decoded, const float *d_weight, const float *d_grad_decoded,
                                  void *d_saved, float *d_grad_params, const pnerf_point_grads *pg, void *d_ws, size_t ws_bytes, void *stream) {
    int rc = check_common(cam, pts, R, SR, K);
    if (rc) return rc;
    if (!d_packed_mlp || !d_params || !d_raydir || !d_sample_loc || !d_sample_pidx || !d_valid_list || !d_counters) return PNERF_E_INVAL;
    if (!d_decoded || !d_weight || !d_grad_decoded || !d_saved || !d_grad_params || !pg || !d_ws) return PNERF_E_INVAL;
    if (ws_bytes < pn_wgrad_partials_bytes()) return PNERF_E_WS;
    if (R == 0 || n_valid == 0) return 0;
    PnSaved sv = pn_saved_carve(d_saved, n_valid, K);
    return pn_agg_backward_launch(cam, pts, d_params, d_packed_mlp, d_raydir, d_sample_loc, d_sample_pidx, d_valid_list, d_counters,
                                  R, SR, K, d_decoded, d_weight, d_grad_decoded, sv, n_valid, d_grad_params, pg, (float *)d_ws,
                                  /*x0_saved=*/true, (hipStream_t)stream);
}

// ray_march(ray_dist, ray_valid, ray_features, radiance, alpha, bg_color)   models/rendering/diff_ray_marching.py:508-554
extern "C" int pnerf_raymarch_forward(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *bg3_host,
                                      int R, int SR, float *d_ray_color, float *d_opacity, float *d_acc_trans, float *d_blend_w,
                                      float *d_bg_trans, void *stream) {
    if (!d_ray_dist || !d_ray_valid || !d_features || !d_ray_color || !d_opacity || !d_acc_trans || !d_blend_w || !d_bg_trans || R < 0 || SR <= 0) return PNERF_E_INVAL;
    if (R == 0) return 0;
    RmArgs ra = {};
    ra.decoded = d_features; ra.ray_dist = d_ray_dist; ra.valid8 = d_ray_valid; ra.R = R; ra.SR = SR;
    ra.cam.has_bg = bg3_host ? 1 : 0;
    if (bg3_host) { ra.cam.bg[0] = bg3_host[0]; ra.cam.bg[1] = bg3_host[1]; ra.cam.bg[2] = bg3_host[2]; }
    hipStream_t s = (hipStream_t)stream;
    PnProfScope prof(PNK_RAYMARCH_FWD, s);
    hipLaunchKernelGGL(k_raymarch_forward, dim3(pn_cdiv(R, TPB / 64)), dim3(TPB), 0, s, ra, d_ray_color, d_opacity, d_bg_trans, d_blend_w, d_acc_trans);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_raymarch_backward(const float *d_ray_dist, const uint8_t *d_ray_valid, const float *d_features, const float *bg3_host,
                                       int R, int SR, const float *d_grad_ray_color, float *d_grad_features, void *stream) {
    if (!d_ray_dist || !d_ray_valid || !d_features || !d_grad_ray_color || !d_grad_features || R < 0 || SR <= 0) return PNERF_E_INVAL;
    if (R == 0) return 0;
    RmArgs ra = {};
    ra.decoded = d_features; ra.ray_dist = d_ray_dist; ra.valid8 = d_ray_valid; ra.R = R; ra.SR = SR;
    ra.cam.has_bg = bg3_host ? 1 : 0;
    if (bg3_host) { ra.cam.bg[0] = bg3_host[0]; ra.cam.bg[1] = bg3_host[1]; ra.cam.bg[2] = bg3_host[2]; }
    hipStream_t s = (hipStream_t)stream;
    PnProfScope prof(PNK_RAYMARCH_BWD, s);
    hipLaunchKernelGGL(k_raymarch_backward, dim3(pn_cdiv(R, TPB / 64)), dim3(TPB), 0, s, ra, d_grad_ray_color, d_grad_features);
    PN_CHECK_LAUNCH();
    return 0;
}
