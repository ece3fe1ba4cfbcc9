// optim.hip -- the parameter update that follows the hot path every training step.
//
// The reference steps two torch.optim.Adam instances (models/mvs_points_volumetric_model.py:80-91,
// lr / plr, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad): ~10 ATen multi-tensor kernels
// over the MLP parameters and over N x 39 point parameters -- at N = 2 M that is 2.2 GB of HBM traffic
// per pass and the largest stream of the step outside the aggregator.  Here: one pass, one kernel per
// tensor (p, g, m, v read once; p, m, v written once = 28 B per element).  HBM-bound by construction.
// The arithmetic follows torch's single-tensor Adam operation by operation so that the two can be
// compared to fp32 rounding:
//   m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g g; denom = sqrt(v) / sqrt(1 - b2^t) + eps;
//   p = p - (lr / (1 - b1^t)) * m / denom
#include "pn_common.h"
#include <math.h>

namespace {
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                              long long n, float b1, float omb1, float b2, float omb2, float eps, float step_size, float inv_sqrt_bc2, int vec) {
    const long long n4 = vec ? n >> 2 : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        mm = mm + (gg - mm) * omb1;
        vv = vv * b2 + omb2 * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp = pp - step_size * (mm / denom);
    };
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4 *>(p)[i], mv = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gv = reinterpret_cast<const float4 *>(g)[i];
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4 *>(p)[i] = pv; reinterpret_cast<float4 *>(m)[i] = mv; reinterpret_cast<float4 *>(v)[i] = vv;
    }
    // scalar tail (or everything, when the four arrays are not all 16-byte aligned: views into a flat parameter vector)
    for (long long t = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) upd(p[t], g[t], m[t], v[t]);
}
// ---- multi-tensor form: ONE launch for a list of tensors (the MLP's 18 parameter tensors are 10 .. 72 704 floats each: as 18 + 4
// launches the two optimizers were 22 launches per step, 18 of them a few microseconds of work behind a ~1.5 us launch boundary each)
constexpr int PN_ADAM_MAX = 24;
struct PnAdamDesc { float *p; const float *g; float *m, *v; long long n; float step_size, inv_sqrt_bc2; int vec, block0, nblocks; };
struct PnAdamTable { PnAdamDesc d[PN_ADAM_MAX]; int count; float b1, omb1, b2, omb2, eps; };

__global__ __launch_bounds__(256) void k_adam_multi(PnAdamTable t) {
    int ti = 0;
#pragma unroll 1
    for (int i = 1; i < t.count; ++i)
        if ((int)blockIdx.x >= t.d[i].block0) ti = i;
    const PnAdamDesc &d = t.d[ti];
    float *__restrict__ p = d.p; const float *__restrict__ g = d.g; float *__restrict__ m = d.m; float *__restrict__ v = d.v;
    const float b2 = t.b2, omb1 = t.omb1, omb2 = t.omb2, eps = t.eps, step_size = d.step_size, inv_sqrt_bc2 = d.inv_sqrt_bc2;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        mm = mm + (gg - mm) * omb1;
        vv = vv * b2 + omb2 * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp = pp - step_size * (mm / denom);
    };
    const long long n = d.n, n4 = d.vec ? n >> 2 : 0;
    const long long first = (long long)(blockIdx.x - d.block0) * 256 + threadIdx.x, stride = (long long)d.nblocks * 256;
    for (long long i = first; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4 *>(p)[i], mv = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gv = reinterpret_cast<const float4 *>(g)[i];
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4 *>(p)[i] = pv; reinterpret_cast<float4 *>(m)[i] = mv; reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (long long e = (n4 << 2) + first; e < n; e += stride) upd(p[e], g[e], m[e], v[e]);
}
}  // namespace

extern "C" int pnerf_adam_step_multi(const pnerf_adam_tensor *tensors, int count, double beta1, double beta2, double eps, void *stream) {
    if (count < 0 || (count > 0 && !tensors)) return PNERF_E_INVAL;
    for (int i = 0; i < count; ++i) {
        const pnerf_adam_tensor &a = tensors[i];
        if (a.n == 0) continue;                       // (an empty tensor has no storage: nothing to check, nothing to do)
        if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq || a.n < 0 || a.step < 1) return PNERF_E_INVAL;
        if (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 3) return PNERF_E_INVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    PnProfScope prof(PNK_ADAM, s);
    int i = 0;
    while (i < count) {
        PnAdamTable t;
        t.count = 0; t.b1 = (float)beta1; t.omb1 = (float)(1.0 - beta1); t.b2 = (float)beta2; t.omb2 = (float)(1.0 - beta2); t.eps = (float)eps;
        int blocks = 0;
        for (; i < count && t.count < PN_ADAM_MAX; ++i) {
            const pnerf_adam_tensor &a = tensors[i];
            if (a.n == 0) continue;
            PnAdamDesc &d = t.d[t.count++];
            d.p = a.param; d.g = a.grad; d.m = a.exp_avg; d.v = a.exp_avg_sq; d.n = a.n;
            // scalars are formed in double like torch forms them from its python floats, then rounded once
            const double bc1 = 1.0 - pow(beta1, (double)a.step), bc2 = 1.0 - pow(beta2, (double)a.step);
            d.step_size = (float)(a.lr / bc1); d.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
            d.vec = (((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) & 15) == 0;
            const long long work = d.vec ? (a.n + 3) / 4 : a.n;
            long long nb = (work + 255) / 256;
            if (nb > 256 * 16) nb = 256 * 16;        // grid-stride inside a tensor: 16 workgroups per CU keep HBM busy
            d.block0 = blocks; d.nblocks = (int)nb;
            blocks += (int)nb;
        }
        if (t.count == 0) continue;
        hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks), dim3(256), 0, s, t);
        PN_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int pnerf_adam_step(float *d_param, const float *d_grad, float *d_exp_avg, float *d_exp_avg_sq, int64_t n,
                               double lr, double beta1, double beta2, double eps, int64_t step, void *stream) {
    if (!d_param || !d_grad || !d_exp_avg || !d_exp_avg_sq || n < 0 || step < 1) return PNERF_E_INVAL;
    if (((uintptr_t)d_param | (uintptr_t)d_grad | (uintptr_t)d_exp_avg | (uintptr_t)d_exp_avg_sq) & 3) return PNERF_E_INVAL;
    const int vec = (((uintptr_t)d_param | (uintptr_t)d_grad | (uintptr_t)d_exp_avg | (uintptr_t)d_exp_avg_sq) & 15) == 0;   // float4 path
    if (n == 0) return 0;
    // scalars are formed in double like torch forms them from its python floats, then rounded once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipStream_t s = (hipStream_t)stream;
    const long long work = vec ? (n + 3) / 4 : n;
    long long blocks = (work + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;       // grid-stride: 16 workgroups per CU keep HBM busy
    if (blocks < 1) blocks = 1;
    PnProfScope prof(PNK_ADAM, s);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, s, d_param, d_grad, d_exp_avg, d_exp_avg_sq, (long long)n, (float)beta1, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, inv_sqrt_bc2, vec);
    PN_CHECK_LAUNCH();
    return 0;
}
