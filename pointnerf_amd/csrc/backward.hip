// backward.hip -- backward of the aggregator: colour MLP dgrad, per-neighbor MLP dgrad + gather
// scatter-add, and the weight-gradient GEMMs.  The four 256-wide layers run on the f16 matrix pipe with two-plane operands
// (f16x3.h); all gradients inside the aggregator backward carry a per-call power-of-two scale (k_grad_max) so that they sit
// in the f16 range, and leave it (atomics, weight-gradient reduction) multiplied by its exact inverse.
//
// The reference gets all of this from torch.autograd over ~60 ATen ops (loss.backward() in
// models/mvs_points_volumetric_model.py:98-118): cuBLAS dgrad/wgrad per nn.Linear, dense
// index_select backward into [1,N,F] buffers, boolean-mask scatter backward.  Here:
//   k_color_backward : 64 valid samples per tile; d rgb -> colour chain dgrad on MFMA -> d f[256]
//   k_agg_backward   : TS samples x K rows per tile; (d sigma, d f) -> alpha head, K-weighted sums,
//                      block3/block1 dgrad on MFMA, PE chain rule, atomic scatter-add into the
//                      embedding / colour / dir / conf gradients of the touched points only
//   k_wgrad_f16      : dW = dY^T X as split-K MFMA GEMMs over the saved k-major f16 planes, full dW tile
//                      resident in accumulators, deterministic partial-sum reduction
// LeakyReLU masks: 1 bit per element, written by the forward in the accumulator layout (layers 1-3), sign of the saved h4 (layer 4).
#include "mixq.h"

namespace {
constexpr int WG_CHUNKS = 256;                 // split-K factor of the wgrad GEMMs
constexpr size_t PARTIAL_FLOATS = (size_t)WG_CHUNKS * PN_H * PN_IN1P;

struct BwdArgs {
    pnerf_camera cam;
    const float *params;
    const float4 *packed;
    const float *raydir;
    const int *pidx, *valid_list, *counters;
    const int *cls_list, *cls_info;     // sample classes (aggregate.hip: pn_classify); cls = the class this launch processes
    int cls;
    int SR, K, TS;
    long long cap_samples;
    const float *decoded, *weight, *grad_decoded;
    PnSaved sv;
    const float *emb;
    float *gparams;
    float *g_emb, *g_conf, *g_dir, *g_color;
    // optional (pnerf_point_grads.zero_one_gscale): the zero-one regulariser's conf gradient rides on this kernel's conf atomics
    const float *conf, *zo_gs;
    float zo_eps;
};

// d/d conf of log(v) + log(1 - v), v = clamp(clamp(conf, 1e-4, 1), eps, 1 - eps) with a straight-through inner clamp (the arithmetic of
// render.hip pn_zero_one_value / k_zero_one_backward_rays), times the caller's scale
__device__ __forceinline__ float pn_zero_one_grad(float conf, float eps, float gs) {
    const float c = fminf(fmaxf(conf, 1e-4f), 1.0f);
    if (!(c >= eps && c <= 1.f - eps)) return 0.f;
    return gs * (1.f / c - 1.f / (1.f - c));
}

__device__ __forceinline__ void rot3b(const float *M, float x, float y, float z, bool transpose, float &ox, float &oy, float &oz) {
    if (!transpose) { ox = x * M[0] + y * M[3] + z * M[6]; oy = x * M[1] + y * M[4] + z * M[7]; oz = x * M[2] + y * M[5] + z * M[8]; }
    else { ox = x * M[0] + y * M[1] + z * M[2]; oy = x * M[3] + y * M[4] + z * M[5]; oz = x * M[6] + y * M[7] + z * M[8]; }
}

// ------------------------------------------------------------------------------ gradient scale
// bits of max |d decoded| over the valid samples -> sv.gscale[0] (zeroed by the launcher); every consumer derives the same
// power of two from it: S = 2^(4 - floor(log2 max)), so that the largest scaled gradient lies in [16, 32)
__global__ __launch_bounds__(256) void k_grad_max(const int *__restrict__ list, const int *__restrict__ counters, long long cap, const float *__restrict__ gd,
                                                  unsigned *__restrict__ gscale) {
    __shared__ unsigned red[4];
    const long long Ns = counters[0] < cap ? counters[0] : cap;
    float m = 0.f;
    for (long long vs = (long long)blockIdx.x * 256 + threadIdx.x; vs < Ns; vs += (long long)gridDim.x * 256) {
        const float4 g = *reinterpret_cast<const float4 *>(gd + (long long)list[vs] * 4);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = __float_as_uint(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = red[0];
        for (int i = 1; i < 4; ++i) r = red[i] > r ? red[i] : r;
        if (r != 0u && r < 0x7f800000u) atomicMax(gscale, r);            // (NaN / inf gradients do not pick the scale)
    }
}
__device__ __forceinline__ void pn_scale_from_bits(unsigned mb, float &S, float &invS) {
    const int e = (int)((mb >> 23) & 0xffu);
    int se = (e == 0 || e == 255) ? 127 : 127 + 4 - (e - 127);
    se = se < 2 ? 2 : (se > 252 ? 252 : se);
    S = __uint_as_float((unsigned)se << 23);
    invS = __uint_as_float((unsigned)(254 - se) << 23);
}

// ------------------------------------------------------------------------------ colour backward
// 64 valid samples per tile, the forward's organisation: d rgb -> d(pre-sigmoid) -> d c3 on the VALU (3 x 128 weights), then the
// dgrad chain d c3 x Wc3 -> d c2 x Wc2 -> d c1 x Wc1[:, :256] -> d f as two-plane f16 GEMMs on gradients that carry the call's
// power-of-two scale S (k_grad_max).  The LeakyReLU masks are the forward's sign words (c1, c2) and the saved fp32 c3; d c1..d c3 leave
// k-major as one f16 plane (scaled) for the weight-gradient GEMMs of these layers, d f as fp32 rows (x 1/S) for k_agg_backward.
// LDS: the tile, d raw [64][4], and the workgroup's running sums of d Wc4 [3][128], d bc3, d bc2, d bc1 [128] (LDS float adds:
// kept in registers they are 45 loop-carried values per thread next to the GEMM's working set)
// The tile holds 128 columns at most (d c3, d c2, d c1): its rows are 272 bytes apart (68 dwords = 4 mod 64 banks: the GEMM's 16-byte fragment
// reads of 16 consecutive rows are conflict-free), 35 KB for both planes instead of the 76 KB of the 296-column tile -- THREE workgroups per
// CU instead of two (the kernel is a chain of short phases with 3.5 us of MFMA work per tile: what it lacks is waves to overlap them).
constexpr int CB_XRS = 2 * PN_HC + 16, CB_XPL = PN_CTILE * CB_XRS, CB_XBYTES = 2 * CB_XPL;
constexpr int CB_DRAW = CB_XBYTES, CB_GACC = CB_DRAW + PN_CTILE * 4 * 4, CB_W4 = CB_GACC + 6 * PN_HC * 4, CB_BYTES = CB_W4 + 3 * PN_HC * 4;
static_assert(3 * CB_BYTES <= 160 * 1024, "three colour workgroups must fit the 160 KB LDS");

__device__ __forceinline__ void cb_acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
// d(pre-activation) = acc * LeakyReLU' (sign bits of the forward's pre-activations, same lane -> element map): two planes
// (saturating, high plane rounded to nearest) into the tile
__device__ __forceinline__ void cb_epilogue(const f32x16 (&acc)[2][2], unsigned mw, char *X, int wave, int lane) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = pn_d_feat(wave, g, lane), row = 32 * rb + (lane & 31), e = rb * 16 + g * 4;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[0][rb][4 * g + i] * ((int)(mw << (e + i)) < 0 ? 0.01f : 1.f);
            pn_x_store4<true, CB_XRS, CB_XPL>(X, row, f0, v[0], v[1], v[2], v[3]);
        }
}
// column sums of the tile's first 128 columns (the bias gradient of the layer whose d(pre-activation) the tile holds): thread ->
// 8 columns x 4 rows
__device__ __forceinline__ void cb_bias_sums(const char *X, int tid, float *__restrict__ gsum) {
    const int c0 = 8 * (tid & 15), r0 = 4 * (tid >> 4);
    float gb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const uint4 h = *reinterpret_cast<const uint4 *>(X + (r0 + rr) * CB_XRS + c0 * 2);
        const uint4 m = *reinterpret_cast<const uint4 *>(X + CB_XPL + (r0 + rr) * CB_XRS + c0 * 2);
        gb[0] = pn_fma2_lo(h.x, m.x, 1.f, gb[0]); gb[1] = pn_fma2_hi(h.x, m.x, 1.f, gb[1]);
        gb[2] = pn_fma2_lo(h.y, m.y, 1.f, gb[2]); gb[3] = pn_fma2_hi(h.y, m.y, 1.f, gb[3]);
        gb[4] = pn_fma2_lo(h.z, m.z, 1.f, gb[4]); gb[5] = pn_fma2_hi(h.z, m.z, 1.f, gb[5]);
        gb[6] = pn_fma2_lo(h.w, m.w, 1.f, gb[6]); gb[7] = pn_fma2_hi(h.w, m.w, 1.f, gb[7]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(gsum + c0 + i, gb[i]);
}

// WG2: the two-plane weight-gradient mode (pnerf_set_wgrad_planes(2)): the residual plane of every d c tile leaves as well
template <bool WG2>
__global__ __launch_bounds__(256, 3) void k_color_backward(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_cb[];
    char *X = smem_cb;
    float *draw = reinterpret_cast<float *>(smem_cb + CB_DRAW);          // [64][4] d(pre-sigmoid colour)
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;
    const char *img = reinterpret_cast<const char *>(a.packed);
    float S, invS;
    pn_scale_from_bits(a.sv.gscale[0], S, invS);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.sv.cls_info[PN_CI_CTILES] = (Ns + PN_CTILE - 1) / PN_CTILE;       // for the weight-gradient GEMMs
    float *gacc = reinterpret_cast<float *>(smem_cb + CB_GACC);          // [3][128] d Wc4 | d bc3 | d bc2 (x S) | d bc1 (x S)
    for (int i = threadIdx.x; i < 6 * PN_HC; i += 256) gacc[i] = 0.f;
    float gb4x = 0.f, gb4y = 0.f, gb4z = 0.f;
    // d(pre-sigmoid colour) of this thread's row (threads 0 .. 63) is formed from two dependent gathers (sample id -> decoded / its gradient):
    // requested one tile ahead (the id at the top of the previous tile, the six values behind its first GEMM), so that no tile starts with two
    // HBM round trips; d bc4 = the column sums of d raw accumulate per row thread and are reduced once at the end
    auto row_sample = [&](long long t) -> long long {
        const long long vs = t * PN_CTILE + threadIdx.x;
        return (threadIdx.x < PN_CTILE && vs < Ns) ? (long long)a.valid_list[vs] : -1;
    };
    float po[3] = {0.f, 0.f, 0.f}, pg[3] = {0.f, 0.f, 0.f};
    auto row_values = [&](long long si) {
        if (si >= 0) {
            const float *o = a.decoded + si * 4, *g = a.grad_decoded + si * 4;
            po[0] = o[1]; po[1] = o[2]; po[2] = o[3]; pg[0] = g[1]; pg[1] = g[2]; pg[2] = g[3];
        }
    };
    long long si_cur = row_sample(blockIdx.x);
    row_values(si_cur);
    float *w4s = reinterpret_cast<float *>(smem_cb + CB_W4);             // Wc4 [3][128] (the colour tensors sit at odd float offsets in the flat vector)
    for (int i = threadIdx.x; i < 3 * PN_HC; i += 256) w4s[i] = P[PO_WC4 + i];

    f32x16 acc[2][2];
    for (long long tile = blockIdx.x; tile * PN_CTILE < Ns; tile += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const long long grow0 = tile * PN_CTILE;
        PN_LDS_BARRIER();
        if (tid < PN_CTILE) {
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (si_cur >= 0) {
                // rgb = sigmoid(raw) * 1.002 - 0.001  ->  d raw = d rgb * 1.002 * s (1 - s)
                const float s0 = (po[0] + 0.001f) / 1.002f, s1 = (po[1] + 0.001f) / 1.002f, s2 = (po[2] + 0.001f) / 1.002f;
                d0 = pg[0] * 1.002f * s0 * (1.f - s0); d1 = pg[1] * 1.002f * s1 * (1.f - s1); d2 = pg[2] * 1.002f * s2 * (1.f - s2);
            }
            *reinterpret_cast<float4 *>(draw + tid * 4) = make_float4(d0, d1, d2, 0.f);
            gb4x += d0; gb4y += d1; gb4z += d2;
        }
        const long long si_next = row_sample(tile + gridDim.x);
        PN_LDS_BARRIER();
        // ---- d c3 = (d raw @ Wc4) * lrelu'(c3); d Wc4, d bc4, d bc3 accumulate in registers
        {
            const int c4 = tid & 31, rg = tid >> 5;
            float4 c3v[8];
            float w4[3][4], gw4[3][4], gb3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 w = *reinterpret_cast<const float4 *>(w4s + k * PN_HC + 4 * c4);
                w4[k][0] = w.x; w4[k][1] = w.y; w4[k][2] = w.z; w4[k][3] = w.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) gw4[k][i] = 0.f;
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) c3v[rr] = *reinterpret_cast<const float4 *>(a.sv.c3 + (grow0 + 8 * rg + rr) * PN_HC + 4 * c4);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = 8 * rg + rr;
                const float4 d = *reinterpret_cast<const float4 *>(draw + row * 4);
                const float c[4] = {c3v[rr].x, c3v[rr].y, c3v[rr].z, c3v[rr].w};
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u[i] = (d.x * w4[0][i] + d.y * w4[1][i] + d.z * w4[2][i]) * pn_lrelu_grad(c[i]);
                    gw4[0][i] += d.x * c[i]; gw4[1][i] += d.y * c[i]; gw4[2][i] += d.z * c[i];
                    gb3[i] += u[i];
                }
                pn_x_store4<true, CB_XRS, CB_XPL>(X, row, 4 * c4, u[0] * S, u[1] * S, u[2] * S, u[3] * S);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(gacc + k * PN_HC + 4 * c4 + i, gw4[k][i]);
                atomicAdd(gacc + 3 * PN_HC + 4 * c4 + i, gb3[i]);
            }
        }
        // (round 4: the first weight chunks of each GEMM are requested in front of the barrier that precedes it: see k_color_forward)
        PnGemmW<8, 4, 1, 4, 3> WD3;
        WD3.prefetch(reinterpret_cast<const uint4 *>(img + PKH_DC3), wave, lane);
        PN_LDS_BARRIER();
        // ---- d c2 = (d c3 @ Wc3) * lrelu'(c2)
        pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X, a.sv.dc3k, tile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X + CB_XPL, a.sv.dc3m, tile * 8, tid);
        const unsigned mw2 = a.sv.cmask[(tile * 2 + 1) * 256 + tid], mw1 = a.sv.cmask[(tile * 2 + 0) * 256 + tid];
        cb_acc_zero(acc);
        pn_gemm_f16x3_run<8, 4, 1, 4, 3, CB_XRS, CB_XPL>(X, WD3, lane, acc);
        row_values(si_next);                              // (the next tile's six values: consumed at the top of the next iteration)
        si_cur = si_next;
        PN_LDS_BARRIER();
        cb_epilogue(acc, mw2, X, wave, lane);
        PnGemmW<8, 4, 1, 4, 3> WD2;
        WD2.prefetch(reinterpret_cast<const uint4 *>(img + PKH_DC2), wave, lane);
        PN_LDS_BARRIER();
        cb_bias_sums(X, tid, gacc + 4 * PN_HC);
        // ---- d c1 = (d c2 @ Wc2) * lrelu'(c1)
        pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X, a.sv.dc2k, tile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X + CB_XPL, a.sv.dc2m, tile * 8, tid);
        cb_acc_zero(acc);
        pn_gemm_f16x3_run<8, 4, 1, 4, 3, CB_XRS, CB_XPL>(X, WD2, lane, acc);
        PN_LDS_BARRIER();
        cb_epilogue(acc, mw1, X, wave, lane);
        PnGemmW<8, 8, 2, 1, 3> WD1;
        WD1.prefetch(reinterpret_cast<const uint4 *>(img + PKH_DC1), 2 * wave, lane);
        PN_LDS_BARRIER();
        cb_bias_sums(X, tid, gacc + 5 * PN_HC);
        pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X, a.sv.dc1k, tile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_HC, CB_XRS>(X + CB_XPL, a.sv.dc1m, tile * 8, tid);
        // ---- d f = d c1 @ Wc1[:, :256]
        cb_acc_zero(acc);
        pn_gemm_f16x3_run<8, 8, 2, 1, 3, CB_XRS, CB_XPL>(X, WD1, lane, acc);
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pn_f4 t = {acc[fb][rb][4 * g] * invS, acc[fb][rb][4 * g + 1] * invS, acc[fb][rb][4 * g + 2] * invS, acc[fb][rb][4 * g + 3] * invS};
                    PN_REG_STORE(t, reinterpret_cast<pn_f4 *>(a.sv.dfs + (grow0 + 32 * rb + (lane & 31)) * PN_H + pn_d_feat(2 * wave + fb, g, lane)));
                }
    }
    PN_LDS_BARRIER();
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < 3 * PN_HC; i += 256) atomicAdd(&a.gparams[PO_WC4 + i], gacc[i]);
        if (tid < PN_HC) {
            atomicAdd(&a.gparams[PO_BC3 + tid], gacc[3 * PN_HC + tid]);
            atomicAdd(&a.gparams[PO_BC2 + tid], gacc[4 * PN_HC + tid] * invS);
            atomicAdd(&a.gparams[PO_BC1 + tid], gacc[5 * PN_HC + tid] * invS);
        }
        if (tid < PN_CTILE) {                              // d bc4: the row threads' partial column sums (wave 0)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { gb4x += __shfl_xor(gb4x, off, 64); gb4y += __shfl_xor(gb4y, off, 64); gb4z += __shfl_xor(gb4z, off, 64); }
            if (tid == 0) { atomicAdd(&a.gparams[PO_BC4], gb4x); atomicAdd(&a.gparams[PO_BC4 + 1], gb4y); atomicAdd(&a.gparams[PO_BC4 + 2], gb4z); }
        }
    }
}

// ------------------------------------------------------------------------------ aggregator backward
// One 64-row tile per workgroup at a time, two workgroups per CU (the forward's organisation).  Per tile:
//   load h4 planes + row metadata + sign words -> alpha head backward (d conf, d alpha pre-activation) -> dY4 in place
//   -> four dgrad GEMMs on the f16 pipe, each followed by the LeakyReLU' epilogue that writes the next dY tile; the dY tile of
//   every layer is copied (transposed to k-major planes) to HBM for the weight-gradient GEMM while its own GEMM runs
//   -> layer-3 extras (d colour, d dir) as a ninth feature block, K split over the four waves
//   -> d X0 (224 columns: the embedding and its encoding) as fp32 in LDS -> PE chain rule -> atomics into the touched points.
// Bias gradients are not formed here: they are the ones-column of the weight-gradient GEMMs.
constexpr int TPR = PN_TPR;                    // threads per tile row in the row-wise phases
constexpr int EPT = PN_F / TPR;                // embedding dims per thread
constexpr int BL_ROW = PN_XBYTES, BL_W5 = BL_ROW + 7 * PN_TILE * 4, BL_BYTES = BL_W5 + PN_H * 4;
constexpr int LDDX = 228;                      // fp32 row stride of the d X0 tile (over the activation tile's space)
static_assert(2 * BL_BYTES <= 160 * 1024, "two backward workgroups must fit the 160 KB LDS");
static_assert(PN_TILE * LDDX * 4 <= PN_XBYTES, "d X0 tile");

template <int N> __device__ __forceinline__ float group_sum_b(float v) {
#pragma unroll
    for (int off = 1; off < N; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int AF> __device__ __forceinline__ void b_acc_zero(f32x16 (&acc)[AF][2]) {
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
// dgrad epilogue: accumulators x LeakyReLU' (sign bit of the forward's word of the feature block: element e = (rb * 4 + g) * 4 + i at bit
// 31 - e) -> both planes of the next dY tile
template <bool MIX = false>
__device__ __forceinline__ void b_epilogue(const f32x16 (&acc)[PN_NFB][2], const unsigned (&mask)[PN_NFB], char *X, int wave, int lane) {
#pragma unroll
    for (int fb = 0; fb < PN_NFB; ++fb) {
        const unsigned mw = mask[fb];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = (rb * 4 + g) * 4 + i;
                    v[i] = acc[fb][rb][4 * g + i] * ((int)(mw << e) < 0 ? 0.01f : 1.f);
                }
                if (MIX) pn_xq_store4(X, 32 * rb + (lane & 31), pn_d_feat(PN_NFB * wave + fb, g, lane), v[0], v[1], v[2], v[3]);
                else pn_x_store4<true>(X, 32 * rb + (lane & 31), pn_d_feat(PN_NFB * wave + fb, g, lane), v[0], v[1], v[2], v[3]);
            }
    }
}

// Waves run in lockstep: a value one lane wrote to LDS is visible to the other lanes of the SAME wave after the LDS queue has drained.
// (The host emulation runs lanes as fibers: there it has to be a rendezvous.)
#ifdef PN_EMU
#define PN_WAVE_LDS_SYNC() __syncthreads()
#define PN_EMU_MATCH_WAVE_SYNC() __syncthreads()      // for the threads of a workgroup that skip a phase with a PN_WAVE_LDS_SYNC in it
#else
#define PN_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PN_EMU_MATCH_WAVE_SYNC() ((void)0)
#endif
__device__ __forceinline__ float pn_softplus_b(float x) {
#ifdef PN_EMU
    return x > 20.f ? x : log1pf(expf(x));
#else
    return x > 20.f ? x : __logf(1.0f + __expf(x));
#endif
}
__device__ __forceinline__ float pn_sigmoid_b(float x) {
#ifdef PN_EMU
    return x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
#else
    return x > 20.f ? 1.f : __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#endif
}

// ---- the tile's front in ONE pass over the h4 tile, for K in {1, 2, 4, 8} (the forward's f_tail mapping: thread -> columns
// 8 (tid & 31) .. + 7 of rows 8 (tid >> 5) .. + 7, i.e. whole samples): alpha-head backward (d f . h4 per row by the transposing butterfly,
// d conf, d x), then dY4 = (w d f + d x W5) * LeakyReLU'(h4) written IN PLACE -- a thread only rewrites the elements it alone reads, so
// there is no barrier between the two halves; d x of a row reaches the 32 lanes that share the row through LDS inside the wave.
// dfr[2 j], dfr[2 j + 1] = the thread's 8 columns of the d f row of its sample j (KC >= 4), dfb0 = where they come from.  gw5 = d W5 of the thread's 8 columns (scaled).
// KC = 0 (round 5): any other K (12 / 6 / 3 of the Barn configuration, ...).  Nothing in the front is per SAMPLE except which d f row a tile row
// uses, so the same pass serves every K with that index taken at run time (j = row / K - r0 / K; the thread's 8 rows then span up to three
// samples, whose d f values come straight from memory like KC = 2, 1); the two-pass form it replaces read the tile twice with a barrier between.
// MIX: dY4 leaves in the mixed format of mixq.h: h (nearest f16) in plane 0 and, in place of the thread's 16 bytes of the h4 residual plane, the
// group's e4m3 unit [q8(h) x 8 | q8(m 2^11) x 8] -- a thread still rewrites only bytes it alone reads
template <int KC, bool MIX = false>
__device__ __forceinline__ void b_front(const BwdArgs &a, char *X, const float *w5s, const float *wrow, const float *wnrm, const float *dsg, const float *xrow,
                                        float *draw, const int *sidx, const int *prow, const float4 (&dfr)[4], const float *dfb0, float S, float invS, int tid,
                                        float (&gw5)[8], float &gb5t, unsigned kinv = 0u) {
    const int lane = tid & 63, cg = tid & 31, r0 = 8 * (tid >> 5);
    const int j0 = KC == 0 ? pn_row_div(r0, kinv) : 0;
    // (zero-one regulariser riding on the conf atomic below: the row's confidence, requested now, used behind the butterfly)
    float zo_conf = 0.f;
    if (a.zo_gs && (lane & 3) == 0) {
        const int rz = prow[r0 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)];
        if (rz >= 0) zo_conf = a.conf[rz];
    }
    // the d f values of row i's sample: from the registers filled before the barrier (KC = 8, 4: one or two samples per thread), else
    // (KC = 2, 1: the rare classes, four or eight samples per thread) straight from memory
    auto df_of = [&](int j, float4 &ga, float4 &gb) {
        if (KC >= 4) { ga = dfr[2 * j]; gb = dfr[2 * j + 1]; }
        else { ga = *reinterpret_cast<const float4 *>(dfb0 + j * PN_H); gb = *reinterpret_cast<const float4 *>(dfb0 + j * PN_H + 4); }
    };
    float pd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = r0 + i, j = KC == 0 ? pn_row_div(r, kinv) - j0 : i / (KC == 0 ? 1 : KC);
        const uint4 h = *reinterpret_cast<const uint4 *>(X + r * PN_XRS + cg * 16);
        const uint4 m = *reinterpret_cast<const uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16);
        float4 ga, gb;
        df_of(j, ga, gb);
        float s = 0.f;
        s = pn_fma2_lo(h.x, m.x, ga.x, s); s = pn_fma2_hi(h.x, m.x, ga.y, s); s = pn_fma2_lo(h.y, m.y, ga.z, s); s = pn_fma2_hi(h.y, m.y, ga.w, s);
        s = pn_fma2_lo(h.z, m.z, gb.x, s); s = pn_fma2_hi(h.z, m.z, gb.y, s); s = pn_fma2_lo(h.w, m.w, gb.z, s); s = pn_fma2_hi(h.w, m.w, gb.w, s);
        pd[i] = s;
    }
    {   // transposing butterfly (f_tail): afterwards a lane holds the d f . h4 of row r0 + 4 b4 + 2 b3 + b2
        const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float send = b4 ? pd[j] : pd[j + 4], keep = b4 ? pd[j + 4] : pd[j]; pd[j] = keep + __shfl_xor(send, 16, 64); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const float send = b3 ? pd[j] : pd[j + 2], keep = b3 ? pd[j + 2] : pd[j]; pd[j] = keep + __shfl_xor(send, 8, 64); }
        { const float send = b2 ? pd[0] : pd[1], keep = b2 ? pd[1] : pd[0]; pd[0] = keep + __shfl_xor(send, 4, 64); }
        pd[0] += __shfl_xor(pd[0], 2, 64);
        pd[0] += __shfl_xor(pd[0], 1, 64);
        const int r = r0 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
        if ((lane & 3) == 0) {
            float dr = 0.f;
            if (sidx[r] >= 0) {
                const float x = xrow[r], dotf = pd[0] * S;
                const float alpha = pn_softplus_b(x), sg = pn_sigmoid_b(x);
                const int rp = prow[r];
                // w = wn * clamp(conf) with a straight-through clamp (gradiant_clamp, point_aggregators.py:722-724)
                if (rp >= 0) atomicAdd(&a.g_conf[rp], (dsg[r] * alpha + dotf) * wnrm[r] * invS + (a.zo_gs ? pn_zero_one_grad(zo_conf, a.zo_eps, a.zo_gs[0]) : 0.f));
                dr = dsg[r] * wrow[r] * sg;
            }
            draw[r] = dr;
        }
    }
    PN_WAVE_LDS_SYNC();
    const float4 wa = *reinterpret_cast<const float4 *>(w5s + 8 * cg), wb = *reinterpret_cast<const float4 *>(w5s + 8 * cg + 4);
    const float w5[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    float drsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = r0 + i, j = KC == 0 ? pn_row_div(r, kinv) - j0 : i / (KC == 0 ? 1 : KC);
        const uint4 h = *reinterpret_cast<const uint4 *>(X + r * PN_XRS + cg * 16);
        const uint4 m = *reinterpret_cast<const uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16);
        const bool live = sidx[r] >= 0;          // a row without a sample: its d f values are whatever memory held (0 x NaN is NaN)
        const float dr = draw[r], w = wrow[r] * S;
        drsum += dr;
        const float hv[8] = {pn_h_lo(h.x) + pn_h_lo(m.x), pn_h_hi(h.x) + pn_h_hi(m.x), pn_h_lo(h.y) + pn_h_lo(m.y), pn_h_hi(h.y) + pn_h_hi(m.y),
                             pn_h_lo(h.z) + pn_h_lo(m.z), pn_h_hi(h.z) + pn_h_hi(m.z), pn_h_lo(h.w) + pn_h_lo(m.w), pn_h_hi(h.w) + pn_h_hi(m.w)};
        float4 ga, gb;
        df_of(j, ga, gb);
        const float g[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            o[c] = live ? (w * g[c] + dr * w5[c]) * pn_lrelu_grad(hv[c]) : 0.f;
            gw5[c] += dr * hv[c];
        }
        unsigned oh[4], om[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) pn_split2_sat(o[2 * c], o[2 * c + 1], oh[c], om[c]);
        *reinterpret_cast<uint4 *>(X + r * PN_XRS + cg * 16) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
        if (MIX) *reinterpret_cast<uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16) =
                     make_uint4(pn_q8x4(oh[0], oh[1], 1.0f), pn_q8x4(oh[2], oh[3], 1.0f), pn_q8x4(om[0], om[1], PN_MIX_MSC), pn_q8x4(om[2], om[3], PN_MIX_MSC));
        else *reinterpret_cast<uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16) = make_uint4(om[0], om[1], om[2], om[3]);
    }
    if (cg == 0) gb5t += drsum;
}

#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_bwd);
#endif
// MIX: the four input-gradient GEMMs in the mixed format of mixq.h (f16 h.h + e4m3 cross terms; the default since round 6)
template <bool WG2, bool MIX = false>
__global__ __launch_bounds__(PN_NTHR, PN_NW / 2) void k_agg_backward(BwdArgs a) {
    static_assert(!(MIX && WG2), "the two-plane weight-gradient mode keeps f16x3.h's arithmetic everywhere");
    pn_mode_saturate();
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char *X = smem_b;
    float *wrow = reinterpret_cast<float *>(smem_b + BL_ROW), *wnrm = wrow + PN_TILE, *draw = wnrm + PN_TILE, *dsg = draw + PN_TILE, *xrow = dsg + PN_TILE;
    int *sidx = reinterpret_cast<int *>(xrow + PN_TILE), *prow = sidx + PN_TILE;
    float *w5s = reinterpret_cast<float *>(smem_b + BL_W5);
    float *dx = reinterpret_cast<float *>(smem_b);
    const int tid0 = threadIdx.x;
    const int K = a.K, TS = a.TS;
    const unsigned kinv = pn_kinv(K);
    const int Ns = a.cls_info[PN_CI_COUNT + a.cls];
    const long long vb = a.cls_info[PN_CI_VBASE + a.cls], tb = a.cls_info[PN_CI_TBASE + a.cls];
    a.sv.dfs += vb * PN_H;
    const long long ntiles = ((long long)Ns + TS - 1) / TS;
    const float *P = a.params;
    const char *img = reinterpret_cast<const char *>(a.packed);
    float S, invS;
    pn_scale_from_bits(a.sv.gscale[0], S, invS);
    if (tid0 < PN_H) w5s[tid0] = P[PO_W5 + tid0];
    float gw5[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // d W5 of columns 8 (tid & 31) .. + 7 (scaled)
    float gb5t = 0.f;
    f32x16 acc[PN_NFB][2];
    // row metadata of the tile (threads 0..63: one row each), fetched ONE TILE AHEAD: the d sigma of a row hangs off its sample id, and
    // two dependent HBM round trips at the top of every tile were 4 of the 6 us of the load phase
#ifdef PN_TILE_BLOCKED       // (dev A/B: see the forward)
    const long long per_wg = (ntiles + gridDim.x - 1) / gridDim.x, stride = 1, tile_first = blockIdx.x * per_wg;
    const long long tile_last = tile_first + per_wg < ntiles ? tile_first + per_wg : ntiles;
#else
    const long long stride = gridDim.x, tile_first = blockIdx.x, tile_last = ntiles;
#endif
    int4 rm_cur = make_int4(-1, -1, 0, 0);
    float ar_cur = 0.f;
    if (tid0 < PN_TILE && tile_first < tile_last) {
        rm_cur = a.sv.rmeta[(tb + tile_first) * PN_TILE + tid0];
        ar_cur = a.sv.arow[(tb + tile_first) * PN_TILE + tid0];
    }
    PN_TR_ITER_DECL;
    for (long long tile = tile_first; tile < tile_last; tile += stride) {
        PN_TR_ITER_NEXT;
        int tid = threadIdx.x;                          // (recomputed per tile: see the forward)
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        // roles in the 8-wave organisation (the forward's): the first PN_ETHR threads (waves 0..3) own the front (alpha head backward + dY4),
        // the last PN_ETHR threads (waves 4..7) the rows' embedding gradients (4 threads per tile row: row, q)
        const bool ew = PN_NTHR == PN_ETHR || tid < PN_ETHR, bw = PN_NTHR == PN_ETHR || tid >= PN_NTHR - PN_ETHR;
        const int bt = bw ? tid - (PN_NTHR - PN_ETHR) : 0, row = bt / TPR, q = bt % TPR;
        const long long gtile = tb + tile;
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 0); PN_TR_HWID(pn_trace_bwd);
        // ---- load: sign words, d sigma of the rows, the h4 planes (-> LDS before the barrier), the next tile's row metadata: one burst
        unsigned m1[PN_NFB], m2[PN_NFB], m3[PN_NFB];
#pragma unroll
        for (int fb = 0; fb < PN_NFB; ++fb) {
            const long long o = (long long)(PN_NFB * wave + fb) * 64 + lane;
            m1[fb] = a.sv.lmask[(gtile * 3 + 0) * 512 + o]; m2[fb] = a.sv.lmask[(gtile * 3 + 1) * 512 + o]; m3[fb] = a.sv.lmask[(gtile * 3 + 2) * 512 + o];
        }
        float dsg_v = 0.f;
        // the row's ray direction (for d dir, behind layer 3): requested with the burst -- it hangs off the sample id, and as a dependent load
        // behind the extras' barrier it was 1-2 us of one wave's latency per tile with the other three waves waiting
        float rdx = 0.f, rdy = 0.f, rdz = 0.f;
        if (tid < PN_TILE && rm_cur.x >= 0) {
            dsg_v = a.grad_decoded[(long long)rm_cur.x * 4];
            const int r = rm_cur.x / a.SR;
            rdx = a.raydir[3 * r]; rdy = a.raydir[3 * r + 1]; rdz = a.raydir[3 * r + 2];
        }
        pn_f4 h4v[4096 / PN_NTHR];
#pragma unroll
        for (int i = 0; i < 4096 / PN_NTHR; ++i) {
            const int e = tid + PN_NTHR * i, plane = e >> 11, r = (e >> 5) & 63, u = e & 31;
            h4v[i] = *reinterpret_cast<const pn_f4 *>(a.sv.h4r + ((long long)plane * a.sv.rows + gtile * PN_TILE + r) * 32 + u);
        }
        int4 rm_nxt = make_int4(-1, -1, 0, 0);
        float ar_nxt = 0.f;
        if (tid < PN_TILE && tile + stride < tile_last) {
            rm_nxt = a.sv.rmeta[(gtile + stride) * PN_TILE + tid];
            ar_nxt = a.sv.arow[(gtile + stride) * PN_TILE + tid];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tid < PN_TILE) {
            sidx[tid] = rm_cur.x; prow[tid] = rm_cur.y;
            wnrm[tid] = __int_as_float(rm_cur.z); wrow[tid] = __int_as_float(rm_cur.w);
            xrow[tid] = ar_cur;
            dsg[tid] = dsg_v * S;
        }
#pragma unroll
        for (int i = 0; i < 4096 / PN_NTHR; ++i) {
            const int e = tid + PN_NTHR * i, plane = e >> 11, r = (e >> 5) & 63, u = e & 31;
            *reinterpret_cast<pn_f4 *>(X + plane * PN_XPLANE + r * PN_XRS + u * 16) = h4v[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        // one-pass front: the d f rows of the thread's samples (they follow from the tile index alone; rows past the class's last sample
        // read allocated, unused memory and are ignored): requested before the barrier, consumed behind it
        float4 dfr[4];
        const float *dfb0 = a.sv.dfs + (tile * TS + pn_row_div(8 * ((tid >> 5) & 7), kinv)) * PN_H + 8 * (tid & 31);
        if (ew && (K == 8 || K == 4)) {
            dfr[0] = *reinterpret_cast<const float4 *>(dfb0); dfr[1] = *reinterpret_cast<const float4 *>(dfb0 + 4);
            if (K == 4) { dfr[2] = *reinterpret_cast<const float4 *>(dfb0 + PN_H); dfr[3] = *reinterpret_cast<const float4 *>(dfb0 + PN_H + 4); }
        }
        __builtin_amdgcn_sched_barrier(0);
        rm_cur = rm_nxt; ar_cur = ar_nxt;
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 1);
        const int rp = bw ? prow[row] : -1;
        float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;       // the row's embedding values (for its gradient at the end of the tile)
        if (rp >= 0) {
            const float *ep = a.emb + (long long)rp * PN_F + EPT * q;
            e0 = *reinterpret_cast<const float4 *>(ep); e1 = *reinterpret_cast<const float4 *>(ep + 4);
        }
        {
            // ---- alpha head backward + dY4 in one pass (b_front): K = 8 / 4 / 2 / 1 with compile-time sample boundaries, any other K at run time
            if (!ew) { PN_EMU_MATCH_WAVE_SYNC(); }
            else if (K == 8) b_front<8, MIX>(a, X, w5s, wrow, wnrm, dsg, xrow, draw, sidx, prow, dfr, dfb0, S, invS, tid, gw5, gb5t);
            else if (K == 4) b_front<4, MIX>(a, X, w5s, wrow, wnrm, dsg, xrow, draw, sidx, prow, dfr, dfb0, S, invS, tid, gw5, gb5t);
            else if (K == 2) b_front<2, MIX>(a, X, w5s, wrow, wnrm, dsg, xrow, draw, sidx, prow, dfr, dfb0, S, invS, tid, gw5, gb5t);
            else if (K == 1) b_front<1, MIX>(a, X, w5s, wrow, wnrm, dsg, xrow, draw, sidx, prow, dfr, dfb0, S, invS, tid, gw5, gb5t);
            else b_front<0, MIX>(a, X, w5s, wrow, wnrm, dsg, xrow, draw, sidx, prow, dfr, dfb0, S, invS, tid, gw5, gb5t, kinv);
            PN_TR(pn_trace_bwd, 2);
        }
        // (round 4: the first weight-fragment chunks of every GEMM are requested in FRONT of the barrier that precedes it -- see the forward)
        PnGemmW<16, 8, PN_NFB> W4;
        PnMixW<PN_MIX_NS, 0, 8, PN_NFB> M4;
        if constexpr (MIX) M4.prefetch(img + PKM_D4, PN_NFB * wave, lane);
        else W4.prefetch(reinterpret_cast<const uint4 *>(img + PKH_D4), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 3);
        // ---- layer 4: dY4 -> d h3
        // (every dY tile is copied out BEHIND its GEMM: stores and loads of a wave share one in-order vmcnt queue, see the forward)
        b_acc_zero(acc);
        PN_TR(pn_trace_bwd, 4);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 0, 8, PN_NFB>(X, M4, lane, acc);
        else pn_gemm_f16x3_run<16, 8, PN_NFB>(X, W4, lane, acc);
        pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.dy4k, gtile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X + PN_XPLANE, a.sv.dy4m, gtile * 8, tid);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 5);
        b_epilogue<MIX>(acc, m3, X, wave, lane);
        PnGemmW<16, 9, PN_NFB> W3;
        PnMixW<PN_MIX_NS, 0, 9, PN_NFB> M3;
        if constexpr (MIX) M3.prefetch(img + PKM_D3, PN_NFB * wave, lane);
        else W3.prefetch(reinterpret_cast<const uint4 *>(img + PKH_D3), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 6);
        // ---- layer 3: dY3 -> d h2, and the extras block (input columns 256..262 of W3), K split over the waves
        b_acc_zero(acc);
        PN_TR(pn_trace_bwd, 7);
        // the extras block first, K split over the first four waves; their partial sums go straight to the tile's free bytes (80 per row and
        // plane behind column 255, which no GEMM reads: no barrier needed) -- wave w -> plane w >> 1, 32-byte slot w & 1; a lane holds
        // features 4 (l >> 5) .. + 3 of rows (l & 31), (l & 31) + 32.  Done before the main GEMM so that its accumulators are dead by then
        // (an 8-wave workgroup has 128 registers per wave).
        if (wave < 4) {
            f32x16 acce[1][2];
            b_acc_zero(acce);
            // (mixed format: wave w takes superchunk w of the block -- the same K split, four f16 chunks and two e4m3 MFMAs per row block)
            if constexpr (MIX) pn_gemm_mix<1, 0, 9, 1>(X, img + PKM_D3, 8, lane, acce, wave);
            else pn_gemm_f16x3<4, 9, 1>(X, reinterpret_cast<const uint4 *>(img + PKH_D3), 8, lane, acce, 4 * wave);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
                *reinterpret_cast<float4 *>(X + (wave >> 1) * PN_XPLANE + (32 * rb + (lane & 31)) * PN_XRS + 512 + (wave & 1) * 32 + (lane >> 5) * 16) =
                    make_float4(acce[0][rb][0], acce[0][rb][1], acce[0][rb][2], acce[0][rb][3]);
        }
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 0, 9, PN_NFB>(X, M3, lane, acc);
        else pn_gemm_f16x3_run<16, 9, PN_NFB>(X, W3, lane, acc);
        pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.dy3k, gtile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X + PN_XPLANE, a.sv.dy3m, gtile * 8, tid);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 8);
        b_epilogue<MIX>(acc, m2, X, wave, lane);
        PN_LDS_BARRIER();
        if (tid < PN_TILE) {        // d colour, d dir of the row's point from the extras' gradient
            const int p = prow[tid];
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float4 a0 = *reinterpret_cast<const float4 *>(X + (w >> 1) * PN_XPLANE + tid * PN_XRS + 512 + (w & 1) * 32);
                const float4 a1 = *reinterpret_cast<const float4 *>(X + (w >> 1) * PN_XPLANE + tid * PN_XRS + 512 + (w & 1) * 32 + 16);
                u.x += a0.x; u.y += a0.y; u.z += a0.z; u.w += a0.w; v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w;
            }
            float vx, vy, vz, gx = 0.f, gy = 0.f, gz = 0.f;
            if (p >= 0) {
                rot3b(a.cam.rw2c, rdx, rdy, rdz, true, vx, vy, vz);
                // features (q - v, q . v) with q = dir @ Rw2c^T  ->  d q = dex[3:6] + dex[6] * v ; d dir = d q @ Rw2c
                rot3b(a.cam.rw2c, u.w + v.z * vx, v.x + v.z * vy, v.y + v.z * vz, false, gx, gy, gz);
            }
            // Round 5: the six values go back into the row's (consumed) extras slots and leave three lanes per point: a point's 12 bytes of
            // d colour / d dir are one or two 32-byte sectors instead of three (every lane of the old form hit its own sector)
            float *slot = reinterpret_cast<float *>(X + tid * PN_XRS + 512);
            *reinterpret_cast<float4 *>(slot) = make_float4(u.x * invS, u.y * invS, u.z * invS, gx * invS);
            *reinterpret_cast<float2 *>(slot + 4) = make_float2(gy * invS, gz * invS);
        }
        PN_WAVE_LDS_SYNC();
        if (tid < PN_TILE) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int e = 64 * j + tid, r = (e * 21846) >> 16, c = e - 3 * r;          // e / 3 for e < 192
                const int p = prow[r];
                if (p >= 0) {
                    const float *slot = reinterpret_cast<const float *>(X + r * PN_XRS + 512);
                    atomicAdd(&a.g_color[3 * p + c], slot[c]);
                    atomicAdd(&a.g_dir[3 * p + c], slot[3 + c]);
                }
            }
        }
        PnGemmW<16, 8, PN_NFB> W2;
        PnMixW<PN_MIX_NS, 0, 8, PN_NFB> M2;
        if constexpr (MIX) M2.prefetch(img + PKM_D2, PN_NFB * wave, lane);
        else W2.prefetch(reinterpret_cast<const uint4 *>(img + PKH_D2), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 9);
        // ---- layer 2: dY2 -> d h1
        // (Measured and rejected: pulling the next tile's h4 planes / d f rows / sign words into L2 from here with 4-byte LDS-DMA
        //  reads, one per 128-byte line: 16.65 ms against 16.19 ms -- the load phase is not waiting for HBM.)
        b_acc_zero(acc);
        PN_TR(pn_trace_bwd, 10);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 0, 8, PN_NFB>(X, M2, lane, acc);
        else pn_gemm_f16x3_run<16, 8, PN_NFB>(X, W2, lane, acc);
        pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.dy2k, gtile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X + PN_XPLANE, a.sv.dy2m, gtile * 8, tid);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 11);
        b_epilogue<MIX>(acc, m1, X, wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 12);
        // ---- layer 1: dY1 -> d X0 (columns 0..223), fp32 into LDS
        b_acc_zero(acc);
        PN_TR(pn_trace_bwd, 13);
        if constexpr (MIX) {
            if (PN_NFB == 2) {
                if (wave < 3) pn_gemm_mix<PN_MIX_NS, 0, PN_MB_D1, PN_NFB>(X, img + PKM_D1, 2 * wave, lane, acc);
                else pn_gemm_mix<PN_MIX_NS, 0, PN_MB_D1, 1>(X, img + PKM_D1, 6, lane, acc);
            } else if (wave < PN_MB_D1) {
                pn_gemm_mix<PN_MIX_NS, 0, PN_MB_D1, 1>(X, img + PKM_D1, wave, lane, acc);
            }
        } else if (PN_NFB == 2) {       // seven feature blocks over four waves: 2 2 2 1
            if (wave < 3) pn_gemm_f16x3<16, PN_MB_D1, PN_NFB>(X, reinterpret_cast<const uint4 *>(img + PKH_D1), 2 * wave, lane, acc);
            else pn_gemm_f16x3<16, PN_MB_D1, 1>(X, reinterpret_cast<const uint4 *>(img + PKH_D1), 6, lane, acc);
        } else if (wave < PN_MB_D1) {      // over eight waves: one each, the last wave idle
            pn_gemm_f16x3<16, PN_MB_D1, 1>(X, reinterpret_cast<const uint4 *>(img + PKH_D1), wave, lane, acc);
        }
        pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.dy1k, gtile * 8, tid);
        if (WG2) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X + PN_XPLANE, a.sv.dy1m, gtile * 8, tid);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 14);
#pragma unroll
        for (int fb = 0; fb < PN_NFB; ++fb)
            if (PN_NFB * wave + fb < PN_MB_D1) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(dx + (32 * rb + (lane & 31)) * LDDX + pn_d_feat(PN_NFB * wave + fb, g, lane)) =
                            make_float4(acc[fb][rb][4 * g], acc[fb][rb][4 * g + 1], acc[fb][rb][4 * g + 2], acc[fb][rb][4 * g + 3]);
            }
        PN_LDS_BARRIER();
        PN_TR(pn_trace_bwd, 15);
        // ---- embedding gradient through [e | PE3(e)]: d e = dX[e] + sum_f 2^f (dX[sin_f] cos_f - dX[cos_f] sin_f)
        // Round 5: the 32 values of a row leave as ONE lane-contiguous burst.  The (row, q) threads that form them own 8 columns each, so an
        // atomic instruction used to touch 64 different 32-byte sectors (one dword each; the memory side turns every one into a 32-byte
        // read-modify-write: 1 KB of write traffic per row, 7.3 GB per step at the bench configuration).  The values now go back into the
        // row's own d X0 columns (which only their thread reads), and the wave -- it holds 16 whole rows -- sends them out two rows per
        // instruction: lanes 0..31 = the 128-byte gradient row of one point, lanes 32..63 the next row's.
        if (bw && rp >= 0) {
            float *dr_ = dx + row * LDDX;
            const float e[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float go[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int dd = EPT * q + i;
                float s[3], c[3];
                pn_pe_octaves<3>(e[i], s, c);
                float g = dr_[dd], fr = 1.f;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const float2 t = *reinterpret_cast<const float2 *>(dr_ + PN_F + dd * 6 + 2 * f);
                    g += fr * (t.x * c[f] - t.y * s[f]);
                    fr *= 2.f;
                }
                go[i] = g * invS;
            }
#pragma unroll
            for (int i = 0; i < EPT; i += 4) *reinterpret_cast<float4 *>(dr_ + EPT * q + i) = make_float4(go[i], go[i + 1], go[i + 2], go[i + 3]);
        }
        PN_WAVE_LDS_SYNC();
        if (bw) {
            const int r0w = (row & ~15) + (lane >> 5), col = lane & 31;          // the wave's 16 rows, two per instruction
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0w + 2 * j, p = prow[r];
                if (p >= 0) atomicAdd(&a.g_emb[(long long)p * PN_F + col], dx[r * LDDX + col]);
            }
        }
        PN_TR(pn_trace_bwd, 16);
    }
    // flush the register-resident partial sums: a workgroup that had no tile has nothing to add; the others first add up their eight row
    // sets in LDS -- atomics of every workgroup on the same 256 addresses serialise in L2 (2048 per workgroup made the two small sample
    // classes' launches 1 ms each)
    if (tile_first >= tile_last) return;
    {
        const int cg = tid0 & 31, rs = tid0 >> 5;
        float *red = reinterpret_cast<float *>(smem_b);            // [8 row sets][256 columns] over the tile's space
        PN_LDS_BARRIER();
        if (tid0 < PN_ETHR) {                                      // (the front's threads hold the sums)
#pragma unroll
            for (int c = 0; c < 8; ++c) red[rs * PN_H + 8 * cg + c] = gw5[c];
            if (cg == 0) red[8 * PN_H + rs] = gb5t;
        }
        PN_LDS_BARRIER();
        if (tid0 >= PN_H) return;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r * PN_H + tid0];
        atomicAdd(&a.gparams[PO_W5 + tid0], t * invS);
        if (tid0 == 0) {
            float b = 0.f;
            for (int r = 0; r < 8; ++r) b += red[8 * PN_H + r];
            atomicAdd(&a.gparams[PO_B5], b * invS);
        }
    }
}

// ------------------------------------------------------------------------------ weight gradients (aggregator and colour layers)
// dW[m][n] = sum_rows dY[row][m] X[row][n] on the f16 pipe: both operands arrive as ready-made k-major fragments of ONE f16 plane each,
// rounded to nearest by their producers (f16x3.h), so the kernel is glds -> LDS -> ds_read_b128 -> MFMA with no conversion work:
//   256 x (256 + 32) block (all of dW plus a 32-column tail) in the accumulators of 8 waves (2 (M) x 4 (N), 4 x 2 tiles each +
//   one tail tile), split-K over the rows (one workgroup per CU), a ring of LDS stages filled by global_load_lds_dwordx4 (every operand
//   of a stage is one contiguous run in HBM and in LDS) NST - 1 stages ahead: the kernel is HBM-bound by design (1 KB per row and layer
//   against ~0.17 us of MFMA work per 16 rows), so what matters is that ~100 KB per CU are in flight at all times.
//   The LDS-DMA count is tracked by hand (s_waitcnt vmcnt(N) + raw s_barrier: __syncthreads() would drain the queue).
// Error budget of the one-plane operands: every term dY[r][m] X[r][n] carries two independent, unbiased relative roundings of
// <= 2^-11 (rms 1.1e-4 each); a dW element sums 10^5 .. 10^7 of them, so its error is 1.6e-4 x sqrt(sum t^2) -- for the elements that
// matter (|sum t| of the order of the tensor's largest) 1e-6 .. 1e-5 of their value.  Until round 3 X came as two planes (22 bits) and
// dY as one: the second plane of X doubled the X stream of this kernel and of the forward's stores without changing what the sum's error
// is made of (measured on the benchmark configuration before / after: tests/test_gpu_bench_config.py prints both).  The forward values
// and the input-gradient chain keep 22-bit operands: there every element is ONE dot product of 256 terms, here it is a sum over millions
// of rows.
// Tail: NFB == 288: the operand's own columns 256..287 (distance encoding of X0 / layer-3 extras, and the ONES column whose
// "weight gradient" is the bias gradient); NFB == 256: a constant ones fragment (bias gradient of layers 2 and 4).
#ifdef PN_EMU
#define PN_WAIT_VMCNT(n) ((void)0)
#else
#define PN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif
#ifndef PN_WG_RS
#define PN_WG_RS 32             // rows per stage of the aggregator layers' weight-gradient GEMMs (dev A/B: -DPN_WG_RS=16 -DPN_WG_NST=4 is round 2's)
#endif
#ifndef PN_WG_NST
#define PN_WG_NST 4
#endif
template <int N> __device__ __forceinline__ void pn_wait_vm_stages(int stages) {       // wait until at most `stages` x N of this wave's loads are outstanding
    if (stages >= 3) PN_WAIT_VMCNT(3 * N);
    else if (stages == 2) PN_WAIT_VMCNT(2 * N);
    else if (stages == 1) PN_WAIT_VMCNT(N);
    else PN_WAIT_VMCNT(0);
}
// MF = features of dY (rows of dW): 256 (aggregator layers: 2 (M) x 4 (N) waves, 4 x 2 tiles each) or 128 (colour layers: 2 x 2 tiles each for
// NFB = 288, 2 x 1 for NFB = 128).  Tail tile (one per wave, m-tile by wave): NFB == 288 -> the operand's columns 256..287; MF == 256 and
// NFB == 256 -> the constant ones fragment (bias gradient); MF == 128 and NFB == 128 -> none (those bias sums come from k_color_backward).
// tiles_per = rows per tile of *d_tiles / 16 (the aggregator counts 64-row tiles, and so does the colour MLP).
// RS = rows per stage (16 or 32), NST = ring depth.  A stage is one barrier, one LDS round trip for the fragments and RS / 16 x 9 MFMAs per wave,
// and the two waves of a SIMD run it in lockstep: with 16-row stages a stage took ~1.1 us whatever it streamed -- the kernel was bound by
// the per-stage rendezvous, not by HBM.  32-row stages halve the rendezvous per row; four ring slots of 32 .. 34 KB keep ~100 KB per CU in flight.
// TWO (round 5): the fp32-class mode (pnerf_set_wgrad_planes(2)) in ONE pass -- a stage holds both planes of both operands,
// [dYh | Xh | dYm | Xm], and every fragment pair gets the three products dYh Xh + dYh Xm + dYm Xh.  Round 4 ran the one-plane kernel three times
// per layer (six plane streams, three partial reductions); this streams four.
template <int NFB, int MF, int RS, int NST, bool TWO = false>
__global__ __launch_bounds__(512) void k_wgrad_f16(const uint4 *__restrict__ A, const uint4 *__restrict__ B,
                                                   const int *__restrict__ d_tiles, float *__restrict__ partial,
                                                   const uint4 *__restrict__ Am = nullptr, const uint4 *__restrict__ Bm = nullptr) {
    constexpr int RG = RS / 8;                                // row groups (8 rows) per stage
    constexpr int AU = RG * MF, BU = RG * NFB;                // units (16 B) of one plane of a stage
    constexpr int HALF = AU + BU;                             // [dY | X] of one plane pair
    constexpr int STAGE = TWO ? 2 * HALF : HALF;              // TWO: [dYh | Xh | dYm | Xm]
    constexpr int NI = STAGE / 64, NIW = (NI + 7) / 8;        // wave-instructions per stage, per wave (the last ones are padded)
    static_assert((NST - 2) * NIW <= 63 && NST >= 3 && NST <= 5, "vmcnt is a 6-bit count");
    constexpr int MTW = MF / 64, NTW = NFB >= 256 ? 2 : 1;    // m-tiles / main n-tiles per wave
    constexpr int NMAIN = 4 * NTW * 32;                       // columns covered by the main tiles
    constexpr bool TAIL_B = NFB > NMAIN, TAIL_ONES = !TAIL_B && MF == 256;
    static_assert(STAGE % 64 == 0, "a stage is a whole number of 1 KB wave copies");
    extern __shared__ __attribute__((aligned(16))) uint4 smem_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // 16-row stages are dealt to the workgroups round-robin (stage blockIdx.x + gridDim.x * i): at any moment the chip reads ONE contiguous
    // run of each plane instead of 256 runs that are a fixed, channel-aliasing distance apart.  The tiles actually used are known on the
    // device only.
    const long long stages = (long long)(*d_tiles) * (PN_TILE / RS);
    const int nst = stages > (long long)blockIdx.x ? (int)((stages - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    f32x16 acc[MTW][NTW], acct;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acct[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j][r] = 0.f;
    }
    // stage s -> buffer: wave-instruction j copies units [64 j, 64 j + 64) of the stage's concatenated runs; every wave issues
    // exactly NIW instructions (a wave without a piece of its own re-reads the stage's first KB into the pad slot)
    auto issue = [&](int s, int buf) {
        const long long rg = (long long)RG * ((long long)blockIdx.x + (long long)gridDim.x * s);
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const bool pad = wave + 8 * i >= NI;
            const int j = pad ? 0 : wave + 8 * i, u0 = 64 * j;
            const uint4 *dst = smem_w + (pad ? NST * STAGE : buf * STAGE + u0);
            const uint4 *src;
            if (u0 < AU) src = A + rg * MF + u0;
            else if (u0 < HALF) src = B + rg * NFB + (u0 - AU);
            else if (u0 < HALF + AU) src = Am + rg * MF + (u0 - HALF);
            else src = Bm + rg * NFB + (u0 - HALF - AU);
            __builtin_amdgcn_global_load_lds(src + lane, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    pn_h8 ones = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((lane & 31) == 0) ones = pn_h8{1, 1, 1, 1, 1, 1, 1, 1};
    // the wave's tail tile: m-tile 4 wm + wn of 8 (MF = 256), m-tile 2 wm + wn of 4 for the waves wn < 2 (MF = 128)
    const bool has_tail = (TAIL_B || TAIL_ONES) && (MF == 256 || wn < 2);
    if (nst > 0) {
#pragma unroll
        for (int i = 0; i < NST - 1; ++i) if (nst > i) issue(i, i);
        for (int s = 0; s < nst; ++s) {
            // stage s has landed for every wave, and every wave is done with stage s - 1 (whose buffer the next issue overwrites)
            pn_wait_vm_stages<NIW>(nst - 1 - s < NST - 2 ? nst - 1 - s : NST - 2);
            __builtin_amdgcn_s_barrier();
            if (s + NST - 1 < nst) issue(s + NST - 1, (s + NST - 1) % NST);
            const uint4 *st = smem_w + (s % NST) * STAGE;
#pragma unroll
            for (int kk = 0; kk < RS / 16; ++kk) {            // 16 rows (two row groups) per MFMA k-step
                const uint4 *fa = st + (2 * kk + (lane >> 5)) * MF + (lane & 31);
                const uint4 *fb = st + AU + (2 * kk + (lane >> 5)) * NFB + (lane & 31);
                pn_h8 ah[MTW], bh[NTW], am[TWO ? MTW : 1], bm[TWO ? NTW : 1];
#pragma unroll
                for (int i = 0; i < MTW; ++i) ah[i] = __builtin_bit_cast(pn_h8, fa[(MTW * wm + i) * 32]);
#pragma unroll
                for (int i = 0; i < NTW; ++i) bh[i] = __builtin_bit_cast(pn_h8, fb[(NTW * wn + i) * 32]);
                if (TWO) {
#pragma unroll
                    for (int i = 0; i < MTW; ++i) am[i] = __builtin_bit_cast(pn_h8, fa[HALF + (MTW * wm + i) * 32]);
#pragma unroll
                    for (int i = 0; i < NTW; ++i) bm[i] = __builtin_bit_cast(pn_h8, fb[HALF + (NTW * wn + i) * 32]);
                }
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                if (TWO) {
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bm[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[i], bh[j], acc[i][j], 0, 0, 0);
                }
                if (has_tail) {
                    pn_h8 tah, tam = ah[0];
                    if (MF == 256) tah = wn == 0 ? ah[0] : wn == 1 ? ah[1] : wn == 2 ? ah[MTW > 2 ? 2 : 0] : ah[MTW > 3 ? 3 : 0];
                    else tah = wn == 0 ? ah[0] : ah[MTW > 1 ? 1 : 0];
                    if (TWO) {
                        if (MF == 256) tam = wn == 0 ? am[0] : wn == 1 ? am[TWO && MTW > 1 ? 1 : 0] : wn == 2 ? am[TWO && MTW > 2 ? 2 : 0] : am[TWO && MTW > 3 ? 3 : 0];
                        else tam = wn == 0 ? am[0] : am[TWO && MTW > 1 ? 1 : 0];
                    }
                    if (TAIL_B) {
                        acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(pn_h8, fb[NMAIN]), acct, 0, 0, 0);
                        if (TWO) {
                            acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(pn_h8, fb[HALF + NMAIN]), acct, 0, 0, 0);
                            acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tam, __builtin_bit_cast(pn_h8, fb[NMAIN]), acct, 0, 0, 0);
                        }
                    } else {
                        acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, ones, acct, 0, 0, 0);
                        if (TWO) acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tam, ones, acct, 0, 0, 0);      // (the bias gradient takes both planes of dY)
                    }
                }
            }
        }
    }
    float *out = partial + (size_t)blockIdx.x * 256 * 288;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (MTW * wm + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(size_t)m * 288 + (NTW * wn + j) * 32 + (lane & 31)] = acc[i][j][r];
            }
    if (has_tail) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (MTW * wm + wn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(size_t)m * 288 + 256 + (lane & 31)] = acct[r];
        }
    }
}

// grad_w[m * ldc + n] += invS * sum_c partial[c][m][n]  (m < Mreal, n < Nreal);  grad_b[m] += invS * sum_c partial[c][m][bias_col]
__global__ __launch_bounds__(256) void k_wgrad_reduce_f16(const float *__restrict__ partial, int chunks, int Mreal, int Nreal, int bias_col, const unsigned *__restrict__ gscale,
                                                          float *__restrict__ grad, int dst_w, int ldc, int dst_b) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const size_t stride = (size_t)256 * 288;
    float s = 0.f;
    if (e < Mreal * 288) {
#pragma unroll 8
        for (int c = w; c < chunks; c += 4) s += partial[c * stride + e];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < Mreal * 288) {
        float S, invS;
        pn_scale_from_bits(gscale[0], S, invS);
        const int m = e / 288, n = e - m * 288;
        const float v = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * invS;
        if (n < Nreal) grad[dst_w + m * ldc + n] += v;
        else if (n == bias_col) grad[dst_b + m] += v;
    }
}

template <int NFB, int MF, int RS, int NST, bool TWO = false>
int launch_wgrad_f16(const uint4 *A, const uint4 *B, const int *d_tiles, long long rows_max, float *partial, const unsigned *gscale,
                     float *grad, int dst_w, int ldc, int Nreal, int bias_col, int dst_b, hipStream_t s, const uint4 *Am = nullptr, const uint4 *Bm = nullptr) {
    int chunks = WG_CHUNKS;
    const long long tiles = rows_max / PN_TILE;
    if (tiles < chunks) chunks = (int)(tiles > 0 ? tiles : 1);
    if ((size_t)chunks * 256 * 288 > PARTIAL_FLOATS) return PNERF_E_WS;
    if (TWO && (!Am || !Bm)) return PNERF_E_INVAL;
    constexpr size_t lds = ((size_t)NST * (TWO ? 2 : 1) * (RS / 8) * (MF + NFB) + 64) * 16;   // the stages [dY | X] (TWO: both planes of each) + the pad slot
    static_assert(lds <= 160 * 1024, "wgrad ring");
    if (hipFuncSetAttribute((const void *)k_wgrad_f16<NFB, MF, RS, NST, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL((k_wgrad_f16<NFB, MF, RS, NST, TWO>), dim3(chunks), dim3(512), lds, s, A, B, d_tiles, partial, Am, Bm); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce_f16, dim3(pn_cdiv((long long)MF * 288, 64)), dim3(256), 0, s, partial, chunks, MF, Nreal, bias_col, gscale, grad, dst_w, ldc, dst_b);
    PN_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------ layer-1 weight gradients with X0 REBUILT from the embedding
// dW1 = dY1^T X0,  X0 = [e (32) | PE3(e) (192) | PE5(dists) (60) | 1 | 0 0 0].  As a saved f16 plane X0 is 576 bytes per row written by
// the forward and read back here, and both kernels pay for every byte (dropping the forward's saves altogether makes it 21 % faster).  224 of the 288 columns are a function of the point's 32 embedding
// values (128 bytes): the fused path therefore saves only the LAST 64 columns (distance encoding, ones column: x0t, 128 B/row) and this
// kernel rebuilds columns 0 .. 223 of every 16-row stage in LDS from the gathered embedding rows.
// Everything that comes from memory arrives by LDS-DMA with per-lane source addresses, TWO wave-instructions per wave and iteration, so
// that the hand-counted vmcnt waits of the rings stay exact:
//   iteration i issues   dY1 of a later stage (8 pieces of 1 KB, one per wave) | row metadata (sample, point) of a later stage (wave 0) |
//                        the 16 embedding rows of a later stage (waves 1, 2), addressed with metadata that landed D iterations earlier |
//                        the saved last 64 columns of a later stage (waves 0, 1) | a pad piece (the rest)   (distances: WX_D below);
//   iteration i waits for everything issued up to iteration i - D, multiplies stage i (dY1 ring slot x [built columns | saved columns])
//   and builds columns 0 .. 223 of stage i + 1 into the other X0 buffer: thread -> (row group, embedding dim, row in group); the 7 values
//   are rounded to nearest f16 (the forward's saved plane is the nearest f16 of its 22-bit value: the same number except where the
//   22-bit rounding crosses a tie) and written as 2-byte elements of the k-major units.
// Rows without a point (tile padding, the classes' gap tiles whose metadata is whatever memory held) are clamped to a valid point:
// their dY rows are zero, X0 only has to be finite (the gap tiles of x0t are zeroed by k_cls_zero_gaps).
struct WgX0Args {
    const int4 *rmeta;
    const float *emb;
    const uint4 *x0t;                 // [rows / 8][64] units (one plane): columns 224 .. 287 of X0
    int n_points;
};
// Rows per stage WX_RS (16 or 32) and prefetch distance WX_D (iterations): what iteration i consumes was issued at iteration i - WX_D.
//   iteration j issues   dY1 and saved columns of stage j + D | embedding rows of stage j + D + 1 (its X0 is built at iteration j + D) |
//                        row metadata of stage j + 2 D + 2 (read at the END of iteration j + D, for the gather that iteration j + D + 1 issues)
// Measured (round 3, one box, 7.1 M rows).  With two-plane operands: 16-row stages 2.1-2.2 ms with distances of 3 and 7 alike (42 vs 98 KB in
// flight per CU), 1.83 ms with the X0 arithmetic removed; 32-row stages (LDS then held a distance of 2) 2.56 ms.  With one-plane operands
// (half the MFMAs, half the LDS per stage): 16 rows / distance 7 1.42 ms, distance 10 1.45; 32 rows / distance 3 1.28, distance 4 1.29 --
// the per-stage rendezvous is what is left (849 stages of 1.5 us per workgroup; the HBM floor of its 5.4 GB is 1.0 ms).
#ifndef PN_WX_RS
#define PN_WX_RS 32
#endif
#ifndef PN_WX_D
#define PN_WX_D (PN_WX_RS == 32 ? 3 : 7)
#endif
constexpr int WX_RS = PN_WX_RS, WX_RG = WX_RS / 8;  // rows / row groups per stage
constexpr int WX_AU = WX_RG * PN_H;                 // dY1 units of a stage (one plane)
constexpr int WX_NB = 224;                          // columns rebuilt here
constexpr int WX_BU = WX_RG * WX_NB;                // their units of a stage
constexpr int WX_TU = WX_RG * 64;                   // saved-column units of a stage: [rg][64]
constexpr int WX_GU = WX_RS * 8;                    // embedding units of a stage: [rg][piece 8][row 8] x 16 B
constexpr int WX_NA = WX_AU / 512;                  // dY1 pieces (1 KB) per wave and stage
constexpr int WX_L = WX_NA + 2;                     // wave-instructions per wave and iteration: dY1 | a saved-column piece or pad | metadata / embedding rows / pad
constexpr int WX_D = PN_WX_D;
constexpr int WX_NST = WX_D + 1, WX_RMD = 2 * WX_D + 2, WX_GD = WX_D + 1;    // ring depth of dY1 / saved columns; how far ahead metadata / embedding rows are issued
constexpr int WX_RM_SLOTS = WX_D + 2, WX_G_SLOTS = WX_D + 1;
static_assert(WX_L * (WX_D - 1) <= 63 && WX_D >= 2, "vmcnt is a 6-bit count");
static_assert(WX_RS == 16 || WX_RS == 32, "rows per stage");
constexpr int WX_OFF_T = WX_NST * WX_AU, WX_OFF_B = WX_OFF_T + WX_NST * WX_TU, WX_OFF_RM = WX_OFF_B + 2 * WX_BU,
              WX_OFF_G = WX_OFF_RM + WX_RM_SLOTS * WX_RS, WX_OFF_PAD = WX_OFF_G + WX_G_SLOTS * WX_GU, WX_UNITS = WX_OFF_PAD + 64;
static_assert(WX_UNITS * 16 <= 160 * 1024, "k_wgrad_x0 LDS");

__device__ __forceinline__ void wx_store_h16(char *buf, int rg, int f, int rlow, _Float16 v) {
    *reinterpret_cast<_Float16 *>(buf + ((rg * WX_NB + f) * 16) + rlow * 2) = v;
}

__global__ __launch_bounds__(512) void k_wgrad_x0(const uint4 *__restrict__ A, WgX0Args g, const int *__restrict__ d_tiles, float *__restrict__ partial) {
    constexpr int MF = PN_H, MTW = 4;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_x[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const long long stages = (long long)(*d_tiles) * (PN_TILE / WX_RS);
    const int nst = stages > (long long)blockIdx.x ? (int)((stages - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    auto stage_rg = [&](int s) { return (long long)WX_RG * ((long long)blockIdx.x + (long long)gridDim.x * s); };       // first row group (8 rows) of local stage s
    f32x16 acc[MTW][2], acct;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acct[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MTW; ++i) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; }
    }
    char *lds = reinterpret_cast<char *>(smem_x);
    int p_gather = 0;
    // ---- issue helpers: a call is a fixed number of wave-instructions of the calling wave (a stage past the end re-reads a valid source into the pad)
    auto issue_a = [&](int s) {                         // WX_NA + 1 instructions: the wave's dY1 pieces, then its saved-column piece (or a pad piece)
        const bool ok = s < nst;
#pragma unroll
        for (int i = 0; i < WX_NA; ++i) {
            const int piece = wave + 8 * i;
            const uint4 *src = A + (ok ? stage_rg(s) * MF + 64 * piece : 0) + lane;
            const uint4 *dst = smem_x + (ok ? (s % WX_NST) * WX_AU + 64 * piece : WX_OFF_PAD);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        const bool okt = ok && wave < WX_RG;            // piece rg = 64 units of the saved last-64-column plane
        const uint4 *src = g.x0t + (okt ? (stage_rg(s) + wave) * 64 : 0) + lane;
        const uint4 *dst = smem_x + (okt ? WX_OFF_T + (s % WX_NST) * WX_TU + wave * 64 : WX_OFF_PAD);
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto issue_second = [&](int s_rm, int s_g) {        // ONE instruction: metadata (wave 0), embedding rows (waves 1 .. RG), pad (the rest)
        if (wave == 0) {                                // the stage's (sample, point, ., .) records, one per row
            const bool ok = s_rm < nst;
            const int4 *src = g.rmeta + (ok ? stage_rg(s_rm) * 8 : 0) + (lane & (WX_RS - 1));
            const uint4 *dst = smem_x + (ok ? WX_OFF_RM + (s_rm % WX_RM_SLOTS) * WX_RS : WX_OFF_PAD);
            if (lane < WX_RS) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else if (wave <= WX_RG) {                     // embedding rows of row group wave - 1 of stage s_g: lane = piece * 8 + row -> slot [rg][piece][row] x 16 B
            const bool ok = s_g < nst;
            const int p = ok ? p_gather : 0;            // (read from the metadata ring and clamped one iteration ahead: next_point)
            const float *src = g.emb + (long long)p * PN_F + (lane >> 3) * 4;
            const uint4 *dst = smem_x + (ok ? WX_OFF_G + (s_g % WX_G_SLOTS) * WX_GU + (wave - 1) * 64 : WX_OFF_PAD);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else {                                        // a pad piece, so that every wave counts the same number of loads per iteration
            __builtin_amdgcn_global_load_lds(A + lane, (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_PAD), 16, 0, 0);
        }
    };
    // the point of this lane's embedding row of stage s_g (waves 1 .. RG), clamped: looked up at the END of an iteration for the gather the
    // next one issues, so that the LDS round trip is not in front of the next stage's MFMAs
    auto next_point = [&](int s_g) {
        if (wave >= 1 && wave <= WX_RG && s_g < nst) {
            const int p = reinterpret_cast<const int4 *>(smem_x + WX_OFF_RM + (s_g % WX_RM_SLOTS) * WX_RS)[8 * (wave - 1) + (lane & 7)].y;
            p_gather = p < 0 ? 0 : (p >= g.n_points ? g.n_points - 1 : p);
        }
    };
    // ---- columns 0 .. 223 of local stage s -> buffer s & 1: thread -> (row group, embedding dim, row in group), WX_RS / 16 items each
    auto build_into = [&](char *buf, const char *slot) {
#pragma unroll
        for (int it = 0; it < WX_RS / 16; ++it) {
            const int item = tid + 512 * it, rg = item >> 8, d = (item >> 3) & 31, rlow = item & 7;
            // (clamped like every other producer of an f16 plane, pn_split2_sat: a plain cast turns |e| > 65504 into inf, and inf x the zero
            //  dY of a padded row is NaN in dW1)
            const float e = __builtin_amdgcn_fmed3f(*reinterpret_cast<const float *>(slot + rg * 1024 + ((d >> 2) * 8 + rlow) * 16 + (d & 3) * 4), -65504.f, 65504.f);
            float sn[3], cs[3];
            pn_pe_octaves<3>(e, sn, cs);
            const int f = PN_F + d * 6;
            wx_store_h16(buf, rg, d, rlow, (_Float16)e);
#pragma unroll
            for (int o = 0; o < 3; ++o) { wx_store_h16(buf, rg, f + 2 * o, rlow, (_Float16)sn[o]); wx_store_h16(buf, rg, f + 2 * o + 1, rlow, (_Float16)cs[o]); }
        }
    };
    auto build = [&](int s) {
        build_into(lds + (size_t)(WX_OFF_B + (s & 1) * WX_BU) * 16, lds + (size_t)(WX_OFF_G + (s % WX_G_SLOTS) * WX_GU) * 16);
    };
    if (nst > 0) {
        // ---- prologue: the metadata the first embedding gathers need (stages 0 .. D + 1: a ring's worth), drained; then dY1 / saved columns
        // of stages 0 .. D - 1 and the embedding rows of stages 0 .. D, drained; then the metadata of stages D + 2 .. 2 D + 1, drained
        if (wave == 0)
            for (int s = 0; s < WX_RM_SLOTS; ++s) issue_second(s, 0);
        PN_WAIT_VMCNT(0);
        __syncthreads();
        for (int s = 0; s < WX_D; ++s) issue_a(s);
        if (wave >= 1 && wave <= WX_RG)
            for (int s = 0; s < WX_GD; ++s) { next_point(s); issue_second(0, s); }
        PN_WAIT_VMCNT(0);
        __syncthreads();                                  // (the metadata slots of stages 0 .. D have been read: D of them may be overwritten)
        if (wave == 0)
            for (int s = WX_RM_SLOTS; s < WX_RMD; ++s) issue_second(s, 0);
        PN_WAIT_VMCNT(0);
        __syncthreads();
        build(0);
        next_point(WX_GD);                                // (its metadata was issued in the first prologue step: landed)
        // ---- main loop.  The scalar unit is shared by the CU's eight waves: ring slots and source addresses are carried as counters and
        // running pointers (one add per iteration) instead of being recomputed from the stage index (a 64-bit multiply, a modulo and a bounds
        // select each: the loop body had 137 scalar instructions per wave for 18 MFMAs), and the bounds checks exist only in the last
        // 2 D + 2 iterations, which go through the checked issue helpers.
        const long long stride_a = (long long)WX_RG * gridDim.x * MF, stride_t = (long long)WX_RG * gridDim.x * 64, stride_rm = (long long)WX_RG * gridDim.x * 8;
        const uint4 *pa = A + stage_rg(WX_D) * MF + 64 * wave + lane;
        const uint4 *pt = g.x0t + (stage_rg(WX_D) + wave) * 64 + lane;
        const int4 *prm = g.rmeta + stage_rg(WX_RMD) * 8 + (lane & (WX_RS - 1));
        int ia = WX_D % WX_NST, ig = WX_GD % WX_G_SLOTS, irm = WX_RMD % WX_RM_SLOTS;      // slots the issues of iteration 0 write
        int ra = 0, rgs = 1 % WX_G_SLOTS, rrm = (1 + WX_GD) % WX_RM_SLOTS;                  // slots iteration 0 reads: dY1 / saved columns, embedding rows of stage 1, metadata of stage 1 + GD
        const int n_main = nst - WX_RMD > 0 ? nst - WX_RMD : 0;                             // iterations whose every issue is inside the run
        auto bump = [](int &c, int n) { c = c + 1 == n ? 0 : c + 1; };
        for (int s = 0; s < nst; ++s) {
            // everything this wave issued up to iteration s - D has landed (D - 1 iterations' worth of wave-instructions may be outstanding),
            // its X0 writes of iteration s - 1 are done; then everybody's
            PN_WAIT_VMCNT(WX_L * (WX_D - 1));
            PN_LDS_BARRIER();
            if (s < n_main) {
#pragma unroll
                for (int i = 0; i < WX_NA; ++i)
                    __builtin_amdgcn_global_load_lds(pa + 512 * i, (__attribute__((address_space(3))) void *)(smem_x + ia * WX_AU + 64 * (wave + 8 * i)), 16, 0, 0);
                if (wave < WX_RG) __builtin_amdgcn_global_load_lds(pt, (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_T + ia * WX_TU + wave * 64), 16, 0, 0);
                else __builtin_amdgcn_global_load_lds(A + lane, (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_PAD), 16, 0, 0);
                if (wave == 0) {
                    if (lane < WX_RS) __builtin_amdgcn_global_load_lds(prm, (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_RM + irm * WX_RS), 16, 0, 0);
                } else if (wave <= WX_RG) {
                    __builtin_amdgcn_global_load_lds(g.emb + (long long)p_gather * PN_F + (lane >> 3) * 4,
                                                     (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_G + ig * WX_GU + (wave - 1) * 64), 16, 0, 0);
                } else {
                    __builtin_amdgcn_global_load_lds(A + lane, (__attribute__((address_space(3))) void *)(smem_x + WX_OFF_PAD), 16, 0, 0);
                }
            } else {
                issue_a(s + WX_D);
                issue_second(s + WX_RMD, s + WX_GD);
            }
            pa += stride_a; pt += stride_t; prm += stride_rm;
#pragma unroll
            for (int kk = 0; kk < WX_RS / 16; ++kk) {         // 16 rows (two row groups) per MFMA k-step
                const uint4 *fa = smem_x + ra * WX_AU + (2 * kk + (lane >> 5)) * MF + (lane & 31);
                const uint4 *fb = smem_x + WX_OFF_B + (s & 1) * WX_BU + (2 * kk + (lane >> 5)) * WX_NB + (lane & 31);                // built columns
                const uint4 *ft = smem_x + WX_OFF_T + ra * WX_TU + (2 * kk + (lane >> 5)) * 64 + (lane & 31);                         // saved columns
                pn_h8 ah[MTW], bh[2];
#pragma unroll
                for (int i = 0; i < MTW; ++i) ah[i] = __builtin_bit_cast(pn_h8, fa[(MTW * wm + i) * 32]);
                const pn_h8 tah = __builtin_bit_cast(pn_h8, fa[(MTW * wm + wn) * 32]);
                // n-tiles 2 wn, 2 wn + 1 of the 9: tiles 0 .. 6 = built columns, tile 7 = saved columns 224 .. 255, tile 8 (the tail, one m-tile per wave) = 256 .. 287
                bh[0] = __builtin_bit_cast(pn_h8, fb[(2 * wn) * 32]);
                bh[1] = __builtin_bit_cast(pn_h8, wn < 3 ? fb[(2 * wn + 1) * 32] : ft[0]);
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                acct = __builtin_amdgcn_mfma_f32_32x32x16_f16(tah, __builtin_bit_cast(pn_h8, ft[32]), acct, 0, 0, 0);
            }
            // columns 0 .. 223 of stage s + 1 -> the other X0 buffer (unconditional: past the last stage it rebuilds from a stale -- finite --
            // slot into the buffer nobody reads; no branch, so the arithmetic can be scheduled between the MFMAs instead of behind them)
            build_into(lds + (size_t)(WX_OFF_B + ((s + 1) & 1) * WX_BU) * 16, lds + (size_t)(WX_OFF_G + rgs * WX_GU) * 16);
            // the point of the embedding row the NEXT iteration gathers (stage s + 1 + GD; its metadata was issued at iteration s - D: landed)
            if (wave >= 1 && wave <= WX_RG && s + 1 + WX_GD < nst) {
                const int p = reinterpret_cast<const int4 *>(smem_x + WX_OFF_RM + rrm * WX_RS)[8 * (wave - 1) + (lane & 7)].y;
                p_gather = p < 0 ? 0 : (p >= g.n_points ? g.n_points - 1 : p);
            }
            bump(ia, WX_NST); bump(ig, WX_G_SLOTS); bump(irm, WX_RM_SLOTS); bump(ra, WX_NST); bump(rgs, WX_G_SLOTS); bump(rrm, WX_RM_SLOTS);
        }
    }
    float *out = partial + (size_t)blockIdx.x * 256 * 288;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (MTW * wm + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(size_t)m * 288 + (2 * wn + j) * 32 + (lane & 31)] = acc[i][j][r];
            }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (MTW * wm + wn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(size_t)m * 288 + 256 + (lane & 31)] = acct[r];
    }
}

int launch_wgrad_x0(const uint4 *A, const WgX0Args &g, const int *d_tiles, long long rows_max, float *partial, const unsigned *gscale,
                    float *grad, hipStream_t s) {
    int chunks = WG_CHUNKS;
    const long long tiles = rows_max / PN_TILE;
    if (tiles < chunks) chunks = (int)(tiles > 0 ? tiles : 1);
    if ((size_t)chunks * 256 * 288 > PARTIAL_FLOATS) return PNERF_E_WS;
    constexpr size_t lds = (size_t)WX_UNITS * 16;
    if (hipFuncSetAttribute((const void *)k_wgrad_x0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL(k_wgrad_x0, dim3(chunks), dim3(512), lds, s, A, g, d_tiles, partial); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce_f16, dim3(pn_cdiv((long long)PN_H * 288, 64)), dim3(256), 0, s, partial, chunks, PN_H, PN_IN1, PN_ONES1, gscale, grad, (int)PO_W1, (int)PN_IN1, (int)PO_B1);
    PN_CHECK_LAUNCH();
    return 0;
}
}  // namespace

size_t pn_wgrad_partials_bytes() { return pn_align(PARTIAL_FLOATS * sizeof(float)); }

namespace {
// the empty neighbor slots of the hit rays read point 0 (neural_points.py:709): (#rays hit x SR x K - #valid neighbor slots) identical terms of the
// zero-one regulariser's conf gradient, added once (query counters [1] and [3])
__global__ void k_zero_one_empty(const float *__restrict__ conf, const int *__restrict__ counters, long long slots_per_ray, const float *__restrict__ gs, float eps,
                                 float *__restrict__ g_conf) {
    const long long n_empty = (long long)counters[1] * slots_per_ray - (long long)counters[3];
    if (n_empty <= 0) return;
    const float g = pn_zero_one_grad(conf[0], eps, gs[0]);
    if (g != 0.f) atomicAdd(&g_conf[0], g * (float)n_empty);
}
}  // namespace

int pn_agg_backward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                           const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                           const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                           const float *d_decoded, const float *d_weight, const float *d_grad_decoded,
                           const PnSaved &sv, long long n_valid, float *d_grad_params, const pnerf_point_grads *pg,
                           float *d_partials, bool x0_saved, hipStream_t s) {
    BwdArgs a;
    a.cam = *cam; a.params = d_params; a.packed = (const float4 *)d_packed; a.raydir = d_raydir;
    a.pidx = d_sample_pidx; a.valid_list = d_valid_list; a.counters = d_counters;
    a.SR = SR; a.K = K; a.TS = pn_tile_samples(K); a.cap_samples = n_valid;
    a.decoded = d_decoded; a.weight = d_weight; a.grad_decoded = d_grad_decoded; a.sv = sv;
    a.emb = pts->embedding;
    a.gparams = d_grad_params; a.g_emb = pg->embedding; a.g_conf = pg->conf; a.g_dir = pg->dir; a.g_color = pg->color;
    a.conf = pts->conf; a.zo_gs = x0_saved ? nullptr : pg->zero_one_gscale; a.zo_eps = pg->zero_one_eps;       // (the fused render path only)
    if (a.zo_gs && !a.conf) return PNERF_E_INVAL;
    if (!a.g_emb || !a.g_conf || !a.g_dir || !a.g_color) return PNERF_E_INVAL;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const long long ctiles = (n_valid + PN_CTILE - 1) / PN_CTILE;
    const int grid_c = (int)(ctiles < 3 * ncu ? (ctiles > 0 ? ctiles : 1) : 3 * ncu);       // 40 KB of LDS: three workgroups per CU
    const size_t lds_c = CB_BYTES, lds_a = BL_BYTES;
    const bool wg2 = sv.wg2 != 0;                      // two-plane weight-gradient mode (the forward of this step ran in it: same process-wide setting)
    if (wg2) x0_saved = true;
    const void *kcb = wg2 ? (const void *)k_color_backward<true> : (const void *)k_color_backward<false>;
    const bool mix = !wg2 && (pn_mix_mask() & 4);      // mixq.h: e4m3 cross terms in the input-gradient chain
    const void *kab = wg2 ? (const void *)k_agg_backward<true> : mix ? (const void *)k_agg_backward<false, true> : (const void *)k_agg_backward<false>;
    if (hipFuncSetAttribute(kcb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipFuncSetAttribute(kab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    // the forward left the class partition of the valid samples in the saved area (aggregate.hip: pn_classify)
    a.cls_list = sv.cls_list; a.cls_info = sv.cls_info; a.valid_list = sv.cls_list;
    // the scale of this call's gradients (a power of two derived on the device from max |d decoded| over the valid samples)
    if (hipMemsetAsync(sv.gscale, 0, 4 * sizeof(unsigned), s) != hipSuccess) return PNERF_E_LAUNCH;
    {
        const long long blocks = (n_valid + 255) / 256;
        hipLaunchKernelGGL(k_grad_max, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, s, sv.cls_list, d_counters, (long long)n_valid, d_grad_decoded, sv.gscale);
    }
    if (a.zo_gs) hipLaunchKernelGGL(k_zero_one_empty, dim3(1), dim3(1), 0, s, a.conf, d_counters, (long long)SR * K, a.zo_gs, a.zo_eps, a.g_conf);
    { PnProfScope prof(PNK_COLOR_BWD, s);
      if (wg2) hipLaunchKernelGGL(k_color_backward<true>, dim3(grid_c), dim3(256), lds_c, s, a);
      else hipLaunchKernelGGL(k_color_backward<false>, dim3(grid_c), dim3(256), lds_c, s, a); }
    int kc[PN_NCLS];
    const int ncls = pn_class_slots(K, kc);
    { PnProfScope prof(PNK_AGG_BWD, s);
      for (int j = 0; j < ncls; ++j) {
          a.cls = j; a.K = kc[j]; a.TS = pn_tile_samples(kc[j]);
          const long long tiles = (n_valid + a.TS - 1) / a.TS;                    // worst-case grid, two workgroups per CU
          const int grid_a = (int)(tiles < 2LL * ncu ? (tiles > 0 ? tiles : 1) : 2LL * ncu);
          if (wg2) hipLaunchKernelGGL(k_agg_backward<true>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
          else if (mix) hipLaunchKernelGGL((k_agg_backward<false, true>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
          else hipLaunchKernelGGL(k_agg_backward<false>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
      } }
    PN_CHECK_LAUNCH();
    // the point gradients are final here: let a data-parallel caller start their all-reduce behind this event while the
    // weight-gradient GEMMs below still run
    if (pg->ready_event && hipEventRecord((hipEvent_t)pg->ready_event, s) != hipSuccess) return PNERF_E_LAUNCH;
    // weight gradients over every tile of every class (+ their zero padding tiles): the tile count lives on the device, the host
    // bound is the allocation.  samples: only the first n_valid rows of fs / pe / c1.. exist -- the GEMM masks the rest of the
    // last colour tile (0 * stale bits could be NaN)
    const long long rows = sv.rows, smp = n_valid;
    const int *dt = sv.cls_info + PN_CI_TILES;
    int rc;
    float *g = d_grad_params;
    const int *ct = sv.cls_info + PN_CI_CTILES;
    if (wg2) {
        // two-plane weight-gradient mode: dW = dYh^T Xh + dYh^T Xm + dYm^T Xh in ONE pass per layer (k_wgrad_f16<.., TWO>: a stage holds both planes of
        // both operands; round 4 ran the one-plane kernel three times per layer).  16-row stages: four planes of 32 rows would not leave four ring slots.
        // The bias gradients take both planes of dY: the operands' own ones column (layers 1 and 3) has a zero residual, the constant-ones tail
        // (layers 2 and 4) is multiplied with dYh and dYm.
        if ((rc = launch_wgrad_f16<PN_NF1, PN_H, 16, 4, true>(sv.dy1k, sv.x0k, dt, rows, d_partials, sv.gscale, g, PO_W1, PN_IN1, PN_IN1, PN_ONES1, PO_B1, s, sv.dy1m, sv.x0m))) return rc;
        if ((rc = launch_wgrad_f16<PN_H, PN_H, 16, 4, true>(sv.dy2k, sv.h1k, dt, rows, d_partials, sv.gscale, g, PO_W2, PN_H, PN_H, PN_H, PO_B2, s, sv.dy2m, sv.h1m))) return rc;
        if ((rc = launch_wgrad_f16<PN_NF1, PN_H, 16, 4, true>(sv.dy3k, sv.h2k, dt, rows, d_partials, sv.gscale, g, PO_W3, PN_IN3, PN_IN3, PN_ONES3, PO_B3, s, sv.dy3m, sv.h2m))) return rc;
        if ((rc = launch_wgrad_f16<PN_H, PN_H, 16, 4, true>(sv.dy4k, sv.h3k, dt, rows, d_partials, sv.gscale, g, PO_W4, PN_H, PN_H, PN_H, PO_B4, s, sv.dy4m, sv.h3m))) return rc;
        // the three colour layers: samples instead of neighbor rows, 128 output features; their bias gradients were summed by k_color_backward
        if ((rc = launch_wgrad_f16<PN_NF1, PN_HC, 16, 4, true>(sv.dc1k, sv.xck, ct, sv.samples, d_partials, sv.gscale, g, PO_WC1, PN_INC, PN_INC, -1, 0, s, sv.dc1m, sv.xcm))) return rc;
        if ((rc = launch_wgrad_f16<PN_HC, PN_HC, 16, 4, true>(sv.dc2k, sv.c1k, ct, sv.samples, d_partials, sv.gscale, g, PO_WC2, PN_HC, PN_HC, -1, 0, s, sv.dc2m, sv.c1m))) return rc;
        if ((rc = launch_wgrad_f16<PN_HC, PN_HC, 16, 4, true>(sv.dc3k, sv.c2k, ct, sv.samples, d_partials, sv.gscale, g, PO_WC3, PN_HC, PN_HC, -1, 0, s, sv.dc3m, sv.c2m))) return rc;
        (void)smp;
        return 0;
    }
    if (x0_saved) {        // the stand-alone aggregator (perspective coordinates from its caller): X0 planes saved by the forward
        if ((rc = launch_wgrad_f16<PN_NF1, PN_H, PN_WG_RS, PN_WG_NST>(sv.dy1k, sv.x0k, dt, rows, d_partials, sv.gscale, g, PO_W1, PN_IN1, PN_IN1, PN_ONES1, PO_B1, s))) return rc;
    } else {               // the fused path: X0 rebuilt from the gather
        (void)d_sample_loc; (void)R;
        WgX0Args wa;
        wa.rmeta = sv.rmeta; wa.emb = pts->embedding; wa.x0t = sv.x0k; wa.n_points = pts->n;
        if ((rc = launch_wgrad_x0(sv.dy1k, wa, dt, rows, d_partials, sv.gscale, g, s))) return rc;
    }
    if ((rc = launch_wgrad_f16<PN_H, PN_H, PN_WG_RS, PN_WG_NST>(sv.dy2k, sv.h1k, dt, rows, d_partials, sv.gscale, g, PO_W2, PN_H, PN_H, PN_H, PO_B2, s))) return rc;
    if ((rc = launch_wgrad_f16<PN_NF1, PN_H, PN_WG_RS, PN_WG_NST>(sv.dy3k, sv.h2k, dt, rows, d_partials, sv.gscale, g, PO_W3, PN_IN3, PN_IN3, PN_ONES3, PO_B3, s))) return rc;
    if ((rc = launch_wgrad_f16<PN_H, PN_H, PN_WG_RS, PN_WG_NST>(sv.dy4k, sv.h3k, dt, rows, d_partials, sv.gscale, g, PO_W4, PN_H, PN_H, PN_H, PO_B4, s))) return rc;
    // the three colour layers: samples instead of neighbor rows, 128 output features; their bias gradients were summed by k_color_backward
    (void)smp;
    if ((rc = launch_wgrad_f16<PN_NF1, PN_HC, 16, 4>(sv.dc1k, sv.xck, ct, sv.samples, d_partials, sv.gscale, g, PO_WC1, PN_INC, PN_INC, -1, 0, s))) return rc;
    if ((rc = launch_wgrad_f16<PN_HC, PN_HC, 16, 4>(sv.dc2k, sv.c1k, ct, sv.samples, d_partials, sv.gscale, g, PO_WC2, PN_HC, PN_HC, -1, 0, s))) return rc;
    if ((rc = launch_wgrad_f16<PN_HC, PN_HC, 16, 4>(sv.dc3k, sv.c2k, ct, sv.samples, d_partials, sv.gscale, g, PO_WC3, PN_HC, PN_HC, -1, 0, s))) return rc;
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_bwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_bwd), bytes < sizeof(pn_trace_bwd) ? bytes : sizeof(pn_trace_bwd)) == hipSuccess ? 0 : -1;
}
#endif
