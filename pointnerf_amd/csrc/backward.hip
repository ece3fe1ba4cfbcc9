// backward.hip -- backward of the aggregator: colour MLP dgrad, per-neighbor MLP dgrad + gather
// scatter-add, and the weight-gradient GEMMs.
//
// The reference gets all of this from torch.autograd over ~60 ATen ops (loss.backward() in
// models/mvs_points_volumetric_model.py:98-118): cuBLAS dgrad/wgrad per nn.Linear, dense
// index_select backward into [1,N,F] buffers, boolean-mask scatter backward.  Here:
//   k_color_backward : 64 valid samples per tile; d rgb -> colour chain dgrad on MFMA -> d f[256]
//   k_agg_backward   : TS samples x K rows per tile; (d sigma, d f) -> alpha head, K-weighted sums,
//                      block3/block1 dgrad on MFMA, PE chain rule, atomic scatter-add into the
//                      embedding / colour / dir / conf gradients of the touched points only
//   k_wgrad          : dW = dY^T X as split-K MFMA GEMMs over the saved activations, full dW tile
//                      resident in accumulators, deterministic partial-sum reduction
// LeakyReLU masks come from the saved post-activations (sign(post) == sign(pre)).
#include "mlp_common.h"

namespace {
constexpr int LDH = 260;
constexpr int LDC = 132;
constexpr int WG_CHUNKS = 256;                 // split-K factor of the wgrad GEMMs
constexpr size_t PARTIAL_FLOATS = (size_t)WG_CHUNKS * PN_H * PN_H;

struct BwdArgs {
    pnerf_camera cam;
    const float *params;
    const float4 *packed;
    const float *raydir;
    const int *pidx, *valid_list, *counters;
    int SR, K, TS;
    long long cap_samples;
    const float *decoded, *weight, *grad_decoded;
    PnSaved sv;
    float *gparams;
    float *g_emb, *g_conf, *g_dir, *g_color;
};

__device__ __forceinline__ void rot3b(const float *M, float x, float y, float z, bool transpose, float &ox, float &oy, float &oz) {
    if (!transpose) { ox = x * M[0] + y * M[3] + z * M[6]; oy = x * M[1] + y * M[4] + z * M[7]; oz = x * M[2] + y * M[5] + z * M[8]; }
    else { ox = x * M[0] + y * M[1] + z * M[2]; oy = x * M[3] + y * M[4] + z * M[5]; oz = x * M[6] + y * M[7] + z * M[8]; }
}

// ------------------------------------------------------------------------------ colour backward
constexpr int COLB_LDS_FLOATS = 2 * PN_CTILE * LDC + PN_CTILE * 4;

__global__ __launch_bounds__(256, 1) void k_color_backward(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *D1 = smem;                          // [64][LDC]
    float *D2 = D1 + PN_CTILE * LDC;            // [64][LDC]
    float *draw = D2 + PN_CTILE * LDC;          // [64][4] d(pre-sigmoid colour)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;
    const int cc = tid & 127, half = tid >> 7;
    float gw4[3] = {0.f, 0.f, 0.f}, gb3 = 0.f, gb2 = 0.f, gb1 = 0.f, gb4 = 0.f;

    for (long long tile = blockIdx.x; tile * PN_CTILE < Ns; tile += gridDim.x) {
        const long long grow0 = tile * PN_CTILE;
        __syncthreads();
        if (tid < PN_CTILE) {
            const long long vs = grow0 + tid;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            if (vs < Ns) {
                const long long si = a.valid_list[vs];
                const float *o = a.decoded + si * 4, *g = a.grad_decoded + si * 4;
                // rgb = sigmoid(raw) * 1.002 - 0.001  ->  d raw = d rgb * 1.002 * s (1 - s)
                const float s0 = (o[1] + 0.001f) / 1.002f, s1 = (o[2] + 0.001f) / 1.002f, s2 = (o[3] + 0.001f) / 1.002f;
                d0 = g[1] * 1.002f * s0 * (1.f - s0); d1 = g[2] * 1.002f * s1 * (1.f - s1); d2 = g[3] * 1.002f * s2 * (1.f - s2);
            }
            draw[tid * 4] = d0; draw[tid * 4 + 1] = d1; draw[tid * 4 + 2] = d2; draw[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        // d c3 = (d raw @ Wc4) * lrelu'(c3) ; accumulate d Wc4, d bc4
        {
            const float w0 = P[PO_WC4 + cc], w1 = P[PO_WC4 + PN_HC + cc], w2 = P[PO_WC4 + 2 * PN_HC + cc];
            _Pragma("unroll 4") for (int row = half; row < PN_CTILE; row += 2) {
                const float d0 = draw[row * 4], d1 = draw[row * 4 + 1], d2 = draw[row * 4 + 2];
                const float c3 = a.sv.c3[(grow0 + row) * PN_HC + cc];
                const float v = (d0 * w0 + d1 * w1 + d2 * w2) * pn_lrelu_grad(c3);
                D1[row * LDC + cc] = v;
                a.sv.dc3[(grow0 + row) * PN_HC + cc] = v;
                gw4[0] += d0 * c3; gw4[1] += d1 * c3; gw4[2] += d2 * c3;
                gb3 += v;
            }
            if (tid < 3) _Pragma("unroll 4") for (int row = 0; row < PN_CTILE; ++row) gb4 += draw[row * 4 + tid];
        }
        __syncthreads();
        f32x16 acc[2][1];
        pn_acc_init_bias<2, 1>(acc, nullptr, wave, lane);
        pn_tile_gemm<2, 1>(D1, LDC, PN_HC / 8, a.packed + PK_DC3 / 4, wave, lane, acc);
        pn_acc_to_lds<2, 1, false>(acc, D2, LDC, wave, lane);
        __syncthreads();
        pn_tile_mask_pass<PN_CTILE, PN_HC>(D2, LDC, a.sv.c2, PN_HC, a.sv.dc2, PN_HC, grow0, tid);
        __syncthreads();
        _Pragma("unroll 4") for (int row = half; row < PN_CTILE; row += 2) gb2 += D2[row * LDC + cc];
        pn_acc_init_bias<2, 1>(acc, nullptr, wave, lane);
        pn_tile_gemm<2, 1>(D2, LDC, PN_HC / 8, a.packed + PK_DC2 / 4, wave, lane, acc);
        pn_acc_to_lds<2, 1, false>(acc, D1, LDC, wave, lane);
        __syncthreads();
        pn_tile_mask_pass<PN_CTILE, PN_HC>(D1, LDC, a.sv.c1, PN_HC, a.sv.dc1, PN_HC, grow0, tid);
        __syncthreads();
        _Pragma("unroll 4") for (int row = half; row < PN_CTILE; row += 2) gb1 += D1[row * LDC + cc];
        f32x16 acc2[2][2];
        pn_acc_init_bias<2, 2>(acc2, nullptr, wave, lane);
        pn_tile_gemm<2, 2>(D1, LDC, PN_HC / 8, a.packed + PK_DC1 / 4, wave, lane, acc2);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = pn_acc_col<2>(wave, ct, lane);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    a.sv.dfs[(grow0 + pn_acc_row(rt, reg, lane)) * PN_H + col] = acc2[rt][ct][reg];
            }
    }
    atomicAdd(&a.gparams[PO_WC4 + cc], gw4[0]);
    atomicAdd(&a.gparams[PO_WC4 + PN_HC + cc], gw4[1]);
    atomicAdd(&a.gparams[PO_WC4 + 2 * PN_HC + cc], gw4[2]);
    atomicAdd(&a.gparams[PO_BC3 + cc], gb3);
    atomicAdd(&a.gparams[PO_BC2 + cc], gb2);
    atomicAdd(&a.gparams[PO_BC1 + cc], gb1);
    if (tid < 3) atomicAdd(&a.gparams[PO_BC4 + tid], gb4);
}

// ------------------------------------------------------------------------------ aggregator backward
// One LDS buffer updated in place (GEMM -> barrier -> epilogue -> barrier), two workgroups per CU.  A workgroup running
// alone already keeps the MFMA pipe ~86 % busy (tools/mfma_probe.hip), so what matters is how long a workgroup spends
// OUTSIDE its four GEMMs: the phase timeline (tools/gpu_phase_trace.py) showed 102 us of latency-bound element-wise
// phases against 64 us of GEMM per tile.  Hence everything an element-wise phase needs from HBM/L2 is requested at the top
// of the tile (row metadata written by the forward, LeakyReLU sign words, the d f tile, the ray direction) and only
// LDS, registers and fire-and-forget stores/atomics remain after each GEMM.
constexpr int TPR = PN_TPR;                    // threads per tile row
constexpr int EPT = PN_F / TPR;                // embedding dims per thread
constexpr int CPT = PN_H / TPR;                // hidden columns per thread
constexpr int AGGB_UNION_FLOATS = 8 * PN_H;    // d f tile [TS <= 8][256] until dY4 is formed, then W3[:, 256:263] as [7][256]
constexpr int AGGB_LDS_FLOATS = PN_TILE * LDH + PN_TILE * 8 + AGGB_UNION_FLOATS + PN_H + 6 * PN_TILE;
constexpr int AGGB_WG_PER_CU = (160 * 1024) / (AGGB_LDS_FLOATS * 4);

template <int N> __device__ __forceinline__ float group_sum_b(float v) {
#pragma unroll
    for (int off = 1; off < N; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_bwd);
#endif
// DFS_LDS: the tile's d f rows (TS x 256 floats) fit the 8 KB union region (K >= 8); otherwise they are read from HBM/L2.
template <bool DFS_LDS>
__global__ __launch_bounds__(PN_NTHR, 2) void k_agg_backward(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *buf = smem;                         // [PN_TILE][LDH]  h4 -> dY4 -> dY3 -> dY2 -> dY1 -> dX0
    float *exs = buf + PN_TILE * LDH;          // [PN_TILE][8]
    float *uni = exs + PN_TILE * 8;            // [8][256] d f tile, later [7][256] W3[o][256+j]
    float *w5s = uni + AGGB_UNION_FLOATS;      // [256]
    float *wrow = w5s + PN_H;                  // [PN_TILE]
    float *wnrm = wrow + PN_TILE;
    float *draw = wnrm + PN_TILE;              // d(alpha pre-activation)
    float *dsg = draw + PN_TILE;               // d sigma of the row's sample
    int *sidx = reinterpret_cast<int *>(dsg + PN_TILE);   // row -> sample id (or -1)
    int *prow = sidx + PN_TILE;                            // row -> point id (or -1)

    const int tid = threadIdx.x;
    const int K = a.K, TS = a.TS;
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;
    if (tid < PN_H) w5s[tid] = P[PO_W5 + tid];
    // column sums (bias / alpha-head / extras gradients): thread = (column cc, row group cg)
    constexpr int CG = PN_NTHR / PN_H, RPG = PN_TILE / CG;
    const int cc0 = tid % PN_H;
    const float b5 = P[PO_B5];
    float gb1 = 0.f, gb2 = 0.f, gb3 = 0.f, gb4 = 0.f, gw5 = 0.f, gb5 = 0.f;
    float gw3e[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int H4PER = PN_TILE * 64 / PN_NTHR;  // float4 of an [PN_TILE x 256] tile per thread

#ifdef PN_PHASE_TRACE
    int titer = -1;
#endif
    for (long long tile = blockIdx.x; tile * TS < Ns; tile += gridDim.x) {
        const long long grow0 = tile * PN_TILE;
#ifdef PN_PHASE_TRACE
        ++titer;
#endif
        // thread-index-derived offsets are recomputed per tile (a few VALU ops) instead of living in registers across the
        // whole loop: hipcc otherwise hoists ~60 of them, spills, and the scratch reloads queue behind the atomics
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int cc = tl % PN_H, row_lo = (tl / PN_H) * RPG, row_hi = row_lo + RPG;
        const int rrow = tl / TPR, rq = tl % TPR;      // (row, quarter) of the row-wise phases
        const int rls = rrow / K;
        const int lane = tl & 63, wave = tl >> 6;
        __syncthreads();
        PN_TR(pn_trace_bwd, 0); PN_TR_HWID(pn_trace_bwd);
        // ---- everything this tile needs from memory, requested up front ---------------------------
        const unsigned long long m1 = a.sv.lmask[(tile * 3 + 0) * PN_NTHR + tl];
        const unsigned long long m2 = a.sv.lmask[(tile * 3 + 1) * PN_NTHR + tl];
        const unsigned long long m3 = a.sv.lmask[(tile * 3 + 2) * PN_NTHR + tl];
        if (tid < PN_TILE) {
            const int4 rm = a.sv.rmeta[grow0 + tl];
            const int si = rm.x;
            sidx[tl] = si; prow[tl] = rm.y;
            wnrm[tl] = __int_as_float(rm.z); wrow[tl] = __int_as_float(rm.w);
            dsg[tl] = si >= 0 ? a.grad_decoded[(long long)si * 4] : 0.f;
        }
        {   // h4 tile (+ d f tile): all loads issued before the first LDS store
            float4 v[H4PER], g[2];
#pragma unroll
            for (int i = 0; i < H4PER; ++i) {
                const int e = tl + i * PN_NTHR;
                v[i] = *reinterpret_cast<const float4 *>(a.sv.h4 + (grow0 + (e >> 6)) * PN_H + (e & 63) * 4);
            }
            if (DFS_LDS) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = tl + i * PN_NTHR;       // float4 e of the [TS x 256] block (rows past TS: the next tile's, unused)
                    g[i] = (e >> 6) < TS ? *reinterpret_cast<const float4 *>(a.sv.dfs + (tile * TS + (e >> 6)) * PN_H + (e & 63) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < H4PER; ++i) {
                const int e = tl + i * PN_NTHR;
                *reinterpret_cast<float4 *>(buf + (e >> 6) * LDH + (e & 63) * 4) = v[i];
            }
            if (DFS_LDS) {
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<float4 *>(uni + (tl + i * PN_NTHR) * 4) = g[i];
            }
        }
        if (tid < PN_TILE * 2) {
            const int row = tl >> 1, h = tl & 1;
            *reinterpret_cast<float4 *>(exs + row * 8 + h * 4) = *reinterpret_cast<const float4 *>(a.sv.ex + (grow0 + row) * 8 + h * 4);
        }
        __syncthreads();
        PN_TR(pn_trace_bwd, 1);
        // ---- alpha head + weight gradient ------------------------------------------------------
        const int rsi = sidx[rrow], rp = prow[rrow];
        float rdx = 0.f, rdy = 0.f, rdz = 0.f;        // ray direction of the row's sample: used after the next GEMM
        if (rq == 0 && rp >= 0) {
            const int r = rsi / a.SR;
            rdx = a.raydir[3 * r]; rdy = a.raydir[3 * r + 1]; rdz = a.raydir[3 * r + 2];
        }
        {
            const float *h = buf + rrow * LDH + rq * CPT;
            float s = 0.f, dotf = 0.f;
            if (rsi >= 0) {
                const float *df = DFS_LDS ? uni + rls * PN_H + rq * CPT : a.sv.dfs + (tile * TS + rls) * PN_H + rq * CPT;
#pragma unroll 4
                for (int c = 0; c < CPT; c += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(h + c);
                    const float4 g = *reinterpret_cast<const float4 *>(df + c);
                    s += v.x * w5s[rq * CPT + c] + v.y * w5s[rq * CPT + c + 1] + v.z * w5s[rq * CPT + c + 2] + v.w * w5s[rq * CPT + c + 3];
                    dotf += v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
                }
            }
            s = group_sum_b<TPR>(s);
            dotf = group_sum_b<TPR>(dotf);
            if (rq == 0) {
                float dr = 0.f;
                if (rsi >= 0) {
                    const float x = s + b5 - 1.0f;
                    const float alpha = x > 20.f ? x : log1pf(expf(x));
                    const float sg = x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
                    if (rp >= 0) {
                        // w = wn * clamp(conf) with a straight-through clamp (gradiant_clamp, point_aggregators.py:722-724)
                        const float dw = dsg[rrow] * alpha + dotf;
                        atomicAdd(&a.g_conf[rp], dw * wnrm[rrow]);
                    }
                    dr = dsg[rrow] * wrow[rrow] * sg;
                }
                draw[rrow] = dr;
            }
        }
        __syncthreads();
        PN_TR(pn_trace_bwd, 2);
        // ---- d W5 / d b5 (column tid) ------------------------------------------------------------
        {
            float accw = 0.f;
            _Pragma("unroll 8") for (int row = row_lo; row < row_hi; ++row) accw += draw[row] * buf[row * LDH + cc];
            gw5 += accw;
            if (tid == 0) _Pragma("unroll 8") for (int row = 0; row < PN_TILE; ++row) gb5 += draw[row];
        }
        __syncthreads();
        PN_TR(pn_trace_bwd, 3);
        // ---- dY4 = (w * d f + d raw * w5) * lrelu'(h4), in place ---------------------------------
        float w3r[7];                              // W3[o][256 + j] of column o = tid: into the union region once d f is dead
        if (tid < PN_H) {
#pragma unroll
            for (int j = 0; j < 7; ++j) w3r[j] = P[PO_W3 + tl * PN_IN3 + PN_H + j];
        }
#pragma unroll 4
        for (int i = 0; i < H4PER; ++i) {
            const int e = tl + i * PN_NTHR;
            const int row = e >> 6, c4 = e & 63;
            const int si = sidx[row];
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (si >= 0) {
                const int ls = row / K;
                const float4 hv = *reinterpret_cast<const float4 *>(buf + row * LDH + c4 * 4);
                const float4 g = DFS_LDS ? *reinterpret_cast<const float4 *>(uni + ls * PN_H + c4 * 4)
                                         : *reinterpret_cast<const float4 *>(a.sv.dfs + (tile * TS + ls) * PN_H + c4 * 4);
                const float w = wrow[row], dr = draw[row];
                o.x = (w * g.x + dr * w5s[c4 * 4]) * pn_lrelu_grad(hv.x);
                o.y = (w * g.y + dr * w5s[c4 * 4 + 1]) * pn_lrelu_grad(hv.y);
                o.z = (w * g.z + dr * w5s[c4 * 4 + 2]) * pn_lrelu_grad(hv.z);
                o.w = (w * g.w + dr * w5s[c4 * 4 + 3]) * pn_lrelu_grad(hv.w);
            }
            *reinterpret_cast<float4 *>(buf + row * LDH + c4 * 4) = o;
            *reinterpret_cast<float4 *>(a.sv.dy4 + (grow0 + row) * PN_H + c4 * 4) = o;
        }
        __syncthreads();
        PN_TR(pn_trace_bwd, 4);
        if (tid < PN_H) {
#pragma unroll
            for (int j = 0; j < 7; ++j) uni[j * PN_H + tl] = w3r[j];
        }
        // ---- block3 second layer: dY3 = (dY4 @ W4) * lrelu'(h3) ----------------------------------
        _Pragma("unroll 8") for (int row = row_lo; row < row_hi; ++row) gb4 += buf[row * LDH + cc];
        f32x16 acc[PN_MT][PN_NT];
        PN_TR(pn_trace_bwd, 5);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, nullptr, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(buf, LDH, PN_H / 8, a.packed + PK_D4 / 4, wave, lane, acc);
        PN_TR(pn_trace_bwd, 6);
        __syncthreads();
        pn_acc_to_lds<PN_MT, PN_NT, false>(acc, buf, LDH, wave, lane);
        __syncthreads();
        PN_TR(pn_trace_bwd, 7);
        pn_tile_mask_bits<PN_TILE, PN_H, PN_NTHR>(buf, LDH, m3, a.sv.dy3, PN_H, grow0, tl);
        __syncthreads();
        PN_TR(pn_trace_bwd, 8);
        // ---- block3 first layer: extras (colour, dir) + dY2 = (dY3 @ W3[:, :256]) * lrelu'(h2) ----
        _Pragma("unroll 4") for (int row = row_lo; row < row_hi; ++row) {
            const float v = buf[row * LDH + cc];
            gb3 += v;
#pragma unroll
            for (int j = 0; j < 7; ++j) gw3e[j] += v * exs[row * 8 + j];
        }
        {
            float dex[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (rp >= 0) {
                const float *dy = buf + rrow * LDH + rq * CPT;
                _Pragma("unroll 2") for (int c = 0; c < CPT; c += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(dy + c);
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const float4 w = *reinterpret_cast<const float4 *>(uni + j * PN_H + rq * CPT + c);
                        dex[j] += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) dex[j] = group_sum_b<TPR>(dex[j]);
            if (rq == 0 && rp >= 0) {
                atomicAdd(&a.g_color[3 * rp], dex[0]); atomicAdd(&a.g_color[3 * rp + 1], dex[1]); atomicAdd(&a.g_color[3 * rp + 2], dex[2]);
                float vx, vy, vz, gx, gy, gz;
                rot3b(a.cam.rw2c, rdx, rdy, rdz, true, vx, vy, vz);
                // features (q - v, q . v) with q = dir @ Rw2c^T  ->  d q = dex[3:6] + dex[6] * v ; d dir = d q @ Rw2c
                rot3b(a.cam.rw2c, dex[3] + dex[6] * vx, dex[4] + dex[6] * vy, dex[5] + dex[6] * vz, false, gx, gy, gz);
                atomicAdd(&a.g_dir[3 * rp], gx); atomicAdd(&a.g_dir[3 * rp + 1], gy); atomicAdd(&a.g_dir[3 * rp + 2], gz);
            }
        }
        PN_TR(pn_trace_bwd, 9);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, nullptr, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(buf, LDH, PN_H / 8, a.packed + PK_D3 / 4, wave, lane, acc);
        PN_TR(pn_trace_bwd, 10);
        __syncthreads();
        pn_acc_to_lds<PN_MT, PN_NT, false>(acc, buf, LDH, wave, lane);
        __syncthreads();
        PN_TR(pn_trace_bwd, 11);
        pn_tile_mask_bits<PN_TILE, PN_H, PN_NTHR>(buf, LDH, m2, a.sv.dy2, PN_H, grow0, tl);
        __syncthreads();
        PN_TR(pn_trace_bwd, 12);
        // ---- block1 second layer: dY1 = (dY2 @ W2) * lrelu'(h1) ----------------------------------
        _Pragma("unroll 8") for (int row = row_lo; row < row_hi; ++row) gb2 += buf[row * LDH + cc];
        PN_TR(pn_trace_bwd, 13);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, nullptr, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(buf, LDH, PN_H / 8, a.packed + PK_D2 / 4, wave, lane, acc);
        PN_TR(pn_trace_bwd, 14);
        __syncthreads();
        pn_acc_to_lds<PN_MT, PN_NT, false>(acc, buf, LDH, wave, lane);
        __syncthreads();
        PN_TR(pn_trace_bwd, 15);
        pn_tile_mask_bits<PN_TILE, PN_H, PN_NTHR>(buf, LDH, m1, a.sv.dy1, PN_H, grow0, tl);
        __syncthreads();
        PN_TR(pn_trace_bwd, 16);
        // ---- block1 first layer: d X0[:, :256] = dY1 @ W1[:, :256] --------------------------------
        _Pragma("unroll 8") for (int row = row_lo; row < row_hi; ++row) gb1 += buf[row * LDH + cc];
        PN_TR(pn_trace_bwd, 17);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, nullptr, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(buf, LDH, PN_H / 8, a.packed + PK_D1 / 4, wave, lane, acc);
        PN_TR(pn_trace_bwd, 18);
        __syncthreads();
        pn_acc_to_lds<PN_MT, PN_NT, false>(acc, buf, LDH, wave, lane);
        __syncthreads();
        PN_TR(pn_trace_bwd, 19);
        // ---- embedding gradient through [e | PE3(e)]: d e = dX[e] + sum_f 2^f (dX[sin] cos - dX[cos] sin)
        if (rp >= 0) {
            const float *dx = buf + rrow * LDH;
            const float *x0 = a.sv.x0 + (grow0 + rrow) * PN_IN1P + PN_F + 6 * EPT * rq;          // EPT dims * 3 freqs * 2
            float4 xs[6 * EPT / 4];
#pragma unroll
            for (int i = 0; i < 6 * EPT / 4; ++i) xs[i] = *reinterpret_cast<const float4 *>(x0 + 4 * i);
            const float *xf = reinterpret_cast<const float *>(xs);
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int dd = EPT * rq + i;
                float g = dx[dd], fr = 1.f;
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const int o = PN_F + (dd * 3 + f) * 2, l = (i * 3 + f) * 2;
                    g += fr * (dx[o] * xf[l + 1] - dx[o + 1] * xf[l]);
                    fr *= 2.f;
                }
                atomicAdd(&a.g_emb[(long long)rp * PN_F + dd], g);
            }
        }
        PN_TR(pn_trace_bwd, 20);
    }
    const int cc = cc0;
    atomicAdd(&a.gparams[PO_B1 + cc], gb1);
    atomicAdd(&a.gparams[PO_B2 + cc], gb2);
    atomicAdd(&a.gparams[PO_B3 + cc], gb3);
    atomicAdd(&a.gparams[PO_B4 + cc], gb4);
    atomicAdd(&a.gparams[PO_W5 + cc], gw5);
    if (tid == 0) atomicAdd(&a.gparams[PO_B5], gb5);
#pragma unroll
    for (int j = 0; j < 7; ++j) atomicAdd(&a.gparams[PO_W3 + cc * PN_IN3 + PN_H + j], gw3e[j]);
}

// ------------------------------------------------------------------------------ weight gradients
// partial[chunk][m][n] = sum_{r in chunk} A[r][m] B[r][n]   (A = dY [rows,lda], B = X [rows,ldb])
// Block tile (WM*MT*32) x (WN*NT*32); the whole tile lives in MFMA accumulators, operands stream
// straight from HBM/L2 in the MFMA fragment layout (lane l: row r + (l>>5), column base + (l&31):
// two 128-byte segments per load).
template <int MT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM *WN * 64) void k_wgrad(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                       long long rows, int rows_per_chunk, float *__restrict__ partial, int Mtot, int Ntot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = wm * MT * 32, n0 = blockIdx.x * (WN * NT * 32) + wn * NT * 32;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[mt][nt][reg] = 0.f;
    const float *ap = A + m0 + (lane & 31) + (long long)(lane >> 5) * lda;
    const float *bp = B + n0 + (lane & 31) + (long long)(lane >> 5) * ldb;
    constexpr int U = 4;                       // k-steps (2 rows each) per unrolled body
    long long r = r0;
    for (; r + 2 * U <= r1; r += 2 * U) {
        float av[U][MT], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[u][mt] = ap[(r + 2 * u) * lda + mt * 32];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[u][nt] = bp[(r + 2 * u) * ldb + nt * 32];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
    }
    for (; r + 2 <= r1; r += 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[r * lda + mt * 32], bp[r * ldb + nt * 32], acc[mt][nt], 0, 0, 0);
    }
    float *out = partial + (size_t)blockIdx.y * Mtot * Ntot;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int n = n0 + nt * 32 + (lane & 31);
                out[(size_t)m * Ntot + n] = acc[mt][nt][reg];
            }
}

// Same GEMM with both operands staged through LDS (each element leaves L2 once per workgroup instead of once per
// wave that needs it): KB rows of A [KB x Mtot] and B [KB x Ntile] per stage, double-buffered, next stage's global
// loads in flight during the current stage's MFMAs.  Row strides are exact multiples of 32 floats, so the fragment
// reads (lane -> column) are conflict-free and lanes l / l+32 (adjacent rows) never share a service group.
template <int MT, int NT, int WM, int WN, int KB>
__global__ __launch_bounds__(WM *WN * 64) void k_wgrad_lds(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                           long long rows, int rows_per_chunk, float *__restrict__ partial, int Ntot) {
    constexpr int NTHR = WM * WN * 64, MTOT = WM * MT * 32, NTILE = WN * NT * 32;
    constexpr int A4 = KB * MTOT / 4 / NTHR, B4 = KB * NTILE / 4 / NTHR;      // float4 per thread per stage
    static_assert(A4 * 4 * NTHR == KB * MTOT && B4 * 4 * NTHR == KB * NTILE, "stage must divide evenly");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                           // [2][KB][MTOT]
    float *Bs = smem + 2 * KB * MTOT;           // [2][KB][NTILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = wm * MT * 32, n0l = wn * NT * 32, n0 = blockIdx.x * NTILE;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[mt][nt][reg] = 0.f;
    float4 ra[A4], rb[B4];
    auto gload = [&](long long r) {
#pragma unroll
        for (int i = 0; i < A4; ++i) {
            const int e = (tid + i * NTHR) * 4, kr = e / MTOT, c = e - kr * MTOT;
            ra[i] = (r + kr < r1) ? *reinterpret_cast<const float4 *>(A + (r + kr) * lda + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B4; ++i) {
            const int e = (tid + i * NTHR) * 4, kr = e / NTILE, c = e - kr * NTILE;
            rb[i] = (r + kr < r1) ? *reinterpret_cast<const float4 *>(B + (r + kr) * ldb + n0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int bufi) {
#pragma unroll
        for (int i = 0; i < A4; ++i) *reinterpret_cast<float4 *>(As + bufi * KB * MTOT + (tid + i * NTHR) * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B4; ++i) *reinterpret_cast<float4 *>(Bs + bufi * KB * NTILE + (tid + i * NTHR) * 4) = rb[i];
    };
    if (r0 < r1) {
        gload(r0);
        lstore(0);
        __syncthreads();
        int cur = 0;
        for (long long r = r0; r < r1; r += KB) {
            const bool more = r + KB < r1;
            if (more) gload(r + KB);
            const float *ap = As + cur * KB * MTOT + (lane >> 5) * MTOT + m0 + (lane & 31);
            const float *bp = Bs + cur * KB * NTILE + (lane >> 5) * NTILE + n0l + (lane & 31);
#pragma unroll 4
            for (int k = 0; k < KB; k += 2) {
                float av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = ap[k * MTOT + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bp[k * NTILE + nt * 32];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
            if (more) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    float *out = partial + (size_t)blockIdx.y * MTOT * Ntot;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int n = n0 + n0l + nt * 32 + (lane & 31);
                out[(size_t)m * Ntot + n] = acc[mt][nt][reg];
            }
}

// grad[dst + m*ldc + n] += sum_chunk partial[chunk][m][n]   for n < Nreal
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int chunks, int Mtot, int Ntot, int Nreal,
                                                      float *__restrict__ grad, int dst, int ldc) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Mtot * Ntot) return;
    const int m = e / Ntot, n = e - m * Ntot;
    if (n >= Nreal) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * Mtot * Ntot + e];
    grad[dst + m * ldc + n] += s;
}

template <int MT, int NT, int WM, int WN, int KB>
int launch_wgrad_lds(const float *A, int lda, const float *B, int ldb, long long rows, float *partial, int Ntot, int Nreal,
                     float *grad, int dst, int ldc, hipStream_t s) {
    constexpr int Mtot = WM * MT * 32, NTILE = WN * NT * 32;
    const int ntiles = Ntot / NTILE;
    int chunks = WG_CHUNKS / ntiles;
    long long rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + 63) / 64 * 64;
    if (rpc < 64) rpc = 64;
    chunks = (int)((rows + rpc - 1) / rpc);
    if (chunks < 1) chunks = 1;
    const size_t lds = (size_t)2 * KB * (Mtot + NTILE) * sizeof(float);
    if (hipFuncSetAttribute((const void *)k_wgrad_lds<MT, NT, WM, WN, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL((k_wgrad_lds<MT, NT, WM, WN, KB>), dim3(ntiles, chunks), dim3(WM * WN * 64), lds, s, A, lda, B, ldb, rows, (int)rpc, partial, Ntot); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(pn_cdiv((long long)Mtot * Ntot, 256)), dim3(256), 0, s, partial, chunks, Mtot, Ntot, Nreal, grad, dst, ldc);
    PN_CHECK_LAUNCH();
    return 0;
}

template <int MT, int NT, int WM, int WN>
int launch_wgrad(const float *A, int lda, const float *B, int ldb, long long rows, float *partial, int Ntot, int Nreal,
                 float *grad, int dst, int ldc, hipStream_t s) {
    constexpr int Mtot = WM * MT * 32;
    const int ntiles = Ntot / (WN * NT * 32);
    int chunks = WG_CHUNKS / ntiles;
    long long rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + 63) / 64 * 64;                       // whole row tiles per chunk
    if (rpc < 64) rpc = 64;
    chunks = (int)((rows + rpc - 1) / rpc);
    if (chunks < 1) chunks = 1;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL((k_wgrad<MT, NT, WM, WN>), dim3(ntiles, chunks), dim3(WM * WN * 64), 0, s, A, lda, B, ldb, rows, (int)rpc, partial, Mtot, Ntot); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(pn_cdiv((long long)Mtot * Ntot, 256)), dim3(256), 0, s, partial, chunks, Mtot, Ntot, Nreal, grad, dst, ldc);
    PN_CHECK_LAUNCH();
    return 0;
}
}  // namespace

size_t pn_wgrad_partials_bytes() { return pn_align(PARTIAL_FLOATS * sizeof(float)); }

int pn_agg_backward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                           const float *d_raydir, const float *d_sample_loc, const int32_t *d_sample_pidx,
                           const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                           const float *d_decoded, const float *d_weight, const float *d_grad_decoded,
                           const PnSaved &sv, long long n_valid, float *d_grad_params, const pnerf_point_grads *pg,
                           float *d_partials, hipStream_t s) {
    (void)d_sample_loc; (void)R;
    BwdArgs a;
    a.cam = *cam; a.params = d_params; a.packed = (const float4 *)d_packed; a.raydir = d_raydir;
    a.pidx = d_sample_pidx; a.valid_list = d_valid_list; a.counters = d_counters;
    a.SR = SR; a.K = K; a.TS = pn_tile_samples(K); a.cap_samples = n_valid;
    a.decoded = d_decoded; a.weight = d_weight; a.grad_decoded = d_grad_decoded; a.sv = sv;
    a.gparams = d_grad_params; a.g_emb = pg->embedding; a.g_conf = pg->conf; a.g_dir = pg->dir; a.g_color = pg->color;
    if (!a.g_emb || !a.g_conf || !a.g_dir || !a.g_color) return PNERF_E_INVAL;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const long long tiles = (n_valid + a.TS - 1) / a.TS;
    const long long ctiles = (n_valid + PN_CTILE - 1) / PN_CTILE;
    const int wgcu = AGGB_WG_PER_CU < 1 ? 1 : (AGGB_WG_PER_CU > 4 ? 4 : AGGB_WG_PER_CU);
    const int grid_a = (int)(tiles < (long long)wgcu * ncu ? (tiles > 0 ? tiles : 1) : wgcu * ncu);
    const int grid_c = (int)(ctiles < ncu ? (ctiles > 0 ? ctiles : 1) : ncu);
    const size_t lds_c = COLB_LDS_FLOATS * sizeof(float), lds_a = AGGB_LDS_FLOATS * sizeof(float);
    if (hipFuncSetAttribute((const void *)k_color_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
    const bool dfs_lds = a.TS * PN_H <= AGGB_UNION_FLOATS;
    const void *kfn = dfs_lds ? (const void *)k_agg_backward<true> : (const void *)k_agg_backward<false>;
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_COLOR_BWD, s); hipLaunchKernelGGL(k_color_backward, dim3(grid_c), dim3(256), lds_c, s, a); }
    { PnProfScope prof(PNK_AGG_BWD, s);
      if (dfs_lds) hipLaunchKernelGGL(k_agg_backward<true>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
      else hipLaunchKernelGGL(k_agg_backward<false>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a); }
    PN_CHECK_LAUNCH();
    // weight gradients over the rows / samples of the tiles that actually ran
    const long long rows = tiles * PN_TILE, smp = ctiles * PN_CTILE;
    int rc;
    float *g = d_grad_params;
    if ((rc = launch_wgrad_lds<4, 2, 2, 4, 16>(sv.dy1, PN_H, sv.x0, PN_IN1P, rows, d_partials, 256, 256, g, PO_W1, PN_IN1, s))) return rc;
    if ((rc = launch_wgrad_lds<4, 1, 2, 1, 16>(sv.dy1, PN_H, sv.x0 + 256, PN_IN1P, rows, d_partials, 32, PN_IN1 - 256, g, PO_W1 + 256, PN_IN1, s))) return rc;
    if ((rc = launch_wgrad_lds<4, 2, 2, 4, 16>(sv.dy2, PN_H, sv.h1, PN_H, rows, d_partials, 256, 256, g, PO_W2, PN_H, s))) return rc;
    if ((rc = launch_wgrad_lds<4, 2, 2, 4, 16>(sv.dy3, PN_H, sv.h2, PN_H, rows, d_partials, 256, 256, g, PO_W3, PN_IN3, s))) return rc;
    if ((rc = launch_wgrad_lds<4, 2, 2, 4, 16>(sv.dy4, PN_H, sv.h3, PN_H, rows, d_partials, 256, 256, g, PO_W4, PN_H, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 2, 2, 4, 16>(sv.dc1, PN_HC, sv.fs, PN_H, smp, d_partials, 256, 256, g, PO_WC1, PN_INC, s))) return rc;
    if ((rc = launch_wgrad<1, 1, 4, 1>(sv.dc1, PN_HC, sv.pe, 32, smp, d_partials, 32, PN_INC - 256, g, PO_WC1 + 256, PN_INC, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 1, 2, 4, 16>(sv.dc2, PN_HC, sv.c1, PN_HC, smp, d_partials, 128, 128, g, PO_WC2, PN_HC, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 1, 2, 4, 16>(sv.dc3, PN_HC, sv.c2, PN_HC, smp, d_partials, 128, 128, g, PO_WC3, PN_HC, s))) return rc;
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_bwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_bwd), bytes < sizeof(pn_trace_bwd) ? bytes : sizeof(pn_trace_bwd)) == hipSuccess ? 0 : -1;
}
#endif
