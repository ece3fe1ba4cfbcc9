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
//   k_wgrad_lds      : dW = dY^T X as split-K MFMA GEMMs over the saved activations, full dW tile
//                      resident in accumulators, deterministic partial-sum reduction
// LeakyReLU masks come from the saved post-activations (sign(post) == sign(pre)).
#include "mlp_common.h"

namespace {
constexpr int LDH = 260;
constexpr int LDC = 132;
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
    float *gparams;
    float *g_emb, *g_conf, *g_dir, *g_color;
};

__device__ __forceinline__ void rot3b(const float *M, float x, float y, float z, bool transpose, float &ox, float &oy, float &oz) {
    if (!transpose) { ox = x * M[0] + y * M[3] + z * M[6]; oy = x * M[1] + y * M[4] + z * M[7]; oz = x * M[2] + y * M[5] + z * M[8]; }
    else { ox = x * M[0] + y * M[1] + z * M[2]; oy = x * M[3] + y * M[4] + z * M[5]; oz = x * M[6] + y * M[7] + z * M[8]; }
}

// ------------------------------------------------------------------------------ colour backward
constexpr int COLB_LDS_FLOATS = 2 * PN_CTILE * LDC + PN_CTILE * 4;

__global__ __launch_bounds__(256, 2) void k_color_backward(BwdArgs a) {
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
// One workgroup (4 waves, one per SIMD) per CU keeps TWO tiles (A, B) in flight and alternates their layer GEMMs:
//   G(A,4) G(B,4) G(A,3) G(B,3) G(A,2) G(B,2) G(A,1) G(B,1)
// While tile X's GEMM streams through the MFMA pipe, the same wave issues, in the shadow of its own MFMAs
// (pn_tile_gemm_side), the other tile's epilogue (accumulators x LeakyReLU' -> LDS, bias column sums) and the copy-out of
// X's finished dY rows to HBM (plus the W3-extras gradient, which needs exactly those rows).  The phase timeline of the
// previous design (two single-tile workgroups per CU, tools/gpu_phase_trace.py) showed why: a workgroup spent 80-100 us per
// tile outside its 64 us of GEMM, because VALU work of one wave crawls (1 instruction / ~84 cycles) while another wave of
// the same SIMD streams MFMAs -- a second workgroup cannot hide element-wise phases on this hardware, the GEMM wave itself can.
// Everything a tile needs from HBM/L2 is requested at its start (row metadata and 1-bit LeakyReLU masks written by the
// forward, the d f tile, the ray direction).
constexpr int TPR = PN_TPR;                    // threads per tile row in the row-wise phases
constexpr int EPT = PN_F / TPR;                // embedding dims per thread
constexpr int B2_DFS_FLOATS = 8 * PN_H;        // d f tile [TS <= 8][256]
constexpr int B2_TILE_FLOATS = PN_TILE * LDH + B2_DFS_FLOATS + 6 * PN_TILE;
constexpr int AGGB_LDS_FLOATS = 2 * B2_TILE_FLOATS + PN_H + 7 * PN_H;
static_assert(AGGB_LDS_FLOATS * 4 <= 160 * 1024, "two tiles must fit the 160 KB LDS");
static_assert(PN_TILE == 64 && PN_NTHR == 256, "the two-tile backward is written for 64-row tiles and 4 waves");

template <int N> __device__ __forceinline__ float group_sum_b(float v) {
#pragma unroll
    for (int off = 1; off < N; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct B2Tile {            // LDS of one in-flight tile
    float *buf;            // [64][LDH]  h4 -> dY4 -> dY3 -> dY2 -> dY1 -> dX0
    float *dfs;            // [8][256]   d f rows of the tile's samples
    float *wrow, *wnrm, *draw, *dsg;
    int *sidx, *prow;
};
struct B2State {           // registers of one in-flight tile
    unsigned long long m1, m2, m3;
    float rdx, rdy, rdz;
    int rsi, rp;
    long long tile;
    bool valid;
    float4 exv;            // this thread's float4 of the tile's layer-3 extras [64][8] (threads < 128): into the d f region once d f is dead
};

__device__ __forceinline__ B2Tile b2_carve(float *base) {
    B2Tile t;
    t.buf = base; t.dfs = t.buf + PN_TILE * LDH;
    t.wrow = t.dfs + B2_DFS_FLOATS; t.wnrm = t.wrow + PN_TILE; t.draw = t.wnrm + PN_TILE; t.dsg = t.draw + PN_TILE;
    t.sidx = reinterpret_cast<int *>(t.dsg + PN_TILE); t.prow = t.sidx + PN_TILE;
    return t;
}

// P0: request everything the tile needs from memory; h4 / d f / row metadata land in LDS
template <bool DFS_LDS>
__device__ __forceinline__ void b2_load(const BwdArgs &a, const B2Tile &T, B2State &S, long long tile, long long ntiles, int tl, int TS) {
    S.tile = tile; S.valid = tile < ntiles;
    S.rdx = S.rdy = S.rdz = 0.f;
    S.exv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (S.valid && tl < 2 * PN_TILE) S.exv = *reinterpret_cast<const float4 *>(a.sv.ex + tile * PN_TILE * 8 + tl * 4);
    const long long grow0 = tile * PN_TILE;
    if (S.valid) {
        S.m1 = a.sv.lmask[(tile * 3 + 0) * PN_NTHR + tl];
        S.m2 = a.sv.lmask[(tile * 3 + 1) * PN_NTHR + tl];
        S.m3 = a.sv.lmask[(tile * 3 + 2) * PN_NTHR + tl];
        if (tl < PN_TILE) {
            const int4 rm = a.sv.rmeta[grow0 + tl];
            T.sidx[tl] = rm.x; T.prow[tl] = rm.y;
            T.wnrm[tl] = __int_as_float(rm.z); T.wrow[tl] = __int_as_float(rm.w);
            T.dsg[tl] = rm.x >= 0 ? a.grad_decoded[(long long)rm.x * 4] : 0.f;
        }
        // (no staging arrays: hipcc hoists the 16 + 2 loads of the unrolled bodies above the first LDS store by itself)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(a.sv.h4 + (grow0 + (tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4);
            *reinterpret_cast<float4 *>(T.buf + ((tl >> 6) + 4 * i) * LDH + (tl & 63) * 4) = v;
        }
        if (DFS_LDS) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (tl >> 6) + 4 * i;            // row of the [TS x 256] block
                const float4 g = r < TS ? *reinterpret_cast<const float4 *>(a.sv.dfs + (tile * TS + r) * PN_H + (tl & 63) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(T.dfs + r * PN_H + (tl & 63) * 4) = g;
            }
        }
    } else {                                               // the partner slot of the last odd tile: an all-invalid tile of zeros
        S.m1 = S.m2 = S.m3 = 0ull;
        if (tl < PN_TILE) { T.sidx[tl] = -1; T.prow[tl] = -1; T.wnrm[tl] = 0.f; T.wrow[tl] = 0.f; T.dsg[tl] = 0.f; }
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float4 *>(T.buf + ((tl >> 6) + 4 * i) * LDH + (tl & 63) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// P1: alpha head (softplus'), d conf, d(alpha pre-activation) per row.  Row-wise: 4 threads per row, float4 columns interleaved
// (thread q takes float4 q, q+4, ...: conflict-free LDS reads; contiguous 64-column quarters were 8-way bank conflicts)
template <bool DFS_LDS>
__device__ __forceinline__ void b2_alpha(const BwdArgs &a, const B2Tile &T, B2State &S, const float *w5s, float b5, int tl, int TS, int K) {
    const int rrow = tl / TPR, rq = tl % TPR, rls = rrow / K;
    S.rsi = T.sidx[rrow]; S.rp = T.prow[rrow];
    if (rq == 0 && S.rp >= 0) {
        const int r = S.rsi / a.SR;
        S.rdx = a.raydir[3 * r]; S.rdy = a.raydir[3 * r + 1]; S.rdz = a.raydir[3 * r + 2];
    }
    float s = 0.f, dotf = 0.f;
    if (S.rsi >= 0) {
        const float *h = T.buf + rrow * LDH + rq * 4;
        const float *df = DFS_LDS ? T.dfs + rls * PN_H + rq * 4 : a.sv.dfs + (S.tile * TS + rls) * PN_H + rq * 4;
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(h + 16 * j);
            const float4 g = *reinterpret_cast<const float4 *>(df + 16 * j);
            const float4 w = *reinterpret_cast<const float4 *>(w5s + rq * 4 + 16 * j);
            s += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
            dotf += v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
        }
    }
    s = group_sum_b<TPR>(s);
    dotf = group_sum_b<TPR>(dotf);
    if (rq == 0) {
        float dr = 0.f;
        if (S.rsi >= 0) {
            const float x = s + b5 - 1.0f;
            const float alpha = x > 20.f ? x : log1pf(expf(x));
            const float sg = x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
            if (S.rp >= 0) {
                // w = wn * clamp(conf) with a straight-through clamp (gradiant_clamp, point_aggregators.py:722-724)
                const float dw = T.dsg[rrow] * alpha + dotf;
                atomicAdd(&a.g_conf[S.rp], dw * T.wnrm[rrow]);
            }
            dr = T.dsg[rrow] * T.wrow[rrow] * sg;
        }
        T.draw[rrow] = dr;
    }
}

// P2: dY4 = (w * d f + d raw * w5) * lrelu'(h4) in place + HBM; d W5 / d b4 / d b5 partial sums ride along
template <bool DFS_LDS>
__device__ __forceinline__ void b2_dy4(const BwdArgs &a, const B2Tile &T, const B2State &S, const float *w5s, int tl, int TS, int K,
                                       float4 &gb4v, float4 &gw5v, float &gb5t) {
    const int c4 = tl & 63;
    const float4 w5 = *reinterpret_cast<const float4 *>(w5s + c4 * 4);
    const long long grow0 = S.tile * PN_TILE;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int row = (tl >> 6) + 4 * i;
        const int si = T.sidx[row];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (si >= 0) {
            const int ls = row / K;
            const float4 hv = *reinterpret_cast<const float4 *>(T.buf + row * LDH + c4 * 4);
            const float4 g = DFS_LDS ? *reinterpret_cast<const float4 *>(T.dfs + ls * PN_H + c4 * 4)
                                     : *reinterpret_cast<const float4 *>(a.sv.dfs + (S.tile * TS + ls) * PN_H + c4 * 4);
            const float w = T.wrow[row], dr = T.draw[row];
            o.x = (w * g.x + dr * w5.x) * pn_lrelu_grad(hv.x);
            o.y = (w * g.y + dr * w5.y) * pn_lrelu_grad(hv.y);
            o.z = (w * g.z + dr * w5.z) * pn_lrelu_grad(hv.z);
            o.w = (w * g.w + dr * w5.w) * pn_lrelu_grad(hv.w);
            gw5v.x += dr * hv.x; gw5v.y += dr * hv.y; gw5v.z += dr * hv.z; gw5v.w += dr * hv.w;
            gb4v.x += o.x; gb4v.y += o.y; gb4v.z += o.z; gb4v.w += o.w;
        }
        *reinterpret_cast<float4 *>(T.buf + row * LDH + c4 * 4) = o;
        pn_store_stream(a.sv.dy4 + (grow0 + row) * PN_H + c4 * 4, o);     // an invalid partner tile writes zeros into the padding tile
    }
    if (tl < PN_TILE) gb5t += T.draw[tl];
}

// P4: extras of block3's first layer: d colour, d dir from dY3 (row-wise, interleaved columns: conflict-free LDS reads).
// P4 in the MFMA shadows of the tile's own layer-3 GEMM (which only reads the same dY3 rows): float4 column group j of the row,
// sub-piece k, at slot 7 (4 j + k) + 3; reduction at 452 / 456, atomics at 460 / 464
struct B2Ext { float dex[7]; float4 v, w0, w1, w2; };
template <int SLOT>
__device__ __forceinline__ void b2_extras_slot(const BwdArgs &a, const B2Tile &T, const B2State &S, const float *w3ex, B2Ext &E, int tl) {
    const int rrow = tl / TPR, rq = tl % TPR;
    if constexpr (SLOT == 0) {
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) E.dex[jj] = 0.f;
    }
    if constexpr (SLOT >= 3 && SLOT < 3 + 7 * 64 && (SLOT - 3) % 7 == 0) {
        constexpr int q = (SLOT - 3) / 7, j = q / 4, k = q % 4;
        const float *wj = w3ex + rq * 4 + 16 * j;
        auto dot = [](const float4 &x, const float4 &y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; };
        if constexpr (k == 0) {
            E.v = *reinterpret_cast<const float4 *>(T.buf + rrow * LDH + rq * 4 + 16 * j);
            E.w0 = *reinterpret_cast<const float4 *>(wj); E.w1 = *reinterpret_cast<const float4 *>(wj + PN_H);
        }
        if constexpr (k == 1) {
            E.dex[0] += dot(E.v, E.w0); E.dex[1] += dot(E.v, E.w1);
            E.w0 = *reinterpret_cast<const float4 *>(wj + 2 * PN_H); E.w1 = *reinterpret_cast<const float4 *>(wj + 3 * PN_H); E.w2 = *reinterpret_cast<const float4 *>(wj + 4 * PN_H);
            asm volatile("" : "+v"(E.dex[0]), "+v"(E.dex[1]));
        }
        if constexpr (k == 2) {
            E.dex[2] += dot(E.v, E.w0); E.dex[3] += dot(E.v, E.w1); E.dex[4] += dot(E.v, E.w2);
            E.w0 = *reinterpret_cast<const float4 *>(wj + 5 * PN_H); E.w1 = *reinterpret_cast<const float4 *>(wj + 6 * PN_H);
            asm volatile("" : "+v"(E.dex[2]), "+v"(E.dex[3]), "+v"(E.dex[4]));
        }
        if constexpr (k == 3) {
            E.dex[5] += dot(E.v, E.w0); E.dex[6] += dot(E.v, E.w1);
            asm volatile("" : "+v"(E.dex[5]), "+v"(E.dex[6]));
        }
    }
    if constexpr (SLOT == 452) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) E.dex[jj] = group_sum_b<TPR>(E.dex[jj]);
    }
    if constexpr (SLOT == 456) {
#pragma unroll
        for (int jj = 4; jj < 7; ++jj) E.dex[jj] = group_sum_b<TPR>(E.dex[jj]);
    }
    if constexpr (SLOT == 460) {
        if (rq == 0 && S.rp >= 0) {
            atomicAdd(&a.g_color[3 * S.rp], E.dex[0]); atomicAdd(&a.g_color[3 * S.rp + 1], E.dex[1]); atomicAdd(&a.g_color[3 * S.rp + 2], E.dex[2]);
        }
    }
    if constexpr (SLOT == 464) {
        if (rq == 0 && S.rp >= 0) {
            float vx, vy, vz, gx, gy, gz;
            rot3b(a.cam.rw2c, S.rdx, S.rdy, S.rdz, true, vx, vy, vz);
            // features (q - v, q . v) with q = dir @ Rw2c^T  ->  d q = dex[3:6] + dex[6] * v ; d dir = d q @ Rw2c
            rot3b(a.cam.rw2c, E.dex[3] + E.dex[6] * vx, E.dex[4] + E.dex[6] * vy, E.dex[5] + E.dex[6] * vz, false, gx, gy, gz);
            atomicAdd(&a.g_dir[3 * S.rp], gx); atomicAdd(&a.g_dir[3 * S.rp + 1], gy); atomicAdd(&a.g_dir[3 * S.rp + 2], gz);
        }
    }
}

// P8: embedding gradient through [e | PE3(e)]: d e = dX[e] + sum_f 2^f (dX[sin] cos - dX[cos] sin)
__device__ __forceinline__ void b2_emb(const BwdArgs &a, const B2Tile &T, const B2State &S, int tl) {
    const int rrow = tl / TPR, rq = tl % TPR;
    if (S.rp >= 0) {
        const float *dx = T.buf + rrow * LDH;
        const float *x0 = a.sv.x0 + (S.tile * PN_TILE + rrow) * PN_IN1P + PN_F + 6 * EPT * rq;          // EPT dims * 3 freqs * 2
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
            atomicAdd(&a.g_emb[(long long)S.rp * PN_F + dd], g);
        }
    }
}

// Epilogue pieces that run in the MFMA shadow.
// E: element r of the finished accumulators of tile Y, times LeakyReLU' (bit r of m), into Y's LDS tile; bias column sums
template <int R, bool MASK>
__device__ __forceinline__ void b2_epi_piece(const f32x16 (&acc)[2][2], unsigned mlo, unsigned mhi, float *wy, float (&gb)[2]) {
    constexpr int mt = R >> 5, ct = (R >> 4) & 1, reg = R & 15;
    float v = acc[mt][ct][reg];
    if (MASK) {
        const unsigned bit = ((R < 32 ? mlo : mhi) >> (R & 31)) & 1u;
        v *= bit ? 1.f : 0.01f;
        gb[ct] += v;
        asm volatile("" : "+v"(gb[ct]));       // pin the accumulation to this MFMA shadow (pure arithmetic is otherwise sunk to its use)
    }
    wy[(mt * 32 + (reg & 3) + 8 * (reg >> 2)) * LDH + ct * 32] = v;
}

#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_bwd);
#endif
// ---- the tile-boundary program, one slot at a time in the MFMA shadows of the OTHER tile's GEMM --------------------------
// Between a tile's last GEMM (layer 1) and the first GEMM of the tile that replaces it in the same LDS buffer lie: E1
// (d X0 accumulators -> LDS), the embedding gradient, the next tile's loads, its alpha head and its dY4 pass -- four
// workgroup barriers and ~1300 instructions.  All of it is issued by the waves that run the other tile's 512-MFMA GEMM.
// Everything from HBM is requested in ONE burst at slot 0 (a second burst would stall the GEMM's own operand loads a second
// time: vmcnt retires in order).  Slot map:
//   0..8   burst: embedding values of the finished tile (for its embedding gradient), next tile's masks / row metadata / h4 / d f
//   10..73 E1: accumulator element s-10 -> LDS                                  75: barrier
//   77..140 embedding gradient, one embedding dim per 8 slots                   141: barrier (buffer free)
//   143..161 next tile: state, metadata and the staged h4 / d f rows -> LDS     240: d sigma -> LDS   244: barrier
//   246..312 alpha head (one float4 column group per 4 slots, loads two slots ahead of use)  314, 316: reduce, softplus', d conf   320: barrier
//   322..449 dY4 pass (one tile row group per 8 slots)         452: d b5
// A piece never consumes an LDS / HBM value in the slot that requested it (the wave would wait, and the MFMA stream with it),
// and stays under ~12 instructions (the shadow of one MFMA).
struct B2Bnd {
    float4 ev[EPT / 4];          // this thread's EPT embedding values of its row of the finished tile (columns 0..31 of the saved X0)
    float sc[6];                 // (sin, cos) x 3 octaves of the embedding dim being processed: recomputed like the forward computes them
                                 // (one sincosf + double-angle steps) instead of loading 48 floats per thread in the burst
    float4 h4[16], df[2];        // staged rows of the next tile
    unsigned long long nm1, nm2, nm3;
    int4 rm;
    float dsgv, s, dotf;
    float4 hv, g4, o;            // dY4 pass registers
    float4 exn;                  // next tile's extras
    float wv, drv; int siv;
    float dxv[7];
    long long ntile; bool nvalid;
#ifdef PN_PHASE_TRACE
    int titer, trbase;
#endif
};

template <int SLOT, bool DFS_LDS>
__device__ __forceinline__ void b2_boundary_slot(const BwdArgs &a, const B2Tile &T, B2State &S, const f32x16 (&acc)[2][2], float *wy, B2Bnd &C,
                                                 long long next_tile, long long ntiles, const float *w5s, float b5, int tl, int TS, int K,
                                                 float4 &gb4v, float4 &gw5v, float &gb5t) {
    const int rrow = tl / TPR, rq = tl % TPR;
    constexpr int S_E1 = 10, S_EMB = 77, S_NEXT = 143, S_ALPHA = 246, S_DY4 = 322;
    // ---- one burst of requests
    if constexpr (SLOT == 0) {
        const float *x0 = a.sv.x0 + (S.tile * PN_TILE + rrow) * PN_IN1P + EPT * rq;
#pragma unroll
        for (int i = 0; i < EPT / 4; ++i) C.ev[i] = *reinterpret_cast<const float4 *>(x0 + 4 * i);
        C.ntile = next_tile; C.nvalid = next_tile < ntiles;
    }
    if constexpr (SLOT == 3) {
        const long long te = C.nvalid ? C.ntile : ntiles;          // an invalid tile reads the (allocated) padding tile and is masked out below
        C.nm1 = a.sv.lmask[(te * 3 + 0) * PN_NTHR + tl];
        C.nm2 = a.sv.lmask[(te * 3 + 1) * PN_NTHR + tl];
        C.nm3 = a.sv.lmask[(te * 3 + 2) * PN_NTHR + tl];
        C.rm = a.sv.rmeta[te * PN_TILE + (tl & 63)];
        C.exn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (C.nvalid && tl < 2 * PN_TILE) C.exn = *reinterpret_cast<const float4 *>(a.sv.ex + te * PN_TILE * 8 + tl * 4);
    }
    if constexpr (SLOT >= 4 && SLOT < 8) {
        const long long te = C.nvalid ? C.ntile : ntiles;
#pragma unroll
        for (int i = (SLOT - 4) * 4; i < (SLOT - 4) * 4 + 4; ++i) {
            C.h4[i] = make_float4(0.f, 0.f, 0.f, 0.f);     // (an invalid tile must not bring the padding tile's garbage in: 0 * NaN)
            if (C.nvalid) C.h4[i] = *reinterpret_cast<const float4 *>(a.sv.h4 + (te * PN_TILE + (tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4);
        }
    }
    if constexpr (DFS_LDS && SLOT == 8) {
        const long long te = C.nvalid ? C.ntile : ntiles;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            C.df[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (C.nvalid) C.df[i] = *reinterpret_cast<const float4 *>(a.sv.dfs + (te * TS + (tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4);
        }
    }
    // ---- E1 of the finished tile
    if constexpr (SLOT >= S_E1 && SLOT < S_E1 + 64) {
        float gd[2] = {0.f, 0.f};
        b2_epi_piece<SLOT - S_E1, false>(acc, 0u, 0u, wy, gd);
    }
    if constexpr (SLOT == 75 || SLOT == 141 || SLOT == 244 || SLOT == 320) __syncthreads();
#ifdef PN_PHASE_TRACE
    if constexpr (SLOT == 0 || SLOT == 3 || SLOT == 9 || SLOT == 17 || SLOT == 34 || SLOT == 50 || SLOT == 74 || SLOT == 140 || SLOT == 243 || SLOT == 319) {
        constexpr int k = SLOT == 0 ? 0 : SLOT == 3 ? 1 : SLOT == 9 ? 2 : SLOT == 17 ? 3 : SLOT == 34 ? 4 : SLOT == 50 ? 5 : SLOT == 74 ? 6 : SLOT == 140 ? 7 : SLOT == 243 ? 8 : 9;
        const int tid = threadIdx.x, titer = C.titer;
        if (C.trbase >= 0) PN_TR(pn_trace_bwd, C.trbase + k);
    }
#endif
    // ---- embedding gradient of the finished tile: dim i of this thread at slots S_EMB + 8 i (+0 LDS reads, +3 / +5 math, +6 atomic)
    if constexpr (SLOT >= S_EMB && SLOT < S_EMB + 64 && (SLOT - S_EMB) % 8 == 0) {
        constexpr int i = (SLOT - S_EMB) / 8;
        const float *dx = T.buf + rrow * LDH;
        const int dd = EPT * rq + i;
        C.dxv[0] = dx[dd];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            const float2 t = *reinterpret_cast<const float2 *>(dx + PN_F + dd * 6 + 2 * f);
            C.dxv[1 + 2 * f] = t.x; C.dxv[2 + 2 * f] = t.y;
        }
    }
    if constexpr (SLOT >= S_EMB && SLOT < S_EMB + 64 && (SLOT - S_EMB) % 8 == 1) {
        constexpr int i = (SLOT - S_EMB) / 8;
        const float e = i % 4 == 0 ? C.ev[i / 4].x : i % 4 == 1 ? C.ev[i / 4].y : i % 4 == 2 ? C.ev[i / 4].z : C.ev[i / 4].w;
        float sn, cs;
        sincosf(e, &sn, &cs);
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            C.sc[2 * f] = sn; C.sc[2 * f + 1] = cs;
            const float s2 = 2.f * sn * cs;
            cs = 1.f - 2.f * sn * sn; sn = s2;
        }
    }
    if constexpr (SLOT >= S_EMB && SLOT < S_EMB + 64 && (SLOT - S_EMB) % 8 == 3) {
        C.dxv[0] += (C.dxv[1] * C.sc[1] - C.dxv[2] * C.sc[0]) + 2.f * (C.dxv[3] * C.sc[3] - C.dxv[4] * C.sc[2]);
        asm volatile("" : "+v"(C.dxv[0]));
    }
    if constexpr (SLOT >= S_EMB && SLOT < S_EMB + 64 && (SLOT - S_EMB) % 8 == 5) {
        constexpr int i = (SLOT - S_EMB) / 8;
        const float g = C.dxv[0] + 4.f * (C.dxv[5] * C.sc[5] - C.dxv[6] * C.sc[4]);
        if (S.rp >= 0) atomicAdd(&a.g_emb[(long long)S.rp * PN_F + EPT * rq + i], g);
    }
    // ---- the next tile takes over the buffer
    if constexpr (SLOT == S_NEXT) {
        S.tile = C.nvalid ? C.ntile : ntiles; S.valid = C.nvalid;      // an invalid tile lives on the padding tile's storage
        S.m1 = C.nvalid ? C.nm1 : 0ull; S.m2 = C.nvalid ? C.nm2 : 0ull; S.m3 = C.nvalid ? C.nm3 : 0ull;
        S.rdx = S.rdy = S.rdz = 0.f;
        S.exv = C.exn;
        const int si = C.nvalid ? C.rm.x : -1;
        C.dsgv = 0.f;
        if (si >= 0) C.dsgv = a.grad_decoded[(long long)si * 4];
        if (tl < PN_TILE) {
            T.sidx[tl] = si; T.prow[tl] = C.nvalid ? C.rm.y : -1;
            T.wnrm[tl] = C.nvalid ? __int_as_float(C.rm.z) : 0.f; T.wrow[tl] = C.nvalid ? __int_as_float(C.rm.w) : 0.f;
        }
    }
    if constexpr (SLOT > S_NEXT && SLOT <= S_NEXT + 16) {
        constexpr int i = SLOT - S_NEXT - 1;
        *reinterpret_cast<float4 *>(T.buf + ((tl >> 6) + 4 * i) * LDH + (tl & 63) * 4) = C.h4[i];
    }
    if constexpr (DFS_LDS && (SLOT == S_NEXT + 17 || SLOT == S_NEXT + 18)) {
        constexpr int i = SLOT - S_NEXT - 17;
        *reinterpret_cast<float4 *>(T.dfs + ((tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4) = C.df[i];
    }
    if constexpr (SLOT == 240) {
        if (tl < PN_TILE) T.dsg[tl] = C.dsgv;
    }
    // ---- alpha head of the next tile: float4 column group j loaded at S_ALPHA + 1 + 4 j, consumed two slots later
    if constexpr (SLOT == S_ALPHA) {
        S.rsi = T.sidx[rrow]; S.rp = T.prow[rrow];
        C.s = 0.f; C.dotf = 0.f;
        if (rq == 0 && S.rp >= 0) {
            const int r = S.rsi / a.SR;
            S.rdx = a.raydir[3 * r]; S.rdy = a.raydir[3 * r + 1]; S.rdz = a.raydir[3 * r + 2];
        }
    }
    if constexpr (SLOT > S_ALPHA && SLOT <= S_ALPHA + 64 && (SLOT - S_ALPHA - 1) % 4 == 0) {
        constexpr int j = (SLOT - S_ALPHA - 1) / 4;
        const int rls = rrow / K;
        C.hv = *reinterpret_cast<const float4 *>(T.buf + rrow * LDH + rq * 4 + 16 * j);
        C.g4 = DFS_LDS ? *reinterpret_cast<const float4 *>(T.dfs + rls * PN_H + rq * 4 + 16 * j)
                       : *reinterpret_cast<const float4 *>(a.sv.dfs + (S.tile * TS + rls) * PN_H + rq * 4 + 16 * j);
        C.o = *reinterpret_cast<const float4 *>(w5s + rq * 4 + 16 * j);
    }
    if constexpr (SLOT > S_ALPHA && SLOT <= S_ALPHA + 66 && (SLOT - S_ALPHA - 1) % 4 == 2) {
        C.s += C.hv.x * C.o.x + C.hv.y * C.o.y + C.hv.z * C.o.z + C.hv.w * C.o.w;
        C.dotf += C.hv.x * C.g4.x + C.hv.y * C.g4.y + C.hv.z * C.g4.z + C.hv.w * C.g4.w;
        asm volatile("" : "+v"(C.s), "+v"(C.dotf));
    }
    if constexpr (SLOT == 314) {
        C.s = group_sum_b<TPR>(C.s);
        C.dotf = group_sum_b<TPR>(C.dotf);
    }
    if constexpr (SLOT == 316) {
        if (rq == 0) {
            float dr = 0.f;
            if (S.rsi >= 0) {
                const float x = C.s + b5 - 1.0f;
                const float alpha = x > 20.f ? x : log1pf(expf(x));
                const float sg = x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
                if (S.rp >= 0) atomicAdd(&a.g_conf[S.rp], (T.dsg[rrow] * alpha + C.dotf) * T.wnrm[rrow]);
                dr = T.dsg[rrow] * T.wrow[rrow] * sg;
            }
            T.draw[rrow] = dr;
        }
    }
    // ---- dY4 pass of the next tile: row group i at slots S_DY4 + 8 i: +0 LDS reads, +2 / +3 math, +4 partial sums, +5 LDS write, +6 HBM store
    if constexpr (SLOT >= S_DY4 && SLOT < S_DY4 + 128 && (SLOT - S_DY4) % 8 == 0) {
        constexpr int i = (SLOT - S_DY4) / 8;
        const int row = (tl >> 6) + 4 * i, c4 = tl & 63;
        C.siv = T.sidx[row];
        C.hv = *reinterpret_cast<const float4 *>(T.buf + row * LDH + c4 * 4);
        C.g4 = DFS_LDS ? *reinterpret_cast<const float4 *>(T.dfs + (row / K) * PN_H + c4 * 4)
                       : *reinterpret_cast<const float4 *>(a.sv.dfs + (S.tile * TS + row / K) * PN_H + c4 * 4);
        C.wv = T.wrow[row]; C.drv = T.draw[row];
    }
    if constexpr (SLOT >= S_DY4 && SLOT < S_DY4 + 128 && ((SLOT - S_DY4) % 8 == 2 || (SLOT - S_DY4) % 8 == 3)) {
        const float4 w5 = *reinterpret_cast<const float4 *>(w5s + (tl & 63) * 4);
        const float m = C.siv >= 0 ? 1.f : 0.f;                     // invalid rows: kills garbage d f / h4 of the padding tile too
        const float w = C.wv * m, dr = C.drv * m;
        if ((SLOT - S_DY4) % 8 == 2) {
            C.o.x = C.siv >= 0 ? (w * C.g4.x + dr * w5.x) * pn_lrelu_grad(C.hv.x) : 0.f;
            C.o.y = C.siv >= 0 ? (w * C.g4.y + dr * w5.y) * pn_lrelu_grad(C.hv.y) : 0.f;
            asm volatile("" : "+v"(C.o.x), "+v"(C.o.y));
        } else {
            C.o.z = C.siv >= 0 ? (w * C.g4.z + dr * w5.z) * pn_lrelu_grad(C.hv.z) : 0.f;
            C.o.w = C.siv >= 0 ? (w * C.g4.w + dr * w5.w) * pn_lrelu_grad(C.hv.w) : 0.f;
            asm volatile("" : "+v"(C.o.z), "+v"(C.o.w));
        }
    }
    if constexpr (SLOT >= S_DY4 && SLOT < S_DY4 + 128 && (SLOT - S_DY4) % 8 == 4) {
        const float dr = C.siv >= 0 ? C.drv : 0.f;
        gw5v.x += dr * C.hv.x; gw5v.y += dr * C.hv.y; gw5v.z += dr * C.hv.z; gw5v.w += dr * C.hv.w;
        gb4v.x += C.o.x; gb4v.y += C.o.y; gb4v.z += C.o.z; gb4v.w += C.o.w;
        asm volatile("" : "+v"(gw5v.x), "+v"(gw5v.y), "+v"(gw5v.z), "+v"(gw5v.w), "+v"(gb4v.x), "+v"(gb4v.y), "+v"(gb4v.z), "+v"(gb4v.w));
    }
    if constexpr (SLOT >= S_DY4 && SLOT < S_DY4 + 128 && (SLOT - S_DY4) % 8 == 5) {
        constexpr int i = (SLOT - S_DY4) / 8;
        *reinterpret_cast<float4 *>(T.buf + ((tl >> 6) + 4 * i) * LDH + (tl & 63) * 4) = C.o;
    }
    if constexpr (SLOT >= S_DY4 && SLOT < S_DY4 + 128 && (SLOT - S_DY4) % 8 == 6) {
        constexpr int i = (SLOT - S_DY4) / 8;
        pn_store_stream(a.sv.dy4 + (S.tile * PN_TILE + (tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4, C.o);
    }
    if constexpr (SLOT == 452) {
        if (tl < PN_TILE) gb5t += T.draw[tl];
    }
}

// DFS_LDS: the tile's d f rows (TS x 256 floats) fit the 8 KB LDS region (K >= 8); otherwise they are read from HBM/L2.
template <bool DFS_LDS>
__global__ __launch_bounds__(PN_NTHR, 1) void k_agg_backward(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const B2Tile TA = b2_carve(smem), TB = b2_carve(smem + B2_TILE_FLOATS);
    float *w5s = smem + 2 * B2_TILE_FLOATS;    // [256]
    float *w3ex = w5s + PN_H;                  // [7][256]  W3[o][256+j]
    const int tid = threadIdx.x;
    const int K = a.K, TS = a.TS;
    // this launch processes one sample class: its run of tiles, its range of per-sample rows
    const int Ns = a.cls_info[PN_CI_COUNT + a.cls];
    {
        const long long vb = a.cls_info[PN_CI_VBASE + a.cls], tb = a.cls_info[PN_CI_TBASE + a.cls];
        a.sv.dfs += vb * PN_H;
        a.sv.x0 += tb * PN_TILE * PN_IN1P; a.sv.ex += tb * PN_TILE * 8; a.sv.rmeta += tb * PN_TILE; a.sv.lmask += tb * 3 * PN_NTHR;
        a.sv.h4 += tb * PN_TILE * PN_H;
        a.sv.dy1 += tb * PN_TILE * PN_H; a.sv.dy2 += tb * PN_TILE * PN_H; a.sv.dy3 += tb * PN_TILE * PN_H; a.sv.dy4 += tb * PN_TILE * PN_H;
    }
    const long long ntiles = ((long long)Ns + TS - 1) / TS;
    const float *P = a.params;
    if (tid < PN_H) {
        w5s[tid] = P[PO_W5 + tid];
        for (int j = 0; j < 7; ++j) w3ex[j * PN_H + tid] = P[PO_W3 + tid * PN_IN3 + PN_H + j];
    }
    const float b5 = P[PO_B5];
    // gradient partial sums that live in registers for the whole kernel
    float gb[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // d b1..b3: accumulator layout, columns wave*64 + ct*32 + (lane&31)
    float4 gb4v = make_float4(0.f, 0.f, 0.f, 0.f), gw5v = gb4v;  // d b4, d W5: columns 4*lane .. 4*lane+3
    float gw3e[7][4];                                            // d W3[col][256 + j], columns 4*lane .. +3
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) gw3e[j][c] = 0.f;
    float gb5t = 0.f;
    f32x16 accA[2][2], accB[2][2];
    pn_acc_zero(accA); pn_acc_zero(accB);
    B2State SA, SB;
    B2Bnd CB;
    float4 bpre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};   // chunk-0 weight fragments of the next GEMM
    if ((long long)blockIdx.x * 2 < ntiles) {
        // prologue: the first tile of buffer A, plain; buffer B starts as an empty finished tile (zero accumulators, no rows)
        b2_load<DFS_LDS>(a, TA, SA, 2 * (long long)blockIdx.x, ntiles, tid, TS);
        SB.tile = ntiles; SB.valid = false; SB.m1 = SB.m2 = SB.m3 = 0ull; SB.rdx = SB.rdy = SB.rdz = 0.f; SB.rsi = -1; SB.rp = -1;
        __syncthreads();
        b2_alpha<DFS_LDS>(a, TA, SA, w5s, b5, tid, TS, K);
        __syncthreads();
        b2_dy4<DFS_LDS>(a, TA, SA, w5s, tid, TS, K, gb4v, gw5v, gb5t);
    }
#ifdef PN_PHASE_TRACE
    int titer = -1;
#endif
    for (long long pair = blockIdx.x; pair * 2 < ntiles; pair += gridDim.x) {
#ifdef PN_PHASE_TRACE
        ++titer;
#endif
        // thread-index-derived offsets are recomputed per pair (a few VALU ops) instead of living in registers across the loop
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int lane = tl & 63, wave = tl >> 6;
        __syncthreads();
        PN_TR(pn_trace_bwd, 0); PN_TR_HWID(pn_trace_bwd);
        float *wyA = TA.buf + (4 * (lane >> 5)) * LDH + wave * 64 + (lane & 31);     // accumulator-layout write base
        float *wyB = TB.buf + (4 * (lane >> 5)) * LDH + wave * 64 + (lane & 31);
        const float *rxA = TA.buf + wave * LDH + lane * 4, *rxB = TB.buf + wave * LDH + lane * 4;   // copy-out read base (+ 4*i rows)
        const long long gA = SA.tile * PN_TILE + wave;                                  // copy-out row base (+ 4*i)
        const long long nextB = 2 * pair + 1, nextA = 2 * (pair + gridDim.x);
        float4 cpv = make_float4(0.f, 0.f, 0.f, 0.f), exa = cpv, exb2 = cpv;
        B2Ext EX;
        float gnone[2] = {0.f, 0.f};
        if (pair == (long long)blockIdx.x) pn_gemm_prefetch_b0(a.packed + PK_D4 / 4, wave, lane, bpre);
#ifdef PN_PHASE_TRACE
        CB.titer = titer; CB.trbase = 11;
#endif

        // one G step: GEMM of tile X (LDS XB, accumulators ACCX, weight image PK) with, in the MFMA shadows,
        //   E (MASKED: x LeakyReLU' of mask word MY, column sums into GBY) of the other tile's accumulators ACCY -> WY,
        //   the copy-out of X's own finished rows RX -> DST (COPY), plus the W3-extras gradient (EXTRAS), and
        //   optionally the whole boundary program of the other tile (BND)
#define B2_NOBND(s_) (void)0
#define B2_BND_B(s_) b2_boundary_slot<s_, DFS_LDS>(a, TB, SB, accB, wyB, CB, nextB, ntiles, w5s, b5, tl, TS, K, gb4v, gw5v, gb5t)
#define B2_BND_A(s_) b2_boundary_slot<s_, DFS_LDS>(a, TA, SA, accA, wyA, CB, nextA, ntiles, w5s, b5, tl, TS, K, gb4v, gw5v, gb5t)
#define B2_STEP(XB, ACCX, PK, PKNEXT, ACCY, MY, WY, GBY, EPI, MASKED, COPY, EXTRAS, EXW, XT, XS, RX, DST, GROW, BND)                  \
        {                                                                                                                           \
            const unsigned mlo_ = (unsigned)(MY), mhi_ = (unsigned)((MY) >> 32);                                                    \
            pn_acc_zero(ACCX);                                                                                                      \
            pn_tile_gemm_side<PN_H / 8>(XB, LDH, a.packed + (PK) / 4, wave, lane, ACCX, bpre, a.packed + (PKNEXT) / 4, [&](auto ss) { \
                constexpr int s = decltype(ss)::value;                                                                              \
                if constexpr (EPI && s % 8 == 0) b2_epi_piece<s / 8, MASKED>(ACCY, mlo_, mhi_, WY, GBY);                            \
                if constexpr (EXW && s == 1) {        /* the tile's extras take over its d f region (dead since the dY4 pass) */          \
                    if (tl < 2 * PN_TILE) *reinterpret_cast<float4 *>((XT).dfs + tl * 4) = (XS).exv;                                   \
                }                                                                                                                   \
                if constexpr (COPY && EXTRAS && s % 32 == 2) {                                                                      \
                    const float *exr = (XT).dfs + (wave + 4 * (s / 32)) * 8;                                                         \
                    exa = *reinterpret_cast<const float4 *>(exr); exb2 = *reinterpret_cast<const float4 *>(exr + 4);                \
                }                                                                                                                   \
                if constexpr (COPY && s % 32 == 4) cpv = *reinterpret_cast<const float4 *>((RX) + 4 * (s / 32) * LDH);              \
                if constexpr (COPY && s % 32 == 20) pn_store_stream((DST) + ((GROW) + 4 * (s / 32)) * PN_H + lane * 4, cpv);             \
                if constexpr (COPY && EXTRAS && s % 32 >= 21 && s % 32 < 28) {                                                      \
                    constexpr int j_ = s % 32 - 21;                                                                                 \
                    const float e_ = j_ == 0 ? exa.x : j_ == 1 ? exa.y : j_ == 2 ? exa.z : j_ == 3 ? exa.w : j_ == 4 ? exb2.x : j_ == 5 ? exb2.y : exb2.z; \
                    gw3e[j_][0] += cpv.x * e_; gw3e[j_][1] += cpv.y * e_; gw3e[j_][2] += cpv.z * e_; gw3e[j_][3] += cpv.w * e_;      \
                    asm volatile("" : "+v"(gw3e[j_][0]), "+v"(gw3e[j_][1]), "+v"(gw3e[j_][2]), "+v"(gw3e[j_][3]));                    \
                }                                                                                                                   \
                if constexpr (COPY && EXTRAS) b2_extras_slot<s>(a, XT, XS, w3ex, EX, tl);                                          \
                BND(s);                                                                                                             \
            });                                                                                                                     \
            __syncthreads();                                                                                                        \
        }
        //      X-tile   accX  image  next   accY  maskY  writeY gbY    EPI    MASK   COPY   EXTRAS EXW   X   state readX dst       rowbase boundary
        B2_STEP(TA.buf, accA, PK_D4, PK_D4, accB, 0ull, wyB, gnone, false, false, false, false, true, TA, SA, rxA, a.sv.dy3, gA, B2_BND_B)
        PN_TR(pn_trace_bwd, 1);
        const long long gB = SB.tile * PN_TILE + wave;
        B2_STEP(TB.buf, accB, PK_D4, PK_D3, accA, SA.m3, wyA, gb[2], true, true, false, false, true, TB, SB, rxB, a.sv.dy3, gB, B2_NOBND)
        PN_TR(pn_trace_bwd, 2);
        PN_TR(pn_trace_bwd, 3);
        B2_STEP(TA.buf, accA, PK_D3, PK_D3, accB, SB.m3, wyB, gb[2], true, true, true, true, false, TA, SA, rxA, a.sv.dy3, gA, B2_NOBND)
        PN_TR(pn_trace_bwd, 4);
        PN_TR(pn_trace_bwd, 5);
        B2_STEP(TB.buf, accB, PK_D3, PK_D2, accA, SA.m2, wyA, gb[1], true, true, true, true, false, TB, SB, rxB, a.sv.dy3, gB, B2_NOBND)
        PN_TR(pn_trace_bwd, 6);
        B2_STEP(TA.buf, accA, PK_D2, PK_D2, accB, SB.m2, wyB, gb[1], true, true, true, false, false, TA, SA, rxA, a.sv.dy2, gA, B2_NOBND)
        PN_TR(pn_trace_bwd, 7);
        B2_STEP(TB.buf, accB, PK_D2, PK_D1, accA, SA.m1, wyA, gb[0], true, true, true, false, false, TB, SB, rxB, a.sv.dy2, gB, B2_NOBND)
        PN_TR(pn_trace_bwd, 8);
        B2_STEP(TA.buf, accA, PK_D1, PK_D1, accB, SB.m1, wyB, gb[0], true, true, true, false, false, TA, SA, rxA, a.sv.dy1, gA, B2_NOBND)
        PN_TR(pn_trace_bwd, 9);
#ifdef PN_PHASE_TRACE
        CB.trbase = -1;
#endif
        B2_STEP(TB.buf, accB, PK_D1, PK_D4, accA, 0ull, wyA, gnone, false, false, true, false, false, TB, SB, rxB, a.sv.dy1, gB, B2_BND_A)
        PN_TR(pn_trace_bwd, 10);
#undef B2_STEP
#undef B2_BND_A
#undef B2_BND_B
#undef B2_NOBND
    }
    if ((long long)blockIdx.x * 2 < ntiles) {
        // epilogue: d X0 of the last B tile and its embedding gradient (A's were done inside the last step)
        const int lane = tid & 63, wave = tid >> 6;
        float *wyB = TB.buf + (4 * (lane >> 5)) * LDH + wave * 64 + (lane & 31);
        float gnone[2] = {0.f, 0.f};
        pn_static_for<64>([&](auto rr) { b2_epi_piece<decltype(rr)::value, false>(accB, 0u, 0u, wyB, gnone); });
        __syncthreads();
        b2_emb(a, TB, SB, tid);
    }
    // flush the register-resident partial sums
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = wave * 64 + ct * 32 + (lane & 31);
            atomicAdd(&a.gparams[PO_B1 + col], gb[0][ct]);
            atomicAdd(&a.gparams[PO_B2 + col], gb[1][ct]);
            atomicAdd(&a.gparams[PO_B3 + col], gb[2][ct]);
        }
        const float g4[4] = {gb4v.x, gb4v.y, gb4v.z, gb4v.w}, g5[4] = {gw5v.x, gw5v.y, gw5v.z, gw5v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            atomicAdd(&a.gparams[PO_B4 + lane * 4 + c], g4[c]);
            atomicAdd(&a.gparams[PO_W5 + lane * 4 + c], g5[c]);
#pragma unroll
            for (int j = 0; j < 7; ++j) atomicAdd(&a.gparams[PO_W3 + (lane * 4 + c) * PN_IN3 + PN_H + j], gw3e[j][c]);
        }
        if (tid < PN_TILE) atomicAdd(&a.gparams[PO_B5], gb5t);
    }
}

// ------------------------------------------------------------------------------ weight gradients
// partial[chunk][m][n] = sum_{r in chunk} A[r][m] B[r][n]   (A = dY [rows,lda], B = X [rows,ldb])
// Block tile (WM*MT*32) x (WN*NT*32); the whole tile lives in MFMA accumulators.  Both operands are staged through LDS
// (each element leaves L2 once per workgroup instead of once per wave that needs it): KB rows of A [KB x Mtot] and B [KB x Ntile] per stage, double-buffered, next stage's global
// loads in flight during the current stage's MFMAs.  Row strides are exact multiples of 32 floats, so the fragment
// reads (lane -> column) are conflict-free and lanes l / l+32 (adjacent rows) never share a service group.
// TAIL: the B operand has 32 more columns in a second array Bt (x0[:, 256:288] for W1, the view PE for the colour layer).
// They ride in the same pass over A instead of a second kernel that re-reads all of dY: wave w (< MTOT/32) owns the extra
// 32 x 32 tile of row tile w -- one more MFMA per k-step next to its MT*NT.
template <int MT, int NT, int WM, int WN, int KB, bool TAIL>
__global__ __launch_bounds__(WM *WN * 64) void k_wgrad_lds(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
                                                           const float *__restrict__ Bt, int ldbt,
                                                           long long rows, const int *__restrict__ d_tiles, int rows_per_chunk, float *__restrict__ partial, int Ntot) {
    constexpr int NTHR = WM * WN * 64, MTOT = WM * MT * 32, NTILE = WN * NT * 32;
    if (d_tiles) {      // the tiles the aggregator kernels actually used are known on the device only: re-split them over the chunks
        const long long r = (long long)(*d_tiles) * PN_TILE;
        rows = r < rows ? r : rows;
        long long rpc = (rows + gridDim.y - 1) / gridDim.y;
        rpc = (rpc + 63) / 64 * 64;
        rows_per_chunk = (int)(rpc < 64 ? 64 : rpc);
    }
    constexpr int A4 = KB * MTOT / 4 / NTHR, B4 = KB * NTILE / 4 / NTHR;      // float4 per thread per stage
    constexpr int T4 = KB * 32 / 4;                                            // float4 of the tail stage (threads < T4 carry one)
    static_assert(A4 * 4 * NTHR == KB * MTOT && B4 * 4 * NTHR == KB * NTILE, "stage must divide evenly");
    static_assert(T4 <= NTHR, "tail stage: one float4 per thread");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                           // [2][KB][MTOT]
    float *Bs = smem + 2 * KB * MTOT;           // [2][KB][NTILE]
    float *Ts = Bs + 2 * KB * NTILE;            // [2][KB][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = wm * MT * 32, n0l = wn * NT * 32, n0 = blockIdx.x * NTILE;
    const bool tail_wave = TAIL && wave < MTOT / 32;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    f32x16 acc[MT][NT], acct;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) acct[reg] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[mt][nt][reg] = 0.f;
    float4 ra[A4], rb[B4], rt = make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (TAIL && tid < T4) {
            const int kr = tid / 8, c = (tid % 8) * 4;
            rt = (r + kr < r1) ? *reinterpret_cast<const float4 *>(Bt + (r + kr) * ldbt + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int bufi) {
#pragma unroll
        for (int i = 0; i < A4; ++i) *reinterpret_cast<float4 *>(As + bufi * KB * MTOT + (tid + i * NTHR) * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B4; ++i) *reinterpret_cast<float4 *>(Bs + bufi * KB * NTILE + (tid + i * NTHR) * 4) = rb[i];
        if (TAIL && tid < T4) *reinterpret_cast<float4 *>(Ts + bufi * KB * 32 + tid * 4) = rt;
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
            const float *atp = As + cur * KB * MTOT + (lane >> 5) * MTOT + wave * 32 + (lane & 31);
            const float *tp = Ts + cur * KB * 32 + (lane >> 5) * 32 + (lane & 31);
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
                if (tail_wave) acct = __builtin_amdgcn_mfma_f32_32x32x2f32(atp[k * MTOT], tp[k * 32], acct, 0, 0, 0);
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
    if (tail_wave) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            out[(size_t)m * Ntot + NTILE + (lane & 31)] = acct[reg];        // TAIL kernels run with a single column block (gridDim.x == 1)
        }
    }
}

// sum the split-K partials: grad[dst + m*ldc + n] += sum_c partial[c][m][n]   (n < Nreal)
// 256 threads: wave w sums chunks w, w + 4, ... of 64 consecutive outputs (eight loads in flight per lane), the four partial sums are
// combined in a fixed order -- deterministic, and 4 x 8 times the bytes in flight of one thread per output
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ partial, int chunks, int Mtot, int Ntot, int Nreal, float *__restrict__ grad, int dst, int ldc) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const size_t stride = (size_t)Mtot * Ntot;
    float s = 0.f;
    if (e < Mtot * Ntot) {
#pragma unroll 8
        for (int c = w; c < chunks; c += 4) s += partial[c * stride + e];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && e < Mtot * Ntot) {
        const int m = e / Ntot, n = e - m * Ntot;
        if (n < Nreal) grad[dst + m * ldc + n] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    }
}

// Ntot / Nreal: padded / real number of B columns INCLUDING the 32-column tail when Bt != nullptr
template <int MT, int NT, int WM, int WN, int KB, bool TAIL>
int launch_wgrad_lds(const float *A, int lda, const float *B, int ldb, const float *Bt, int ldbt, long long rows, const int *d_tiles, float *partial, int Ntot, int Nreal,
                     float *grad, int dst, int ldc, hipStream_t s) {
    constexpr int Mtot = WM * MT * 32, NTILE = WN * NT * 32;
    const int ntiles = TAIL ? 1 : Ntot / NTILE;
    if (TAIL && Ntot != NTILE + 32) return PNERF_E_INVAL;
    int chunks = WG_CHUNKS / ntiles;
    if ((size_t)chunks * Mtot * Ntot > PARTIAL_FLOATS) chunks = (int)(PARTIAL_FLOATS / ((size_t)Mtot * Ntot));     // the tail widens the tile
    long long rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + 63) / 64 * 64;
    if (rpc < 64) rpc = 64;
    chunks = (int)((rows + rpc - 1) / rpc);
    if (chunks < 1) chunks = 1;
    if ((size_t)chunks * Mtot * Ntot > PARTIAL_FLOATS) return PNERF_E_WS;
    const size_t lds = (size_t)2 * KB * (Mtot + NTILE + (TAIL ? 32 : 0)) * sizeof(float);
    if (hipFuncSetAttribute((const void *)k_wgrad_lds<MT, NT, WM, WN, KB, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL((k_wgrad_lds<MT, NT, WM, WN, KB, TAIL>), dim3(ntiles, chunks), dim3(WM * WN * 64), lds, s, A, lda, B, ldb, Bt, ldbt, rows, d_tiles, (int)rpc, partial, Ntot); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(pn_cdiv((long long)Mtot * Ntot, 64)), dim3(256), 0, s, partial, chunks, Mtot, Ntot, Nreal, grad, dst, ldc);
    PN_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------ weight gradients, split-bf16 form
// The 256 x 256 weight gradients of the four aggregator layers are 1/3 of the step's flops, and an fp32-input MFMA runs at
// 1/16 of the bf16 rate.  k_wgrad_b3 computes the same fp32 product on the bf16 MFMA: every fp32 operand x is split into
// three bf16 numbers by round-to-nearest,
//     h = bf16(x),   m = bf16(x - h),   l = bf16((x - h) - m)      (both subtractions and the last conversion are exact)
// so that x == h + m + l exactly (8 + 8 + 8 significand bits; for |x| below ~1e-33 the residuals are fp32 denormals and the
// split loses its low bits -- of numbers that small), |m| <= 2^-8 |h|, |l| <= 2^-16 |h|, and  a*b  is accumulated (fp32, inside the
// MFMA) as  ah*bl + ah*bm + ah*bh + am*bm + am*bh + al*bh:  the three dropped terms are below 2^-23 of the product, i.e. at
// the level of the fp32 accumulation's own rounding (tests/test_split_bf16_cpu.py restates and checks this arithmetic).  Six
// 32-cycle v_mfma_f32_32x32x16_bf16 (K = 16) replace eight 64-cycle v_mfma_f32_32x32x2_f32: 2.67x on the matrix pipe.
// Measured (profiles/r01_pmc_wgrad_split.json): the MFMA pipe is busy ~65 % of the kernel, at a clock the bf16 MFMA load
// pulls down to ~1.7 GHz (the fp32-MFMA kernels run at ~2.25 GHz).
//
// Block tile 256 x 256 (all of dW), 8 waves as 2 (M) x 4 (N), each 4 x 2 tiles of 32 x 32.  The operands are k-major in HBM
// (row = k) and the MFMA wants 8 consecutive k per lane, so the loader thread owns ONE column and 8 consecutive rows
// (8 dword loads, each coalesced over the wave), splits them in registers and writes one 16-byte [8 x bf16] fragment slot per
// plane: the LDS image is [plane][k-half][column][8 k] and a fragment read is one conflict-free ds_read_b128.
// The split (about 90 VALU operations per thread and k-step) is placed by hand between the step's MFMA groups -- a wave's
// own VALU issues in the shadow of its MFMAs, another wave's does not -- and the global loads run two
// k-steps ahead in two register sets (about 70 KB in flight per CU).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct B3Set { float a[8], b[8]; };

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float pn_f32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> one dword holding bf16(x0) | bf16(x1) << 16, round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned b3_pack(float x0, float x1) {
    const pn_f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float b3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float b3_up(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 4 floats (k, k+1, k+2, k+3 of one column) -> two packed dwords of each plane
__device__ __forceinline__ void b3_split4(const float *x, unsigned *h, unsigned *m, unsigned *l) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        h[p] = b3_pack(x0, x1);
        const float r0 = x0 - b3_lo(h[p]), r1 = x1 - b3_up(h[p]);
        m[p] = b3_pack(r0, r1);
        l[p] = b3_pack(r0 - b3_lo(m[p]), r1 - b3_up(m[p]));
    }
}

// slot -> split piece (0..19) of a k-step, or -1: slots that carry a fragment read carry no piece
constexpr bool b3_slot_reads(int s) { return s <= 5 || s == 12 || s == 13 || s == 20 || s == 21 || s == 24 || s == 25 || s == 36 || s == 37; }
constexpr int b3_slot_work(int s) {
    if (b3_slot_reads(s)) return -1;
    int n = 0;
    for (int i = 0; i < s; ++i) n += b3_slot_reads(i) ? 0 : 1;
    return n < 20 ? n : -1;
}

constexpr int B3_PLANE = 2 * 256 + 2 * 256;      // uint4 slots of one plane: A [2][256], B [2][256]
constexpr int B3_STAGE = 3 * B3_PLANE;           // 3072 slots = 48 KB
constexpr size_t B3_LDS_BYTES = (size_t)2 * B3_STAGE * 16;

template <int LDB>
__global__ __launch_bounds__(512) void k_wgrad_b3(const float *__restrict__ A, const float *__restrict__ B, long long rows,
                                                  const int *__restrict__ d_tiles, int rows_per_chunk, float *__restrict__ partial) {
    constexpr int MT = 4, NT = 2, WN = 4, KB = 16, LDA = PN_H;
    if (d_tiles) {
        const long long r = (long long)(*d_tiles) * PN_TILE;
        rows = r < rows ? r : rows;
        long long rpc = (rows + gridDim.y - 1) / gridDim.y;
        rpc = (rpc + 63) / 64 * 64;
        rows_per_chunk = (int)(rpc < 64 ? 64 : rpc);
    }
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int col = tid & 255, kh = tid >> 8;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;                     // rows and chunk bounds are multiples of 64: every 16-row step is full
    const int nsteps = r1 > r0 ? (int)((r1 - r0) / KB) : 0;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[mt][nt][reg] = 0.f;
    const float *pa = A + (r0 + kh * 8) * LDA + col, *pb = B + (r0 + kh * 8) * LDB + col;     // this thread's column, rows of the step to load
    auto gload = [&](B3Set &S, bool advance) {         // !advance: past the end of the chunk, re-read the last step (never used)
#pragma unroll
        for (int j = 0; j < 8; ++j) { S.a[j] = pa[j * LDA]; S.b[j] = pb[j * LDB]; }
        pa += advance ? KB * LDA : 0; pb += advance ? KB * LDB : 0;
    };
    uint2 *const wbase = reinterpret_cast<uint2 *>(smem4) + 2 * (kh * 256 + col);
    // split 4 k of one operand's column into the three planes of LDS buffer `buf` (half a fragment slot each)
    auto half = [&](const float *x, int buf, int operand, int hf) {
        unsigned h[2], m[2], l[2];
        b3_split4(x + 4 * hf, h, m, l);
        uint2 *d = wbase + 2 * (buf * B3_STAGE + operand * 512) + hf;
        d[0] = make_uint2(h[0], h[1]); d[2 * B3_PLANE] = make_uint2(m[0], m[1]); d[4 * B3_PLANE] = make_uint2(l[0], l[1]);
    };
    const uint4 *const fbase = smem4 + (lane >> 5) * 256 + (lane & 31);
    auto frag = [&](int buf, int plane, int operand, int tile) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, fbase[buf * B3_STAGE + plane * B3_PLANE + operand * 512 + tile * 32]);
    };
    // One k-step = 48 MFMAs on LDS buffer CUR.  The three B planes of the wave's two column tiles stay in registers for the
    // whole step; the A fragments stream through two 2-tile register buffers X / Y:
    //     row-tile pair p (slots 24 p ..):  12 x  ah * {bl, bm, bh}   |   8 x  am * {bm, bh}   |   4 x  al * bh
    // (every accumulator is touched once in four MFMAs).  After every MFMA one small piece of other work is issued
    // (sched_barrier pins it there): the fragment reads of a later group, and the split of register set Sn into
    // buffer CUR ^ 1 as 20 pieces (per half fragment slot: high plane of pair 0, its middle + low planes, the same for pair 1, 3 x ds_write_b64).
    auto step = [&](auto cur_c, B3Set &Sn) {
        constexpr int CUR = decltype(cur_c)::value;
        bf16x8 ax[2], ay[2], bl[NT], bm[NT], bh[NT];
        ax[0] = frag(CUR, 0, 0, wm * MT); ax[1] = frag(CUR, 0, 0, wm * MT + 1);
        bl[0] = frag(CUR, 2, 1, wn * NT); bl[1] = frag(CUR, 2, 1, wn * NT + 1);
        unsigned ph[2], pm[2], pl[2];
        float r0 = 0.f, r1 = 0.f;
        pn_static_for<48>([&](auto ss) {
            constexpr int sl = decltype(ss)::value, pr = sl / 24, q = sl % 24;
            constexpr int grp = q < 12 ? 0 : (q < 20 ? 1 : 2), qi = q - (grp == 0 ? 0 : grp == 1 ? 12 : 20);
            constexpr int bp = grp == 0 ? qi / 4 : (grp == 1 ? 1 + qi / 4 : 2);            // 0: bl, 1: bm, 2: bh
            constexpr int ml = qi % 2, nt = (qi % 4) / 2, mt = 2 * pr + ml;
            constexpr bool use_x = (grp == 1) == (pr == 1);                                 // p0: X Y X, p1: Y X Y
            const bf16x8 av = use_x ? ax[ml] : ay[ml];
            const bf16x8 bv = bp == 0 ? bl[nt] : (bp == 1 ? bm[nt] : bh[nt]);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mt][nt], 0, 0, 0);
            constexpr int w = b3_slot_work(sl);
            if constexpr (sl == 0 || sl == 1) bm[sl] = frag(CUR, 1, 1, wn * NT + sl);
            else if constexpr (sl == 2 || sl == 3) bh[sl - 2] = frag(CUR, 0, 1, wn * NT + sl - 2);
            else if constexpr (sl == 4 || sl == 5) ay[sl - 4] = frag(CUR, 1, 0, wm * MT + sl - 4);             // am, pair 0
            else if constexpr (sl == 12 || sl == 13) ax[sl - 12] = frag(CUR, 2, 0, wm * MT + sl - 12);          // al, pair 0
            else if constexpr (sl == 20 || sl == 21) ay[sl - 20] = frag(CUR, 0, 0, wm * MT + 2 + sl - 20);      // ah, pair 1
            else if constexpr (sl == 24 || sl == 25) ax[sl - 24] = frag(CUR, 1, 0, wm * MT + 2 + sl - 24);      // am, pair 1
            else if constexpr (sl == 36 || sl == 37) ay[sl - 36] = frag(CUR, 2, 0, wm * MT + 2 + sl - 36);      // al, pair 1
            else if constexpr (w >= 0) {
                constexpr int hfi = w / 5, k = w % 5, operand = hfi / 2, hf = hfi % 2;      // per half slot: A0 B0 A1 B1 W
                if constexpr (k == 0 || k == 2) {
                    constexpr int pp = k / 2, j = 4 * hf + 2 * pp;
                    const float x0 = operand == 0 ? Sn.a[j] : Sn.b[j], x1 = operand == 0 ? Sn.a[j + 1] : Sn.b[j + 1];
                    ph[pp] = b3_pack(x0, x1);
                    r0 = x0 - b3_lo(ph[pp]); r1 = x1 - b3_up(ph[pp]);
                } else if constexpr (k == 1 || k == 3) {
                    constexpr int pp = k / 2;
                    pm[pp] = b3_pack(r0, r1);
                    pl[pp] = b3_pack(r0 - b3_lo(pm[pp]), r1 - b3_up(pm[pp]));
                } else {
                    uint2 *d = wbase + 2 * ((CUR ^ 1) * B3_STAGE + operand * 512) + hf;
                    d[0] = make_uint2(ph[0], ph[1]); d[2 * B3_PLANE] = make_uint2(pm[0], pm[1]); d[4 * B3_PLANE] = make_uint2(pl[0], pl[1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    if (nsteps > 0) {                             // nsteps is a multiple of 4 (chunks are multiples of 64 rows)
        B3Set S0, S1;
        gload(S0, true);
        gload(S1, true);
        half(S0.a, 0, 0, 0); half(S0.a, 0, 0, 1); half(S0.b, 0, 1, 0); half(S0.b, 0, 1, 1);
        __syncthreads();
        // step i computes buffer i & 1, splits the set holding step i + 1 into the other buffer and, before that, refills the
        // set step i was split from with step i + 2 (past the end: a harmless re-read whose split is never consumed)
        for (int i = 0; i < nsteps; i += 2) {
            gload(S0, i + 3 < nsteps);
            step(C0{}, S1);
            __syncthreads();
            gload(S1, i + 4 < nsteps);
            step(C1{}, S0);
            __syncthreads();
        }
    }
    float *out = partial + (size_t)blockIdx.y * 256 * 256;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = (wm * MT + mt) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int n = (wn * NT + nt) * 32 + (lane & 31);
                out[(size_t)m * 256 + n] = acc[mt][nt][reg];
            }
}

// dW[256 x 256] (+)= A^T B over `rows` rows; same partial / reduce scheme as launch_wgrad_lds
template <int LDB>
int launch_wgrad_b3(const float *A, const float *B, long long rows, const int *d_tiles, float *partial, float *grad, int dst, int ldc, hipStream_t s) {
    if (rows % PN_TILE) return PNERF_E_INVAL;
    int chunks = WG_CHUNKS;
    long long rpc = (rows + chunks - 1) / chunks;
    rpc = (rpc + 63) / 64 * 64;
    if (rpc < 64) rpc = 64;
    chunks = (int)((rows + rpc - 1) / rpc);
    if (chunks < 1) chunks = 1;
    if ((size_t)chunks * 256 * 256 > PARTIAL_FLOATS) return PNERF_E_WS;
    if (hipFuncSetAttribute((const void *)k_wgrad_b3<LDB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3_LDS_BYTES) != hipSuccess) return PNERF_E_LAUNCH;
    { PnProfScope prof(PNK_WGRAD, s);
    hipLaunchKernelGGL(k_wgrad_b3<LDB>, dim3(1, chunks), dim3(512), B3_LDS_BYTES, s, A, B, rows, d_tiles, (int)rpc, partial); }
    PnProfScope prof(PNK_WGRAD_REDUCE, s);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(pn_cdiv(256LL * 256, 64)), dim3(256), 0, s, partial, chunks, 256, 256, 256, grad, dst, ldc);
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
    const long long ctiles = (n_valid + PN_CTILE - 1) / PN_CTILE;
    const int grid_c = (int)(ctiles < 2 * ncu ? (ctiles > 0 ? ctiles : 1) : 2 * ncu);       // 69 KB of LDS: two workgroups per CU
    const size_t lds_c = COLB_LDS_FLOATS * sizeof(float), lds_a = AGGB_LDS_FLOATS * sizeof(float);
    if (hipFuncSetAttribute((const void *)k_color_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipFuncSetAttribute((const void *)k_agg_backward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipFuncSetAttribute((const void *)k_agg_backward<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    // the forward left the class partition of the valid samples in the saved area (aggregate.hip: pn_classify)
    a.cls_list = sv.cls_list; a.cls_info = sv.cls_info; a.valid_list = sv.cls_list;
    { PnProfScope prof(PNK_COLOR_BWD, s); hipLaunchKernelGGL(k_color_backward, dim3(grid_c), dim3(256), lds_c, s, a); }
    int kc[PN_NCLS];
    const int ncls = pn_class_slots(K, kc);
    { PnProfScope prof(PNK_AGG_BWD, s);
      for (int j = 0; j < ncls; ++j) {
          a.cls = j; a.K = kc[j]; a.TS = pn_tile_samples(kc[j]);
          const long long pairs = ((n_valid + a.TS - 1) / a.TS + 1) / 2;          // one workgroup per CU, two tiles in flight each; worst-case grid
          const int grid_a = (int)(pairs < (long long)ncu ? (pairs > 0 ? pairs : 1) : ncu);
          if (a.TS * PN_H <= B2_DFS_FLOATS) hipLaunchKernelGGL(k_agg_backward<true>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
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
    if ((rc = launch_wgrad_b3<PN_IN1P>(sv.dy1, sv.x0, rows, dt, d_partials, g, PO_W1, PN_IN1, s))) return rc;
    // columns 256..283 of W1 (the distance encoding): a 256 x 32 tile on the fp32 MFMA
    if ((rc = launch_wgrad_lds<1, 1, 8, 1, 64, false>(sv.dy1, PN_H, sv.x0 + 256, PN_IN1P, nullptr, 0, rows, dt, d_partials, 32, PN_IN1 - 256, g, PO_W1 + 256, PN_IN1, s))) return rc;
    if ((rc = launch_wgrad_b3<PN_H>(sv.dy2, sv.h1, rows, dt, d_partials, g, PO_W2, PN_H, s))) return rc;
    if ((rc = launch_wgrad_b3<PN_H>(sv.dy3, sv.h2, rows, dt, d_partials, g, PO_W3, PN_IN3, s))) return rc;
    if ((rc = launch_wgrad_b3<PN_H>(sv.dy4, sv.h3, rows, dt, d_partials, g, PO_W4, PN_H, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 2, 2, 4, 16, true>(sv.dc1, PN_HC, sv.fs, PN_H, sv.pe, 32, smp, nullptr, d_partials, 288, PN_INC, g, PO_WC1, PN_INC, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 1, 2, 4, 16, false>(sv.dc2, PN_HC, sv.c1, PN_HC, nullptr, 0, smp, nullptr, d_partials, 128, 128, g, PO_WC2, PN_HC, s))) return rc;
    if ((rc = launch_wgrad_lds<2, 1, 2, 4, 16, false>(sv.dc3, PN_HC, sv.c2, PN_HC, nullptr, 0, smp, nullptr, d_partials, 128, 128, g, PO_WC3, PN_HC, s))) return rc;
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_bwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_bwd), bytes < sizeof(pn_trace_bwd) ? bytes : sizeof(pn_trace_bwd)) == hipSuccess ? 0 : -1;
}
#endif
