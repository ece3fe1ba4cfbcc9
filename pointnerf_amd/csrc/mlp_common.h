// mlp_common.h -- layout of the aggregator MLP, the fp32 MFMA-fragment weight images and the 64-row fp32 tile GEMM
// (v_mfma_f32_32x32x2_f32) of the COLOUR MLP, and the saved-activation area.  The four 256-wide aggregator layers run on
// the f16 matrix pipe with two-plane operands: f16x3.h.
#pragma once
#include "pn_common.h"
#include <type_traits>
#include <utility>

// ---- architecture (reference viewmlp_init, models/aggregators/point_aggregators.py:276-348, lego flags)
#define PN_F      32                 // point_features_dim
#define PN_IN1    284                // 32 + 2*3*32 + 2*5*6
#define PN_IN1P   288                // padded to a multiple of 8 (zero columns)
#define PN_H      256                // shading_feature_num
#define PN_IN3    263                // 256 + colour 3 + (dir - view) 3 + dir.view 1
#define PN_INC    280                // 256 + view PE 24
#define PN_HC     128
#define PN_TILE   64                 // neighbor rows per aggregator tile (2 MFMA row tiles).  Measured: 32-row tiles are 10 % slower (twice the weight-fragment traffic per MFMA)
#define PN_MT     (PN_TILE / 32)
#define PN_NTHR   256                // threads per aggregator workgroup: 4 waves (one per SIMD) x (64 rows x 64 cols), up to 512 registers each
#define PN_NW     (PN_NTHR / 64)     // waves per aggregator workgroup
#define PN_NT     (PN_H / (PN_NW * 32))   // MFMA column tiles per wave (1 with 8 waves, 2 with 4)
#define PN_TPR    (PN_NTHR / PN_TILE)     // threads per tile row in the element-wise phases
#define PN_CTILE  64                 // valid samples per colour-MLP tile

// flat parameter vector (state_dict order, torch [out,in] row-major)
enum : int {
    PO_W1 = 0, PO_B1 = PO_W1 + PN_H * PN_IN1, PO_W2 = PO_B1 + PN_H, PO_B2 = PO_W2 + PN_H * PN_H,
    PO_W3 = PO_B2 + PN_H, PO_B3 = PO_W3 + PN_H * PN_IN3, PO_W4 = PO_B3 + PN_H, PO_B4 = PO_W4 + PN_H * PN_H,
    PO_W5 = PO_B4 + PN_H, PO_B5 = PO_W5 + PN_H, PO_WC1 = PO_B5 + 1, PO_BC1 = PO_WC1 + PN_HC * PN_INC,
    PO_WC2 = PO_BC1 + PN_HC, PO_BC2 = PO_WC2 + PN_HC * PN_HC, PO_WC3 = PO_BC2 + PN_HC, PO_BC3 = PO_WC3 + PN_HC * PN_HC,
    PO_WC4 = PO_BC3 + PN_HC, PO_BC4 = PO_WC4 + 3 * PN_HC, PO_TOTAL = PO_BC4 + 3
};
static_assert(PO_TOTAL == 341764, "parameter count of the lego-script aggregator");

// packed images (float offsets).  An image of a B operand [Kpad x N] is stored as
//   float4 img[c][w][ct][lane] ,  element i = B[8c + 4*(lane>>5) + i][w*NT*32 + ct*32 + (lane&31)]
// (the same bytes serve 4 waves x NT=2 and 8 waves x NT=1: fragment index = column / 32)
// i.e. exactly what lane `lane` of wave `w` feeds to 4 consecutive 32x32x2 MFMAs of column tile ct.
enum : int {
    PK_F1 = 0, PK_F2 = PK_F1 + PN_IN1P * PN_H, PK_F3 = PK_F2 + PN_H * PN_H, PK_F4 = PK_F3 + (PN_H + 8) * PN_H,
    PK_C1 = PK_F4 + PN_H * PN_H, PK_C2 = PK_C1 + PN_INC * PN_HC, PK_C3 = PK_C2 + PN_HC * PN_HC,
    PK_D4 = PK_C3 + PN_HC * PN_HC, PK_D3 = PK_D4 + PN_H * PN_H, PK_D2 = PK_D3 + PN_H * PN_H, PK_D1 = PK_D2 + PN_H * PN_H,
    PK_DC3 = PK_D1 + PN_H * PN_H, PK_DC2 = PK_DC3 + PN_HC * PN_HC, PK_DC1 = PK_DC2 + PN_HC * PN_HC,
    PK_TOTAL = PK_DC1 + PN_HC * PN_H
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// streaming store of saved activations: ~70 GB per step pass through the L2 that also holds the 2.7 MB of packed weights every
// wave re-reads continuously; non-temporal keeps them from displacing the weights
typedef float pn_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pn_store_stream(float *p, const float4 &v) {
#ifdef PN_PLAIN_STREAM_STORES       // dev: A/B of the store policy (tools/_build only)
    *reinterpret_cast<float4 *>(p) = v;
#else
    pn_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<pn_f4 *>(p));
#endif
}

__device__ __forceinline__ float pn_lrelu(float v) { return v > 0.f ? v : 0.01f * v; }
__device__ __forceinline__ float pn_lrelu_grad(float post) { return post > 0.f ? 1.f : 0.01f; }

// C[(MT*32) x (4 waves * NT * 32)] += A[(MT*32) x 8*nchunks] * B   (A in LDS, row stride lda floats, lda % 4 == 0;
// B = packed image).  Each wave owns NT column tiles x all MT row tiles.  K order inside a chunk is
// {0,4},{1,5},{2,6},{3,7} (lanes 0-31 / 32-63), identical for A and B, so the sum is a permutation of the
// textbook order.  One chunk is prefetched ahead.
template <int MT, int NT>
__device__ __forceinline__ void pn_mfma_chunk(const float4 (&a)[MT], const float4 (&b)[NT], f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const float bv = i == 0 ? b[ct].x : (i == 1 ? b[ct].y : (i == 2 ? b[ct].z : b[ct].w));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float av = i == 0 ? a[mt].x : (i == 1 ? a[mt].y : (i == 2 ? a[mt].z : a[mt].w));
                acc[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt][ct], 0, 0, 0);
            }
        }
    }
}

// Two explicit register sets, the loop unrolled by two: chunk c+1's operands are requested before chunk c's 16 MFMAs
// and first used after them, with no register copies in between.  (A single-set "next = load; ...; cur = next" form made
// hipcc sink the copies into the middle of the MFMA block behind s_waitcnt vmcnt(0): every chunk then waited for an L2
// round trip after ~6 MFMAs, and a workgroup running alone reached only ~80 % of the MFMA rate.)
template <int MT, int NT, int NW = 4>
__device__ __forceinline__ void pn_tile_gemm(const float *__restrict__ A, int lda, int nchunks,
                                             const float4 *__restrict__ Wp, int wave, int lane, f32x16 (&acc)[MT][NT]) {
    const float *ap = A + (lane & 31) * lda + 4 * (lane >> 5);
    const float4 *wp = Wp + (wave * NT) * 64 + lane;
    float4 a0[MT], b0[NT], a1[MT], b1[NT];
    auto load = [&](int c, float4 (&a)[MT], float4 (&b)[NT]) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) b[ct] = wp[(c * NW * NT + ct) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4 *>(ap + mt * 32 * lda + 8 * c);
    };
    load(0, a0, b0);
#pragma unroll 1
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks) load(c + 1, a1, b1);
        pn_mfma_chunk<MT, NT>(a0, b0, acc);
        if (c + 2 < nchunks) load(c + 2, a0, b0);
        if (c + 1 < nchunks) pn_mfma_chunk<MT, NT>(a1, b1, acc);
    }
}

// accumulator element (rt, ct, reg) of wave `wave` sits at row / col:
__device__ __forceinline__ int pn_acc_row(int rt, int reg, int lane) { return rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
template <int NT> __device__ __forceinline__ int pn_acc_col(int wave, int ct, int lane) { return wave * NT * 32 + ct * 32 + (lane & 31); }

template <int MT, int NT>
__device__ __forceinline__ void pn_acc_init_bias(f32x16 (&acc)[MT][NT], const float *__restrict__ bias, int wave, int lane) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const float bv = bias ? bias[pn_acc_col<NT>(wave, ct, lane)] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[mt][ct][reg] = bv;
    }
}

// ---- wide epilogues: accumulators -> LDS (dword, conflict-free), then whole-row float4 traffic LDS <-> HBM.
// A C-fragment lane owns 16*MT*NT scattered dwords; storing them straight to HBM costs that many dword stores per lane
// per layer (store-issue bound).  Going through the LDS tile that the next layer needs anyway turns that into
// dwordx4 traffic.
template <int MT, int NT, bool LRELU>
__device__ __forceinline__ void pn_acc_to_lds(f32x16 (&acc)[MT][NT], float *__restrict__ H, int ldh, int wave, int lane) {
#pragma unroll
    for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int col = pn_acc_col<NT>(wave, ct, lane);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = acc[rt][ct][reg];
                H[pn_acc_row(rt, reg, lane) * ldh + col] = LRELU ? pn_lrelu(v) : v;
            }
        }
}

// G[grow0 + row][0..W) = H[row][0..W) for the ROWS rows of the tile (W = 256 or 128), float4 per lane
template <int ROWS, int W, int NTHR = 256>
__device__ __forceinline__ void pn_tile_copy_out(const float *__restrict__ H, int ldh, float *__restrict__ G, int ldg, long long grow0, int tid) {
    constexpr int PER = ROWS * W / 4 / NTHR;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + i * NTHR, row = e / (W / 4), c4 = e - row * (W / 4);
        *reinterpret_cast<float4 *>(G + (grow0 + row) * ldg + c4 * 4) = *reinterpret_cast<const float4 *>(H + row * ldh + c4 * 4);
    }
}

// in place: H = H * LeakyReLU'(S) with S the saved post-activation in HBM; the result also goes to D (HBM)
template <int ROWS, int W, int NTHR = 256>
__device__ __forceinline__ void pn_tile_mask_pass(float *__restrict__ H, int ldh, const float *__restrict__ S, int lds_, float *__restrict__ D,
                                                  int ldd, long long grow0, int tid) {
    constexpr int PER = ROWS * W / 4 / NTHR;
    float4 sv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + i * NTHR, row = e / (W / 4), c4 = e - row * (W / 4);
        sv[i] = *reinterpret_cast<const float4 *>(S + (grow0 + row) * lds_ + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + i * NTHR, row = e / (W / 4), c4 = e - row * (W / 4);
        float4 v = *reinterpret_cast<const float4 *>(H + row * ldh + c4 * 4);
        v.x *= pn_lrelu_grad(sv[i].x); v.y *= pn_lrelu_grad(sv[i].y); v.z *= pn_lrelu_grad(sv[i].z); v.w *= pn_lrelu_grad(sv[i].w);
        *reinterpret_cast<float4 *>(H + row * ldh + c4 * 4) = v;
        *reinterpret_cast<float4 *>(D + (grow0 + row) * ldd + c4 * 4) = v;
    }
}

// ---- 1-bit LeakyReLU masks, in the accumulator layout.  A lane owns the same 64 (row, feature) elements of a layer's output in
// the forward (where it applies bias + LeakyReLU to its accumulators) and in the backward (where it multiplies its dgrad
// accumulators by LeakyReLU'), so the sign bits travel as ONE 8-byte word per thread per layer.
// Written and read fully coalesced ([tile][layer][thread]).

template <int... I, class F> __device__ __forceinline__ void pn_static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void pn_static_for(F &&f) { pn_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- saved-activation area (training) -------------------------------------------------------
struct PnSaved {
    // per neighbor row (rows = row tiles * 64), f16 plane pairs (f16x3.h):
    uint4 *x0k, *h1k, *h2k, *h3k;       // k-major [2][rows / 8][NF] inputs of the four layers (NF = 288, 256, 288, 256): what the weight-gradient GEMM streams
    uint4 *dy1k, *dy2k, *dy3k, *dy4k;   // k-major [2][rows / 8][256] output gradients of the four layers, SCALED by the backward's power-of-two scale
    uint4 *h4r;                         // row-major [2][rows][32] last activation (alpha head / K-weighted sums of the backward)
    float *arow;                        // per row: pre-activation of the alpha head
    int4 *rmeta;                        // per row: {sample id or -1, point id or -1, bits(normalised weight), bits(final weight)}
    unsigned long long *lmask;          // [row tiles][3 layers h1..h3][PN_NTHR]: LeakyReLU sign bits in the accumulator layout
    unsigned *gscale;                   // [4]: bits of max |d decoded| over the valid samples (the backward derives its scale from it)
    // per valid sample (padded to colour tiles * 64)
    float *fs, *pe, *c1, *c2, *c3, *dfs, *dc1, *dc2, *dc3;
    // sample classes (aggregate.hip: pn_classify): the valid samples re-listed class by class, and where each class lives
    int *cls_list;                      // [samples] sample ids, class 0 first
    int *cls_info;                      // PN_CI_* words
    int *cls_tmp;                       // scratch of the partition: flags [samples] + positions [samples] + scan scratch
    long long rows, samples;
};
// cls_info words: per class c (< PN_NCLS): number of samples, first position in cls_list, first tile; then totals
enum : int { PN_NCLS = 3, PN_CI_COUNT = 0, PN_CI_VBASE = 4, PN_CI_TBASE = 8, PN_CI_TILES = 12, PN_CI_WORDS = 16 };
int pn_class_slots(int K, int kc[3]);
int pn_classify(const PnSaved &sv, const int32_t *d_valid_list, const int32_t *d_counters, const int32_t *d_pidx, int K, long long n_valid, bool train, hipStream_t s);
size_t pn_cls_bytes(long long samples);
void pn_cls_carve(void *base, long long samples, PnSaved &s);
size_t pn_saved_bytes(long long n_valid, int K, long long *rows_out, long long *samples_out);
PnSaved pn_saved_carve(void *base, long long n_valid, int K);

__host__ __device__ inline int pn_tile_samples(int K) { return PN_TILE / K; }


// ---- dev-only phase timeline (build with EXTRA_DEFS=-DPN_PHASE_TRACE into tools/_build; tools/gpu_phase_trace.py reads it) ----------
// thread 0 of the first PN_TR_WGS workgroups stamps s_memrealtime (100 MHz) at the phase boundaries of PN_TR_ITERS tile iterations
#ifdef PN_PHASE_TRACE
#define PN_TR_WGS   512
#define PN_TR_IT0   20
#define PN_TR_ITERS 6
#define PN_TR_SLOTS 24
#define PN_TR_DECL(name) __device__ unsigned long long name[PN_TR_WGS * PN_TR_ITERS * PN_TR_SLOTS]
#define PN_TR(buf, ph)                                                                                              \
    do {                                                                                                            \
        if (tid == 0 && blockIdx.x < PN_TR_WGS && titer >= PN_TR_IT0 && titer < PN_TR_IT0 + PN_TR_ITERS)           \
            buf[((size_t)blockIdx.x * PN_TR_ITERS + (titer - PN_TR_IT0)) * PN_TR_SLOTS + (ph)] = wall_clock64();    \
    } while (0)
#define PN_TR_HWID(buf)                                                                                             \
    do {                                                                                                            \
        if (tid == 0 && blockIdx.x < PN_TR_WGS && titer >= PN_TR_IT0 && titer < PN_TR_IT0 + PN_TR_ITERS)           \
            buf[((size_t)blockIdx.x * PN_TR_ITERS + (titer - PN_TR_IT0)) * PN_TR_SLOTS + PN_TR_SLOTS - 1] =         \
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                                    \
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);                             \
    } while (0)
#define PN_TR_ITER_DECL int titer = -1
#define PN_TR_ITER_NEXT ++titer
#else
#define PN_TR(buf, ph)
#define PN_TR_HWID(buf)
#define PN_TR_ITER_DECL
#define PN_TR_ITER_NEXT
#endif
