// mlp_common.h -- layout of the aggregator / colour MLP parameters and the saved-activation area.  Every GEMM of the forward and
// of the dgrad chain runs on the f16 matrix pipe with two-plane operands: f16x3.h.
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
// Organisation of an aggregator workgroup (forward and backward tile kernels), two workgroups per CU either way:
//   PN_NTHR 256 ("A", shipped): 4 waves x (2 feature blocks x 2 row blocks of the 64-row tile), up to 256 registers per wave, two waves per SIMD
//   PN_NTHR 512 ("B", dev: EXTRA_DEFS=-DPN_NTHR=512): 8 waves, each ONE feature block x both row blocks in the GEMM phases (32 accumulator
//                registers, 128 registers per wave, four waves per SIMD); the row-wise phases keep A's 4-threads-per-row mapping, split by role:
//                waves 4..7 gather / build / embedding gradient, waves 0..3 tail / front.  Round 4 built and measured it on request of the
//                round-3 review (tools/gemm_probe.hip had it 7 % ahead on a synthetic chain): parity green, but the REAL kernels are slower,
//                forward 12.78 -> 14.56 ms, backward 13.85 -> 14.82 ms, render-only 5.18 -> 4.68 M rays/s -- every wave reads BOTH row blocks of
//                the tile from LDS for half as many MFMAs, the LDS pipe is busy 33 % longer and the waves wait 3.7 x as long for an LDS issue
//                slot (SQ_WAIT_INST_LDS), MFMA busy 0.49 -> 0.42 (profiles/r04_orgB_vs_orgA.txt).  One source for both.
#ifndef PN_NTHR
#define PN_NTHR   256
#endif
#define PN_NW     (PN_NTHR / 64)     // waves per aggregator workgroup
#define PN_NFB    (8 / PN_NW)        // 32-feature blocks per wave in the 256-wide GEMMs
#define PN_ETHR   256                // threads of the row-wise element-wise phases
#define PN_TPR    (PN_ETHR / PN_TILE)     // threads per tile row in those phases
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

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float pn_f4 __attribute__((ext_vector_type(4)));
typedef float pn_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pn_lrelu_grad(float post) { return post > 0.f ? 1.f : 0.01f; }

template <int... I, class F> __device__ __forceinline__ void pn_static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void pn_static_for(F &&f) { pn_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- saved-activation area (training) -------------------------------------------------------
struct PnSaved {
    // per neighbor row (rows = row tiles * 64):
    uint4 *x0k, *h1k, *h2k, *h3k;       // k-major [rows / 8][NF] inputs of the four layers (NF = 288, 256, 288, 256) as ONE f16 plane (round to nearest):
                                        // what the weight-gradient GEMM streams (x0k on the fused path: only X0's last 64 columns, [rows / 8][64] --
                                        // k_wgrad_x0 rebuilds the other 224)
    uint4 *dy1k, *dy2k, *dy3k, *dy4k;   // k-major [rows / 8][256] output gradients of the four layers (ONE f16 plane, round to nearest), SCALED by the backward's power-of-two scale
    uint4 *h4r;                         // row-major [2][rows][32] last activation, both planes (alpha head / K-weighted sums of the backward)
    float *arow;                        // per row: pre-activation of the alpha head
    int4 *rmeta;                        // per row: {sample id or -1, point id or -1, bits(normalised weight), bits(final weight)}
    unsigned *lmask;                    // [row tiles][3 layers h1..h3][8 feature blocks][64 lanes]: LeakyReLU sign bits in the accumulator layout
    unsigned *gscale;                   // [4]: bits of max |d decoded| over the valid samples (the backward derives its scale from it)
    // per valid sample (padded to colour tiles * 64)
    float *fs, *dfs, *c3;               // fp32 rows: aggregated feature [256] and its gradient, last colour post-activation [128]
    uint4 *xck, *c1k, *c2k;             // k-major [samples / 8][288 | 128 | 128] inputs of the three colour layers ([f | view encoding | 0], c1, c2), one plane
    uint4 *dc1k, *dc2k, *dc3k;          // k-major [samples / 8][128] their output gradients (one plane, scaled)
    unsigned *cmask;                    // [colour tiles][2 layers c1, c2][256]: LeakyReLU sign bits in the accumulator layout
    // two-plane weight-gradient mode only (pnerf_set_wgrad_planes(2); null otherwise): the RESIDUAL plane of every array above that the
    // weight-gradient GEMMs stream (value = plane + residual to 22 bits), same layouts; x0k then always holds all 288 columns
    uint4 *x0m, *h1m, *h2m, *h3m, *dy1m, *dy2m, *dy3m, *dy4m;
    uint4 *xcm, *c1m, *c2m, *dc1m, *dc2m, *dc3m;
    int wg2;                            // 1 in that mode
    // sample classes (aggregate.hip: pn_classify): the valid samples re-listed class by class, and where each class lives
    int *cls_list;                      // [samples] sample ids, class 0 first
    int *cls_info;                      // PN_CI_* words
    int *cls_tmp;                       // scratch of the partition: flags [samples] + positions [samples] + scan scratch
    long long rows, samples;
};
// cls_info words: per class c (< PN_NCLS): number of samples, first position in cls_list, first tile; then totals
enum : int { PN_NCLS = 3, PN_CI_COUNT = 0, PN_CI_VBASE = 4, PN_CI_TBASE = 8, PN_CI_TILES = 12, PN_CI_CTILES = 13 /* colour tiles of the step */, PN_CI_WORDS = 16 };
int pn_class_slots(int K, int kc[3]);
int pn_classify(const PnSaved &sv, const int32_t *d_valid_list, const int32_t *d_counters, const int32_t *d_pidx, int K, long long n_valid, bool train, bool save_x0, hipStream_t s);
size_t pn_cls_bytes(long long samples);
void pn_cls_carve(void *base, long long samples, PnSaved &s);
size_t pn_saved_bytes(long long n_valid, int K, long long *rows_out, long long *samples_out);
int pn_cross_terms();                   // 8 (default: mixq.h, e4m3 cross terms in the aggregator's tile GEMMs) or 16 (f16x3.h's three f16 products)
int pn_mix_mask();                      // bit 0 / 1 / 2: the inference forward / training forward / backward tile kernels run the mixed format (0 when the cross terms are f16)
int pn_wgrad_planes();                  // 1 (default: one f16 plane per weight-gradient operand) or 2 (both operands as two planes, three products)
PnSaved pn_saved_carve(void *base, long long n_valid, int K);

__host__ __device__ inline int pn_tile_samples(int K) { return PN_TILE / K; }
// row / K for a tile row (row < 64, K <= 64) without an integer division (a runtime division is ~35 instructions, and the tile kernels
// did one per row they touch): kinv = ceil(2^16 / K), row / K = (row * kinv) >> 16 exactly for row * K < 2^16
__device__ __forceinline__ unsigned pn_kinv(int K) { return (65536u + (unsigned)K - 1u) / (unsigned)K; }
__device__ __forceinline__ int pn_row_div(int row, unsigned kinv) { return (int)(((unsigned)row * kinv) >> 16); }


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
