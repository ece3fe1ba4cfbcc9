// aggregate.hip -- per-sample feature aggregator (forward): gather + feature build + MFMA MLP chain +
// K-weighted reduction, then the colour MLP, for the lego-script architecture.
//
// Replaces NeuralPoints.forward's index_select gathers (models/neural_points/neural_points.py:706-717),
// PointAggregator.forward / viewmlp (models/aggregators/point_aggregators.py:727-814, 488-644) and
// positional_encoding (models/helpers/networks.py:175-190).  The reference runs this as ~60 ATen kernels
// that move every [Nv,256] activation through HBM between layers and compact/scatter rows by boolean
// masks; here a workgroup keeps a 64-row tile (TS samples x K neighbor slots) in LDS across the whole chain:
//   gather (embedding 128 B + xyz/dir/colour/conf) -> X0[64x288] in LDS (sin/cos PE computed in place)
//   -> 284->256->256 -> (+7) ->256->256 on the f16 matrix pipe with two-plane operands (f16x3.h: fp32-accurate, 5.3x the
//   fp32 MFMA rate), weights streamed from an L2-resident fragment-ordered image -> alpha head + K-weighted sums
//   (sigma, f[256]) -> f to HBM -> colour kernel: 64 samples per tile, 280->128->128->128->3 on the same two-plane GEMMs.
// Two workgroups (4 waves each) share a CU: the second one hides the first one's latencies (its VALU / LDS work does not run under
// the first one's MFMAs: tools/gemm_probe.hip).  In training mode the operands of the weight-gradient GEMMs are written once, already
// split into their f16 planes and transposed to the k-major order that kernel streams.
#include "mixq.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------ layout / packing
extern "C" int pnerf_mlp_layout(int feat_dim, int64_t *offsets) {
    if (feat_dim != PN_F || !offsets) return PNERF_E_UNSUP;
    const int64_t o[PNERF_MLP_NTENSORS + 1] = {PO_W1, PO_B1, PO_W2, PO_B2, PO_W3, PO_B3, PO_W4, PO_B4, PO_W5, PO_B5,
                                               PO_WC1, PO_BC1, PO_WC2, PO_BC2, PO_WC3, PO_BC3, PO_WC4, PO_BC4, PO_TOTAL};
    for (int i = 0; i <= PNERF_MLP_NTENSORS; ++i) offsets[i] = o[i];
    return 0;
}
extern "C" size_t pnerf_mlp_packed_bytes(void) { return (size_t)PKM_END; }

namespace {
// two-plane f16 images of the aggregator and colour layers (f16x3.h): forward W[m][k] (trans = 0: m = output unit, k = input column)
// and dgrad W^T[m][k] (trans = 1: m = input column, k = output unit)
struct PackHDesc { int src, ld, trans, Mreal, Kreal, NCH, MB, dst; };
struct PackHTable { PackHDesc d[14]; };
__global__ __launch_bounds__(256) void k_pack_h(PackHTable t, const float *__restrict__ params, char *__restrict__ packed) {
    const PackHDesc d = t.d[blockIdx.y];
    const int total = d.NCH * d.MB * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, mb = (e >> 6) % d.MB, c = (e >> 6) / d.MB;
        const int m = 32 * mb + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
        unsigned h[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = k0 + 2 * j + i;
                v[i] = 0.f;
                if (m < d.Mreal && k < d.Kreal) v[i] = d.trans ? params[d.src + k * d.ld + m] : params[d.src + m * d.ld + k];
            }
            // weights: high plane rounded to NEAREST (v_cvt_pk_f16_f32), residual as always -- h + m is the same 22-bit number for the
            // three-product GEMMs, and h alone is the unbiased best single f16 of the weight, which is what the two-product inference
            // option multiplies with (a round-toward-zero plane shrinks every weight by 2^-12 on average: measured 7e-4 on RGB after
            // seven layers); |w| <= 65504 is checked on the host (PointAggregator.check_range), the clamp here is that bound
            pn_split2_sat(v[0], v[1], h[j], lo[j]);
        }
        uint4 *o = reinterpret_cast<uint4 *>(packed + d.dst) + ((size_t)(c * d.MB + mb) * 2) * 64 + lane;
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// mixed-format images of the eight aggregator GEMMs (mixq.h): per (superchunk, feature block, lane) four f16 h fragments, two e4m3 fragments
// [q8(wm 2^11 / 2^e) x 8 | q8(wh / 2^e) x 8] per 8-column group with the lane's block scale e, then the classic two-plane tail chunks
struct PackMDesc { int src, ld, trans, Mreal, Kreal, NT, MB, dst; };
struct PackMTable { PackMDesc d[8]; };
__global__ __launch_bounds__(256) void k_pack_mix(PackMTable t, const float *__restrict__ params, char *__restrict__ packed) {
    pn_mode_saturate();
    const PackMDesc d = t.d[blockIdx.y];
    auto W = [&](int m, int k) -> float { return (m < d.Mreal && k < d.Kreal) ? (d.trans ? params[d.src + k * d.ld + m] : params[d.src + m * d.ld + k]) : 0.f; };
    uint4 *img = reinterpret_cast<uint4 *>(packed + d.dst);
    unsigned *sc = reinterpret_cast<unsigned *>(img + PN_MIMG_U4(d.NT, d.MB));
    const int nsup = PN_MIX_NS * d.MB * 64, ntail = d.NT * d.MB * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nsup + ntail; e += gridDim.x * 256) {
        if (e < nsup) {
            const int lane = e & 63, mb = (e >> 6) % d.MB, s = (e >> 6) / d.MB, m = 32 * mb + (lane & 31), hf = lane >> 5;
            uint4 *o = img + ((size_t)(s * d.MB + mb) * 8) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 64 * s + 16 * r + 8 * hf;
                unsigned h[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pn_split2_rne(W(m, k0 + 2 * j), W(m, k0 + 2 * j + 1), h[j], lo[j]);
                o[r * 64] = make_uint4(h[0], h[1], h[2], h[3]);
            }
            unsigned scw = 0u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // the lane's 16 columns (two 8-column groups), and the largest slot over ALL 32 columns of the fragment (both lane halves): the
                // instruction's scale blocks are registers 0..3 of BOTH halves (scaled by lanes 0..31's byte) and registers 4..7 of both (lanes 32..63's),
                // not "a lane's 32 slots" -- measured, tools/gpu_mix_diag.py; one exponent per (row, fragment) is right under either reading
                float wh[16], wm[16], mx = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const float w = W(m, 64 * s + 32 * j + i);
#ifdef PN_EMU
                    const float wc = fmaxf(fminf(w, 65504.f), -65504.f);
#else
                    const float wc = w;
#endif
                    const float h = (float)(_Float16)wc;
                    mx = fmaxf(mx, fmaxf(fabsf(h), fabsf(w - h) * 2048.f));
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float w = W(m, 8 * (8 * s + 4 * j + 2 * hf + (i >> 3)) + (i & 7));
#ifdef PN_EMU
                    const float wc = fmaxf(fminf(w, 65504.f), -65504.f);
#else
                    const float wc = w;
#endif
                    wh[i] = (float)(_Float16)wc; wm[i] = w - wh[i];
                }
                int ex = 0;
                const float fr = frexpf(mx, &ex);
                int be = mx > 0.f ? (fr > 0.875f ? ex - 8 : ex - 9) : 0;            // the largest slot / 2^be lies in (224, 448]
                be = be < -100 ? -100 : (be > 100 ? 100 : be);
                const float sh = __uint_as_float((unsigned)(be + 127) << 23), sm = __uint_as_float((unsigned)(be + 127 - 11) << 23);
                scw |= (unsigned)(be + 127) << (8 * j);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    unsigned q[4];
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        const float *src = (w4 < 2 ? wm : wh) + 8 * tt + 4 * (w4 & 1);
                        const float scl = w4 < 2 ? sm : sh;
                        pn_s2 r = {0, 0};
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, src[0], src[1], scl, false);
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, src[2], src[3], scl, true);
                        q[w4] = __builtin_bit_cast(unsigned, r);
                    }
                    o[(4 + 2 * j + tt) * 64] = make_uint4(q[0], q[1], q[2], q[3]);
                }
            }
            sc[(size_t)(s * d.MB + mb) * 64 + lane] = scw;
        } else {
            const int e2 = e - nsup, lane = e2 & 63, mb = (e2 >> 6) % d.MB, tc = (e2 >> 6) / d.MB, m = 32 * mb + (lane & 31);
            const int k0 = 256 + 16 * tc + 8 * (lane >> 5);
            unsigned h[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pn_split2_sat(W(m, k0 + 2 * j), W(m, k0 + 2 * j + 1), h[j], lo[j]);
            uint4 *o = img + (size_t)PN_MIX_NS * d.MB * 8 * 64 + ((size_t)(tc * d.MB + mb) * 2) * 64 + lane;
            o[0] = make_uint4(h[0], h[1], h[2], h[3]);
            o[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
}
}  // namespace

extern "C" int pnerf_mlp_pack(const float *d_params, void *d_packed, void *stream) {
    if (!d_params || !d_packed) return PNERF_E_INVAL;
    PackHTable th = {{
        {PO_W1, PN_IN1, 0, PN_H, PN_IN1, 18, 8, PKH_F1},
        {PO_W2, PN_H, 0, PN_H, PN_H, 16, 8, PKH_F2},
        {PO_W3, PN_IN3, 0, PN_H, PN_IN3, 17, 8, PKH_F3},
        {PO_W4, PN_H, 0, PN_H, PN_H, 16, 8, PKH_F4},
        {PO_W4, PN_H, 1, PN_H, PN_H, 16, 8, PKH_D4},
        {PO_W3, PN_IN3, 1, PN_IN3, PN_H, 16, 9, PKH_D3},
        {PO_W2, PN_H, 1, PN_H, PN_H, 16, 8, PKH_D2},
        {PO_W1, PN_IN1, 1, 32 * PN_MB_D1, PN_H, 16, PN_MB_D1, PKH_D1},
        // colour MLP: forward W[m][k], dgrad W^T[m][k] (first 256 input columns of layer 1: the view encoding has no gradient)
        {PO_WC1, PN_INC, 0, PN_HC, PN_INC, 18, 4, PKH_FC1},
        {PO_WC2, PN_HC, 0, PN_HC, PN_HC, 8, 4, PKH_FC2},
        {PO_WC3, PN_HC, 0, PN_HC, PN_HC, 8, 4, PKH_FC3},
        {PO_WC3, PN_HC, 1, PN_HC, PN_HC, 8, 4, PKH_DC3},
        {PO_WC2, PN_HC, 1, PN_HC, PN_HC, 8, 4, PKH_DC2},
        {PO_WC1, PN_INC, 1, PN_H, PN_HC, 8, 8, PKH_DC1},
    }};
    PnProfScope prof(PNK_PACK, (hipStream_t)stream);
    hipLaunchKernelGGL(k_pack_h, dim3(40, 14), dim3(256), 0, (hipStream_t)stream, th, d_params, (char *)d_packed);
    PackMTable tm = {{
        {PO_W1, PN_IN1, 0, PN_H, PN_IN1, 2, 8, PKM_F1},
        {PO_W2, PN_H, 0, PN_H, PN_H, 0, 8, PKM_F2},
        {PO_W3, PN_IN3, 0, PN_H, PN_IN3, 1, 8, PKM_F3},
        {PO_W4, PN_H, 0, PN_H, PN_H, 0, 8, PKM_F4},
        {PO_W4, PN_H, 1, PN_H, PN_H, 0, 8, PKM_D4},
        {PO_W3, PN_IN3, 1, PN_IN3, PN_H, 0, 9, PKM_D3},
        {PO_W2, PN_H, 1, PN_H, PN_H, 0, 8, PKM_D2},
        {PO_W1, PN_IN1, 1, 32 * PN_MB_D1, PN_H, 0, PN_MB_D1, PKM_D1},
    }};
    hipLaunchKernelGGL(k_pack_mix, dim3(12, 8), dim3(256), 0, (hipStream_t)stream, tm, d_params, (char *)d_packed);
    PN_CHECK_LAUNCH();
    return 0;
}

namespace {
// ---- debug: one mixed tile GEMM D[64][256] = X[64][K] W[256][K]^T through the tile's LDS format and a packed image (tests: against float64 on the
// device, against the numpy restatement of the format on the host emulator)
template <int NT>
__global__ __launch_bounds__(256) void k_debug_mix_gemm(const float *__restrict__ x, const char *__restrict__ img, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    pn_mode_saturate();
    constexpr int K = 256 + 16 * NT;
    char *X = smem_d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < PN_TILE * (K / 4); e += 256) {
        const int row = e / (K / 4), col = 4 * (e % (K / 4));
        const float4 v = *reinterpret_cast<const float4 *>(x + row * K + col);
        if (col < 256) pn_xq_store4(X, row, col, v.x, v.y, v.z, v.w);
        else pn_xt_store4(X, row, col, v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    pn_gemm_mix<PN_MIX_NS, NT, 8, 2>(X, img, 2 * wave, lane, acc);
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) out[(32 * rb + (lane & 31)) * PN_H + pn_d_feat(2 * wave + fb, g, lane) + i] = acc[fb][rb][4 * g + i];
}
}  // namespace

extern "C" int pnerf_debug_mix_gemm(const float *d_w, int K, const float *d_x, void *d_img, float *d_out, void *stream) {
    if (!d_w || !d_x || !d_img || !d_out || (K != 256 && K != 272 && K != 288)) return PNERF_E_INVAL;
    hipStream_t s = (hipStream_t)stream;
    const int NT = (K - 256) / 16;
    PackMTable tm = {{{0, K, 0, PN_H, K, NT, 8, 0}}};
    hipLaunchKernelGGL(k_pack_mix, dim3(12, 1), dim3(256), 0, s, tm, d_w, (char *)d_img);
    const void *fn = NT == 0 ? (const void *)k_debug_mix_gemm<0> : NT == 1 ? (const void *)k_debug_mix_gemm<1> : (const void *)k_debug_mix_gemm<2>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PN_XBYTES) != hipSuccess) return PNERF_E_LAUNCH;
    if (NT == 0) hipLaunchKernelGGL(k_debug_mix_gemm<0>, dim3(1), dim3(256), PN_XBYTES, s, d_x, (const char *)d_img, d_out);
    else if (NT == 1) hipLaunchKernelGGL(k_debug_mix_gemm<1>, dim3(1), dim3(256), PN_XBYTES, s, d_x, (const char *)d_img, d_out);
    else hipLaunchKernelGGL(k_debug_mix_gemm<2>, dim3(1), dim3(256), PN_XBYTES, s, d_x, (const char *)d_img, d_out);
    PN_CHECK_LAUNCH();
    return 0;
}

namespace {
__global__ __launch_bounds__(64) void k_debug_mfma_f16(const uint4 *__restrict__ a, const uint4 *__restrict__ b, float *__restrict__ d) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pn_h8, a[threadIdx.x]), __builtin_bit_cast(pn_h8, b[threadIdx.x]), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[threadIdx.x * 16 + r] = acc[r];
}
}  // namespace
namespace {
__global__ __launch_bounds__(256) void k_debug_split(const float *__restrict__ x, long long n, unsigned *__restrict__ h, unsigned *__restrict__ m, int sat) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; 2 * i < n; i += (long long)gridDim.x * 256) {
        unsigned hh, mm;
        if (sat) pn_split2_sat(x[2 * i], x[2 * i + 1], hh, mm);
        else pn_split2(x[2 * i], x[2 * i + 1], hh, mm);
        h[i] = hh; m[i] = mm;
    }
}
}  // namespace
extern "C" int pnerf_debug_split(const float *d_x, int64_t n, void *d_h, void *d_m, int sat, void *stream) {
    if (!d_x || !d_h || !d_m || n < 0 || (n & 1)) return PNERF_E_INVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_debug_split, dim3(64), dim3(256), 0, (hipStream_t)stream, d_x, (long long)n, (unsigned *)d_h, (unsigned *)d_m, sat);
    PN_CHECK_LAUNCH();
    return 0;
}

namespace {
template <int NF> __global__ __launch_bounds__(256) void k_debug_pe(const float *__restrict__ x, long long n, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s[NF], c[NF];
        pn_pe_octaves<NF>(x[i], s, c);
#pragma unroll
        for (int f = 0; f < NF; ++f) { out[(i * NF + f) * 2] = s[f]; out[(i * NF + f) * 2 + 1] = c[f]; }
    }
}
}  // namespace
extern "C" int pnerf_debug_pe(const float *d_x, int64_t n, int nfreq, float *d_out, void *stream) {
    if (!d_x || !d_out || n < 0 || nfreq < 1 || nfreq > 5) return PNERF_E_INVAL;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (nfreq) {
        case 1: hipLaunchKernelGGL(k_debug_pe<1>, dim3(64), dim3(256), 0, s, d_x, (long long)n, d_out); break;
        case 2: hipLaunchKernelGGL(k_debug_pe<2>, dim3(64), dim3(256), 0, s, d_x, (long long)n, d_out); break;
        case 3: hipLaunchKernelGGL(k_debug_pe<3>, dim3(64), dim3(256), 0, s, d_x, (long long)n, d_out); break;
        case 4: hipLaunchKernelGGL(k_debug_pe<4>, dim3(64), dim3(256), 0, s, d_x, (long long)n, d_out); break;
        default: hipLaunchKernelGGL(k_debug_pe<5>, dim3(64), dim3(256), 0, s, d_x, (long long)n, d_out); break;
    }
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_debug_mfma_f16(const void *d_a, const void *d_b, float *d_out, void *stream) {
    if (!d_a || !d_b || !d_out) return PNERF_E_INVAL;
    hipLaunchKernelGGL(k_debug_mfma_f16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint4 *)d_a, (const uint4 *)d_b, d_out);
    PN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ saved activations
size_t pn_saved_bytes(long long n_valid, int K, long long *rows_out, long long *samples_out) {
    const int TS = pn_tile_samples(K);
    // + PN_NCLS tiles of rounding (each class ends in a partial tile) + PN_NCLS gap tiles (every class owns a padding tile) + 1
    const long long tiles = (n_valid + TS - 1) / TS + 2 * PN_NCLS + 1;
    const long long rows = tiles * PN_TILE;
    const long long samples = ((tiles * TS + PN_CTILE - 1) / PN_CTILE + 1) * PN_CTILE;
    if (rows_out) *rows_out = rows;
    if (samples_out) *samples_out = samples;
    size_t b = 0;
    b += 2 * pn_align((size_t)rows / 8 * PN_NF1 * 16) + 2 * pn_align((size_t)rows / 8 * PN_H * 16);      // x0k, h2k | h1k, h3k (one plane)
    b += 4 * pn_align((size_t)rows / 8 * PN_H * 16) + pn_align((size_t)rows * 32 * 32);                // dy1k..dy4k (one plane) | h4r
    b += pn_align((size_t)rows * 4) + pn_align((size_t)rows * 16) + pn_align((size_t)tiles * 3 * 512 * 4) + pn_align(16);
    b += pn_cls_bytes(samples);
    b += 2 * pn_align((size_t)samples * PN_H * 4) + pn_align((size_t)samples * PN_HC * 4);                         // fs, dfs | c3
    b += pn_align((size_t)samples / 8 * PN_NF1 * 16) + 2 * pn_align((size_t)samples / 8 * PN_HC * 16);              // xck | c1k, c2k (one plane)
    b += 3 * pn_align((size_t)samples / 8 * PN_HC * 16) + pn_align((size_t)samples / PN_CTILE * 2 * 256 * 4);       // dc1k..dc3k | cmask
    if (pn_wgrad_planes() == 2) {       // the residual planes of the fourteen streamed arrays
        b += 2 * pn_align((size_t)rows / 8 * PN_NF1 * 16) + 6 * pn_align((size_t)rows / 8 * PN_H * 16);
        b += pn_align((size_t)samples / 8 * PN_NF1 * 16) + 5 * pn_align((size_t)samples / 8 * PN_HC * 16);
    }
    return b;
}

PnSaved pn_saved_carve(void *base, long long n_valid, int K) {
    PnSaved s;
    size_t total = pn_saved_bytes(n_valid, K, &s.rows, &s.samples);
    PnCarver cv(base, total);
    const size_t rg = (size_t)s.rows / 8;
    s.x0k = cv.take<uint4>(rg * PN_NF1); s.h2k = cv.take<uint4>(rg * PN_NF1);
    s.h1k = cv.take<uint4>(rg * PN_H); s.h3k = cv.take<uint4>(rg * PN_H);
    s.dy1k = cv.take<uint4>(rg * PN_H); s.dy2k = cv.take<uint4>(rg * PN_H);
    s.dy3k = cv.take<uint4>(rg * PN_H); s.dy4k = cv.take<uint4>(rg * PN_H);
    s.h4r = cv.take<uint4>((size_t)s.rows * 32 * 2);
    s.arow = cv.take<float>((size_t)s.rows); s.rmeta = cv.take<int4>((size_t)s.rows);
    s.lmask = cv.take<unsigned>((size_t)(s.rows / PN_TILE) * 3 * 512);
    s.gscale = cv.take<unsigned>(4);
    s.fs = cv.take<float>((size_t)s.samples * PN_H); s.dfs = cv.take<float>((size_t)s.samples * PN_H);
    s.c3 = cv.take<float>((size_t)s.samples * PN_HC);
    const size_t rgc = (size_t)s.samples / 8;
    s.xck = cv.take<uint4>(rgc * PN_NF1); s.c1k = cv.take<uint4>(rgc * PN_HC); s.c2k = cv.take<uint4>(rgc * PN_HC);
    s.dc1k = cv.take<uint4>(rgc * PN_HC); s.dc2k = cv.take<uint4>(rgc * PN_HC); s.dc3k = cv.take<uint4>(rgc * PN_HC);
    s.cmask = cv.take<unsigned>((size_t)s.samples / PN_CTILE * 2 * 256);
    s.wg2 = pn_wgrad_planes() == 2 ? 1 : 0;
    s.x0m = s.h1m = s.h2m = s.h3m = s.dy1m = s.dy2m = s.dy3m = s.dy4m = nullptr;
    s.xcm = s.c1m = s.c2m = s.dc1m = s.dc2m = s.dc3m = nullptr;
    if (s.wg2) {
        s.x0m = cv.take<uint4>(rg * PN_NF1); s.h2m = cv.take<uint4>(rg * PN_NF1);
        s.h1m = cv.take<uint4>(rg * PN_H); s.h3m = cv.take<uint4>(rg * PN_H);
        s.dy1m = cv.take<uint4>(rg * PN_H); s.dy2m = cv.take<uint4>(rg * PN_H);
        s.dy3m = cv.take<uint4>(rg * PN_H); s.dy4m = cv.take<uint4>(rg * PN_H);
        s.xcm = cv.take<uint4>(rgc * PN_NF1); s.c1m = cv.take<uint4>(rgc * PN_HC); s.c2m = cv.take<uint4>(rgc * PN_HC);
        s.dc1m = cv.take<uint4>(rgc * PN_HC); s.dc2m = cv.take<uint4>(rgc * PN_HC); s.dc3m = cv.take<uint4>(rgc * PN_HC);
    }
    pn_cls_carve(cv.take<char>(pn_cls_bytes(s.samples)), s.samples, s);
    return s;
}

size_t pn_cls_bytes(long long samples) {
    return pn_align((size_t)samples * 4) + pn_align(PN_CI_WORDS * 4) + pn_align(((size_t)2 * samples + pn_scan_scratch_ints(samples) + 8) * 4);
}
void pn_cls_carve(void *base, long long samples, PnSaved &s) {
    PnCarver cv(base, pn_cls_bytes(samples));
    s.cls_list = cv.take<int>((size_t)samples);
    s.cls_info = cv.take<int>(PN_CI_WORDS);
    s.cls_tmp = cv.take<int>((size_t)2 * samples + pn_scan_scratch_ints(samples) + 8);
}

extern "C" size_t pnerf_agg_saved_bytes(int64_t n_valid_samples, int K) {
    if (K <= 0 || K > PNERF_MAX_K || n_valid_samples < 0) return 0;
    return pn_saved_bytes(n_valid_samples, K, nullptr, nullptr);
}

// ------------------------------------------------------------------------------ forward kernels
namespace {
constexpr int TPR = PN_TPR; // threads per tile row in the element-wise phases
constexpr int EPT = PN_F / TPR;              // embedding dims per thread in the feature build

struct FwdArgs {
    pnerf_camera cam;
    const float *xyz, *emb, *conf, *dir, *color;
    const float *params;
    const float4 *packed;
    const float *raydir, *sample_loc;
    const float *xyz_pers, *loc_pers;   // optional: perspective coords supplied by the caller (stand-alone aggregator)
    const int *pidx, *valid_list, *counters;
    const int *cls_list, *cls_info;     // sample classes (pn_classify); cls = the class this launch processes
    int cls, Kstride;                   // K = neighbor slots PROCESSED per sample of this class, Kstride = slots per sample in pidx / weight
    int save_x0;                        // training: also save the X0 planes (only the stand-alone aggregator, whose perspective coordinates come
                                        // from the caller; the fused path's weight-gradient GEMM rebuilds X0 from the gather: k_wgrad_x0)
    int R, SR, K, TS;
    long long cap_samples;      // capacity (in valid samples) of fs / saved buffers
    float *decoded, *weight;
    PnSaved sv;                 // fs always valid; the rest only when TRAIN
};

__device__ __forceinline__ void rot3(const float *M /*row-major*/, float x, float y, float z, bool transpose, float &ox, float &oy, float &oz) {
    // transpose=false: out_j = sum_i v_i M[i][j] (v @ M);  true: out_j = sum_i v_i M[j][i] (v @ M^T)
    if (!transpose) { ox = x * M[0] + y * M[3] + z * M[6]; oy = x * M[1] + y * M[4] + z * M[7]; oz = x * M[2] + y * M[5] + z * M[8]; }
    else { ox = x * M[0] + y * M[1] + z * M[2]; oy = x * M[3] + y * M[4] + z * M[5]; oz = x * M[6] + y * M[7] + z * M[8]; }
}

template <int N> __device__ __forceinline__ float group_sum(float v) {      // sum over N adjacent lanes (N = 4 or 8)
#pragma unroll
    for (int off = 1; off < N; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------ sample classes
// 6 % of the neighbor rows of valid samples are empty slots (a sample near the surface's edge has 1..K-1 neighbors); the
// reference drops them by boolean masking (point_aggregators.py:523-538).  Tiles need whole samples of equal row count, so the
// valid samples are partitioned by the position of their LAST occupied slot into up to three classes that are processed with
// K, K/2 and K/4 rows per sample (K % 4 == 0; the query fills slots front to back, and for a caller-supplied mask the last
// occupied slot is what counts).  The partition is stable (ascending sample id inside a class): results are identical to the
// one-class launch, rows processed drop from 7.49 M to 7.1 M at the bench configuration.  Every class owns a run of tiles
// followed by one padding tile (the partner slot of an odd tile count works on it).
namespace {
struct ClsArgs { const int *valid_list, *counters, *pidx; int K, n, ncls; int kc[PN_NCLS], ts[PN_NCLS]; };

__global__ void k_cls_flags(ClsArgs c, int which, int *__restrict__ flags) {
    const int vs = blockIdx.x * blockDim.x + threadIdx.x;
    if (vs >= c.n) return;
    const int Ns = c.counters[0] < c.n ? c.counters[0] : c.n;
    int f = 0;
    if (vs < Ns) {
        const long long si = c.valid_list[vs];
        int hv = 0;
        for (int k = 0; k < c.K; ++k)
            if (c.pidx[si * c.K + k] >= 0) hv = k + 1;
        int cl = 0;
        for (int j = 1; j < c.ncls; ++j)
            if (hv <= c.kc[j]) cl = j;
        f = cl == which;
    }
    flags[vs] = f;
}

__global__ void k_cls_gather(ClsArgs c, int which, const int *__restrict__ pos, const int *__restrict__ count, int *__restrict__ cls_list,
                             int *__restrict__ info) {
    const int n = *count, base = info[PN_CI_VBASE + which];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cls_list[base + i] = c.valid_list[pos[i]];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int tiles = (n + c.ts[which] - 1) / c.ts[which], tb = info[PN_CI_TBASE + which];
        info[PN_CI_COUNT + which] = n;
        if (which + 1 < PN_NCLS) { info[PN_CI_VBASE + which + 1] = base + n; info[PN_CI_TBASE + which + 1] = tb + tiles + 1; }
        info[PN_CI_TILES] = tb + tiles + 1;
    }
}

// the padding tile of every class: zero rows in everything the weight-gradient GEMMs read
__global__ void k_cls_zero_gaps(PnSaved sv, int ncls, int save_x0) {
    const int c = blockIdx.y;
    if (c >= ncls) return;
    // x0k holds EITHER the whole X0 (288-column layout: the stand-alone aggregator) OR its last 64 columns (64-column layout: the fused
    // path) -- the two layouts overlap, only the one in use may be zeroed
    if ((blockIdx.x == 0 && !save_x0) || (blockIdx.x == 8 && save_x0)) return;
    const long long gap = (c + 1 < PN_NCLS ? sv.cls_info[PN_CI_TBASE + c + 1] : sv.cls_info[PN_CI_TILES]) - 1;
    uint4 *arrs[9] = {sv.x0k, sv.h2k, sv.h1k, sv.h3k, sv.dy1k, sv.dy2k, sv.dy3k, sv.dy4k, sv.x0k};
    uint4 *arrm[9] = {sv.x0m, sv.h2m, sv.h1m, sv.h3m, sv.dy1m, sv.dy2m, sv.dy3m, sv.dy4m, nullptr};
    const int which = blockIdx.x;                      // 8 arrays (one k-major plane each) + x0k in its 64-column layout (the fused path: k_wgrad_x0)
    const int nf = which == 8 ? 64 : which < 2 ? PN_NF1 : PN_H;
    uint4 *p = arrs[which] + gap * 8 * nf;
    for (int i = threadIdx.x; i < 8 * nf; i += blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (sv.wg2 && arrm[which]) {                       // two-plane weight-gradient mode: the residual planes too
        uint4 *pm = arrm[which] + gap * 8 * nf;
        for (int i = threadIdx.x; i < 8 * nf; i += blockDim.x) pm[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}
}  // namespace

}  // namespace

// class c processes kc[c] slots per sample
int pn_class_slots(int K, int kc[PN_NCLS]) {
    if (K % 4 == 0) { kc[0] = K; kc[1] = K / 2; kc[2] = K / 4; return 3; }
    kc[0] = K; kc[1] = kc[2] = 0;
    return 1;
}

int pn_classify(const PnSaved &sv, const int32_t *d_valid_list, const int32_t *d_counters, const int32_t *d_pidx, int K, long long n_valid,
                bool train, bool save_x0, hipStream_t s) {
    ClsArgs c;
    c.valid_list = d_valid_list; c.counters = d_counters; c.pidx = d_pidx; c.K = K; c.n = (int)n_valid;
    c.ncls = pn_class_slots(K, c.kc);
    for (int j = 0; j < PN_NCLS; ++j) c.ts[j] = c.kc[j] > 0 ? pn_tile_samples(c.kc[j]) : 1;
    if (hipMemsetAsync(sv.cls_info, 0, PN_CI_WORDS * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n_valid <= 0) return 0;
    int *flags = sv.cls_tmp, *pos = flags + n_valid, *cnt = pos + n_valid, *scratch = cnt + 8;
    PnProfScope prof(PNK_COMPACT, s);
    for (int j = 0; j < c.ncls; ++j) {
        hipLaunchKernelGGL(k_cls_flags, dim3(pn_cdiv(n_valid, 256)), dim3(256), 0, s, c, j, flags);
        int rc = pn_compact_gt0_i32(flags, n_valid, pos, cnt, scratch, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_cls_gather, dim3(256), dim3(256), 0, s, c, j, pos, cnt, sv.cls_list, sv.cls_info);
    }
    if (train) hipLaunchKernelGGL(k_cls_zero_gaps, dim3(9, c.ncls), dim3(256), 0, s, sv, c.ncls, save_x0 ? 1 : 0);
    PN_CHECK_LAUNCH();
    return 0;
}


namespace {

// ------------------------------------------------------------------------------ aggregator forward
// LDS of a workgroup (80 896 bytes: two workgroups per CU):
//   X     [2][64][PN_XRS]   the activation tile, two f16 planes (f16x3.h): X0 -> h1 -> [h2 | extras] -> h3 -> h4
//   exb   [64][8] f32       layer-3 extras of the tile's rows until they move next to h2
//   w5s   [256] f32         alpha head weights
//   rowf  wraw, wrow, wnrm [64] f32, sidx [64] int
constexpr int FL_EX = PN_XBYTES, FL_W5 = FL_EX + PN_TILE * 8 * 4, FL_ROW = FL_W5 + PN_H * 4, FL_BYTES = FL_ROW + 4 * PN_TILE * 4;
static_assert(2 * FL_BYTES <= 160 * 1024, "two forward workgroups must fit the 160 KB LDS");

struct FGather {               // what a thread holds of one tile row (4 threads per row, q = thread in row)
    float4 e0, e1;             // its 8 embedding dims
    float px, py, pz, lx, ly, lz, cf;
    float ppx, ppy, ppz, lpx, lpy, lpz;   // optional caller-supplied perspective coordinates
    float dxv, dyv, dzv, cx, cy, cz, rx, ry, rz;   // q == 0 only
};

// sample id of row `row` of tile `tile` (or -1)
__device__ __forceinline__ int f_sample_of(const FwdArgs &a, long long tile, int row, int Ns, unsigned kinv) {
    const int ls = pn_row_div(row, kinv);
    const long long vs = tile * a.TS + ls;
    return (ls < a.TS && vs < Ns) ? a.valid_list[vs] : -1;
}

template <bool PERS>
__device__ __forceinline__ void f_gather(const FwdArgs &a, FGather &G, int si_, int p_, int q) {
    const int p = p_ > 0 ? p_ : 0, si = si_ > 0 ? si_ : 0;     // empty slots / rows read point 0 / sample 0 like the reference (neural_points.py:709); their weight is 0
    const float *ep = a.emb + (long long)p * PN_F + EPT * q;
    G.e0 = *reinterpret_cast<const float4 *>(ep); G.e1 = *reinterpret_cast<const float4 *>(ep + 4);
    G.px = a.xyz[3 * p]; G.py = a.xyz[3 * p + 1]; G.pz = a.xyz[3 * p + 2];
    G.lx = a.sample_loc[(long long)si * 3]; G.ly = a.sample_loc[(long long)si * 3 + 1]; G.lz = a.sample_loc[(long long)si * 3 + 2];
    if (PERS) {
        G.ppx = a.xyz_pers[3 * p]; G.ppy = a.xyz_pers[3 * p + 1]; G.ppz = a.xyz_pers[3 * p + 2];
        G.lpx = a.loc_pers[(long long)si * 3]; G.lpy = a.loc_pers[(long long)si * 3 + 1]; G.lpz = a.loc_pers[(long long)si * 3 + 2];
    }
    G.cf = a.conf[p];
    if (q == 0) {
        const int r = si / a.SR;
        G.dxv = a.dir[3 * p]; G.dyv = a.dir[3 * p + 1]; G.dzv = a.dir[3 * p + 2];
        G.cx = a.color[3 * p]; G.cy = a.color[3 * p + 1]; G.cz = a.color[3 * p + 2];
        G.rx = a.raydir[3 * r]; G.ry = a.raydir[3 * r + 1]; G.rz = a.raydir[3 * r + 2];
    }
}

// X0 row of the tile: [e(32) | PE3(e) (192) | PE5(dists) (60) | 1 | 0 0 0], the row's raw weight and layer-3 extras
// (point_aggregators.py:773-784, :425-428, :506, :566; networks.py:175-190)
// MIX: the tile in the mixed format of mixq.h (columns < 256: h plane + e4m3 units; from 256 on the two f16 planes, h to nearest)
// HR: f16x3.h's two planes with h rounded to NEAREST (the training forward: its k-major copy-outs then read the h plane only)
template <bool PERS, bool MIX = false, bool HR = false>
__device__ __forceinline__ void f_build(const FwdArgs &a, const FGather &G, char *X, float *exb, float *wraw, int *sidx, int si, int p, int row, int q) {
    const float dwx = G.px - G.lx, dwy = G.py - G.ly, dwz = G.pz - G.lz;
    float ppx, ppy, pcz, spx, spy, scz;
    if (PERS) {
        ppx = G.ppx; ppy = G.ppy; pcz = G.ppz; spx = G.lpx; spy = G.lpy; scz = G.lpz;
    } else {
        float pcx, pcy, scx, scy;
        rot3(a.cam.camrot, G.px - a.cam.campos[0], G.py - a.cam.campos[1], G.pz - a.cam.campos[2], false, pcx, pcy, pcz);
        rot3(a.cam.camrot, G.lx - a.cam.campos[0], G.ly - a.cam.campos[1], G.lz - a.cam.campos[2], false, scx, scy, scz);
        ppx = pcx / pcz; ppy = pcy / pcz; spx = scx / scz; spy = scy / scz;
    }
    float d0, d1, d2;
    rot3(a.cam.rw2c, dwx, dwy, dwz, true, d0, d1, d2);
    const float d3 = ppx * pcz - spx * scz, d4 = ppy * pcz - spy * scz, d5 = pcz - scz;
    const float da = q == 0 ? d0 : q == 1 ? d1 : q == 2 ? d2 : d3;
    const float db = q == 0 ? d4 : d5;
    // the thread's 8 embedding dims and their 3 octaves
    const float e[8] = {G.e0.x, G.e0.y, G.e0.z, G.e0.w, G.e1.x, G.e1.y, G.e1.z, G.e1.w};
    if (MIX) {
        pn_xq_store4(X, row, EPT * q, e[0], e[1], e[2], e[3]);
        pn_xq_store4(X, row, EPT * q + 4, e[4], e[5], e[6], e[7]);
        // two dims = 12 consecutive columns (a multiple of four from column 32 + 48 q on): three 4-column stores
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const int col = PN_F + (EPT * q + i) * 6;
            float s0[3], c0[3], s1[3], c1[3];
            pn_pe_octaves<3>(e[i], s0, c0);
            pn_pe_octaves<3>(e[i + 1], s1, c1);
            pn_xq_store4(X, row, col, s0[0], c0[0], s0[1], c0[1]);
            pn_xq_store4(X, row, col + 4, s0[2], c0[2], s1[0], c1[0]);
            pn_xq_store4(X, row, col + 8, s1[1], c1[1], s1[2], c1[2]);
        }
    } else {
    if (HR) { pn_xt_store4(X, row, EPT * q, e[0], e[1], e[2], e[3]); pn_xt_store4(X, row, EPT * q + 4, e[4], e[5], e[6], e[7]); }
    else { pn_x_store4<false>(X, row, EPT * q, e[0], e[1], e[2], e[3]); pn_x_store4<false>(X, row, EPT * q + 4, e[4], e[5], e[6], e[7]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int dd = EPT * q + i;
        float s[3], c[3];
        pn_pe_octaves<3>(e[i], s, c);
        if (HR) {
            pn_xt_store2(X, row, PN_F + dd * 6, s[0], c[0]);
            pn_xt_store2(X, row, PN_F + dd * 6 + 2, s[1], c[1]);
            pn_xt_store2(X, row, PN_F + dd * 6 + 4, s[2], c[2]);
        } else {
            pn_x_store2(X, row, PN_F + dd * 6, s[0], c[0]);
            pn_x_store2(X, row, PN_F + dd * 6 + 2, s[1], c[1]);
            pn_x_store2(X, row, PN_F + dd * 6 + 4, s[2], c[2]);
        }
        if (PN_NW == 8 && (i & 1)) __builtin_amdgcn_sched_barrier(0);      // (register budget of the 8-wave organisation)
    }
    }
    // PE5 of distance components q and q + 4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int comp = q + 4 * j;
        if (comp < 6) {
            float s[5], c[5];
            pn_pe_octaves<5>(j == 0 ? da : db, s, c);
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                if (MIX) pn_xa_store2(X, row, PN_F * 7 + (comp * 5 + f) * 2, s[f], c[f]);
                else if (HR) pn_xt_store2(X, row, PN_F * 7 + (comp * 5 + f) * 2, s[f], c[f]);
                else pn_x_store2(X, row, PN_F * 7 + (comp * 5 + f) * 2, s[f], c[f]);
            }
        }
    }
    if (q == 3) {
        if (MIX || HR) pn_xt_store4(X, row, PN_ONES1, 1.f, 0.f, 0.f, 0.f);
        else pn_x_store4<false>(X, row, PN_ONES1, 1.f, 0.f, 0.f, 0.f);
    }
    if (q == 0) {
        float vx, vy, vz, qx, qy, qz;
        rot3(a.cam.rw2c, G.rx, G.ry, G.rz, true, vx, vy, vz);
        rot3(a.cam.rw2c, G.dxv, G.dyv, G.dzv, true, qx, qy, qz);
        float *ex = exb + row * 8;
        *reinterpret_cast<float4 *>(ex) = make_float4(G.cx, G.cy, G.cz, qx - vx);
        *reinterpret_cast<float4 *>(ex + 4) = make_float4(qy - vy, qz - vz, qx * vx + qy * vy + qz * vz, 1.f);
        wraw[row] = p >= 0 ? 1.0f / fmaxf(sqrtf(dwx * dwx + dwy * dwy + dwz * dwz), 1e-6f) : 0.f;
        sidx[row] = si;
    }
}

// epilogue of a layer: accumulators + bias, LeakyReLU, sign bits, both planes -> the tile (columns 0..255).
// Sign word: one v_alignbit_b32 per element shifts the element's sign bit into a 32-bit accumulator (MSB first), element
// e = ((fb * 2 + rb) * 4 + g) * 4 + i of a lane -> bit 31 - (e & 31) of half e >> 5; bit set = negative = slope 0.01 in the backward
// The bias is the accumulators' INITIAL value (D = W X^T + b: a lane's element (fb, rb, g, i) belongs to feature pn_d_feat(..) + i, the
// same for both row blocks): requested before the tile's copy-out, in the accumulators when the GEMM starts -- the epilogue then holds
// no global data at all (round 2 loaded it behind the GEMM, and behind the next tile's gather in the in-order vmcnt queue).
__device__ __forceinline__ void f_acc_bias(const float *__restrict__ bias, int wave, int lane, f32x16 (&acc)[PN_NFB][2]) {
#pragma unroll
    for (int fb = 0; fb < PN_NFB; ++fb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4 *>(bias + pn_d_feat(PN_NFB * wave + fb, g, lane));
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) { acc[fb][rb][4 * g] = b.x; acc[fb][rb][4 * g + 1] = b.y; acc[fb][rb][4 * g + 2] = b.z; acc[fb][rb][4 * g + 3] = b.w; }
        }
}
// epilogue of a layer: LeakyReLU of the accumulators (bias included), sign bits, both planes -> the tile (columns 0..255).
// Sign word: one v_alignbit_b32 per element shifts the element's sign bit into a 32-bit accumulator (MSB first), element
// e = (rb * 4 + g) * 4 + i of a lane's feature block -> bit 31 - e of the block's word; bit set = negative = slope 0.01 in the backward.
// mw[fb] = the word of the wave's feature block fb (global block PN_NFB * wave + fb): stored as lmask[tile][layer][block][lane].
template <bool BITS, bool MIX = false, bool HR = false>
__device__ __forceinline__ void f_epilogue(const f32x16 (&acc)[PN_NFB][2], char *X, int wave, int lane, unsigned (&mw)[PN_NFB]) {
#pragma unroll
    for (int fb = 0; fb < PN_NFB; ++fb) mw[fb] = 0u;
#pragma unroll
    for (int fb = 0; fb < PN_NFB; ++fb)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = pn_d_feat(PN_NFB * wave + fb, g, lane);
                float v[4] = {acc[fb][rb][4 * g], acc[fb][rb][4 * g + 1], acc[fb][rb][4 * g + 2], acc[fb][rb][4 * g + 3]};
                if (BITS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) mw[fb] = __builtin_amdgcn_alignbit(mw[fb], __float_as_uint(v[i]), 31);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.01f * v[i]);
                if (MIX) pn_xq_store4(X, 32 * rb + (lane & 31), f0, v[0], v[1], v[2], v[3]);
                else if (HR) pn_xt_store4(X, 32 * rb + (lane & 31), f0, v[0], v[1], v[2], v[3]);
                else pn_x_store4<false>(X, 32 * rb + (lane & 31), f0, v[0], v[1], v[2], v[3]);
            }
}
// the sign words of a layer -> lmask[gtile][layer][block][lane]
__device__ __forceinline__ void f_store_masks(unsigned *__restrict__ lmask, long long gtile, int layer, int wave, int lane, const unsigned (&mw)[PN_NFB]) {
#pragma unroll
    for (int fb = 0; fb < PN_NFB; ++fb) lmask[((gtile * 3 + layer) * 8 + PN_NFB * wave + fb) * 64 + lane] = mw[fb];
}

__device__ __forceinline__ void f_acc_zero(f32x16 (&acc)[PN_NFB][2]) {
#pragma unroll
    for (int i = 0; i < PN_NFB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// softplus(x) = log(1 + e^x) (raw2out_density, networks.py:262-267) on the hardware exponential / logarithm: absolute error <= 2e-7 for
// x <= 20 (log of 1 + e^x, e^x >= 0: the argument is >= 1, v_log_f32 / v_exp_f32 are good to an ulp), the identity above
__device__ __forceinline__ float pn_softplus(float x) {
#ifdef PN_EMU
    return x > 20.f ? x : log1pf(expf(x));
#else
    return x > 20.f ? x : __logf(1.0f + __expf(x));
#endif
}

// ---- the tile's tail in ONE pass over the h4 tile, for K in {1, 2, 4, 8} (the rows of a sample then never straddle a thread's 8 rows):
// thread -> columns 8 (tid & 31) .. + 7 of rows 8 (tid >> 5) .. + 7.  Per row: both planes read once (ds_read_b128), streamed to HBM as
// the backward's h4 (training), dotted with the alpha head's weights and added, weighted, to the thread's per-sample sums f.  The rows'
// alpha dot products are then reduced over the 32 lanes that share the rows by a transposing butterfly (8 -> 4 -> 2 -> 1 values per
// lane: 7 + 2 shuffles instead of 8 x 5), softplus runs once per row, sigma is summed over a sample's rows by log2(KC) more shuffles.
// Round 2 made three passes (alpha head, h4 copy, K-sums: 64 LDS reads per thread and two barriers); this is 16 reads and no barrier.
// KC = 0 (round 5): any other K (12 / 6 / 3 of the Barn configuration, ...): the per-ROW half of the pass -- both planes read once, streamed
// out as the backward's h4, the alpha head's dot product, butterfly, softplus -- with the rows' weighted alpha left in `wraw` for the K-sums
// pass that follows behind one barrier (a sample's rows then straddle the threads' 8-row strips).  The three-pass form it replaces read the
// tile three times with two barriers.
template <int KC, bool TRAIN>
__device__ __forceinline__ void f_tail(const FwdArgs &a, const char *X, const float *w5s, const float *wrow, const int *sidx, float b5,
                                       long long tile, long long gtile, int tid, float *wraw = nullptr) {
    constexpr int NS = KC == 0 ? 1 : 8 / (KC == 0 ? 1 : KC);                      // samples per thread
    const int lane = tid & 63, cg = tid & 31, r0 = 8 * (tid >> 5);
    const float4 wa = *reinterpret_cast<const float4 *>(w5s + 8 * cg), wb = *reinterpret_cast<const float4 *>(w5s + 8 * cg + 4);
    float pa[8];
    float4 fa[NS], fb[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) fa[j] = fb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = r0 + i;
        const uint4 h = *reinterpret_cast<const uint4 *>(X + r * PN_XRS + cg * 16);
        const uint4 m = *reinterpret_cast<const uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16);
        if (TRAIN) {
            pn_f4 th = {__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(h.z), __uint_as_float(h.w)};
            pn_f4 tm = {__uint_as_float(m.x), __uint_as_float(m.y), __uint_as_float(m.z), __uint_as_float(m.w)};
            PN_STREAM_STORE(th, reinterpret_cast<pn_f4 *>(a.sv.h4r + (gtile * PN_TILE + r) * 32 + cg));
            PN_STREAM_STORE(tm, reinterpret_cast<pn_f4 *>(a.sv.h4r + (a.sv.rows + gtile * PN_TILE + r) * 32 + cg));
        }
        float s = 0.f;
        s = pn_fma2_lo(h.x, m.x, wa.x, s); s = pn_fma2_hi(h.x, m.x, wa.y, s); s = pn_fma2_lo(h.y, m.y, wa.z, s); s = pn_fma2_hi(h.y, m.y, wa.w, s);
        s = pn_fma2_lo(h.z, m.z, wb.x, s); s = pn_fma2_hi(h.z, m.z, wb.y, s); s = pn_fma2_lo(h.w, m.w, wb.z, s); s = pn_fma2_hi(h.w, m.w, wb.w, s);
        pa[i] = s;
        if (KC != 0) {
            const float w = wrow[r];
            float4 &f0 = fa[i / (KC == 0 ? 1 : KC)], &f1 = fb[i / (KC == 0 ? 1 : KC)];
            f0.x = pn_fma2_lo(h.x, m.x, w, f0.x); f0.y = pn_fma2_hi(h.x, m.x, w, f0.y); f0.z = pn_fma2_lo(h.y, m.y, w, f0.z); f0.w = pn_fma2_hi(h.y, m.y, w, f0.w);
            f1.x = pn_fma2_lo(h.z, m.z, w, f1.x); f1.y = pn_fma2_hi(h.z, m.z, w, f1.y); f1.z = pn_fma2_lo(h.w, m.w, w, f1.z); f1.w = pn_fma2_hi(h.w, m.w, w, f1.w);
        }
        // 128 registers per wave in the 8-wave organisation: at most two rows' planes in flight (the scheduler otherwise hoists all 16 reads)
        if (PN_NW == 8 && (i & 1)) __builtin_amdgcn_sched_barrier(0);
    }
    // f rows of the thread's samples (class-ordered list: the colour MLP reads them in that order)
#pragma unroll
    for (int j = 0; j < (KC == 0 ? 0 : NS); ++j) {
        const long long vs = tile * (PN_TILE / (KC == 0 ? 1 : KC)) + (r0 / (KC == 0 ? 1 : KC)) + j;
        if (vs < a.cap_samples) {
            *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + 8 * cg) = fa[j];
            *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + 8 * cg + 4) = fb[j];
        }
    }
    // transposing butterfly over the 32 lanes of the row set: after the three halving steps a lane holds ONE row's partial sum,
    // row index i = 4 b4 + 2 b3 + b2 (bits of the lane), then two full steps over bits 1, 0
    {
        const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float send = b4 ? pa[j] : pa[j + 4], keep = b4 ? pa[j + 4] : pa[j]; pa[j] = keep + __shfl_xor(send, 16, 64); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const float send = b3 ? pa[j] : pa[j + 2], keep = b3 ? pa[j + 2] : pa[j]; pa[j] = keep + __shfl_xor(send, 8, 64); }
        { const float send = b2 ? pa[0] : pa[1], keep = b2 ? pa[1] : pa[0]; pa[0] = keep + __shfl_xor(send, 4, 64); }
        pa[0] += __shfl_xor(pa[0], 2, 64);
        pa[0] += __shfl_xor(pa[0], 1, 64);
        const int i = ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1), r = r0 + i;
        const float x = pa[0] + b5 - 1.0f;
        float sg = pn_softplus(x) * wrow[r];
        if (TRAIN && (lane & 3) == 0) a.sv.arow[gtile * PN_TILE + r] = x;
        if (KC == 0) {              // the row's share of sigma: summed per sample by the K-sums pass
            if ((lane & 3) == 0) wraw[r] = sg;
            return;
        }
        // sigma of a sample = sum over its KC rows: rows i differ in the low log2(KC) bits of i = lane bits 2 .. (b2 is i's bit 0)
        if (KC >= 2) sg += __shfl_xor(sg, 4, 64);
        if (KC >= 4) sg += __shfl_xor(sg, 8, 64);
        if (KC >= 8) sg += __shfl_xor(sg, 16, 64);
        if ((lane & 3) == 0 && (i % (KC == 0 ? 1 : KC)) == 0) {
            const int si = sidx[r];
            if (si >= 0) a.decoded[(long long)si * 4] = sg;
        }
    }
}

#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_fwd);
#endif
// WG2: the two-plane weight-gradient mode (pnerf_set_wgrad_planes(2)): every saved GEMM operand also leaves its residual plane, X0 whole
// NP = 4: the mixed format of mixq.h (f16 h.h + e4m3 cross terms: the default since round 6); NP = 3 / 2: f16x3.h's three / two f16 products
template <bool TRAIN, bool PERS, int NP, bool WG2 = false>
__global__ __launch_bounds__(PN_NTHR, PN_NW / 2) void k_agg_forward(FwdArgs a) {
    constexpr bool MIX = NP == 4;
    constexpr bool HR = !MIX;                       // f16x3 tiles with h rounded to nearest (inference and training alike: the same bits); training: h-only copy-outs
    constexpr bool HC = TRAIN && !WG2;              // the k-major copy-outs read the h plane alone
    constexpr int NPC = MIX ? 3 : NP;          // (what the classic templates are instantiated with where MIX compiles them away)
    static_assert(!(MIX && WG2), "the two-plane weight-gradient mode keeps f16x3.h's arithmetic everywhere");
    pn_mode_saturate();
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    char *X = smem_f;
    float *exb = reinterpret_cast<float *>(smem_f + FL_EX), *w5s = reinterpret_cast<float *>(smem_f + FL_W5);
    float *wraw = reinterpret_cast<float *>(smem_f + FL_ROW), *wrow = wraw + PN_TILE, *wnrm = wrow + PN_TILE;
    int *sidx = reinterpret_cast<int *>(wnrm + PN_TILE);
    const int tid0 = threadIdx.x;
    const int K = a.K, TS = a.TS;
    const unsigned kinv = pn_kinv(K);
    // this launch processes one sample class: its list, its run of tiles, its range of per-sample rows
    const int Ns = a.cls_info[PN_CI_COUNT + a.cls];
    const long long vb = a.cls_info[PN_CI_VBASE + a.cls], tb = a.cls_info[PN_CI_TBASE + a.cls];
    a.valid_list = a.cls_list + vb; a.cap_samples = Ns;
    a.sv.fs += vb * PN_H;
    const long long ntiles = ((long long)Ns + TS - 1) / TS;
    const float *P = a.params;
    const char *img = reinterpret_cast<const char *>(a.packed);
    if (tid0 < PN_H) w5s[tid0] = P[PO_W5 + tid0];
    const float b5 = P[PO_B5];
    // tiles of a workgroup: blockIdx.x, blockIdx.x + gridDim.x, ... (the chip works on one moving window of the saved area);
    // dev A/B -DPN_TILE_BLOCKED: one contiguous run of tiles per workgroup (consecutive tiles = consecutive samples of a ray on ONE CU)
#ifdef PN_TILE_BLOCKED
    const long long per_wg = (ntiles + gridDim.x - 1) / gridDim.x, stride = 1, tile_first = blockIdx.x * per_wg;
    const long long tile_last = tile_first + per_wg < ntiles ? tile_first + per_wg : ntiles;
#else
    const long long stride = gridDim.x, tile_first = blockIdx.x, tile_last = ntiles;
#endif
    if (tile_first >= tile_last) return;

    // index pipeline of this thread's row: (si0, p0) current tile, (si1, p1) next, si2 the one after
    long long tile = tile_first;
    int si0, si1, si2, p0, p1;
    FGather G;
    // Roles in the 8-wave organisation: the LAST PN_ETHR threads (waves 4..7) own the gather + feature build + row weights (4 threads per
    // tile row), the FIRST PN_ETHR threads (waves 0..3) own the tail.  The next tile's gathered point data is in flight across the tail:
    // with both roles on the same waves it does not fit 128 registers beside the tail's rows (the compiler spilled it, i.e. waited for it).
    si0 = si1 = si2 = p0 = p1 = -1;
    if (tid0 >= PN_NTHR - PN_ETHR) {
        const int bt = tid0 - (PN_NTHR - PN_ETHR), row = bt / TPR, q = bt % TPR, k = row - pn_row_div(row, kinv) * K;
        si0 = f_sample_of(a, tile, row, Ns, kinv); si1 = tile + stride < tile_last ? f_sample_of(a, tile + stride, row, Ns, kinv) : -1;
        si2 = tile + 2 * stride < tile_last ? f_sample_of(a, tile + 2 * stride, row, Ns, kinv) : -1;
        p0 = si0 >= 0 ? a.pidx[(long long)si0 * a.Kstride + k] : -1;
        p1 = si1 >= 0 ? a.pidx[(long long)si1 * a.Kstride + k] : -1;
        f_gather<PERS>(a, G, si0, p0, q);
    }

    f32x16 acc[PN_NFB][2];
    PN_TR_ITER_DECL;
    for (; tile < tile_last; tile += stride) {
        PN_TR_ITER_NEXT;
        // thread-index-derived offsets are recomputed per tile: hoisted out of the loop they become hundreds of loop-carried
        // registers (every LDS / bias / image address of every unrolled store) and spill
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool ew = PN_NTHR == PN_ETHR || tid < PN_ETHR;               // this thread works in the tail
        const bool bw = PN_NTHR == PN_ETHR || tid >= PN_NTHR - PN_ETHR;    // this thread works in the gather / feature build
        const int bt = bw ? tid - (PN_NTHR - PN_ETHR) : 0, row = bt / TPR, q = bt % TPR, k = row - pn_row_div(row, kinv) * K;
        const long long gtile = tb + tile;               // tile index inside the saved area
        PN_LDS_BARRIER();                                 // the previous tile's readers are done with X and the row arrays
        PN_TR(pn_trace_fwd, 0); PN_TR_HWID(pn_trace_fwd);
        if (bw) f_build<PERS, MIX, HR>(a, G, X, exb, wraw, sidx, si0, p0, row, q);
        // Round 4: what the next GEMM needs from GLOBAL memory -- its bias (the accumulators' initial value) and its first weight-fragment
        // chunks -- is requested in front of the barrier that precedes it, not behind: neither depends on LDS, and the L2 round trip
        // (0.6 .. 1.1 us per layer in profiles/r03_phase_trace.json: the "acc = bias" phases) passes under the barrier wait.
        f_acc_bias(P + PO_B1, wave, lane, acc);
        PnGemmW<18, 8, PN_NFB, PN_WPF, NPC> W1;
        PnMixW<PN_MIX_NS, 2, 8, PN_NFB> M1;
        if constexpr (MIX) M1.prefetch(img + PKM_F1, PN_NFB * wave, lane);
        else W1.prefetch(reinterpret_cast<const uint4 *>(img + PKH_F1), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        if (bw && q == 0) {      // weights of the row: normalise over the K slots, multiply by the clamped confidence (:801-811)
            const int ls = pn_row_div(row, kinv);
            float wn = 0.f, w = 0.f;
            if (si0 >= 0) {
                float sum = 0.f;
                for (int kk = 0; kk < K; ++kk) sum += wraw[ls * K + kk];
                wn = wraw[row] / fmaxf(sum, 1e-8f);
                w = wn * fminf(fmaxf(G.cf, 1e-4f), 1.0f);
                a.weight[(long long)si0 * a.Kstride + k] = wn;
            }
            wnrm[row] = wn; wrow[row] = w;
            if (TRAIN) a.sv.rmeta[gtile * PN_TILE + row] = make_int4(si0, si0 >= 0 ? p0 : -1, __float_as_int(wn), __float_as_int(w));
        }
        unsigned mask[PN_NFB];
        PN_TR(pn_trace_fwd, 1);
        // ---- layer 1: 288 -> 256.  Training: the layer's input tile is copied out (k-major planes for the weight-gradient GEMM) BEHIND the
        // layer's GEMM, not in front of it: vmcnt counts loads and stores of a wave in ONE in-order queue, so a GEMM that starts right
        // behind 20 stores waits for their acknowledgements from HBM before its first weight fragment counts as arrived (round 2 order:
        // every GEMM phase carried 1 .. 3 us of that).  Behind the GEMM the stores have the epilogue and two barriers to drain.
        PN_TR(pn_trace_fwd, 2);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 2, 8, PN_NFB>(X, M1, lane, acc);
        else pn_gemm_f16x3_run<18, 8, PN_NFB, PN_WPF, NPC>(X, W1, lane, acc);
        if (TRAIN) {           // (behind the GEMM: see above)
            if (HC) {         // the tile's h plane IS the nearest f16: the k-major plane is its transpose
                if (a.save_x0) pn_copy_out_kmajor_h<PN_NF1, PN_XRS, PN_NW>(X, a.sv.x0k, gtile * 8, tid);
                else pn_copy_out_kmajor_cols64_h<224, PN_NW>(X, a.sv.x0k, gtile * 8, tid);
            } else if (WG2) pn_copy_out_kmajor<PN_NF1, true, PN_NW>(X, a.sv.x0k, gtile * 8, tid, a.sv.x0m);
            else if (a.save_x0) pn_copy_out_kmajor<PN_NF1, false, PN_NW>(X, a.sv.x0k, gtile * 8, tid);
            else pn_copy_out_kmajor_cols64<224, PN_NW>(X, a.sv.x0k, gtile * 8, tid);       // the fused path: only the last 64 columns (k_wgrad_x0)
        }
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 3);
        f_epilogue<TRAIN, MIX, HR>(acc, X, wave, lane, mask);
        if (TRAIN) f_store_masks(a.sv.lmask, gtile, 0, wave, lane, mask);
        f_acc_bias(P + PO_B2, wave, lane, acc);
        PnGemmW<16, 8, PN_NFB, PN_WPF, NPC> W2;
        PnMixW<PN_MIX_NS, 0, 8, PN_NFB> M2;
        if constexpr (MIX) M2.prefetch(img + PKM_F2, PN_NFB * wave, lane);
        else W2.prefetch(reinterpret_cast<const uint4 *>(img + PKH_F2), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 4);
        // ---- layer 2: 256 -> 256
        PN_TR(pn_trace_fwd, 5);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 0, 8, PN_NFB>(X, M2, lane, acc);
        else pn_gemm_f16x3_run<16, 8, PN_NFB, PN_WPF, NPC>(X, W2, lane, acc);
        if (TRAIN) {      // (behind the GEMM: see below)
            if (HC) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.h1k, gtile * 8, tid);
            else pn_copy_out_kmajor<PN_H, WG2, PN_NW>(X, a.sv.h1k, gtile * 8, tid, a.sv.h1m);
        }
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 6);
        f_epilogue<TRAIN, MIX, HR>(acc, X, wave, lane, mask);
        if (TRAIN) f_store_masks(a.sv.lmask, gtile, 1, wave, lane, mask);
        if (tid < PN_TILE) {     // the row's extras next to h2: columns 256..262, the ones column, zeros up to 271
            const float4 u = *reinterpret_cast<const float4 *>(exb + tid * 8), v = *reinterpret_cast<const float4 *>(exb + tid * 8 + 4);
            if (MIX || HR) {
                pn_xt_store4(X, tid, PN_H, u.x, u.y, u.z, u.w);
                pn_xt_store4(X, tid, PN_H + 4, v.x, v.y, v.z, v.w);
                pn_xt_store4(X, tid, PN_H + 8, 0.f, 0.f, 0.f, 0.f);
                pn_xt_store4(X, tid, PN_H + 12, 0.f, 0.f, 0.f, 0.f);
            } else {
                pn_x_store4<false>(X, tid, PN_H, u.x, u.y, u.z, u.w);
                pn_x_store4<false>(X, tid, PN_H + 4, v.x, v.y, v.z, v.w);
                pn_x_store4<false>(X, tid, PN_H + 8, 0.f, 0.f, 0.f, 0.f);
                pn_x_store4<false>(X, tid, PN_H + 12, 0.f, 0.f, 0.f, 0.f);
            }
        }
        f_acc_bias(P + PO_B3, wave, lane, acc);
        PnGemmW<17, 8, PN_NFB, PN_WPF, NPC> W3;
        PnMixW<PN_MIX_NS, 1, 8, PN_NFB> M3;
        if constexpr (MIX) M3.prefetch(img + PKM_F3, PN_NFB * wave, lane);
        else W3.prefetch(reinterpret_cast<const uint4 *>(img + PKH_F3), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 7);
        // ---- layer 3: 256 + 7 -> 256
        PN_TR(pn_trace_fwd, 8);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 1, 8, PN_NFB>(X, M3, lane, acc);
        else pn_gemm_f16x3_run<17, 8, PN_NFB, PN_WPF, NPC>(X, W3, lane, acc);
        if (TRAIN) {      // (behind the GEMM: see below)
            if (HC) pn_copy_out_kmajor_h<PN_NF1, PN_XRS, PN_NW>(X, a.sv.h2k, gtile * 8, tid);
            else pn_copy_out_kmajor<PN_NF1, WG2, PN_NW>(X, a.sv.h2k, gtile * 8, tid, a.sv.h2m);
        }
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 9);
        f_epilogue<TRAIN, MIX, HR>(acc, X, wave, lane, mask);
        if (TRAIN) f_store_masks(a.sv.lmask, gtile, 2, wave, lane, mask);
        f_acc_bias(P + PO_B4, wave, lane, acc);
        PnGemmW<16, 8, PN_NFB, PN_WPF, NPC> W4;
        PnMixW<PN_MIX_NS, 0, 8, PN_NFB> M4;
        if constexpr (MIX) M4.prefetch(img + PKM_F4, PN_NFB * wave, lane);
        else W4.prefetch(reinterpret_cast<const uint4 *>(img + PKH_F4), PN_NFB * wave, lane);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 10);
        // ---- layer 4: 256 -> 256
        PN_TR(pn_trace_fwd, 11);
        if constexpr (MIX) pn_gemm_mix_run<PN_MIX_NS, 0, 8, PN_NFB>(X, M4, lane, acc);
        else pn_gemm_f16x3_run<16, 8, PN_NFB, PN_WPF, NPC>(X, W4, lane, acc);
        if (TRAIN) {
            if (HC) pn_copy_out_kmajor_h<PN_H, PN_XRS, PN_NW>(X, a.sv.h3k, gtile * 8, tid);
            else pn_copy_out_kmajor<PN_H, WG2, PN_NW>(X, a.sv.h3k, gtile * 8, tid, a.sv.h3m);
        }
        PN_TR(pn_trace_fwd, 12);
        // the next tile's point data and the indices of the two after it: requested here, consumed at the top of the next
        // iteration -- their HBM latency passes under the element-wise tail of this tile (nothing of this tile waits for memory any more)
        const float cf_cur = G.cf;
        (void)cf_cur;
        const int si_next = si1, p_next = p1;
        // (8 waves: the gather is issued by the build waves WHILE the tail waves run the tail, in the other arm of that branch -- then the
        //  tail's code never holds the gathered registers)
        if (PN_NW != 8 && tile + stride < tile_last) f_gather<PERS>(a, G, si1, p1, q);
        const int p2 = si2 >= 0 ? a.pidx[(long long)si2 * a.Kstride + k] : -1;
        const int si3 = (bw && tile + 3 * stride < tile_last) ? f_sample_of(a, tile + 3 * stride, row, Ns, kinv) : -1;
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 13);
        f_epilogue<false>(acc, X, wave, lane, mask);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 14);
        if (K == 8 || K == 4 || K == 2 || K == 1) {
            // ---- alpha head + h4 copy + K-weighted sums + sigma in one pass (f_tail)
            if (!ew) { if (PN_NW == 8 && tile + stride < tile_last) f_gather<PERS>(a, G, si1, p1, q); }
            else if (K == 8) f_tail<8, TRAIN>(a, X, w5s, wrow, sidx, b5, tile, gtile, tid);
            else if (K == 4) f_tail<4, TRAIN>(a, X, w5s, wrow, sidx, b5, tile, gtile, tid);
            else if (K == 2) f_tail<2, TRAIN>(a, X, w5s, wrow, sidx, b5, tile, gtile, tid);
            else f_tail<1, TRAIN>(a, X, w5s, wrow, sidx, b5, tile, gtile, tid);
            PN_TR(pn_trace_fwd, 15);
        } else {
        // ---- (any other K: two passes) the per-row half in f_tail's mapping (alpha head 256 -> 1, softplus(x - 1), raw2out_density :262-265; h4 planes
        // streamed out for the backward), then the K-weighted sums
        if (PN_NW == 8 && bw && tile + stride < tile_last) f_gather<PERS>(a, G, si1, p1, q);      // (8 waves: the build waves' prefetch, see above)
        if (ew) f_tail<0, TRAIN>(a, X, w5s, wrow, sidx, b5, tile, gtile, tid, wraw);
        PN_LDS_BARRIER();
        PN_TR(pn_trace_fwd, 15);
        // ---- K-weighted sums -> f[256] per sample (HBM), sigma
        for (int e = tid; e < TS * 32; e += PN_NTHR) {          // item = (sample, 8 columns): 16-byte reads of both planes, K rows
            const int ls = e >> 5, cg = e & 31;
            float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
            for (int kk = 0; kk < K; ++kk) {
                const int r = ls * K + kk;
                const uint4 h = *reinterpret_cast<const uint4 *>(X + r * PN_XRS + cg * 16);
                const uint4 m = *reinterpret_cast<const uint4 *>(X + PN_XPLANE + r * PN_XRS + cg * 16);
                const float w = wrow[r];
                f0.x = pn_fma2_lo(h.x, m.x, w, f0.x); f0.y = pn_fma2_hi(h.x, m.x, w, f0.y); f0.z = pn_fma2_lo(h.y, m.y, w, f0.z); f0.w = pn_fma2_hi(h.y, m.y, w, f0.w);
                f1.x = pn_fma2_lo(h.z, m.z, w, f1.x); f1.y = pn_fma2_hi(h.z, m.z, w, f1.y); f1.z = pn_fma2_lo(h.w, m.w, w, f1.z); f1.w = pn_fma2_hi(h.w, m.w, w, f1.w);
            }
            const long long vs = tile * TS + ls;
            if (vs < a.cap_samples) {
                *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + 8 * cg) = f0;
                *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + 8 * cg + 4) = f1;
            }
        }
        if (tid < TS) {
            const int si = sidx[tid * K];
            if (si >= 0) {
                float sg = 0.f;
                for (int kk = 0; kk < K; ++kk) sg += wraw[tid * K + kk];
                a.decoded[(long long)si * 4] = sg;
            }
        }
        }
        si0 = si_next; p0 = p_next; si1 = si2; p1 = p2; si2 = si3;
        PN_TR(pn_trace_fwd, 16);
    }
}

// ------------------------------------------------------------------------------ colour MLP forward
// 64 valid samples per tile: [f (256) | view-direction encoding (24) | 0 (8)] -> 128 -> 128 -> 128 -> 3, the three 128-wide layers as
// two-plane f16 GEMMs (f16x3.h: one 32-feature block per wave, 18 / 8 / 8 chunks), the 3-wide output layer and the sigmoid on the
// VALU.  One activation tile in LDS, reused layer after layer; the post-activations the backward needs (LeakyReLU masks, the fp32
// weight-gradient GEMMs of these three layers) leave from the accumulator registers as fp32 rows.
constexpr int CL_W4 = PN_XBYTES, CL_BYTES = CL_W4 + 3 * PN_HC * 4;
static_assert(2 * CL_BYTES <= 160 * 1024, "two colour workgroups must fit the 160 KB LDS");

__device__ __forceinline__ void c_load_bias(const float *__restrict__ bias, int wave, int lane, float4 (&b)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {         // (scalar loads: the colour tensors sit at odd float offsets of the parameter vector)
        const float *p = bias + pn_d_feat(wave, g, lane);
        b[g] = make_float4(p[0], p[1], p[2], p[3]);
    }
}
// bias + LeakyReLU of the wave's feature block, two planes into the tile.  Training: SAVE32 -> the fp32 rows to `save` (the last layer:
// the backward needs its values); else the sign bits of the lane's 32 pre-activations (rb, g, i order, first element in bit 31)
template <bool TRAIN, bool SAVE32>
__device__ __forceinline__ unsigned c_epilogue(const f32x16 (&acc)[2][2], const float4 (&bias)[4], char *X, int wave, int lane, float *__restrict__ save, long long grow0) {
    unsigned mw = 0u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = pn_d_feat(wave, g, lane), row = 32 * rb + (lane & 31);
            const float4 b = bias[g];
            float v[4] = {acc[0][rb][4 * g] + b.x, acc[0][rb][4 * g + 1] + b.y, acc[0][rb][4 * g + 2] + b.z, acc[0][rb][4 * g + 3] + b.w};
            if (TRAIN && !SAVE32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(v[i]), 31);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.01f * v[i]);
            if (TRAIN && SAVE32) {
                pn_f4 t = {v[0], v[1], v[2], v[3]};
                PN_REG_STORE(t, reinterpret_cast<pn_f4 *>(save + (grow0 + row) * PN_HC + f0));
            }
            pn_x_store4<false>(X, row, f0, v[0], v[1], v[2], v[3]);
        }
    return mw;
}
__device__ __forceinline__ void c_acc_zero(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
}

template <bool TRAIN, int NP, bool WG2 = false>
__global__ __launch_bounds__(256, 2) void k_color_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char *X = smem_c;
    float *w4s = reinterpret_cast<float *>(smem_c + CL_W4);       // [3][128] output layer
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;
    const char *img = reinterpret_cast<const char *>(a.packed);
    for (int i = threadIdx.x; i < 3 * PN_HC; i += 256) w4s[i] = P[PO_WC4 + i];
    const float b40 = P[PO_BC4], b41 = P[PO_BC4 + 1], b42 = P[PO_BC4 + 2];

    f32x16 acc[2][2];
    for (long long tile = blockIdx.x; tile * PN_CTILE < Ns; tile += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), row = tid >> 2, q = tid & 3;
        const long long grow0 = tile * PN_CTILE, vs = grow0 + row;
        const int si = vs < Ns ? a.valid_list[vs] : -1;
        // (round 4: the first weight-fragment chunks of a GEMM -- here 7 of a layer's 8 or 18 -- are requested in front of the barrier that
        //  precedes it; a colour tile has 3.5 us of MFMA work in ~25 us, the L2 round trips at the start of its three GEMMs were exposed)
        PnGemmW<18, 4, 1, 7, NP> WC1;
        WC1.prefetch(reinterpret_cast<const uint4 *>(img + PKH_FC1), wave, lane);
        PN_LDS_BARRIER();                                 // the previous tile's readers are done with X
        if (si >= 0) {
            // the row's four threads take interleaved 16-byte pieces (piece 4 c + q): a wave's load instruction reads whole 64-byte segments of
            // 16 rows, its LDS stores land on 8 consecutive banks per row (round 4; one contiguous quarter row per thread touched 64 different
            // segments per instruction and put threads 0 / 2 and 1 / 3 of a row on the same banks: half of this kernel's LDS cycles were conflicts)
            const float4 *f = reinterpret_cast<const float4 *>(a.sv.fs + vs * PN_H);
            float4 fv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) fv[c] = f[4 * c + q];
#pragma unroll
            for (int c = 0; c < 16; ++c) pn_x_store4<false>(X, row, 16 * c + 4 * q, fv[c].x, fv[c].y, fv[c].z, fv[c].w);
            {   // positional_encoding(viewdirs, 4, ori=True)[..., 3:] = [sin(v_d 2^f) (d-major) | cos(...)]   networks.py:185-187
                // thread q < 3 of the row: direction component q, its four frequencies (columns 256 + 4 q .. and 268 + 4 q ..)
                const int r = si / a.SR;
                float v[3], sn[4], cs[4];
                rot3(a.cam.rw2c, a.raydir[3 * r], a.raydir[3 * r + 1], a.raydir[3 * r + 2], true, v[0], v[1], v[2]);
                const float vq = q == 0 ? v[0] : (q == 1 ? v[1] : v[2]);
                pn_pe_octaves<4>(vq, sn, cs);
                if (q < 3) {
                    pn_x_store4<false>(X, row, PN_H + 4 * q, sn[0], sn[1], sn[2], sn[3]);
                    pn_x_store4<false>(X, row, PN_H + 12 + 4 * q, cs[0], cs[1], cs[2], cs[3]);
                } else {
                    pn_x_store4<false>(X, row, PN_H + 24, 0.f, 0.f, 0.f, 0.f);
                    pn_x_store4<false>(X, row, PN_H + 28, 0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 18; ++c) pn_x_store4<false>(X, row, 16 * c + 4 * q, 0.f, 0.f, 0.f, 0.f);
        }
        PN_LDS_BARRIER();
        float4 bias[4];
        // ---- layer 1: 280 (288) -> 128.  Training: every layer's input tile leaves k-major for the weight-gradient GEMM
        if (TRAIN) pn_copy_out_kmajor<PN_NF1, WG2>(X, a.sv.xck, tile * 8, tid, a.sv.xcm);
        c_acc_zero(acc);
        pn_gemm_f16x3_run<18, 4, 1, 7, NP>(X, WC1, lane, acc);
        c_load_bias(P + PO_BC1, wave, lane, bias);
        PN_LDS_BARRIER();                                 // every wave is done reading the input tile
        unsigned mw = c_epilogue<TRAIN, false>(acc, bias, X, wave, lane, nullptr, grow0);
        if (TRAIN) a.sv.cmask[(tile * 2 + 0) * 256 + tid] = mw;
        PnGemmW<8, 4, 1, 7, NP> WC2;
        WC2.prefetch(reinterpret_cast<const uint4 *>(img + PKH_FC2), wave, lane);
        PN_LDS_BARRIER();
        // ---- layer 2
        if (TRAIN) pn_copy_out_kmajor<PN_HC, WG2>(X, a.sv.c1k, tile * 8, tid, a.sv.c1m);
        c_acc_zero(acc);
        pn_gemm_f16x3_run<8, 4, 1, 7, NP>(X, WC2, lane, acc);
        c_load_bias(P + PO_BC2, wave, lane, bias);
        PN_LDS_BARRIER();
        mw = c_epilogue<TRAIN, false>(acc, bias, X, wave, lane, nullptr, grow0);
        if (TRAIN) a.sv.cmask[(tile * 2 + 1) * 256 + tid] = mw;
        PnGemmW<8, 4, 1, 7, NP> WC3;
        WC3.prefetch(reinterpret_cast<const uint4 *>(img + PKH_FC3), wave, lane);
        PN_LDS_BARRIER();
        // ---- layer 3
        if (TRAIN) pn_copy_out_kmajor<PN_HC, WG2>(X, a.sv.c2k, tile * 8, tid, a.sv.c2m);
        c_acc_zero(acc);
        pn_gemm_f16x3_run<8, 4, 1, 7, NP>(X, WC3, lane, acc);
        c_load_bias(P + PO_BC3, wave, lane, bias);
        PN_LDS_BARRIER();
        c_epilogue<TRAIN, true>(acc, bias, X, wave, lane, a.sv.c3, grow0);
        PN_LDS_BARRIER();
        // ---- output layer 128 -> 3 and the colour activation: 4 threads per row, 32 columns each
        {
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = q * 32 + 8 * j;
                o0 = pn_x_dot8(X, row, c0, *reinterpret_cast<const float4 *>(w4s + c0), *reinterpret_cast<const float4 *>(w4s + c0 + 4), o0);
                o1 = pn_x_dot8(X, row, c0, *reinterpret_cast<const float4 *>(w4s + PN_HC + c0), *reinterpret_cast<const float4 *>(w4s + PN_HC + c0 + 4), o1);
                o2 = pn_x_dot8(X, row, c0, *reinterpret_cast<const float4 *>(w4s + 2 * PN_HC + c0), *reinterpret_cast<const float4 *>(w4s + 2 * PN_HC + c0 + 4), o2);
            }
            o0 = group_sum<4>(o0); o1 = group_sum<4>(o1); o2 = group_sum<4>(o2);
            if (q == 0 && si >= 0) {
                float *o = a.decoded + (long long)si * 4;
                o[1] = 1.0f / (1.0f + expf(-(o0 + b40))) * 1.002f - 0.001f;                                     // raw2out_color :269-273
                o[2] = 1.0f / (1.0f + expf(-(o1 + b41))) * 1.002f - 0.001f;
                o[3] = 1.0f / (1.0f + expf(-(o2 + b42))) * 1.002f - 0.001f;
            }
        }
    }
}
}  // namespace

// products per multiply-add of the INFERENCE forward (training always runs three): see f16x3.h NP and include/pnerf.h
static int pn_inference_products = 3;
extern "C" int pnerf_set_inference_products(int n) {
    if (n != 2 && n != 3) return PNERF_E_INVAL;
    const int old = pn_inference_products;
    pn_inference_products = n;
    return old;
}

// planes per operand of the weight-gradient GEMMs (include/pnerf.h: pnerf_set_wgrad_planes)
static int pn_wgrad_planes_ = 1;
int pn_wgrad_planes() { return pn_wgrad_planes_; }
extern "C" int pnerf_set_wgrad_planes(int n) {
    if (n != 1 && n != 2) return PNERF_E_INVAL;
    const int old = pn_wgrad_planes_;
    pn_wgrad_planes_ = n;
    return old;
}

// e4m3 cross terms (mixq.h; 8: the default) or f16 cross terms (16: f16x3.h's three f16 products, the round-2..5 arithmetic) in the aggregator's
// tile GEMMs, forward and input-gradient chain (include/pnerf.h: pnerf_set_cross_terms)
static int pn_cross_terms_ = 8;
int pn_cross_terms() { return pn_cross_terms_; }
extern "C" int pnerf_set_cross_terms(int bits) {
    if (bits != 8 && bits != 16) return PNERF_E_INVAL;
    const int old = pn_cross_terms_;
    pn_cross_terms_ = bits;
    return old;
}
// which tile kernels run the mixed format when the cross terms are e4m3 (include/pnerf.h: pnerf_set_cross_terms_where): bit 0 = inference
// forward, bit 1 = training forward, bit 2 = backward (the input-gradient chain).  Default 4: the forward keeps f16 cross terms -- with e4m3 ones
// its sigma / RGB are 1e-5 .. 6e-5 from the fp32 oracle instead of 1e-6 (inside the 1e-4 bar), but pre-activations within that distance of zero
// take the other LeakyReLU branch than the oracle's, and the gradient tests against the oracle see those flips (profiles/r06_cross_terms_ab.json)
#ifndef PN_MIX_DEFAULT_MASK
#define PN_MIX_DEFAULT_MASK 4
#endif
static int pn_mix_mask_ = -1;
int pn_mix_mask() {
    if (pn_mix_mask_ < 0) {
        const char *e = getenv("PNERF_MIX_MASK");          // (dev A/B)
        pn_mix_mask_ = e ? atoi(e) & 7 : PN_MIX_DEFAULT_MASK;
    }
    return pn_cross_terms_ == 8 ? pn_mix_mask_ : 0;
}
extern "C" int pnerf_set_cross_terms_where(int mask) {
    if (mask < 0 || mask > 7) return PNERF_E_INVAL;
    (void)pn_mix_mask();
    const int old = pn_mix_mask_;
    pn_mix_mask_ = mask;
    return old;
}

// shared with render.hip
int pn_agg_forward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                          const float *d_raydir, const float *d_sample_loc, const float *d_xyz_pers, const float *d_loc_pers,
                          const int32_t *d_sample_pidx,
                          const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                          float *d_decoded, float *d_weight, const PnSaved &sv, long long cap_samples, bool train, bool save_x0,
                          hipStream_t s) {
    FwdArgs a;
    const bool wg2 = train && sv.wg2;                  // two-plane weight-gradient mode: residual planes, X0 saved whole
    if (wg2) save_x0 = true;
    a.save_x0 = save_x0 ? 1 : 0;
    a.cam = *cam;
    a.xyz = pts->xyz; a.emb = pts->embedding; a.conf = pts->conf; a.dir = pts->dir; a.color = pts->color;
    a.params = d_params; a.packed = (const float4 *)d_packed;
    a.raydir = d_raydir; a.sample_loc = d_sample_loc; a.xyz_pers = d_xyz_pers; a.loc_pers = d_loc_pers; a.pidx = d_sample_pidx; a.valid_list = d_valid_list; a.counters = d_counters;
    a.R = R; a.SR = SR; a.K = K; a.TS = pn_tile_samples(K);
    a.cap_samples = cap_samples;
    a.decoded = d_decoded; a.weight = d_weight; a.sv = sv;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const long long ctiles = (cap_samples + PN_CTILE - 1) / PN_CTILE;
    const int grid_c = (int)(ctiles < 2 * ncu ? (ctiles > 0 ? ctiles : 1) : 2 * ncu);       // two workgroups per CU
    const size_t lds_a = FL_BYTES, lds_c = CL_BYTES;
    const bool pers = d_xyz_pers != nullptr;
    const bool np2 = !train && pn_inference_products == 2;      // inference with the weights' high plane only (f16x3.h: NP)
    const bool mix = !wg2 && !np2 && (pn_mix_mask() & (train ? 2 : 1));      // mixq.h: f16 h.h + e4m3 cross terms
    const void *kfn = wg2   ? (pers ? (const void *)k_agg_forward<true, true, 3, true> : (const void *)k_agg_forward<true, false, 3, true>)
                    : mix   ? (train ? (pers ? (const void *)k_agg_forward<true, true, 4> : (const void *)k_agg_forward<true, false, 4>)
                                     : (pers ? (const void *)k_agg_forward<false, true, 4> : (const void *)k_agg_forward<false, false, 4>))
                    : train ? (pers ? (const void *)k_agg_forward<true, true, 3> : (const void *)k_agg_forward<true, false, 3>)
                    : np2   ? (pers ? (const void *)k_agg_forward<false, true, 2> : (const void *)k_agg_forward<false, false, 2>)
                            : (pers ? (const void *)k_agg_forward<false, true, 3> : (const void *)k_agg_forward<false, false, 3>);
    const void *cfn = wg2 ? (const void *)k_color_forward<true, 3, true> : train ? (const void *)k_color_forward<true, 3> : np2 ? (const void *)k_color_forward<false, 2> : (const void *)k_color_forward<false, 3>;
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipFuncSetAttribute(cfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
    int rc = pn_classify(sv, d_valid_list, d_counters, d_sample_pidx, K, cap_samples, train, save_x0, s);
    if (rc) return rc;
    a.cls_list = sv.cls_list; a.cls_info = sv.cls_info; a.Kstride = K;
    int kc[PN_NCLS];
    const int ncls = pn_class_slots(K, kc);
    {
        PnProfScope prof(PNK_AGG_FWD, s);
        for (int j = 0; j < ncls; ++j) {            // class sizes are only known on the device: the grid covers the worst case, surplus workgroups return at once
            a.cls = j; a.K = kc[j]; a.TS = pn_tile_samples(kc[j]);
            const long long tiles = (cap_samples + a.TS - 1) / a.TS;
            const int grid_a = (int)(tiles < 2LL * ncu ? (tiles > 0 ? tiles : 1) : 2LL * ncu);     // two workgroups per CU
            if (wg2 && pers) hipLaunchKernelGGL((k_agg_forward<true, true, 3, true>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (wg2) hipLaunchKernelGGL((k_agg_forward<true, false, 3, true>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (mix && train && pers) hipLaunchKernelGGL((k_agg_forward<true, true, 4>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (mix && train) hipLaunchKernelGGL((k_agg_forward<true, false, 4>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (mix && pers) hipLaunchKernelGGL((k_agg_forward<false, true, 4>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (mix) hipLaunchKernelGGL((k_agg_forward<false, false, 4>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (train && pers) hipLaunchKernelGGL((k_agg_forward<true, true, 3>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (train) hipLaunchKernelGGL((k_agg_forward<true, false, 3>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (np2 && pers) hipLaunchKernelGGL((k_agg_forward<false, true, 2>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (np2) hipLaunchKernelGGL((k_agg_forward<false, false, 2>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (pers) hipLaunchKernelGGL((k_agg_forward<false, true, 3>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else hipLaunchKernelGGL((k_agg_forward<false, false, 3>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
        }
    }
    a.K = K; a.TS = pn_tile_samples(K); a.valid_list = sv.cls_list;      // the colour MLP walks the class-ordered list: f rows are in that order
    {
        PnProfScope prof(PNK_COLOR_FWD, s);
        if (wg2) hipLaunchKernelGGL((k_color_forward<true, 3, true>), dim3(grid_c), dim3(256), lds_c, s, a);
        else if (train) hipLaunchKernelGGL((k_color_forward<true, 3>), dim3(grid_c), dim3(256), lds_c, s, a);
        else if (np2) hipLaunchKernelGGL((k_color_forward<false, 2>), dim3(grid_c), dim3(256), lds_c, s, a);
        else hipLaunchKernelGGL((k_color_forward<false, 3>), dim3(grid_c), dim3(256), lds_c, s, a);
    }
    PN_CHECK_LAUNCH();
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_fwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_fwd), bytes < sizeof(pn_trace_fwd) ? bytes : sizeof(pn_trace_fwd)) == hipSuccess ? 0 : -1;
}
#endif
