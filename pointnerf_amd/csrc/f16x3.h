// f16x3.h -- fp32-accurate GEMMs of the aggregator MLP on the f16 matrix pipe of gfx950 (round 2).
//
// gfx950 has no xf32/TF32 path: an fp32-input MFMA runs at 1/16 of the 16-bit rate (157 TFLOP/s against 2.5 PFLOP/s), and
// the parity bar (1e-4 on sigma / RGB against an fp32 reference) rules out plain f16/bf16 inputs.  Every fp32 operand x
// is therefore carried as TWO f16 numbers
//       h = f16_rtz(x)           (v_cvt_pkrtz_f16_f32: never overflows to inf)
//       m = f16_rne(x - h)       (the subtraction is exact)
// so that x = h + m up to 2^-22 |x| (11 + 11 significand bits; below |x| ~ 6e-5 the planes are f16 subnormals and the
// ABSOLUTE error is <= 2^-25), and a product is accumulated in fp32 by three v_mfma_f32_32x32x16_f16:
//       a*b  ~=  ah*bh + ah*bm + am*bh          (the dropped am*bm is <= 2^-20 |a*b|)
// i.e. 3 x 32 cycles per 16 columns of K instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32: 5.3x on the matrix pipe at
// ~fp32 accuracy (tests/test_split_f16_cpu.py restates the arithmetic in numpy; the GPU parity bars are unchanged).
// The split is done ONCE by whoever produces a value (epilogues write both planes to LDS, the pack kernel splits the
// weights, the training forward / backward store the planes the weight-gradient GEMM streams), never by the consumer.
//
// Orientation: D[feature][row] = W[feature][k] * X^T[k][row] -- the weights are the MFMA "A" operand, the activation tile
// the "B" operand.  A lane of the 32x32 accumulator then owns ONE tile row and features {4 (l>>5) + 8 g + i}: four
// consecutive features = one 8-byte store per plane into the row-major activation tile the next layer reads.
//
// Layouts
//   activation tile in LDS   [plane 2][row 64][k] f16, row stride PN_XRS bytes (37 x 16: odd, so the ds_read_b128 of a
//                            fragment -- lane -> row (l & 31), 8 consecutive k at 8 (l >> 5) -- is conflict-free)
//   weight image (L2)        uint4 img[chunk][mblock][plane][lane] : lane l of feature block mb holds
//                            W[32 mb + (l & 31)][16 chunk + 8 (l >> 5) .. + 7]; wave w owns blocks 2w, 2w + 1
//   k-major planes in HBM    uint4 u[plane][row / 8][feature] = rows 8 g .. 8 g + 7 of one feature: what the weight-gradient
//                            GEMM (k = rows) loads as ready-made fragments, written by the producers' transposing copy-out
#pragma once
#include "mlp_common.h"

typedef _Float16 pn_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 pn_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 pn_h8 __attribute__((ext_vector_type(8)));

#define PN_XRS    592                      // bytes per tile row and plane (up to 288 columns + 16 bytes of padding)
#define PN_XPLANE (PN_TILE * PN_XRS)       // 37 888
#define PN_XBYTES (2 * PN_XPLANE)          // 75 776
#define PN_K1     288                      // layer-1 input columns: 284 + ones column (284) + 3 zeros
#define PN_K3     272                      // layer-3 input columns: 256 + 7 extras + ones column (263) + 8 zeros
#define PN_ONES1  284                      // column of X0 that holds 1.0 (its weight-gradient column is the bias gradient)
#define PN_ONES3  263
#define PN_NF1    288                      // features of the saved k-major X0 and [h2 | extras] planes
#define PN_MB_D1  7                        // feature blocks of d X0 that are needed (embedding + its encoding: 224 columns)

// ---- images (byte offsets inside the packed buffer)
#define PN_IMG(nch, mb) ((nch) * (mb) * 2048)
enum : int {
    PKH_BASE = 0,
    PKH_F1 = PKH_BASE, PKH_F2 = PKH_F1 + PN_IMG(18, 8), PKH_F3 = PKH_F2 + PN_IMG(16, 8), PKH_F4 = PKH_F3 + PN_IMG(17, 8),
    PKH_D4 = PKH_F4 + PN_IMG(16, 8), PKH_D3 = PKH_D4 + PN_IMG(16, 8), PKH_D2 = PKH_D3 + PN_IMG(16, 9), PKH_D1 = PKH_D2 + PN_IMG(16, 8),
    // colour MLP: forward 128 x (288 | 128 | 128), dgrad (128 x 128) x 2 and 256 x 128
    PKH_FC1 = PKH_D1 + PN_IMG(16, PN_MB_D1), PKH_FC2 = PKH_FC1 + PN_IMG(18, 4), PKH_FC3 = PKH_FC2 + PN_IMG(8, 4),
    PKH_DC3 = PKH_FC3 + PN_IMG(8, 4), PKH_DC2 = PKH_DC3 + PN_IMG(8, 4), PKH_DC1 = PKH_DC2 + PN_IMG(8, 4),
    PKH_END = PKH_DC1 + PN_IMG(8, 8)
};

// Streaming stores of the saved planes (whole 1 KB runs per wave-instruction, written once, read by another kernel much later):
// non-temporal stores.  Round 3 compared the cache-policy bits on one box, whole step: plain 42.1 ms, nt 41.3 (shipped), sc1 (write-through,
// line dropped from L2) 41.8, nt sc1 41.0 -- inside the box-to-box noise, so the builtin stays (the sc bits have no builtin; the dev variants
// below go through inline asm, which needs its own s_nop against the store-data hazard: without it the data registers are overwritten
// while the 16-byte store still reads them -- the NaN-poisoned parity tests caught exactly that).
// dev A/B (tools/_build only): -DPN_STREAM_STORE_ASM='"nt sc1"' etc., -DPN_PLAIN_STREAM_STORES, -DPN_NO_STREAM_STORES (drops the stores:
// results are garbage, the timing tells what they cost: forward -21 %, backward -10 %).
#if defined(PN_NO_STREAM_STORES)
#define PN_STREAM_STORE(val, ptr) ((void)(val), (void)(ptr))
#elif defined(PN_PLAIN_STREAM_STORES) || defined(PN_EMU)
#define PN_STREAM_STORE(val, ptr) (*(ptr) = (val))
#elif defined(PN_STREAM_STORE_ASM)
__device__ __forceinline__ void pn_stream_store_asm(pn_f4 v, pn_f4 *p) {
    // (s_nop 1: a 16-byte store reads its data registers over several cycles; the compiler pads its own stores against a following write of
    //  those registers, but it does not know that this asm is one)
    asm volatile("global_store_dwordx4 %0, %1, off " PN_STREAM_STORE_ASM "\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
#define PN_STREAM_STORE(val, ptr) pn_stream_store_asm((val), (ptr))
#else
#define PN_STREAM_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
#endif

// stores of fp32 rows straight from the accumulator layout (a lane pair writes 32 bytes, the wave's four stores of a row fill one
// 128-byte line): plain stores, so that L2 merges them into whole lines (dev A/B: -DPN_REG_STORE_NT)
#ifdef PN_REG_STORE_NT
#define PN_REG_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
#else
#define PN_REG_STORE(val, ptr) (*(ptr) = (val))
#endif

// Workgroup barrier for LDS hazards only.  __syncthreads() carries a full fence: hipcc emits s_waitcnt vmcnt(0) in front of the
// s_barrier, i.e. every barrier would also wait for the fire-and-forget stores of the copy-outs (HBM round trips) and for the next
// tile's prefetched gathers.  The tile kernels only order LDS traffic with their barriers.
#ifdef PN_EMU
#define PN_LDS_BARRIER() __syncthreads()
#else
#define PN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// sin / cos of the positional encodings (networks.py:175-190: sin(x 2^f), cos(x 2^f), f = 0 .. NF-1).  Every octave is reduced on its
// own to revolutions r in [-0.5, 0.5] with a two-term 1 / 2pi -- x 2^f is exact, k = rint(x 2^f / 2pi), r = fma(x 2^f, HI, -k) + x 2^f LO:
// the angle error is <= 2 x 2 pi 2^-25 = 3.7e-7 rad for EVERY octave and every |x| < 2^20 (learned embeddings reach several units) -- and
// then evaluated by v_sin_f32 / v_cos_f32 (input in revolutions).  Round 2 took one __sinf / __cosf of x (a single-term x / 2pi, error
// growing with |x|) and doubled the angle by recurrence (error doubling per octave); same instruction count within 30 %.
// -DPN_EXACT_SINCOS: libm's sincosf per octave (dev A/B only).
template <int NF>
__device__ __forceinline__ void pn_pe_octaves(float x, float (&s)[NF], float (&c)[NF]) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const float xf = x * (float)(1 << f);
#if defined(PN_EXACT_SINCOS) || defined(PN_EMU)
        sincosf(xf, &s[f], &c[f]);
#else
        const float k = __builtin_rintf(xf * 0.15915494f);
        const float r = __builtin_fmaf(xf, 6.4206382e-9f, __builtin_fmaf(xf, 0.15915494f, -k));     // 1 / 2pi = 0.15915494f (= 0.159154936671257) + 6.4206382e-9
        s[f] = __builtin_amdgcn_sinf(r); c[f] = __builtin_amdgcn_cosf(r);
#endif
    }
}

// RANGE of the two-plane form: |x| <= 65504 (the largest finite f16).  v_cvt_pkrtz never produces inf, so beyond that h saturates at
// 65504 and the residual x - h overflows the f16 range for |x| > ~1.3e5 (inf in the residual plane = NaN out of the MFMA), where the fp32
// GEMM this replaces would still have been finite.  Activations of this network are O(1..100); weights are checked when they are packed
// (PointAggregator.check_range, when a checkpoint is re-homed into the flat parameter vector); gradients go through pn_split2_sat.
// (x0, x1) -> packed high plane (round toward zero) and packed residual plane (round to nearest): three instructions,
// v_cvt_pkrtz_f16_f32 and one v_fma_mix{lo,hi}_f16 per element (m = f16(x * 1.0 - h), the fused form of "convert back, subtract,
// convert"; hipcc does not form it by itself: 5 instructions).  tests: pnerf_debug_split against the numpy restatement, bit for bit
__device__ __forceinline__ void pn_split2(float x0, float x1, unsigned &h, unsigned &m) {
    const auto hh = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    h = __builtin_bit_cast(unsigned, hh);
#ifdef PN_EMU
    pn_h2 mm;
    mm[0] = (_Float16)(x0 - (float)hh[0]);
    mm[1] = (_Float16)(x1 - (float)hh[1]);
    m = __builtin_bit_cast(unsigned, mm);
#else
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(m) : "v"(x0), "v"(x1), "v"(h));
#endif
}
// Gradients: the value is clamped to the f16 range first (their scale is chosen per call, an outlier must saturate, not poison) and
// the high plane is rounded to NEAREST (v_cvt_pk_f16_f32, new on gfx950), the residual as before.  h + m is the same 22-bit number
// for the dgrad GEMMs; h ALONE is then the best single f16 for the value (|x - h| <= 2^-11 |x|, unbiased), which is what the
// weight-gradient GEMM streams for its dY operand (one plane instead of two: see k_wgrad_f16 for the error budget).
__device__ __forceinline__ void pn_split2_sat(float x0, float x1, unsigned &h, unsigned &m) {
    x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
    x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
    pn_h2 hh;
    hh[0] = (_Float16)x0; hh[1] = (_Float16)x1;
    h = __builtin_bit_cast(unsigned, hh);
#ifdef PN_EMU
    pn_h2 mm;
    mm[0] = (_Float16)(x0 - (float)hh[0]);
    mm[1] = (_Float16)(x1 - (float)hh[1]);
    m = __builtin_bit_cast(unsigned, mm);
#else
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(m) : "v"(x0), "v"(x1), "v"(h));
#endif
}
__device__ __forceinline__ float pn_h_lo(unsigned p) { return (float)__builtin_bit_cast(pn_h2, p)[0]; }
__device__ __forceinline__ float pn_h_hi(unsigned p) { return (float)__builtin_bit_cast(pn_h2, p)[1]; }

// one value at tile position (row, col): both planes
__device__ __forceinline__ void pn_x_store1(char *X, int row, int col, float v) {
    unsigned h, m;
    pn_split2(v, 0.f, h, m);
    *reinterpret_cast<unsigned short *>(X + row * PN_XRS + col * 2) = (unsigned short)h;
    *reinterpret_cast<unsigned short *>(X + PN_XPLANE + row * PN_XRS + col * 2) = (unsigned short)m;
}
// two values at (row, col), (row, col + 1), col even
__device__ __forceinline__ void pn_x_store2(char *X, int row, int col, float v0, float v1) {
    unsigned h, m;
    pn_split2(v0, v1, h, m);
    *reinterpret_cast<unsigned *>(X + row * PN_XRS + col * 2) = h;
    *reinterpret_cast<unsigned *>(X + PN_XPLANE + row * PN_XRS + col * 2) = m;
}
// four values at (row, col .. col + 3), col % 4 == 0.  XRS / XPL: row stride and plane distance of the tile (the colour backward keeps a
// narrower one)
template <bool SAT, int XRS = PN_XRS, int XPL = PN_XPLANE>
__device__ __forceinline__ void pn_x_store4(char *X, int row, int col, float v0, float v1, float v2, float v3) {
    unsigned h0, m0, h1, m1;
    if (SAT) { pn_split2_sat(v0, v1, h0, m0); pn_split2_sat(v2, v3, h1, m1); }
    else { pn_split2(v0, v1, h0, m0); pn_split2(v2, v3, h1, m1); }
    *reinterpret_cast<uint2 *>(X + row * XRS + col * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(X + XPL + row * XRS + col * 2) = make_uint2(m0, m1);
}
// four values back: h + m
__device__ __forceinline__ float4 pn_x_load4(const char *X, int row, int col) {
    const uint2 h = *reinterpret_cast<const uint2 *>(X + row * PN_XRS + col * 2);
    const uint2 m = *reinterpret_cast<const uint2 *>(X + PN_XPLANE + row * PN_XRS + col * 2);
    return make_float4(pn_h_lo(h.x) + pn_h_lo(m.x), pn_h_hi(h.x) + pn_h_hi(m.x), pn_h_lo(h.y) + pn_h_lo(m.y), pn_h_hi(h.y) + pn_h_hi(m.y));
}

// acc + (h + m) * w for the two packed values of a plane pair: fmaf of a converted half is ONE v_fma_mix_f32 (no conversion, no add).
// Written as inline asm since round 4: where a half feeds two sums (the tail's alpha dot product and K-weighted sum, the backward front's
// two passes) hipcc converts it once (v_cvt_f32_f16) and keeps the fp32 copy -- 128 conversions and 128 more live registers per thread in
// f_tail, which an 8-wave workgroup (128 registers per wave) spills.  Same arithmetic: fma(f32(half), w, acc), the conversion is exact.
#ifdef PN_EMU
__device__ __forceinline__ float pn_mix_lo(unsigned p, float w, float acc) { return __builtin_fmaf((float)__builtin_bit_cast(pn_h2, p)[0], w, acc); }
__device__ __forceinline__ float pn_mix_hi(unsigned p, float w, float acc) { return __builtin_fmaf((float)__builtin_bit_cast(pn_h2, p)[1], w, acc); }
#else
__device__ __forceinline__ float pn_mix_lo(unsigned p, float w, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(p), "v"(w));
    return acc;
}
__device__ __forceinline__ float pn_mix_hi(unsigned p, float w, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(p), "v"(w));
    return acc;
}
#endif
__device__ __forceinline__ float pn_fma2_lo(unsigned h, unsigned m, float w, float acc) { return pn_mix_lo(m, w, pn_mix_lo(h, w, acc)); }
__device__ __forceinline__ float pn_fma2_hi(unsigned h, unsigned m, float w, float acc) { return pn_mix_hi(m, w, pn_mix_hi(h, w, acc)); }
// dot product of 8 tile columns (row, col .. col + 7, col % 8 == 0) with 8 floats
__device__ __forceinline__ float pn_x_dot8(const char *X, int row, int col, const float4 &w0, const float4 &w1, float acc) {
    const uint4 h = *reinterpret_cast<const uint4 *>(X + row * PN_XRS + col * 2);
    const uint4 m = *reinterpret_cast<const uint4 *>(X + PN_XPLANE + row * PN_XRS + col * 2);
    acc = pn_fma2_lo(h.x, m.x, w0.x, acc); acc = pn_fma2_hi(h.x, m.x, w0.y, acc);
    acc = pn_fma2_lo(h.y, m.y, w0.z, acc); acc = pn_fma2_hi(h.y, m.y, w0.w, acc);
    acc = pn_fma2_lo(h.z, m.z, w1.x, acc); acc = pn_fma2_hi(h.z, m.z, w1.y, acc);
    acc = pn_fma2_lo(h.w, m.w, w1.z, acc); acc = pn_fma2_hi(h.w, m.w, w1.w, acc);
    return acc;
}
// f += w * (4 tile columns at (row, col))
__device__ __forceinline__ void pn_x_axpy4(const char *X, int row, int col, float w, float4 &f) {
    const uint2 h = *reinterpret_cast<const uint2 *>(X + row * PN_XRS + col * 2);
    const uint2 m = *reinterpret_cast<const uint2 *>(X + PN_XPLANE + row * PN_XRS + col * 2);
    f.x = pn_fma2_lo(h.x, m.x, w, f.x); f.y = pn_fma2_hi(h.x, m.x, w, f.y);
    f.z = pn_fma2_lo(h.y, m.y, w, f.z); f.w = pn_fma2_hi(h.y, m.y, w, f.w);
}

// ---- the tile GEMM: acc[fb][rb] (feature block fb of this wave x row block rb) += W[.., 16 NC columns from chunk c0] * X^T
// Three products per (fb, rb) and chunk, ordered so that an accumulator is touched once in four MFMAs.  The weight fragments
// come from the L2-resident image (500+ cycles under load) and are requested PN_WPF chunks (x 384 cycles of MFMA) ahead in a
// ring of PN_WPF + 1 register sets; the activation fragments come from LDS one chunk ahead.  The chunk loop is unrolled
// completely (compile-time register sets).  MB = feature blocks of the image; NFB = blocks this wave computes from fb0 on.
#ifndef PN_WPF
#define PN_WPF 2
#endif
// dev experiments (tools/_build variants only): issue-slot spacing after every MFMA / wave priority during the GEMM
#if defined(PN_MFMA_NOPS) && !defined(PN_EMU)
#define PN_MFMA_GAP() do { asm volatile("s_nop %0\n\ts_nop %0" ::"n"(PN_MFMA_NOPS)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PN_MFMA_GAP() ((void)0)
#endif
#if defined(PN_GEMM_PRIO) && !defined(PN_EMU)
#define PN_GEMM_PRIO_BEGIN() __builtin_amdgcn_s_setprio(PN_GEMM_PRIO)
#define PN_GEMM_PRIO_END() __builtin_amdgcn_s_setprio(0)
#else
#define PN_GEMM_PRIO_BEGIN() ((void)0)
#define PN_GEMM_PRIO_END() ((void)0)
#endif
// (Measured, round 2: with every chunk reading chunk 0's fragments -- the weight stream then hits L1 -- the forward is 11 % and the
//  backward 15 % faster: that is what the L2 -> L1 weight traffic (1.07 MB per 64-row tile and workgroup) costs, and the bound on what
//  a larger register blocking could recover.  A prefetch distance of 3 chunks instead of 2, starting the CU's second workgroup half
//  a tile late, streaming loads of the saved activations: no change.)
// (Measured and rejected: running the tile's transposing copy-out as a side job between the chunks' MFMA groups -- its stores share the
//  in-order vmcnt queue with the weight fragments, every chunk then waits for HBM writes: +3 us per GEMM against 2.4 us saved.)
// WPF = chunks of weight fragments requested ahead: 2 covers an L2 round trip when a chunk is 24 MFMAs (two feature blocks); the
// colour MLP's waves own ONE feature block (6 MFMAs = 0.1 us per chunk) and ask for 7.
// NP = products per multiply-add: 3 (h*h + h*m + m*h, fp32-class accuracy: training and the gradient chain) or 2 (the WEIGHTS' residual
// plane dropped: an inference OPTION (pnerf_set_inference_products) -- rendered ray colour within 2e-5 of fp32, per-sample sigma / RGB only
// within 4e-4 (outside the 1e-4 bar: not the default); rejected for training because a systematic
// perturbation of the weights flips LeakyReLU sides between forward and backward; a third of the MFMAs and half of the weight stream less).
// The ring of weight-fragment register sets of one tile GEMM.  prefetch() requests the first PF chunks; a caller may issue it BEFORE the
// barrier in front of the GEMM (round 4): the fragments come from L2 and depend on nothing in LDS, so their round trip (0.6 .. 1 us under
// load, paid at the start of every GEMM phase in rounds 2-3) passes under the barrier wait and the tail of the previous phase.
template <int NC, int MB, int NFB, int WPF = PN_WPF, int NP = 3>
struct PnGemmW {
    static constexpr int PF = WPF < NC ? WPF : NC - 1, NS = PF + 1;
    uint4 wh[NS][NFB], wm[NS][NFB];
    // Round 6: buffer addressing -- the image's descriptor in scalar registers + the lane's 32-bit offset (one vector register for every load of the
    // GEMM) + a wave-uniform scalar offset per load.  With flat pointers the chunk offsets (2 KB per chunk and feature block: beyond the 4 KB
    // immediate range after two chunks) became 64-bit vector address pairs formed by v_add_co / v_addc in the loop.  fb0 / c0 must be wave-uniform.
#ifdef PN_EMU
    const char *base;
#else
    __amdgpu_buffer_rsrc_t rs;
#endif
    int wo, lo;
    __device__ __forceinline__ uint4 ld16(int soff) const {
#ifdef PN_EMU
        return *reinterpret_cast<const uint4 *>(base + soff + lo);
#else
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, soff, 0));
#endif
    }
    template <int C> __device__ __forceinline__ void load() {
        constexpr int s = C % NS;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) { wh[s][fb] = ld16(wo + (C * MB + fb) * 2048); if (NP == 3) wm[s][fb] = ld16(wo + (C * MB + fb) * 2048 + 1024); }
    }
    __device__ __forceinline__ void prefetch(const uint4 *__restrict__ img, int fb0, int lane, int c0 = 0) {
#ifdef PN_EMU
        base = reinterpret_cast<const char *>(img);
#else
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(img), 0, 0x7fffffff, 0x00020000);
#endif
        wo = (c0 * MB + fb0) * 2048;
        lo = lane * 16;
        pn_static_for<PF>([&](auto cc) { load<decltype(cc)::value>(); });
    }
};

// the GEMM proper, on a ring whose first PF chunks have been requested (W.prefetch with the same img / fb0 / lane / c0)
template <int NC, int MB, int NFB, int WPF = PN_WPF, int NP = 3, int XRS = PN_XRS, int XPL = PN_XPLANE, int AF>
__device__ __forceinline__ void pn_gemm_f16x3_run(const char *X, PnGemmW<NC, MB, NFB, WPF, NP> &W, int lane, f32x16 (&acc)[AF][2], int c0 = 0) {
    static_assert(NP == 2 || NP == 3, "two or three products");
    static_assert(NFB <= AF, "accumulator blocks");
    constexpr int PF = PnGemmW<NC, MB, NFB, WPF, NP>::PF, NS = PF + 1;
    const char *xb = X + (lane & 31) * XRS + (lane >> 5) * 16 + c0 * 32;
    uint4 xh[2][2], xm[2][2];
    auto load_x = [&](auto cc) {
        constexpr int c = decltype(cc)::value, s = c & 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            xh[s][rb] = *reinterpret_cast<const uint4 *>(xb + rb * 32 * XRS + c * 32);
            xm[s][rb] = *reinterpret_cast<const uint4 *>(xb + XPL + rb * 32 * XRS + c * 32);
        }
    };
    load_x(std::integral_constant<int, 0>{});
    PN_GEMM_PRIO_BEGIN();
    pn_static_for<NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value, sw = c % NS, sx = c & 1;
        if constexpr (c + PF < NC) W.template load<c + PF>();
        if constexpr (c + 1 < NC) load_x(std::integral_constant<int, c + 1>{});
        // (without the fences hipcc sinks every load down to its first use to shorten live ranges: "load; s_waitcnt; mfma" --
        //  an exposed L2 round trip per chunk, measured 30 % of the MFMA rate)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const pn_h8 a = __builtin_bit_cast(pn_h8, p == 2 ? W.wm[sw][fb] : W.wh[sw][fb]);
                    const pn_h8 b = __builtin_bit_cast(pn_h8, p == 1 ? xm[sx][rb] : xh[sx][rb]);
                    acc[fb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[fb][rb], 0, 0, 0);
                    PN_MFMA_GAP();
                }
        __builtin_amdgcn_sched_barrier(0);
    });
    PN_GEMM_PRIO_END();
}

template <int NC, int MB, int NFB, int WPF = PN_WPF, int NP = 3, int XRS = PN_XRS, int XPL = PN_XPLANE, int AF>
__device__ __forceinline__ void pn_gemm_f16x3(const char *X, const uint4 *__restrict__ img, int fb0, int lane, f32x16 (&acc)[AF][2], int c0 = 0) {
    PnGemmW<NC, MB, NFB, WPF, NP> W;
    W.prefetch(img, fb0, lane, c0);
    pn_gemm_f16x3_run<NC, MB, NFB, WPF, NP, XRS, XPL>(X, W, lane, acc, c0);
}

// accumulator element (fb, rb, g, i) of a lane: feature = 32 fbg + 8 g + 4 (l >> 5) + i (fbg = global block), row = 32 rb + (l & 31)
__device__ __forceinline__ int pn_d_feat(int fbg, int g, int lane) { return 32 * fbg + 8 * g + 4 * (lane >> 5); }

// ---- transposing copy-out: the tile's NF columns -> k-major planes (training).  Unit (plane, row group rg, feature f) = rows
// 8 rg .. 8 rg + 7 of column f.  gfx950's LDS transpose read does the 16-bit transposition: per 16-lane group,
//     ds_read_b64_tr_b16:  lane i, element j  <-  element (i & 3) of the 8-byte slot addressed by lane 4 j + (i >> 2)
// (semantics pinned on the hardware by tools/trb16_probe.hip), so with lane i pointing at row (i >> 2) & 3, columns 4 (i & 3) .. + 3
// of a [4 rows][16 columns] block every lane receives 4 consecutive ROWS of column i: two reads = one 16-byte unit, stored
// coalesced (lane -> column).  dst = base of the array ([2][rg_total][NF] units), rg0 = first row group of the tile.
typedef short pn_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 pn_lds_read_tr16(const char *p) {
    const pn_s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pn_s4 *)p);
    return __builtin_bit_cast(uint2, r);
}
// the high plane only ([rg_total][NF] units): the dY operand of the weight-gradient GEMM
// NW = waves of the calling workgroup: wave w copies the row groups 8 / NW * w .. (8 of them in a 64-row tile)
template <int NF, int XRS = PN_XRS, int NW = 4>
__device__ __forceinline__ void pn_copy_out_kmajor_h(const char *X, uint4 *__restrict__ dst, long long rg0, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = ((lane >> 2) & 3) * XRS + ((lane >> 4) * 16 + (lane & 3) * 4) * 2;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int rg = wave * (8 / NW) + i;
        const char *src = X + rg * 8 * XRS + blk;
        uint4 *d = dst + (rg0 + rg) * NF;
        // (all transposing reads of the run first, then its stores: the stores are inline asm, which the scheduler does not move loads across)
        uint2 lo[(NF + 63) / 64], hi[(NF + 63) / 64];
#pragma unroll
        for (int j = 0; j < (NF + 63) / 64; ++j) { lo[j] = pn_lds_read_tr16(src + j * 128); hi[j] = pn_lds_read_tr16(src + 4 * XRS + j * 128); }
#pragma unroll
        for (int j = 0; j < (NF + 63) / 64; ++j) {
            const int f = lane + 64 * j;
            if (f < NF) {
                pn_f4 t = {__uint_as_float(lo[j].x), __uint_as_float(lo[j].y), __uint_as_float(hi[j].x), __uint_as_float(hi[j].y)};
                PN_STREAM_STORE(t, reinterpret_cast<pn_f4 *>(d + f));
            }
        }
    }
}
// the tile's value planes -> ONE k-major plane ([rg_total][NF] units): what the weight-gradient GEMM streams for its X operand.  The two
// LDS planes are read transposed and summed by v_pk_add_f16: h + m is exact in 22 bits and the packed add rounds it ONCE to the nearest
// f16 -- the best single f16 for the value (|x - f16| <= 2^-11 |x|, unbiased; h alone is rounded toward zero).  See k_wgrad_f16 for the
// error budget of the one-plane operands.
typedef _Float16 pn_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 pn_rne_sum(uint2 h, uint2 m) { return __builtin_bit_cast(uint2, __builtin_bit_cast(pn_h4, h) + __builtin_bit_cast(pn_h4, m)); }
// what the nearest f16 s = rne(h + m) leaves of h + m: r = m - (s - h).  Both subtractions are exact in f16 (|h| >= |m|: Fast2Sum), so
// s + r == h + m, the tile's 22-bit value: the second plane of the two-plane weight-gradient mode (pnerf_set_wgrad_planes(2))
__device__ __forceinline__ uint2 pn_rne_rest(uint2 h, uint2 m, uint2 s) {
    const pn_h4 hh = __builtin_bit_cast(pn_h4, h), mm = __builtin_bit_cast(pn_h4, m), ss = __builtin_bit_cast(pn_h4, s);
    return __builtin_bit_cast(uint2, mm - (ss - hh));
}
// TWO: also write the residual plane to dstm (same layout) -- the fp32-class weight-gradient mode
template <int NF, bool TWO = false, int NW = 4>
__device__ __forceinline__ void pn_copy_out_kmajor(const char *X, uint4 *__restrict__ dst, long long rg0, int tid, uint4 *__restrict__ dstm = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = ((lane >> 2) & 3) * PN_XRS + ((lane >> 4) * 16 + (lane & 3) * 4) * 2;      // this lane's slot inside a [4][64-column] block
    constexpr int NJ = (NF + 63) / 64;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int rg = wave * (8 / NW) + i;
        const char *src = X + rg * 8 * PN_XRS + blk;
        uint4 *d = dst + (rg0 + rg) * NF;
        uint2 lo[NJ], hi[NJ], lom[NJ], him[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            lo[j] = pn_lds_read_tr16(src + j * 128); hi[j] = pn_lds_read_tr16(src + 4 * PN_XRS + j * 128);
            lom[j] = pn_lds_read_tr16(src + PN_XPLANE + j * 128); him[j] = pn_lds_read_tr16(src + PN_XPLANE + 4 * PN_XRS + j * 128);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int f = lane + 64 * j;
            if (f < NF) {
                const uint2 a = pn_rne_sum(lo[j], lom[j]), b = pn_rne_sum(hi[j], him[j]);
                pn_f4 t = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y)};
                PN_STREAM_STORE(t, reinterpret_cast<pn_f4 *>(d + f));
                if (TWO) {
                    const uint2 ra = pn_rne_rest(lo[j], lom[j], a), rb = pn_rne_rest(hi[j], him[j], b);
                    pn_f4 u = {__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(rb.x), __uint_as_float(rb.y)};
                    PN_STREAM_STORE(u, reinterpret_cast<pn_f4 *>(dstm + (rg0 + rg) * NF + f));
                }
            }
        }
    }
}

// the tile's columns C0 .. C0 + 63 -> one k-major plane of 64 features ([rg_total][64] units): the saved part of X0 on the fused path
// (the weight-gradient kernel rebuilds the rest from the embedding: backward.hip k_wgrad_x0)
template <int C0, int NW = 4>
__device__ __forceinline__ void pn_copy_out_kmajor_cols64(const char *X, uint4 *__restrict__ dst, long long rg0, int tid) {
    static_assert(C0 % 16 == 0, "a 16-column group boundary");
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = ((lane >> 2) & 3) * PN_XRS + ((lane >> 4) * 16 + (lane & 3) * 4) * 2 + C0 * 2;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int rg = wave * (8 / NW) + i;
        const char *src = X + rg * 8 * PN_XRS + blk;
        const uint2 lo = pn_lds_read_tr16(src), hi = pn_lds_read_tr16(src + 4 * PN_XRS);
        const uint2 lom = pn_lds_read_tr16(src + PN_XPLANE), him = pn_lds_read_tr16(src + PN_XPLANE + 4 * PN_XRS);
        const uint2 a = pn_rne_sum(lo, lom), b = pn_rne_sum(hi, him);
        pn_f4 t = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y)};
        PN_STREAM_STORE(t, reinterpret_cast<pn_f4 *>(dst + (rg0 + rg) * 64 + lane));
    }
}
