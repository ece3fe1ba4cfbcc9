// mixq.h -- the mixed-format tile GEMM of the aggregator MLP (round 6): fp32-class products at 2 instead of 3 matrix-pipe passes.
//
// f16x3.h carries every fp32 operand as two f16 planes x = h + m and forms a product as h*h + h*m + m*h on v_mfma_f32_32x32x16_f16.
// The two cross terms are 2^-11 of the result: they do not need f16 factors.  Here they run on gfx950's block-scaled 8-bit instruction,
//       a*b  ~=  ah*bh  (f16, four K = 16 MFMAs per 64 columns)  +  [q8(ah) * q8(bm 2^11) + q8(am 2^11) * q8(bh)] 2^-11  (e4m3, two K = 64 MFMAs)
// -- 4 x 32 + 2 x 64 = 256 pipe cycles per 64 columns and 32 x 32 block instead of 12 x 32 = 384 (measured with toggling register operands, two
// waves per SIMD: 1170 against 1820 ns per 64-column step, tools/mx_probe.hip / profiles/r06_mx_probe.jsonl).  Each e4m3 factor carries a relative
// rounding of <= 2^-4, so a cross term is good to ~2^-4 of 2^-11 of the product: error 1.1e-6 rms of sum |terms| over a 256-term dot product
// measured on the device by a probe (profiles/r05_fp8_gemm_probe.jsonl) and 1.3e-6 rms / 5.6e-6 max through this format (tests/test_gpu_mix.py); three
// f16 products: 3e-8; the parity bar downstream is 1e-4.
//
// Instruction semantics pinned on the hardware (tools/mx_probe.hip): v_cvt_scalef32_pk_fp8_f16 converts value / scale, round to nearest even,
// subnormals kept (step 2^-9), and overflow gives NaN unless MODE.FP16_OVFL is set, in which case it SATURATES at +-448 (and v_cvt_f16_f32 at
// +-65504) -- the tile kernels set that bit (pn_mode_saturate), so an outlier degrades a cross term instead of poisoning the tile;
// v_mfma_scale_f32_32x32x64_f8f6f4 multiplies byte j of a lane's A registers with byte j of the same lane's B registers (any assignment of the
// 64 k to (lane half, byte) works if both operands use it) and applies 2^(s - 127) from a byte of a VGPR chosen by op_sel -- per SCALE BLOCK, and
// the blocks are registers 0..3 of both lane halves (scaled by the byte lanes 0..31 hold) and registers 4..7 of both halves (lanes 32..63's
// byte), not "a lane's 32 slots": a row's two lanes must agree wherever one exponent has to cover slots of both (measured: tools/gpu_mix_diag.py).
//
// Layout of a tile row in LDS (the two planes of f16x3.h keep their places and strides):
//   plane 0  [row][k] f16     h = f16_rne(x), all columns                                (row stride PN_XRS)
//   plane 1  columns < 256:   per group of 8 columns G one 16-byte unit  [q8(h) x 8 | q8(m 2^11) x 8]   at 16 G
//            columns >= 256:  the f16 residual plane of f16x3.h at 2 k (bytes 512..591): those 16 / 32 columns (layer-3 extras, the distance
//                             encoding's tail and the ones column) run the classic three f16 products
//   so that a producer thread rewrites exactly the bytes it owns in BOTH planes (the backward's front works in place), and the B fragment of
//   the e4m3 MFMA j of superchunk s (64 columns) is 32 contiguous bytes of the row at 128 s + 64 j + 32 (lane >> 5).
// Weight image of a layer (MB feature blocks, NT classic tail chunks), uint4 units, lane-contiguous:
//   superchunk s, block mb:  [((s MB + mb) 8 + r) 64 + lane]   r = 0..3: f16 h fragment of chunk 4 s + r (f16x3.h's A fragment, high plane)
//                                                               r = 4, 5: e4m3 fragment of MFMA 0;  r = 6, 7: of MFMA 1 -- per 8-column group
//                                                               [q8(wm 2^11 / 2^e) x 8 | q8(wh / 2^e) x 8]: slot-wise the partner of the row's unit
//   tail chunk t:            [NS MB 8 64 + ((t MB + mb) 2 + plane) 64 + lane]          (f16x3.h's two-plane chunk)
//   block scales:            uint32 [(s MB + mb) 64 + lane] behind the units: byte j = e + 127 of the ROW's 64 slots of MFMA j (both lanes of a row hold
//                            the same byte; e chosen so that the largest slot lies in (224, 448]: the weights are packed once per step, their
//                            block scale costs nothing in the loop)
// The activations use fixed scales (1 for h, 2^-11 for m): a lane's 32 slots are 16 columns x (h, m 2^11), magnitudes |x| and <= |x| / 2.
#pragma once
#include "f16x3.h"

typedef int pn_i8v __attribute__((ext_vector_type(8)));
typedef short pn_s2 __attribute__((ext_vector_type(2)));

#define PN_MIX_NS 4                          // superchunks of 64 columns: columns 0..255 of every aggregator layer
#define PN_MIX_MSC 0.00048828125f            // 2^-11: scale of the residual slots (value / scale = m 2^11)
#define PN_MIX_MSC_BYTE 116                  // 127 - 11
#ifndef PN_MIX_LEAD
#define PN_MIX_LEAD 2                        // units of weight fragments requested ahead
#endif
#define PN_MIMG_U4(nt, mb) ((PN_MIX_NS * 8 + (nt) * 2) * (mb) * 64)
#define PN_MIMG(nt, mb) (PN_MIMG_U4(nt, mb) * 16 + PN_MIX_NS * (mb) * 256)
enum : int {
    PKM_F1 = PKH_END, PKM_F2 = PKM_F1 + PN_MIMG(2, 8), PKM_F3 = PKM_F2 + PN_MIMG(0, 8), PKM_F4 = PKM_F3 + PN_MIMG(1, 8),
    PKM_D4 = PKM_F4 + PN_MIMG(0, 8), PKM_D3 = PKM_D4 + PN_MIMG(0, 8), PKM_D2 = PKM_D3 + PN_MIMG(0, 9), PKM_D1 = PKM_D2 + PN_MIMG(0, 8),
    PKM_END = PKM_D1 + PN_MIMG(0, PN_MB_D1)
};

// MODE.FP16_OVFL = 1: f16 and fp8 conversions saturate instead of producing inf / NaN (per wave; every wave of a tile kernel runs this first)
__device__ __forceinline__ void pn_mode_saturate() {
#ifndef PN_EMU
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
#endif
}

// (x0, x1) -> packed f16 nearest (saturating under pn_mode_saturate) and packed f16 residual
__device__ __forceinline__ void pn_split2_rne(float x0, float x1, unsigned &h, unsigned &m) {
#ifdef PN_EMU
    x0 = fmaxf(fminf(x0, 65504.f), -65504.f); x1 = fmaxf(fminf(x1, 65504.f), -65504.f);
#endif
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
// four f16 (two packed pairs) -> four e4m3 bytes of value / scale
__device__ __forceinline__ unsigned pn_q8x4(unsigned p01, unsigned p23, float scale) {
    pn_s2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(pn_h2, p01), scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(pn_h2, p23), scale, true);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned short pn_q8x2(unsigned p01, float scale) {
    pn_s2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(pn_h2, p01), scale, false);
    return (unsigned short)r[0];
}
// byte offset of column col's e4m3 slot inside the row's plane-1 area (col < 256); the residual's slot is 8 bytes further
__device__ __forceinline__ int pn_q_off(int col) { return (col >> 3) * 16 + (col & 7); }

// four values at tile position (row, col .. col + 3), col % 4 == 0, col < 256: h plane + the group's e4m3 unit
template <int XRS = PN_XRS, int XPL = PN_XPLANE>
__device__ __forceinline__ void pn_xq_store4(char *X, int row, int col, float v0, float v1, float v2, float v3) {
    unsigned h0, m0, h1, m1;
    pn_split2_rne(v0, v1, h0, m0); pn_split2_rne(v2, v3, h1, m1);
    *reinterpret_cast<uint2 *>(X + row * XRS + col * 2) = make_uint2(h0, h1);
    char *q = X + XPL + row * XRS + pn_q_off(col);
    *reinterpret_cast<unsigned *>(q) = pn_q8x4(h0, h1, 1.0f);
    *reinterpret_cast<unsigned *>(q + 8) = pn_q8x4(m0, m1, PN_MIX_MSC);
}
// two values at (row, col), (row, col + 1), col even, col < 256
__device__ __forceinline__ void pn_xq_store2(char *X, int row, int col, float v0, float v1) {
    unsigned h, m;
    pn_split2_rne(v0, v1, h, m);
    *reinterpret_cast<unsigned *>(X + row * PN_XRS + col * 2) = h;
    char *q = X + PN_XPLANE + row * PN_XRS + pn_q_off(col);
    *reinterpret_cast<unsigned short *>(q) = pn_q8x2(h, 1.0f);
    *reinterpret_cast<unsigned short *>(q + 8) = pn_q8x2(m, PN_MIX_MSC);
}
// any column: the mixed form below 256, f16x3.h's two planes (h to nearest) from 256 on
__device__ __forceinline__ void pn_xa_store2(char *X, int row, int col, float v0, float v1) {
    if (col < 256) { pn_xq_store2(X, row, col, v0, v1); return; }
    unsigned h, m;
    pn_split2_rne(v0, v1, h, m);
    *reinterpret_cast<unsigned *>(X + row * PN_XRS + col * 2) = h;
    *reinterpret_cast<unsigned *>(X + PN_XPLANE + row * PN_XRS + col * 2) = m;
}
// two values, f16x3.h's two planes with h rounded to nearest (any column)
__device__ __forceinline__ void pn_xt_store2(char *X, int row, int col, float v0, float v1) {
    unsigned h, m;
    pn_split2_rne(v0, v1, h, m);
    *reinterpret_cast<unsigned *>(X + row * PN_XRS + col * 2) = h;
    *reinterpret_cast<unsigned *>(X + PN_XPLANE + row * PN_XRS + col * 2) = m;
}
// four values (col % 4 == 0): f16x3.h's two planes with h rounded to nearest -- the mixed tile's columns >= 256, and every column of the training
// forward's f16x3 tiles (round 6: h IS then the nearest f16 of the value, so the k-major plane the weight-gradient GEMM streams is the transpose
// of the h plane alone: the copy-out reads one plane instead of two and adds nothing)
__device__ __forceinline__ void pn_xt_store4(char *X, int row, int col, float v0, float v1, float v2, float v3) {
    unsigned h0, m0, h1, m1;
    pn_split2_rne(v0, v1, h0, m0); pn_split2_rne(v2, v3, h1, m1);
    *reinterpret_cast<uint2 *>(X + row * PN_XRS + col * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(X + PN_XPLANE + row * PN_XRS + col * 2) = make_uint2(m0, m1);
}

// ---- the weight-fragment registers of one mixed tile GEMM.  Units of the static schedule, per superchunk: HA (f16 chunks 0, 1) HB (chunks 2, 3)
// Q0 Q1 (one e4m3 MFMA each per accumulator block); then the NT classic tail chunks (h plane, m plane).  Every unit is TWO 16-byte fragments per
// feature block on the weight side and two per row block on the tile side, and 8 or 4 MFMAs per (2 x 2 block) wave = 256 .. 320 pipe cycles; its
// weight fragments are requested LEAD units ahead into a ring of LEAD + 1 register sets (48 registers with two feature blocks: f16x3.h's budget),
// the tile's fragments one unit ahead into a ring of two.  The image is addressed as a wave-uniform base (scalar registers) + the lane's 32-bit
// offset, so that the unit offsets cost scalar adds, not 64-bit vector address pairs.
// load<U>() requests unit U, prefetch() the first LEAD units.
// NSR = superchunks this GEMM runs (4, or 1: the layer-3 extras block of the backward, one superchunk per wave).
template <int NSR, int NT, int MB, int NFB, int LEAD = PN_MIX_LEAD>
struct PnMixW {
    static constexpr int NU = 4 * NSR + NT, PF = LEAD < NU ? LEAD : NU, NS = LEAD + 1;
    uint4 r[NS][NFB][2];
    unsigned wsc[NFB];
    // buffer addressing: descriptor of the layer's image (scalar registers) + the lane's 32-bit offset (ONE vector register for every load of the
    // GEMM) + a wave-uniform scalar offset per load: no 64-bit vector address pairs and no vector adds for the unit offsets, which exceed the
    // 4 KB immediate range of a global load (the compiler otherwise forms every one of them with v_add_co / v_addc pairs)
#ifdef PN_EMU
    const char *base;
#else
    __amdgpu_buffer_rsrc_t rs;
#endif
    int wo, to, so;        // wave-uniform byte offsets inside the image: (superchunk s0, block fb0), the tail chunks at fb0, the block scales at (s0, fb0)
    int lo;                // lane * 16
    __device__ __forceinline__ uint4 ld16(int soff) const {
#ifdef PN_EMU
        return *reinterpret_cast<const uint4 *>(base + soff + lo);
#else
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, soff, 0));
#endif
    }
    __device__ __forceinline__ unsigned ld4(int soff) const {
#ifdef PN_EMU
        return *reinterpret_cast<const unsigned *>(base + soff + (lo >> 2));
#else
        return __builtin_amdgcn_raw_buffer_load_b32(rs, lo >> 2, soff, 0);
#endif
    }
    template <int U> __device__ __forceinline__ void load() {
        constexpr int sl = U % NS;
        if constexpr (U < 4 * NSR) {
            constexpr int s = U / 4, j = U % 4;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                r[sl][fb][0] = ld16(wo + ((s * MB + fb) * 8 + 2 * j) * 1024);
                r[sl][fb][1] = ld16(wo + ((s * MB + fb) * 8 + 2 * j + 1) * 1024);
                if constexpr (j == 2) wsc[fb] = ld4(so + (s * MB + fb) * 256);
            }
        } else {
            constexpr int t = U - 4 * NSR;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                r[sl][fb][0] = ld16(to + ((t * MB + fb) * 2) * 1024);
                r[sl][fb][1] = ld16(to + ((t * MB + fb) * 2 + 1) * 1024);
            }
        }
    }
    // img = the layer's image (uniform); fb0 = first feature block of this wave (wave-uniform); s0 = first superchunk (NSR < 4 only; wave-uniform)
    __device__ __forceinline__ void prefetch(const char *img, int fb0, int lane, int s0 = 0) {
#ifdef PN_EMU
        base = img;
#else
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(img), 0, PN_MIMG(NT, MB), 0x00020000);
#endif
        wo = (s0 * MB + fb0) * 8 * 1024;
        to = PN_MIX_NS * MB * 8 * 1024 + fb0 * 2 * 1024;
        so = PN_MIMG_U4(NT, MB) * 16 + (s0 * MB + fb0) * 256;
        lo = lane * 16;
        pn_static_for<PF>([&](auto uu) { load<decltype(uu)::value>(); });
    }
};

// the GEMM proper: acc[fb][rb] += W[.., columns 64 s0 ..] X^T over NSR superchunks and the NT tail chunks, on a ring whose first units have
// been requested (W.prefetch with the same s0)
template <int NSR, int NT, int MB, int NFB, int LEAD = PN_MIX_LEAD, int AF>
__device__ __forceinline__ void pn_gemm_mix_run(const char *X, PnMixW<NSR, NT, MB, NFB, LEAD> &W, int lane, f32x16 (&acc)[AF][2], int s0 = 0) {
    static_assert(NFB <= AF, "accumulator blocks");
    constexpr int NU = 4 * NSR + NT, NS = LEAD + 1;
    const char *xb = X + (lane & 31) * PN_XRS + (lane >> 5) * 16 + s0 * 128;                  // h plane: chunk c at 32 c
    const char *qb = X + PN_XPLANE + (lane & 31) * PN_XRS + (lane >> 5) * 32 + s0 * 128;      // plane 1: MFMA j of superchunk s at 128 s + 64 j
    uint4 x[2][2][2];
    auto load_x = [&](auto uu) {
        constexpr int u = decltype(uu)::value, sl = u & 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if constexpr (u < 4 * NSR) {
                constexpr int s = u / 4, j = u % 4;
                if constexpr (j < 2) {
                    x[sl][rb][0] = *reinterpret_cast<const uint4 *>(xb + rb * 32 * PN_XRS + (4 * s + 2 * j) * 32);
                    x[sl][rb][1] = *reinterpret_cast<const uint4 *>(xb + rb * 32 * PN_XRS + (4 * s + 2 * j + 1) * 32);
                } else {
                    x[sl][rb][0] = *reinterpret_cast<const uint4 *>(qb + rb * 32 * PN_XRS + 128 * s + 64 * (j - 2));
                    x[sl][rb][1] = *reinterpret_cast<const uint4 *>(qb + rb * 32 * PN_XRS + 128 * s + 64 * (j - 2) + 16);
                }
            } else {
                constexpr int t = u - 4 * NSR;
                x[sl][rb][0] = *reinterpret_cast<const uint4 *>(xb + rb * 32 * PN_XRS + 512 + 32 * t);
                x[sl][rb][1] = *reinterpret_cast<const uint4 *>(xb + PN_XPLANE + rb * 32 * PN_XRS + 512 + 32 * t);
            }
        }
    };
    load_x(std::integral_constant<int, 0>{});
    PN_GEMM_PRIO_BEGIN();
    pn_static_for<NU>([&](auto uu) {
        constexpr int u = decltype(uu)::value, sw = u % NS, sx = u & 1;
        if constexpr (u + LEAD < NU) W.template load<u + LEAD>();
        if constexpr (u + 1 < NU) load_x(std::integral_constant<int, u + 1>{});
        __builtin_amdgcn_sched_barrier(0);      // (loads stay in front of the unit's MFMAs: see f16x3.h)
        if constexpr (u < 4 * NSR) {
            constexpr int j = u % 4;
            if constexpr (j < 2) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb)
                            acc[fb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pn_h8, W.r[sw][fb][c]), __builtin_bit_cast(pn_h8, x[sx][rb][c]), acc[fb][rb], 0, 0, 0);
            } else {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        struct { uint4 a, b; } wa = {W.r[sw][fb][0], W.r[sw][fb][1]}, xa = {x[sx][rb][0], x[sx][rb][1]};
                        acc[fb][rb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(pn_i8v, wa), __builtin_bit_cast(pn_i8v, xa), acc[fb][rb], 0, 0,
                                                                                      j - 2, (int)W.wsc[fb], 0, PN_MIX_MSC_BYTE);
                    }
            }
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        acc[fb][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pn_h8, W.r[sw][fb][p == 2 ? 1 : 0]), __builtin_bit_cast(pn_h8, x[sx][rb][p == 1 ? 1 : 0]),
                                                                             acc[fb][rb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    PN_GEMM_PRIO_END();
}

template <int NSR, int NT, int MB, int NFB, int LEAD = PN_MIX_LEAD, int AF>
__device__ __forceinline__ void pn_gemm_mix(const char *X, const char *img, int fb0, int lane, f32x16 (&acc)[AF][2], int s0 = 0) {
    PnMixW<NSR, NT, MB, NFB, LEAD> W;
    W.prefetch(img, fb0, lane, s0);
    pn_gemm_mix_run<NSR, NT, MB, NFB, LEAD>(X, W, lane, acc, s0);
}

// the tile's columns C0 .. C0 + 63 of the HIGH plane -> one k-major plane of 64 features (pn_copy_out_kmajor_cols64 without the residual: the mixed
// tile's h is already the nearest f16)
template <int C0, int NW = 4>
__device__ __forceinline__ void pn_copy_out_kmajor_cols64_h(const char *X, uint4 *__restrict__ dst, long long rg0, int tid) {
    static_assert(C0 % 16 == 0, "a 16-column group boundary");
    const int lane = tid & 63, wave = tid >> 6;
    const int blk = ((lane >> 2) & 3) * PN_XRS + ((lane >> 4) * 16 + (lane & 3) * 4) * 2 + C0 * 2;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
        const int rg = wave * (8 / NW) + i;
        const char *src = X + rg * 8 * PN_XRS + blk;
        const uint2 lo = pn_lds_read_tr16(src), hi = pn_lds_read_tr16(src + 4 * PN_XRS);
        pn_f4 t = {__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y)};
        PN_STREAM_STORE(t, reinterpret_cast<pn_f4 *>(dst + (rg0 + rg) * 64 + lane));
    }
}
