// fp8_gemm_probe.hip -- dev tool, "what comes next" (DESIGN 4.2 item 24): the block-scaled fp8 matrix instruction of gfx950 on the hardware.
//   1. v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands against a host product of the very bytes the device converted: the byte format (OCP e4m3fn,
//      not fnuz), the e8m0 scale's meaning (2^(s - 127)), and that byte j of a lane's A registers meets byte j of the same lane half's B registers --
//      i.e. ANY assignment of the 64 k to (lane half, byte) works as long as both operands use the same one (the two candidates below both match);
//   2. a 32 x 32 x 256 product with fp32-class operands three ways -- the shipped three f16 products (h.h + h.m + m.h), the candidate
//      (f16 h.h) + (fp8 h.m) + (fp8 m.h) with the residual planes scaled by 2^11 and the instruction's scale operand taking it back, and one f16 product --
//      each against the float64 product of the fp32 inputs: the error the candidate scheme has ON THE DEVICE'S OWN conversions and accumulation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 fp8_gemm_probe.hip -o fp8_gemm_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// k index of byte j (0..31) of lane-half `half` (0, 1) under the two candidate layouts
__host__ __device__ inline int k_of(int layout, int half, int j) {
    return layout == 0 ? 32 * half + j                       // contiguous: lanes 0..31 hold k 0..31, lanes 32..63 k 32..63
                       : 32 * (j >> 4) + 16 * half + (j & 15);   // two K = 32 halves: registers 0..3 hold k 16 half .. + 15, registers 4..7 k 32 + 16 half .. + 15
}
__device__ inline i8v pack_fp8(const float (&v)[32]) {
    i8v r;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        int x = 0;
        x = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * w], v[4 * w + 1], x, false);
        x = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * w + 2], v[4 * w + 3], x, true);
        r[w] = x;
    }
    return r;
}

// ---- 1: D = A (32 x 64) B (64 x 32), e4m3; the device writes back the bytes it formed so that the host multiplies exactly those
__global__ void k_layout(const float *A, const float *B, int layout, int scale_a, int scale_b, float *D, unsigned char *A8, unsigned char *B8) {
    const int l = threadIdx.x, row = l & 31, half = l >> 5;
    float va[32], vb[32];
    for (int j = 0; j < 32; ++j) { const int k = k_of(layout, half, j); va[j] = A[row * 64 + k]; vb[j] = B[k * 32 + row]; }
    const i8v a = pack_fp8(va), b = pack_fp8(vb);
    for (int j = 0; j < 32; ++j) {
        const int k = k_of(layout, half, j);
        A8[row * 64 + k] = (unsigned char)((unsigned)a[j >> 2] >> (8 * (j & 3)));
        B8[k * 32 + row] = (unsigned char)((unsigned)b[j >> 2] >> (8 * (j & 3)));
    }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (scale_a == 127 && scale_b == 127) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 127, 0, 116);          // B scaled by 2^(116 - 127) = 2^-11
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + row] = acc[r];
}

// ---- 2: 32 x 32 x 256 with fp32-class operands.  mode 0: three f16 products; 1: f16 h.h + fp8 h.m + fp8 m.h; 2: one f16 product
__device__ inline float h_of(float x) { return (float)(_Float16)x; }
__global__ void k_schemes(const float *A, const float *B, int layout, int mode, float *D) {
    const int l = threadIdx.x, row = l & 31, half = l >> 5;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c = 0; c < 256; c += 16) {                         // f16 K = 16 steps: lane holds k = c + 8 half .. + 7
        h8 ah, am, bh, bm;
        for (int j = 0; j < 8; ++j) {
            const int k = c + 8 * half + j;
            const float a = A[row * 256 + k], b = B[k * 32 + row];
            ah[j] = (_Float16)a; am[j] = (_Float16)(a - (float)ah[j]); bh[j] = (_Float16)b; bm[j] = (_Float16)(b - (float)bh[j]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if (mode == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc, 0, 0, 0);
        }
    }
    if (mode == 1) {
        for (int c = 0; c < 256; c += 64) {                     // fp8 K = 64 steps: the cross terms, residual planes x 2^11, taken back by the scale operand
            float ah[32], am[32], bh[32], bm[32];
            for (int j = 0; j < 32; ++j) {
                const int k = c + k_of(layout, half, j);
                const float a = A[row * 256 + k], b = B[k * 32 + row];
                ah[j] = h_of(a); am[j] = (a - ah[j]) * 2048.f; bh[j] = h_of(b); bm[j] = (b - bh[j]) * 2048.f;
            }
            const i8v a8 = pack_fp8(ah), am8 = pack_fp8(am), b8 = pack_fp8(bh), bm8 = pack_fp8(bm);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, bm8, acc, 0, 0, 0, 127, 0, 116);       // h . m
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am8, b8, acc, 0, 0, 0, 116, 0, 127);       // m . h
        }
    }
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + row] = acc[r];
}

static double dec_e4m3(unsigned char b, bool fnuz) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    if (fnuz) { if (b == 0x80) return NAN; const double v = e == 0 ? ldexp(m / 8.0, -7) : ldexp(1.0 + m / 8.0, e - 8); return s ? -v : v; }
    if (e == 15 && m == 7) return NAN;
    const double v = e == 0 ? ldexp(m / 8.0, -6) : ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}

int main() {
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
    // ---- 1
    std::vector<float> A(32 * 64), B(64 * 32), D(32 * 32);
    for (auto &v : A) v = (float)(rnd() * 4.0);
    for (auto &v : B) v = (float)(rnd() * 4.0);
    float *dA, *dB, *dD; unsigned char *dA8, *dB8;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&dA8, 32 * 64); hipMalloc(&dB8, 64 * 32);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    int good_layout = -1;
    for (int layout = 0; layout < 2; ++layout)
        for (int sc = 0; sc < 2; ++sc) {
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, layout, 127, sc ? 116 : 127, dD, dA8, dB8);
            std::vector<unsigned char> a8(32 * 64), b8(64 * 32);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(a8.data(), dA8, a8.size(), hipMemcpyDeviceToHost); hipMemcpy(b8.data(), dB8, b8.size(), hipMemcpyDeviceToHost);
            for (int fnuz = 0; fnuz < 2; ++fnuz) {
                double worst = 0, mag = 0;
                for (int i = 0; i < 32; ++i)
                    for (int n = 0; n < 32; ++n) {
                        double ref = 0;
                        for (int k = 0; k < 64; ++k) ref += dec_e4m3(a8[i * 64 + k], fnuz) * dec_e4m3(b8[k * 32 + n], fnuz);
                        if (sc) ref *= ldexp(1.0, -11);
                        worst = fmax(worst, fabs(ref - D[i * 32 + n])); mag = fmax(mag, fabs(ref));
                    }
                printf("{\"probe\": \"layout\", \"candidate\": %d, \"b_scale\": \"%s\", \"decode\": \"%s\", \"max_abs_err\": %.3g, \"max_abs_ref\": %.3g, \"matches\": %s}\n", layout,
                       sc ? "2^-11" : "1", fnuz ? "e4m3fnuz" : "e4m3fn (OCP)", worst, mag, worst <= 1e-4 * mag ? "true" : "false");
                if (worst <= 1e-4 * mag && good_layout < 0) good_layout = layout;
            }
        }
    if (good_layout < 0) { printf("{\"probe\": \"layout\", \"error\": \"no candidate layout reproduces the host product\"}\n"); return 1; }
    // ---- 2: activation-like A (half of the entries 100 x smaller: LeakyReLU), weight-like B; error relative to sum |a b| per output
    std::vector<float> A2(32 * 256), B2(256 * 32), D2(32 * 32);
    for (auto &v : A2) v = (float)(rnd() * 3.0 * (rand() & 1 ? 1.0 : 0.01));
    for (auto &v : B2) v = (float)(rnd() * 0.25);
    float *dA2, *dB2;
    hipMalloc(&dA2, A2.size() * 4); hipMalloc(&dB2, B2.size() * 4);
    hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB2, B2.data(), B2.size() * 4, hipMemcpyHostToDevice);
    const char *names[3] = {"three f16 products (shipped)", "f16 h.h + fp8 h.m + fp8 m.h (residual planes x 2^11, scale operand 2^-11)", "one f16 product"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_schemes, dim3(1), dim3(64), 0, 0, dA2, dB2, good_layout, mode, dD);
        hipMemcpy(D2.data(), dD, D2.size() * 4, hipMemcpyDeviceToHost);
        double se = 0, worst = 0;
        for (int i = 0; i < 32; ++i)
            for (int n = 0; n < 32; ++n) {
                double ref = 0, sab = 0;
                for (int k = 0; k < 256; ++k) { const double t = (double)A2[i * 256 + k] * B2[k * 32 + n]; ref += t; sab += fabs(t); }
                const double e = (D2[i * 32 + n] - ref) / sab;
                se += e * e; worst = fmax(worst, fabs(e));
            }
        printf("{\"probe\": \"32 x 32 x 256 product on the device\", \"scheme\": \"%s\", \"rms_err_over_sum_abs_terms\": %.3g, \"max\": %.3g}\n", names[mode], sqrt(se / 1024), worst);
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
