// mx_probe.hip -- dev tool (round 6): the gfx950 instructions behind the mixed-format tile GEMM (f16 h.h + low-precision cross terms), pinned on the hardware.
//   1. conversions: v_cvt_scalef32_pk_fp8_{f32,f16}, v_cvt_pk_fp8_f32, v_cvt_scalef32_2xpk16_fp6_f32, v_cvt_scalef32_pk32_fp6_f16 on a table of
//      values x scales, with MODE.FP16_OVFL clear and set: raw bytes out (decoded offline: scale direction, rounding, saturation, slot order)
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands formed by those conversions, and with per-lane block scales taken from a VGPR
//      (op_sel byte selection): inputs, operand registers and D dumped
//   3. rates with toggling register operands, two waves per SIMD on every CU: f16 only, fp8 only, fp6 only, and the mixed per-64-k groups
//      (4 f16 + 2 fp8) / (4 f16 + 2 fp6) against 12 f16
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mx_probe.hip -o mx_probe ; prints JSON lines
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i8v __attribute__((ext_vector_type(8)));
typedef unsigned u6v __attribute__((ext_vector_type(6)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef short s2 __attribute__((ext_vector_type(2)));

#define SET_OVFL(v) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), (v))

// ---- 1: one lane converts in[0..31] (pairs / vectors) with `scale`; out: [0..15] fp8 from f32 (scaled), [16..31] fp8 from f16 (scaled), [32..47] plain
// cvt_pk_fp8_f32, [48..53] fp6 from 2 x 16 f32, [54..59] fp6 from 32 f16, [60..75] f16 conversions of the inputs (cvt_pk_f16_f32 = RNE), [76..91] pkrtz
__global__ void k_cvt(const float *in, float scale, int ovfl, unsigned *out) {
    if (threadIdx.x != 0) return;
    if (ovfl) SET_OVFL(1);
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = in[i];
    for (int i = 0; i < 16; ++i) {
        s2 z = {0, 0};
        s2 a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, v[2 * i], v[2 * i + 1], scale, false);
        out[i] = (unsigned short)a[0];
        h2 hh; hh[0] = (_Float16)v[2 * i]; hh[1] = (_Float16)v[2 * i + 1];
        s2 b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, hh, scale, false);
        out[16 + i] = (unsigned short)b[0];
        out[32 + i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2 * i], v[2 * i + 1], 0, false) & 0xffffu;
        out[60 + i] = __builtin_bit_cast(unsigned, hh);
        out[76 + i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]));
    }
    f16v fa, fb; h32 hv;
    for (int i = 0; i < 16; ++i) { fa[i] = v[i]; fb[i] = v[16 + i]; }
    for (int i = 0; i < 32; ++i) hv[i] = (_Float16)v[i];
    u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(fa, fb, scale);
    u6v q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, scale);
    for (int i = 0; i < 6; ++i) { out[48 + i] = r[i]; out[54 + i] = q[i]; }
    if (ovfl) SET_OVFL(0);
}

// ---- 2: D = A (32 x 64) B (64 x 32) with operands converted on the device.  fmt 0: fp8 e4m3 (cvt_scalef32_pk_fp8_f32), 2: fp6 e2m3 (2xpk16 from f32).
// Lane l: row l & 31, values k = 32 (l >> 5) + j, j = 0..31 in register order.  sa_mode 0: immediate 127; 1: scale_a from a VGPR holding
// bytes (120 + (l & 7)) | (125 << 8) | (130 << 16) | (118 << 24) with op_sel byte `sel`; scale_b = 127 always.
template <int FMT, int SEL>
__global__ void k_mfma_q(const float *A, const float *B, float cscale, int sa_mode, unsigned *RA, unsigned *RB, float *D) {
    const int l = threadIdx.x, row = l & 31, half = l >> 5;
    float va[32], vb[32];
    for (int j = 0; j < 32; ++j) { va[j] = A[row * 64 + 32 * half + j]; vb[j] = B[(32 * half + j) * 32 + row]; }
    i8v a, b;
    for (int i = 0; i < 8; ++i) a[i] = b[i] = 0;
    if (FMT == 0) {
        for (int w = 0; w < 8; ++w) {
            s2 z = {0, 0};
            s2 lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, va[4 * w], va[4 * w + 1], cscale, false);
            s2 x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(lo, va[4 * w + 2], va[4 * w + 3], cscale, true);
            a[w] = __builtin_bit_cast(int, x);
            lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, vb[4 * w], vb[4 * w + 1], cscale, false);
            x = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(lo, vb[4 * w + 2], vb[4 * w + 3], cscale, true);
            b[w] = __builtin_bit_cast(int, x);
        }
    } else {
        f16v a0, a1, b0, b1;
        for (int i = 0; i < 16; ++i) { a0[i] = va[i]; a1[i] = va[16 + i]; b0[i] = vb[i]; b1[i] = vb[16 + i]; }
        u6v ra = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, cscale), rb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(b0, b1, cscale);
        for (int i = 0; i < 6; ++i) { a[i] = (int)ra[i]; b[i] = (int)rb[i]; }
    }
    for (int i = 0; i < 8; ++i) { RA[l * 8 + i] = (unsigned)a[i]; RB[l * 8 + i] = (unsigned)b[i]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int sv = (120 + (l & 7)) | (125 << 8) | (130 << 16) | (118 << 24);
    if (sa_mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT, FMT, 0, 127, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FMT, FMT, SEL, sv, 0, 127);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + row] = acc[r];
}

// ---- 3: rates
__device__ __forceinline__ unsigned prn(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// KIND 0: per group 12 f16 MFMAs per accumulator-quad step (today's three products on 16 k) x 4 = one 64-k step; 1: 4 x (4 f16) + 2 x (4 fp8 K64);
// 2: the same with fp6; 3: fp8 only; 4: fp6 only; 5: f16 only (4 per step).  One "step" = the work of 64 k for four 32 x 32 accumulators.
template <int KIND>
__global__ __launch_bounds__(256, 2) void k_rate(int iters, float *out) {
    // operand sets live in registers and are indexed at COMPILE time (the first form of this probe indexed a register array with the loop counter:
    // scratch traffic, not MFMA, was what it timed): 4 f16 A / B fragments and 4 low-precision fragments
    uint4 fa[4], fb[4];
    i8v q[4];
    for (int i = 0; i < 4; ++i) {
        unsigned w[16];
        for (int j = 0; j < 16; ++j) w[j] = prn(threadIdx.x * 64 + blockIdx.x * 7919 + i * 16 + j);
        fa[i] = make_uint4((w[0] & 0x83ff83ffu) | 0x38003800u, (w[1] & 0x83ff83ffu) | 0x38003800u, (w[2] & 0x83ff83ffu) | 0x38003800u, (w[3] & 0x83ff83ffu) | 0x38003800u);
        fb[i] = make_uint4((w[4] & 0x83ff83ffu) | 0x38003800u, (w[5] & 0x83ff83ffu) | 0x38003800u, (w[6] & 0x83ff83ffu) | 0x38003800u, (w[7] & 0x83ff83ffu) | 0x38003800u);
        for (int j = 0; j < 8; ++j) q[i][j] = (int)(w[8 + j] & 0xbfbfbfbfu);
    }
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 || KIND == 1 || KIND == 2 || KIND == 5) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const h8 a = __builtin_bit_cast(h8, fa[c]), b = __builtin_bit_cast(h8, fb[c]);
                const h8 a2 = __builtin_bit_cast(h8, fa[(c + 1) & 3]), b2 = __builtin_bit_cast(h8, fb[(c + 2) & 3]);
                // accumulator k <-> (weight block k >> 1, row block k & 1): two distinct A and two distinct B fragments per chunk
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16((k >> 1) ? a2 : a, (k & 1) ? b2 : b, acc[k], 0, 0, 0);
                if (KIND == 0) {
                    const h8 a3 = __builtin_bit_cast(h8, fb[(c + 3) & 3]), b3 = __builtin_bit_cast(h8, fa[(c + 2) & 3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16((k >> 1) ? a2 : a, (k & 1) ? b3 : a3, acc[k], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16((k >> 1) ? a3 : b3, (k & 1) ? b2 : b, acc[k], 0, 0, 0);
                }
            }
        }
        if (KIND == 1 || KIND == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((k >> 1) ? q[0] : q[1], (k & 1) ? q[2] : q[3], acc[k], 0, 0, 0, 127, 0, 116);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((k >> 1) ? q[3] : q[2], (k & 1) ? q[1] : q[0], acc[k], 0, 0, 0, 116, 0, 127);
        }
        if (KIND == 2 || KIND == 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((k >> 1) ? q[0] : q[1], (k & 1) ? q[2] : q[3], acc[k], 2, 2, 0, 127, 0, 116);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((k >> 1) ? q[3] : q[2], (k & 1) ? q[1] : q[0], acc[k], 2, 2, 0, 116, 0, 127);
        }
        if ((it & 63) == 63) for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] *= 1e-3f;
    }
    float sum = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) sum += acc[k][r];
    if (sum == 123.456f) out[threadIdx.x] = sum;
}
template <int KIND> static void run_rate(const char *name, float *out) {
    const int iters = KIND == 0 ? 250000 : 500000, wgs = 512;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(wgs), dim3(256), 0, 0, iters, out);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0.f; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    // ns per 64-k step of four accumulators, per wave (two waves share a SIMD)
    printf("{\"probe\": \"rate\", \"group\": \"%s\", \"ms\": %.1f, \"ns_per_64k_step_per_wave\": %.1f}\n", name, best, best * 1e6 / iters);
    fflush(stdout);
}

static void dump_u(const char *key, const unsigned *p, int n) {
    printf("\"%s\": [", key);
    for (int i = 0; i < n; ++i) printf("%s%u", i ? ", " : "", p[i]);
    printf("]");
}
static void dump_f(const char *key, const float *p, int n) {
    printf("\"%s\": [", key);
    for (int i = 0; i < n; ++i) printf("%s%.9g", i ? ", " : "", p[i]);
    printf("]");
}

int main(int argc, char **argv) {
    const bool rates_only = argc > 1;
    if (!rates_only) {
    // ---- 1
    const float tab[32] = {0.f, 1.f, -1.f, 1.0625f, 1.1875f, 1.125f, 0.3f, -0.3f, 0.0156f, 0.017f, 0.001953125f, 0.001f, 0.0009f, 3.f, 100.f, 447.f,
                           448.f, 449.f, 464.f, 480.f, 500.f, 1000.f, 70000.f, -70000.f, 7.5f, 7.75f, 8.f, 0.125f, 0.0625f, 0.06f, 5.f, 6.5f};
    float *din; unsigned *dout;
    hipMalloc(&din, 32 * 4); hipMalloc(&dout, 128 * 4);
    hipMemcpy(din, tab, sizeof(tab), hipMemcpyHostToDevice);
    const float scales[5] = {1.f, 1.f / 2048.f, 16.f, 2048.f, 0.25f};
    for (int ov = 0; ov < 2; ++ov)
        for (int s = 0; s < 5; ++s) {
            hipMemset(dout, 0, 128 * 4);
            hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, scales[s], ov, dout);
            unsigned h[128];
            hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
            printf("{\"probe\": \"cvt\", \"fp16_ovfl\": %d, \"scale\": %.9g, ", ov, scales[s]);
            dump_f("in", tab, 32); printf(", ");
            dump_u("fp8_f32", h, 16); printf(", "); dump_u("fp8_f16", h + 16, 16); printf(", "); dump_u("fp8_plain", h + 32, 16); printf(", ");
            dump_u("fp6_2x16_f32", h + 48, 6); printf(", "); dump_u("fp6_32_f16", h + 54, 6); printf(", ");
            dump_u("f16_rne", h + 60, 16); printf(", "); dump_u("f16_rtz", h + 76, 16);
            printf("}\n");
        }
    // ---- 2
    srand(2);
    std::vector<float> A(32 * 64), B(64 * 32), D(1024);
    for (auto &v : A) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 6.0);
    for (auto &v : B) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 6.0);
    float *dA, *dB, *dD; unsigned *dRA, *dRB;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096); hipMalloc(&dRA, 64 * 8 * 4); hipMalloc(&dRB, 64 * 8 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int fmt = cfg < 2 ? 0 : 2, mode = cfg == 0 || cfg == 2 ? 0 : 1, sel = cfg == 4 ? 1 : cfg == 5 ? 3 : 0;
        if (fmt == 0 && mode == 0) hipLaunchKernelGGL((k_mfma_q<0, 0>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 0, dRA, dRB, dD);
        else if (fmt == 0) hipLaunchKernelGGL((k_mfma_q<0, 0>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 1, dRA, dRB, dD);
        else if (mode == 0) hipLaunchKernelGGL((k_mfma_q<2, 0>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 0, dRA, dRB, dD);
        else if (sel == 0) hipLaunchKernelGGL((k_mfma_q<2, 0>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 1, dRA, dRB, dD);
        else if (sel == 1) hipLaunchKernelGGL((k_mfma_q<2, 1>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 1, dRA, dRB, dD);
        else hipLaunchKernelGGL((k_mfma_q<2, 3>), dim3(1), dim3(64), 0, 0, dA, dB, 1.f, 1, dRA, dRB, dD);
        std::vector<unsigned> ra(512), rb(512);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(ra.data(), dRA, 2048, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), dRB, 2048, hipMemcpyDeviceToHost);
        printf("{\"probe\": \"mfma_q\", \"fmt\": %d, \"scale_a_mode\": %d, \"op_sel\": %d, ", fmt, mode, sel);
        if (cfg == 0) { dump_f("A", A.data(), 2048); printf(", "); dump_f("B", B.data(), 2048); printf(", "); }
        dump_u("RA", ra.data(), 512); printf(", "); dump_u("RB", rb.data(), 512); printf(", "); dump_f("D", D.data(), 1024);
        printf("}\n");
    }
    }
    // ---- 3
    float *out;
    hipMalloc(&out, 4096);
    run_rate<5>("4 f16 (one product)", out);
    run_rate<0>("12 f16 (three products: shipped)", out);
    run_rate<1>("4 f16 + 2 fp8 K64", out);
    run_rate<2>("4 f16 + 2 fp6 K64", out);
    run_rate<3>("2 fp8 K64", out);
    run_rate<4>("2 fp6 K64", out);
    run_rate<0>("12 f16 (three products: shipped), again", out);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
