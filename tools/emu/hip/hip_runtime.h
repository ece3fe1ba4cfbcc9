// hip_runtime.h (HOST EMULATION SHIM) -- test infrastructure, never part of the product.
//
// Lets the .hip kernel sources of pointnerf_amd/csrc be compiled as plain C++ for x86 and executed on the host, one
// workgroup at a time, every GPU thread as a fiber (tools/emu/emu_runtime.cpp): __syncthreads, the wave-level
// exchanges (__shfl*, __ballot) and the MFMA builtins are rendezvous points of the fibers of a workgroup / wave.
// Purpose: this container has no GPU and a round has 90 GPU-minutes; the emulator runs the REAL kernel code on tiny
// inputs against the oracle, so index / layout / synchronisation bugs are found on the CPU (tests/test_emu_*.py).
// The MFMA fragment layouts emulated here are the ones the round-1 kernels were validated with on hardware
// (32x32x2 f32 and 32x32x16 bf16; the f16 form shares the bf16 layout).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define PN_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

namespace emu {
struct Idx3 { unsigned x, y, z; };
struct Fiber;
extern Fiber *g_cur;
extern Idx3 g_tid, g_bid, g_bdim, g_gdim;
char *lds();
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body);
void syncthreads();
int lane_id();
// every active lane of the wave deposits n (<= 64) bytes; returns the wave's 64 x 64-byte table (valid until the lane's next rendezvous)
const char *wave_exchange(const void *mine, size_t n);
unsigned long long wave_active_mask();
int ncu();
}  // namespace emu

#define threadIdx (emu::g_tid)
#define blockIdx (emu::g_bid)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)
#define hipLaunchKernelGGL(kern, grid, block, ldsb, stream, ...) emu::launch((grid), (block), (ldsb), [=]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { emu::syncthreads(); }

// ---- host API used by the launch code
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = emu::ncu(); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ---- bit casts / integer intrinsics
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline unsigned long long wall_clock64() { return 0ull; }

// ---- HIP's global min / max
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ---- atomics (workgroups and fibers run one at a time: plain read-modify-write)
template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// ---- wave-level data exchange
template <class T> static inline T emu_lane_value(const char *tab, int l) { T v; memcpy(&v, tab + 64 * l, sizeof(T)); return v; }
template <class T> static inline T __shfl(T v, int src, int = 64) { const char *t = emu::wave_exchange(&v, sizeof(T)); return emu_lane_value<T>(t, src & 63); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { const int l = emu::lane_id(); const char *t = emu::wave_exchange(&v, sizeof(T)); return emu_lane_value<T>(t, (l ^ m) & 63); }
template <class T> static inline T __shfl_up(T v, unsigned d, int = 64) { const int l = emu::lane_id(); const char *t = emu::wave_exchange(&v, sizeof(T)); return l >= (int)d ? emu_lane_value<T>(t, l - (int)d) : v; }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) { const int l = emu::lane_id(); const char *t = emu::wave_exchange(&v, sizeof(T)); return l + (int)d < 64 ? emu_lane_value<T>(t, l + (int)d) : v; }
static inline unsigned long long __ballot(int pred) {
    const unsigned long long act = emu::wave_active_mask();
    const int p = pred ? 1 : 0;
    const char *t = emu::wave_exchange(&p, 4);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (((act >> l) & 1) && emu_lane_value<int>(t, l)) m |= 1ull << l;
    return m;
}

// ---- MFMA
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x2_f32: A[i][k] in lane i + 32 k, B[k][j] in lane j + 32 k; D reg r of lane l = element
// (i = (r & 3) + 8 (r >> 2) + 4 (l >> 5), j = l & 31); numerically a k-ordered fmaf chain.
static inline emu_f32x16 emu_mfma_32x32x2f32(float a, float b, emu_f32x16 c) {
    const int l = emu::lane_id();
    float ab[2] = {a, b};
    const char *t = emu::wave_exchange(ab, 8);
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float A[2], B[2];
            memcpy(A, t + 64 * (i + 32 * k), 8); memcpy(B, t + 64 * (j + 32 * k), 8);
            acc = fmaf(A[0], B[1], acc);
        }
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_32x32x16_{bf16,f16}: lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7] and B[k = 8 (l >> 5) .. + 7][j = l & 31]
template <class V, class E> static inline emu_f32x16 emu_mfma_32x32x16(V a, V b, emu_f32x16 c) {
    const int l = emu::lane_id();
    struct { V a, b; } ab = {a, b};
    static_assert(sizeof(ab) == 32, "two 16-byte fragments");
    const char *t = emu::wave_exchange(&ab, 32);
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = c[r];
        for (int kh = 0; kh < 2; ++kh) {
            E A[8], B[8];
            memcpy(A, t + 64 * (i + 32 * kh), 16); memcpy(B, t + 64 * (j + 32 * kh) + 16, 16);
            for (int q = 0; q < 8; ++q) acc += (double)(float)A[q] * (double)(float)B[q];
        }
        c[r] = (float)acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32x2f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16<emu_bf16x8, __bf16>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16<emu_f16x8, _Float16>((a), (b), (c))
// ---- gfx950 e4m3 conversions and the block-scaled 8-bit MFMA (semantics measured by tools/mx_probe.hip: value / scale, round to nearest even,
// subnormal step 2^-9, saturation at +-448 as under MODE.FP16_OVFL = 1; byte j of a lane's A registers meets byte j of the same lane's B
// registers; 2^(byte - 127) per lane from the byte of the scale operand that op_sel names)
static inline unsigned char emu_e4m3_enc(float v) {
    if (std::isnan(v)) return 0x7f;
    const unsigned char sg = std::signbit(v) ? 0x80 : 0;
    double a = std::fabs((double)v);
    if (a >= 448.0) return sg | 0x7e;
    if (a < 0.015625) {                                   // subnormal: multiples of 2^-9
        const int q = (int)std::nearbyint(a * 512.0);
        return q >= 8 ? (sg | 0x08) : (sg | (unsigned char)q);
    }
    int e; std::frexp(a, &e); e -= 1;                       // a in [2^e, 2^(e + 1))
    int q = (int)std::nearbyint(std::ldexp(a, 3 - e));    // 8 .. 16
    if (q == 16) { q = 8; e += 1; }
    if (e > 8 || (e == 8 && q > 14)) return sg | 0x7e;
    return sg | (unsigned char)(((e + 7) << 3) | (q - 8));
}
static inline float emu_e4m3_dec(unsigned char b) {
    const int e = (b >> 3) & 15, m = b & 7;
    if (e == 15 && m == 7) return NAN;
    const float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return (b & 0x80) ? -v : v;
}
typedef short emu_s2 __attribute__((ext_vector_type(2)));
static inline emu_s2 emu_cvt_scalef32_pk_fp8(emu_s2 old, float a, float b, float scale, bool hi) {
    const unsigned short w = (unsigned short)(emu_e4m3_enc(a / scale) | (emu_e4m3_enc(b / scale) << 8));
    old[hi ? 1 : 0] = (short)w;
    return old;
}
#define __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, src, scale, hi) emu_cvt_scalef32_pk_fp8((old), (float)(src)[0], (float)(src)[1], (scale), (hi))
#define __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, scale, hi) emu_cvt_scalef32_pk_fp8((old), (a), (b), (scale), (hi))
typedef int emu_i8v __attribute__((ext_vector_type(8)));
static inline emu_f32x16 emu_mfma_scale_32x32x64_e4m3(emu_i8v a, emu_i8v b, emu_f32x16 c, int osa, int sa, int osb, int sb) {
    const int l = emu::lane_id();
    struct { emu_i8v a, b; } ab = {a, b};
    static_assert(sizeof(ab) == 64, "two 32-byte fragments");
    char mine[64 * 64];
    memcpy(mine, emu::wave_exchange(&ab, 64), sizeof(mine));
    const int sc[2] = {(sa >> (8 * osa)) & 255, (sb >> (8 * osb)) & 255};
    const char *t2 = emu::wave_exchange(sc, 8);
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = c[r];
        // scale blocks (measured on gfx950, tools/gpu_mix_diag.py): registers 0..3 (bytes 0..15) of BOTH lane halves form K block 0, scaled by the
        // byte of lanes 0..31; registers 4..7 (bytes 16..31) of both halves form K block 1, scaled by the byte of lanes 32..63
        int sA[2][2], sB[2][2];
        for (int kb = 0; kb < 2; ++kb) { memcpy(sA[kb], t2 + 64 * (i + 32 * kb), 8); memcpy(sB[kb], t2 + 64 * (j + 32 * kb), 8); }
        for (int kh = 0; kh < 2; ++kh) {
            const unsigned char *A = (const unsigned char *)mine + 64 * (i + 32 * kh), *B = (const unsigned char *)mine + 64 * (j + 32 * kh) + 32;
            for (int kb = 0; kb < 2; ++kb) {
                double s = 0.0;
                for (int q = 16 * kb; q < 16 * kb + 16; ++q) s += (double)emu_e4m3_dec(A[q]) * (double)emu_e4m3_dec(B[q]);
                acc += std::ldexp(s, (sA[kb][0] - 127) + (sB[kb][1] - 127));
            }
        }
        c[r] = (float)acc;
    }
    return c;
}
#define __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, cbsz, blgp, osa, sa, osb, sb) emu_mfma_scale_32x32x64_e4m3((a), (b), (c), (osa), (sa), (osb), (sb))
#define __builtin_amdgcn_s_setreg(a, b) ((void)0)
// v_cvt_pkrtz_f16_f32: two floats -> two halfs, round toward zero (finite inputs never become inf)
typedef _Float16 emu_h2 __attribute__((ext_vector_type(2)));
static inline _Float16 emu_f16_rtz(float x) {
    _Float16 r = (_Float16)x;                       // round to nearest even
    unsigned short b; memcpy(&b, &r, 2);
    if ((b & 0x7fff) == 0x7c00 && !std::isinf(x)) { b = (b & 0x8000) | 0x7bff; memcpy(&r, &b, 2); return r; }
    if (std::fabs((float)r) > std::fabs(x)) { b -= 1; memcpy(&r, &b, 2); }      // one ulp toward zero
    return r;
}
static inline emu_h2 emu_cvt_pkrtz(float a, float b) { emu_h2 r; r[0] = emu_f16_rtz(a); r[1] = emu_f16_rtz(b); return r; }
#define __builtin_amdgcn_cvt_pkrtz(a, b) emu_cvt_pkrtz((a), (b))
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
#define __expf(x) expf(x)
#define __builtin_amdgcn_fmed3f(a, b, c) fmaxf(fminf(fmaxf((a), (b)), (c)), fminf((a), (b)))
// global_load_lds_dwordx4 & co: LDS destination = wave-uniform base + lane * size (+ offset), global source per lane
static inline void emu_global_load_lds(const void *g, void *lds_base, unsigned size, unsigned offset) {
    memcpy((char *)lds_base + offset + (size_t)emu::lane_id() * size, g, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((g), (void *)(l), (size), (off))
// ds_read_b64_tr_b16 (semantics measured on gfx950 by tools/trb16_probe.hip): per 16-lane group, lane i element j = element (i & 3) of
// the 8-byte slot addressed by lane 4 j + (i >> 2)
typedef short emu_s4 __attribute__((ext_vector_type(4)));
static inline emu_s4 emu_ds_read_tr16_b64(const void *p) {
    const int l = emu::lane_id(), g = l & ~15, i = l & 15;
    const char *t = emu::wave_exchange(&p, sizeof(p));
    emu_s4 r;
    for (int j = 0; j < 4; ++j) {
        const char *src = emu_lane_value<const char *>(t, g + 4 * j + (i >> 2));
        short v; memcpy(&v, src + 2 * (i & 3), 2);
        r[j] = v;
    }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu_ds_read_tr16_b64((const void *)(p))
#define __builtin_amdgcn_alignbit(hi, lo, sh) ((unsigned)(((((unsigned long long)(hi)) << 32) | (unsigned)(lo)) >> ((sh) & 31)))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_s_barrier() emu::syncthreads()
