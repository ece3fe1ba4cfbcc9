// gemm_probe.hip -- dev tool (not part of the product): the layer chain of the aggregator tile kernels in isolation -- four 64 x 256 x 256
// two-plane f16 GEMMs (pn_gemm_f16x3: activation tile in LDS, weight fragments from the L2-resident image) each followed by an epilogue
// that writes the next layer's input planes -- for two organisations of the same tile at two workgroups per CU:
//   A: 4 waves, each 2 feature blocks x 2 row blocks (the shipped kernels)          -> 2 waves per SIMD
//   B: 8 waves, each 1 feature block  x 2 row blocks (twice the LDS fragment reads) -> 4 waves per SIMD
//   C: 8 waves = two halves of 4 waves, each half on its own 64-row tile (two tiles in LDS, ONE workgroup per CU), barriers common: the
//      halves request the same weight fragments at the same time (do the CU's L1 / L2 requests merge?)    -> 2 waves per SIMD
//   D: ONE 128-row tile per workgroup, 8 waves, each 1 feature block x 4 row blocks: every weight fragment is loaded once per 128 rows (half the
//      L2 -> L1 weight stream per row), twice the LDS fragment reads of A (like B), one workgroup per CU                 -> 2 waves per SIMD
// Prints microseconds per tile and workgroup.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pointnerf_amd/csrc -I../include gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "f16x3.h"

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_chain(const char *__restrict__ img, int tiles, int other_sleeps, int other_valu, int other_lds, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char *X = smem_p;
    constexpr int NFB = 8 / NW;                      // feature blocks per wave
    const int tid0 = threadIdx.x;
    for (int i = tid0; i < PN_XBYTES / 4; i += NW * 64) reinterpret_cast<unsigned *>(X)[i] = 0x2c002c00u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    float keep = 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            pn_gemm_f16x3<16, 8, NFB>(X, reinterpret_cast<const uint4 *>(img + (size_t)layer * PN_IMG(16, 8)), NFB * wave, lane, acc);
            PN_LDS_BARRIER();
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v[i] = acc[fb][rb][4 * g + i] * 1e-3f + 0.0625f; v[i] = fmaxf(v[i], 0.01f * v[i]); }
                        pn_x_store4<false>(X, 32 * rb + (lane & 31), pn_d_feat(NFB * wave + fb, g, lane), v[0], v[1], v[2], v[3]);
                    }
            PN_LDS_BARRIER();
        }
        keep += acc[0][0][0];
        for (int i = 0; i < other_sleeps; ++i) __builtin_amdgcn_s_sleep(127);          // stand-in for the tile's non-GEMM phases (8128 cycles each, pipe idle)
        {   // ... as VALU work (four independent fma chains) / as LDS traffic (8-byte writes + reads of the thread's own slots in the spare KB)
            float f0 = keep, f1 = keep + 1.f, f2 = keep + 2.f, f3 = keep + 3.f;
            for (int i = 0; i < other_valu; ++i) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
            keep += (f0 + f1) + (f2 + f3);
            volatile unsigned long long *slot = reinterpret_cast<volatile unsigned long long *>(X) + (threadIdx.x & 63) * 74 + (threadIdx.x >> 6) * 8;      // inside the tile (contents are dummies)
            unsigned long long v = (unsigned long long)tid * 0x100000001ull;
            for (int i = 0; i < other_lds; ++i) { slot[i & 7] = v; const unsigned long long r = slot[(i + 3) & 7]; v += r; }
            keep += (float)(unsigned)v;
        }
    }
    if (keep == 123.456f) out[threadIdx.x] = keep;
}

template <int NW>
static void run(const char *name, const char *img, float *out, int wgs, int tiles_per_wg, int sl, int va, int ld) {
    const size_t lds = PN_XBYTES + 1024 + (wgs == 256 ? 40 * 1024 : 0);        // 256 workgroups: LDS sized so that only ONE fits a CU
    hipFuncSetAttribute((const void *)k_chain<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int tiles = wgs * tiles_per_wg;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain<NW>, dim3(wgs), dim3(NW * 64), lds, 0, img, tiles, sl, va, ld, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("{\"variant\": \"%s\", \"wgs_per_cu\": %d, \"other\": {\"sleeps\": %d, \"valu_x4\": %d, \"lds_rw\": %d}, \"us_per_tile_and_wg\": %.2f, \"us_per_tile_and_cu\": %.2f}\n", name, wgs / 256,
           sl, va, ld, best * 1e3 / tiles_per_wg, best * 1e3 / tiles_per_wg / (wgs / 256));
}

__global__ __launch_bounds__(512) void k_chain_pair(const char *__restrict__ img, int tiles, int other_sleeps, int other_valu, int other_lds, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    char *X = smem_p + half * (PN_XBYTES + 1024);
    const int tid0 = threadIdx.x & 255;
    for (int i = tid0; i < PN_XBYTES / 4; i += 256) reinterpret_cast<unsigned *>(X)[i] = 0x2c002c00u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    float keep = 0.f;
    for (int t = 2 * blockIdx.x; t < tiles; t += 2 * gridDim.x) {
        int tid = threadIdx.x & 255;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            pn_gemm_f16x3<16, 8, 2>(X, reinterpret_cast<const uint4 *>(img + (size_t)layer * PN_IMG(16, 8)), 2 * wave, lane, acc);
            PN_LDS_BARRIER();
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { v[i] = acc[fb][rb][4 * g + i] * 1e-3f + 0.0625f; v[i] = fmaxf(v[i], 0.01f * v[i]); }
                        pn_x_store4<false>(X, 32 * rb + (lane & 31), pn_d_feat(2 * wave + fb, g, lane), v[0], v[1], v[2], v[3]);
                    }
            PN_LDS_BARRIER();
        }
        keep += acc[0][0][0];
        for (int i = 0; i < other_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
        {
            float f0 = keep, f1 = keep + 1.f, f2 = keep + 2.f, f3 = keep + 3.f;
            for (int i = 0; i < other_valu; ++i) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
            keep += (f0 + f1) + (f2 + f3);
            volatile unsigned long long *slot = reinterpret_cast<volatile unsigned long long *>(X) + (tid & 63) * 74 + (tid >> 6) * 8;
            unsigned long long v = (unsigned long long)tid * 0x100000001ull;
            for (int i = 0; i < other_lds; ++i) { slot[i & 7] = v; const unsigned long long r = slot[(i + 3) & 7]; v += r; }
            keep += (float)(unsigned)v;
        }
    }
    if (keep == 123.456f) out[threadIdx.x] = keep;
}

static void run_pair(const char *name, const char *img, float *out, int tiles_per_wg, int sl, int va, int ld) {
    const size_t lds = 2 * (PN_XBYTES + 1024);
    hipFuncSetAttribute((const void *)k_chain_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int tiles = 512 * tiles_per_wg;              // the same number of 64-row tiles as 512 workgroups of variant A
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain_pair, dim3(256), dim3(512), lds, 0, img, tiles, sl, va, ld, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("{\"variant\": \"%s\", \"wgs_per_cu\": 1, \"other\": {\"sleeps\": %d, \"valu_x4\": %d, \"lds_rw\": %d}, \"us_per_tile_and_wg\": %.2f, \"us_per_tile_and_cu\": %.2f}\n", name,
           sl, va, ld, best * 1e3 / tiles_per_wg, best * 1e3 / tiles_per_wg / 2);
}

// ---- variant D: probe-local 128-row tile (plane stride D_XPL), GEMM with one feature block x four row blocks per wave
constexpr int D_ROWS = 128, D_XPL = D_ROWS * PN_XRS, D_XBYTES = 2 * D_XPL;
template <int NC, int MB>
__device__ __forceinline__ void gemm_d(const char *X, const uint4 *__restrict__ img, int fb, int lane, f32x16 (&acc)[4]) {
    constexpr int PF = 2, NS = PF + 1;
    const char *xb = X + (lane & 31) * PN_XRS + (lane >> 5) * 16;
    const uint4 *wp = img + (size_t)fb * 128 + lane;
    uint4 wh[NS], wm[NS], xh[2][4], xm[2][4];
    auto load_w = [&](auto cc) { constexpr int c = decltype(cc)::value, s = c % NS; wh[s] = wp[(c * MB) * 128]; wm[s] = wp[(c * MB) * 128 + 64]; };
    auto load_x = [&](auto cc) {
        constexpr int c = decltype(cc)::value, s = c & 1;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            xh[s][rb] = *reinterpret_cast<const uint4 *>(xb + rb * 32 * PN_XRS + c * 32);
            xm[s][rb] = *reinterpret_cast<const uint4 *>(xb + D_XPL + rb * 32 * PN_XRS + c * 32);
        }
    };
    pn_static_for<PF>([&](auto cc) { load_w(cc); });
    load_x(std::integral_constant<int, 0>{});
    pn_static_for<NC>([&](auto cc) {
        constexpr int c = decltype(cc)::value, sw = c % NS, sx = c & 1;
        if constexpr (c + PF < NC) load_w(std::integral_constant<int, c + PF>{});
        if constexpr (c + 1 < NC) load_x(std::integral_constant<int, c + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const pn_h8 a = __builtin_bit_cast(pn_h8, p == 2 ? wm[sw] : wh[sw]);
                const pn_h8 b = __builtin_bit_cast(pn_h8, p == 1 ? xm[sx][rb] : xh[sx][rb]);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[rb], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    });
}
__global__ __launch_bounds__(512) void k_chain_d(const char *__restrict__ img, int tiles, int other_sleeps, int other_valu, int other_lds, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char *X = smem_p;
    for (int i = threadIdx.x; i < D_XBYTES / 4; i += 512) reinterpret_cast<unsigned *>(X)[i] = 0x2c002c00u + (i & 7);
    __syncthreads();
    f32x16 acc[4];
    float keep = 0.f;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 1
        for (int layer = 0; layer < 4; ++layer) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            gemm_d<16, 8>(X, reinterpret_cast<const uint4 *>(img + (size_t)layer * PN_IMG(16, 8)), wave, lane, acc);
            PN_LDS_BARRIER();
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { v[i] = acc[rb][4 * g + i] * 1e-3f + 0.0625f; v[i] = fmaxf(v[i], 0.01f * v[i]); }
                    unsigned h0, m0, h1, m1;
                    pn_split2(v[0], v[1], h0, m0); pn_split2(v[2], v[3], h1, m1);
                    char *d = X + (32 * rb + (lane & 31)) * PN_XRS + pn_d_feat(wave, g, lane) * 2;
                    *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(d + D_XPL) = make_uint2(m0, m1);
                }
            PN_LDS_BARRIER();
        }
        keep += acc[0][0];
        for (int i = 0; i < other_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
        {   // the same other work PER ROW as variant A: twice the rows with twice the threads = the same per-thread counts
            float f0 = keep, f1 = keep + 1.f, f2 = keep + 2.f, f3 = keep + 3.f;
            for (int i = 0; i < other_valu; ++i) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
            keep += (f0 + f1) + (f2 + f3);
            volatile unsigned long long *slot = reinterpret_cast<volatile unsigned long long *>(X) + (tid & 63) * 74 + (tid >> 6) * 8;
            unsigned long long v = (unsigned long long)tid * 0x100000001ull;
            for (int i = 0; i < other_lds; ++i) { slot[i & 7] = v; const unsigned long long r = slot[(i + 3) & 7]; v += r; }
            keep += (float)(unsigned)v;
        }
    }
    if (keep == 123.456f) out[threadIdx.x] = keep;
}
static void run_d(const char *name, const char *img, float *out, int tiles_per_wg, int sl, int va, int ld) {
    const size_t lds = D_XBYTES + 1024;
    hipFuncSetAttribute((const void *)k_chain_d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int tiles = 256 * tiles_per_wg;              // 128-row tiles: the same number of ROWS as 512 workgroups x tiles_per_wg of variant A
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain_d, dim3(256), dim3(512), lds, 0, img, tiles, sl, va, ld, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("{\"variant\": \"%s\", \"wgs_per_cu\": 1, \"other\": {\"sleeps\": %d, \"valu_x4\": %d, \"lds_rw\": %d}, \"us_per_128_rows_and_wg\": %.2f, \"us_per_tile_and_cu\": %.2f}\n", name,
           sl, va, ld, best * 1e3 / tiles_per_wg, best * 1e3 / tiles_per_wg / 2);
}

int main() {
    char *img; float *out;
    hipMalloc(&img, 4 * PN_IMG(16, 8)); hipMemset(img, 0x2c, 4 * PN_IMG(16, 8));
    hipMalloc(&out, 4096);
    // the same TOTAL other-phase work per tile for both organisations: B has twice the threads, so half the per-thread counts
    const int cfg[5][3] = {{0, 0, 0}, {6, 0, 0}, {0, 2500, 0}, {0, 0, 600}, {0, 1250, 300}};
    for (int c = 0; c < 5; ++c) {
        run<4>("A: 4 waves x (2 fb x 2 rb)", img, out, 256, 100, cfg[c][0], cfg[c][1], cfg[c][2]);
        run<4>("A: 4 waves x (2 fb x 2 rb)", img, out, 512, 100, cfg[c][0], cfg[c][1], cfg[c][2]);
        run<8>("B: 8 waves x (1 fb x 2 rb), same total other work", img, out, 512, 100, cfg[c][0], cfg[c][1] / 2, cfg[c][2] / 2);
        run_pair("C: 8 waves = 2 halves x 4 waves on two tiles, one workgroup per CU", img, out, 100, cfg[c][0], cfg[c][1], cfg[c][2]);
        run_d("D: 128-row tile, 8 waves x (1 fb x 4 rb), one workgroup per CU", img, out, 100, cfg[c][0], cfg[c][1], cfg[c][2]);
    }
    return 0;
}
