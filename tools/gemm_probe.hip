// gemm_probe.hip -- dev tool (not part of the product): the layer chain of the aggregator tile kernels in isolation -- four 64 x 256 x 256
// two-plane f16 GEMMs (pn_gemm_f16x3: activation tile in LDS, weight fragments from the L2-resident image) each followed by an epilogue
// that writes the next layer's input planes -- for two organisations of the same tile at two workgroups per CU:
//   A: 4 waves, each 2 feature blocks x 2 row blocks (the shipped kernels)          -> 2 waves per SIMD
//   B: 8 waves, each 1 feature block  x 2 row blocks (twice the LDS fragment reads) -> 4 waves per SIMD
// Prints microseconds per tile and workgroup.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pointnerf_amd/csrc -I../include gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "f16x3.h"

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_chain(const char *__restrict__ img, int tiles, float *__restrict__ out) {
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
    }
    if (keep == 123.456f) out[threadIdx.x] = keep;
}

template <int NW>
static void run(const char *name, const char *img, float *out, int tiles) {
    const size_t lds = PN_XBYTES + 1024;
    hipFuncSetAttribute((const void *)k_chain<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_chain<NW>, dim3(512), dim3(NW * 64), lds, 0, img, tiles, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        const double per = ms * 1e3 / (tiles / 512.0);
        printf("{\"variant\": \"%s\", \"waves_per_wg\": %d, \"ms\": %.3f, \"us_per_tile_and_wg\": %.2f, \"tflops_f16_products\": %.1f}\n", name, NW, ms, per,
               3.0 * 2.0 * 64 * 256 * 256 * 4 * tiles / (ms * 1e-3) / 1e12);
    }
}

int main() {
    char *img; float *out;
    hipMalloc(&img, 4 * PN_IMG(16, 8)); hipMemset(img, 0x2c, 4 * PN_IMG(16, 8));
    hipMalloc(&out, 4096);
    const int tiles = 512 * 200;
    run<4>("A: 4 waves x (2 fb x 2 rb)", img, out, tiles);
    run<8>("B: 8 waves x (1 fb x 2 rb)", img, out, tiles);
    return 0;
}
