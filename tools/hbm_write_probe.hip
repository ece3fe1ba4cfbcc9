// dev probe: HBM WRITE bandwidth of the forms the tile kernels use for their saved planes (16 B per lane, 1 KB per wave-instruction),
// non-temporal and plain, alone and next to a read stream of equal size.   hipcc --offload-arch=gfx950 -O3 -o hbm_write_probe hbm_write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k(f4 *__restrict__ dst, const f4 *__restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        f4 v = {1.f, 2.f, 3.f, (float)i};
        if (MODE == 2 || MODE == 3) { v = src[i]; }
        if (MODE == 4) { acc += src[i]; continue; }
        if (MODE == 0 || MODE == 2) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
    if (MODE == 4 && acc.x == 123.456f) dst[0] = acc;
}
int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    f4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[5] = {"write nontemporal", "write plain", "copy (read + nt write)", "copy (read + plain write)", "read"};
    for (int grid : {512, 2048, 8192})
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, a, b, n);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, a, b, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double moved = (mode == 2 || mode == 3) ? 2.0 * bytes : (double)bytes;
        printf("{\"probe\": \"hbm_write\", \"mode\": \"%s\", \"grid\": %d, \"ms\": %.3f, \"TBps_total\": %.3f}\n", names[mode], grid, best, moved / best / 1e9);
    }
    return 0;
}
