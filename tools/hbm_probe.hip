// hbm_probe.hip -- what does the LOAD PATTERN of the split-bf16 weight-gradient kernel cost?  (dev tool, not part of the product)
// hipcc --offload-arch=gfx950 -O3 hbm_probe.hip -o hbm_probe.  256 workgroups (one per CU) of 512 threads stream two operands of
// 1 KB rows, 16 rows per step, the way k_wgrad_b3 does (every thread one column, 8 rows = 8 dword loads per operand), with one or
// two steps of loads in flight, against the same bytes fetched as 8-byte and 16-byte loads.  The consumer is a few adds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int W, int SETS>   // W dwords per load; a step is 16 rows x 256 columns x 2 operands = 32 KB per workgroup
__global__ __launch_bounds__(512) void k_stream(const float *__restrict__ A, const float *__restrict__ B, long long rows_per_wg, float *__restrict__ out) {
    constexpr int LD = 256, NL = 8 / W;           // loads per thread, operand and step
    const int tid = threadIdx.x;
    // W == 1: column tid & 255, rows (tid >> 8) * 8 + j.  W > 1: the 4096 floats of the step as contiguous W-vectors
    const long long r0 = (long long)blockIdx.x * rows_per_wg;
    float acc = 0.f;
    typedef float vec __attribute__((ext_vector_type(W)));
    vec sa[SETS][NL], sb[SETS][NL];
    auto load = [&](int s, long long r) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            long long off;
            if (W == 1) off = (r + (tid >> 8) * 8 + j) * LD + (tid & 255);
            else off = r * LD + ((long long)j * 512 + tid) * W;
            sa[s][j] = *reinterpret_cast<const vec *>(A + off);
            sb[s][j] = *reinterpret_cast<const vec *>(B + off);
        }
    };
    auto use = [&](int s) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
#pragma unroll
            for (int w = 0; w < W; ++w) acc += sa[s][j][w] * 1.0001f + sb[s][j][w];
    };
    const int nsteps = (int)(rows_per_wg / 16);
#pragma unroll
    for (int s = 0; s < SETS; ++s) load(s, r0 + 16LL * s);
    for (int i = 0; i < nsteps; i += SETS) {
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            use(s);
            if (i + s + SETS < nsteps) load(s, r0 + 16LL * (i + s + SETS));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (acc == 123.456f) out[blockIdx.x * 512 + tid] = acc;
}

template <int W, int SETS>
void run(const float *A, const float *B, long long rows, float *out) {
    const long long rpw = rows / 256 / 48 * 48;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_stream<W, SETS>), dim3(256), dim3(512), 0, 0, A, B, rpw, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * 256 * rpw * 1024.0;
    printf("{\"probe\": \"hbm_stream\", \"load_bytes\": %d, \"sets_in_flight\": %d, \"ms\": %.3f, \"TBps\": %.3f}\n", W * 4, SETS, ms, bytes / ms * 1e-9);
}

int main() {
    const long long rows = 7000000;               // the bench's row count: 7.2 GB per operand
    float *A, *B, *out;
    if (hipMalloc(&A, rows * 1024) != hipSuccess || hipMalloc(&B, rows * 1024) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 256 * 512 * 4);
    hipMemset(A, 0, rows * 1024); hipMemset(B, 0, rows * 1024);
    run<1, 1>(A, B, rows, out); run<1, 2>(A, B, rows, out); run<1, 3>(A, B, rows, out);
    run<2, 2>(A, B, rows, out); run<4, 2>(A, B, rows, out); run<4, 3>(A, B, rows, out);
    return 0;
}
