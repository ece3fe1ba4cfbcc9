// scan.hip -- exclusive scan / stream compaction over int32 arrays (three-kernel, deterministic).
// Used for the CSR offsets of the voxel grid (cells -> point ranges) and for the aggregator's
// work list (samples with >=1 neighbor), so that neither needs a host round trip.
#include "pn_common.h"

namespace {
constexpr int TPB = 256;
constexpr int EPT = 8;
constexpr int CHUNK = TPB * EPT;   // 2048 elements per block

template <int MODE> __device__ __forceinline__ int xf(int v) { return MODE == 0 ? v : (v > 0 ? 1 : 0); }

// exclusive scan of one value per thread across a 256-thread block
__device__ __forceinline__ int block_excl_scan(int v, int *block_total, int *lds4) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) lds4[wave] = x;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) {
        int t = lds4[w];
        if (w < wave) woff += t;
        tot += t;
    }
    *block_total = tot;
    __syncthreads();
    return woff + x - v;
}

template <int MODE>
__global__ __launch_bounds__(TPB) void k_scan_reduce(const int *__restrict__ in, long long n, int *__restrict__ bsum) {
    __shared__ int lds4[4];
    long long base = (long long)blockIdx.x * CHUNK + (long long)threadIdx.x * EPT;
    int s = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        long long i = base + e;
        if (i < n) s += xf<MODE>(in[i]);
    }
    int tot;
    block_excl_scan(s, &tot, lds4);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: exclusive scan of bsum[0..nb) in place, grand total to bsum[nb]
__global__ __launch_bounds__(TPB) void k_scan_bsums(int *bsum, int nb) {
    __shared__ int lds4[4];
    int carry = 0;
    for (int base = 0; base < nb; base += TPB) {
        int i = base + threadIdx.x;
        int v = i < nb ? bsum[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot, lds4);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}

// MODE 0: out[i] = exclusive prefix, out[n] = total.  MODE 1: list[prefix] = i where in[i] > 0, *count = total.
template <int MODE>
__global__ __launch_bounds__(TPB) void k_scan_apply(const int *__restrict__ in, long long n, const int *__restrict__ bsum,
                                                    int nb, int *__restrict__ out, int *__restrict__ count) {
    __shared__ int lds4[4];
    long long base = (long long)blockIdx.x * CHUNK + (long long)threadIdx.x * EPT;
    int v[EPT];
    int s = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        long long i = base + e;
        v[e] = i < n ? xf<MODE>(in[i]) : 0;
        s += v[e];
    }
    int tot;
    int ex = block_excl_scan(s, &tot, lds4) + bsum[blockIdx.x];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        long long i = base + e;
        if (i < n) {
            if (MODE == 0) out[i] = ex;
            else if (v[e]) out[ex] = (int)i;
        }
        ex += v[e];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (MODE == 0) out[n] = bsum[nb];
        else *count = bsum[nb];
    }
}
}  // namespace

size_t pn_scan_scratch_ints(long long n) { return (size_t)pn_cdiv(n > 0 ? n : 1, CHUNK) + 2; }

int pn_exclusive_scan_i32(const int *in, int *out, long long n, int *scratch, hipStream_t s) {
    int nb = pn_cdiv(n > 0 ? n : 1, CHUNK);
    hipLaunchKernelGGL(k_scan_reduce<0>, dim3(nb), dim3(TPB), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(TPB), 0, s, scratch, nb);
    hipLaunchKernelGGL(k_scan_apply<0>, dim3(nb), dim3(TPB), 0, s, in, n, scratch, nb, out, (int *)nullptr);
    PN_CHECK_LAUNCH();
    return 0;
}

int pn_compact_gt0_i32(const int *in, long long n, int *list, int *count_out, int *scratch, hipStream_t s) {
    int nb = pn_cdiv(n > 0 ? n : 1, CHUNK);
    hipLaunchKernelGGL(k_scan_reduce<1>, dim3(nb), dim3(TPB), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(TPB), 0, s, scratch, nb);
    hipLaunchKernelGGL(k_scan_apply<1>, dim3(nb), dim3(TPB), 0, s, in, n, scratch, nb, list, count_out);
    PN_CHECK_LAUNCH();
    return 0;
}
