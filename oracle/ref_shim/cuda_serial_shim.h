// oracle/ref_shim/cuda_serial_shim.h -- TEST INFRASTRUCTURE.
// Lets g++ compile the __global__ kernels of the reference's
// models/neural_points/cuda/query_worldcoords.cu for the HOST, unmodified, so
// that oracle/ref_driver.cpp can run each kernel's threads one after another in
// ascending global thread index (the canonical serial order of SURVEY.md 8c).
// Nothing here is reference code: it only supplies the handful of CUDA names the
// kernels use.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>

#define __global__
#define __device__
#define __host__

struct pnerf_dim3 { unsigned x, y, z; };
extern thread_local pnerf_dim3 blockIdx, blockDim, threadIdx;
extern int pnerf_ref_curand_hits;   // counts entries into the curand reservoir paths

using std::max;
using std::min;
using std::abs;

static inline int atomicCAS(int *addr, int compare, int val) {
    int old = *addr; if (old == compare) *addr = val; return old;
}
static inline int atomicAdd(int *addr, int val) { int old = *addr; *addr = old + val; return old; }

// The reservoir-replacement paths (max_o / P overflow) are seeded from the wall
// clock in the reference; parity is undefined there.  The shim makes them a
// detectable no-op: curand_uniform returns 1.0 so insrtidx == tmp >= cap.
struct curandState { int dummy; };
static inline void curand_init(unsigned long long, unsigned long long, unsigned long long, curandState *) {
    pnerf_ref_curand_hits++;
}
static inline float curand_uniform(curandState *) { return 1.0f; }
