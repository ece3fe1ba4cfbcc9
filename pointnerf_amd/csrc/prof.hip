// prof.hip -- per-kernel wall time measured with HIP events recorded on the stream the kernels are launched on
// (torch.cuda.Event would only see torch's view of that stream).  Off by default; bench.py switches it on for the
// timed region and reads the per-kernel totals afterwards.
#include <vector>
#include "pn_common.h"

int pn_prof_enabled = 0;
namespace {
struct Rec { int id; hipEvent_t a, b; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[PNK_COUNT];
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
const char *kNames[PNK_COUNT] = {"grid_build", "probe", "neighbors", "compact", "mlp_pack", "agg_forward", "color_forward",
                                 "raymarch_forward", "raymarch_backward", "color_backward", "agg_backward", "wgrad",
                                 "wgrad_reduce", "gather", "adam"};
}  // namespace

void pn_prof_mark(int id, bool begin, hipStream_t s) {
    if (id < 0 || id >= PNK_COUNT) return;
    hipEvent_t e = get_event();
    if (!e) return;
    (void)hipEventRecord(e, s);
    if (begin) g_open[id] = e;
    else { g_recs.push_back({id, g_open[id], e}); g_open[id] = nullptr; }
}

extern "C" int pnerf_prof_enable(int on) {
    pn_prof_enabled = on ? 1 : 0;
    return 0;
}

extern "C" int pnerf_prof_kernel_count(void) { return PNK_COUNT; }
extern "C" const char *pnerf_prof_kernel_name(int id) { return (id >= 0 && id < PNK_COUNT) ? kNames[id] : ""; }

// synchronous: waits for the device, adds the elapsed ms of every recorded launch to total_ms[id] and its count to
// launches[id] (arrays of pnerf_prof_kernel_count() entries), then clears the records.
extern "C" int pnerf_prof_collect(double *total_ms, int64_t *launches) {
    if (!total_ms || !launches) return PNERF_E_INVAL;
    if (hipDeviceSynchronize() != hipSuccess) return PNERF_E_LAUNCH;
    for (int i = 0; i < PNK_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
    for (const Rec &r : g_recs) {
        float ms = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.id] += ms; launches[r.id] += 1; }
        if (r.a) g_pool.push_back(r.a);
        if (r.b) g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}

// ---- the matrix pipe's sustained rate with operands that TOGGLE like real data (bench.py: roofline.peak_measured) ---------------------------
// Register-resident v_mfma_f32_32x32x16_f16 only (no LDS, no memory in the loop), four independent accumulators per wave, two 256-thread
// workgroups per CU.  mode 0: all-zero operands, 1: one constant, 2: pseudo-random f16 in +-[0.5, 1) from a register ring.  Round 5 measured
// 2.33-2.36 PFLOP/s for modes 0 / 1 and 1.70 PFLOP/s for mode 2 on an MI355X (profiles/r05_mfma_power_probe.jsonl): the power management holds the
// clock down when the pipe's inputs switch, so the guide's 2.5 PFLOP/s is not what a GEMM on real data can reach on this chip.
namespace {
typedef _Float16 pk_h8 __attribute__((ext_vector_type(8)));
typedef float pk_f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned pk_prn(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256, 2) void k_mfma_rate(int mode, int iters, float *__restrict__ out) {
    uint4 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) {
        unsigned w[8];
        for (int j = 0; j < 8; ++j) {
            const unsigned r = pk_prn(threadIdx.x * 64 + blockIdx.x * 7919 + i * 8 + j);
            w[j] = mode == 0 ? 0u : mode == 1 ? 0x2c002c00u : ((r & 0x83ff83ffu) | 0x38003800u);
        }
        ra[i] = make_uint4(w[0], w[1], w[2], w[3]); rb[i] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    pk_f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const pk_h8 a = __builtin_bit_cast(pk_h8, ra[s]), b = __builtin_bit_cast(pk_h8, rb[(s + it) & 3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[k], 0, 0, 0);
        }
        if ((it & 63) == 63) for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] *= 1e-3f;      // keep the sums finite
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}
}  // namespace

extern "C" int pnerf_debug_mfma_rate(int mode, int iters, float *d_scratch, double *flop_out, void *stream) {
    if (mode < 0 || mode > 2 || iters <= 0 || !d_scratch) return PNERF_E_INVAL;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int wgs = 2 * cus;
    hipLaunchKernelGGL(k_mfma_rate, dim3(wgs), dim3(256), 0, (hipStream_t)stream, mode, iters, d_scratch);
    PN_CHECK_LAUNCH();
    if (flop_out) *flop_out = (double)wgs * 4.0 * iters * 32.0 * (2.0 * 32 * 32 * 16);
    return 0;
}
