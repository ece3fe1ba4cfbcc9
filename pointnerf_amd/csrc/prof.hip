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
