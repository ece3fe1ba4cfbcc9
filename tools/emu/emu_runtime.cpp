// emu_runtime.cpp -- fiber scheduler of the host emulation (see hip/hip_runtime.h).  Test infrastructure only.
// One workgroup runs at a time; each of its threads is a ucontext fiber with its own stack; the scheduler resumes the
// fibers round-robin, a fiber gives the CPU back only inside a rendezvous (workgroup barrier, wave exchange) or when it
// ends.  A full pass over the fibers without any progress is a deadlock (divergent barrier / shuffle) and aborts.
#include <ucontext.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstdlib>
#include <vector>
#include "hip/hip_runtime.h"

namespace emu {
struct Wave { int arrived = 0, gen = 0, active = 0; unsigned long long mask = 0; alignas(64) char buf[2][64 * 64]; };
struct Fiber { ucontext_t ctx; char *stack = nullptr; bool done = true; Idx3 tid; int lin = 0; Wave *wave = nullptr; };

Fiber *g_cur = nullptr;
Idx3 g_tid, g_bid, g_bdim, g_gdim;
static ucontext_t g_sched;
static std::vector<Fiber> g_fibers;
static std::vector<Wave> g_waves;
static std::vector<char> g_lds;
static const std::function<void()> *g_body = nullptr;
static int g_alive = 0, g_bar_count = 0, g_bar_gen = 0;
static unsigned long long g_progress = 0;
static constexpr size_t STACK = 256 * 1024;

char *lds() { return g_lds.data(); }
int lane_id() { return g_cur->lin & 63; }
unsigned long long wave_active_mask() { return g_cur->wave->mask; }
int ncu() { const char *e = getenv("PN_EMU_NCU"); return e ? atoi(e) : 2; }

static void yield() { Fiber *f = g_cur; swapcontext(&f->ctx, &g_sched); g_cur = f; g_tid = f->tid; }

static void fiber_exit_bookkeeping(Fiber *f) {
    f->done = true; --g_alive; ++g_progress;
    Wave *w = f->wave;
    w->active--; w->mask &= ~(1ull << (f->lin & 63));
    if (w->active > 0 && w->arrived == w->active) { w->arrived = 0; w->gen++; }
    if (g_alive > 0 && g_bar_count == g_alive) { g_bar_count = 0; g_bar_gen++; }
}

static void trampoline() {
    Fiber *f = g_cur;
    (*g_body)();
    g_cur = f;
    fiber_exit_bookkeeping(f);
    swapcontext(&f->ctx, &g_sched);
}

void syncthreads() {
    const int gen = g_bar_gen;
    ++g_progress;
    if (++g_bar_count == g_alive) { g_bar_count = 0; g_bar_gen++; return; }
    while (g_bar_gen == gen) yield();
}

const char *wave_exchange(const void *mine, size_t n) {
    Fiber *f = g_cur;
    Wave *w = f->wave;
    const int gen = w->gen;
    char *tab = w->buf[gen & 1];
    memcpy(tab + 64 * (f->lin & 63), mine, n);
    ++g_progress;
    if (++w->arrived == w->active) { w->arrived = 0; w->gen++; }
    else while (w->gen == gen) yield();
    return tab;
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
    const int nthr = (int)(block.x * block.y * block.z);
    if (nthr <= 0 || nthr > 1024) { fprintf(stderr, "emu: bad block size %d\n", nthr); abort(); }
    if ((int)g_fibers.size() < nthr) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nthr);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char *)aligned_alloc(64, STACK);
    }
    g_waves.resize((nthr + 63) / 64);
    if (g_lds.size() < 160 * 1024 + 64) g_lds.resize(160 * 1024 + 64);
    if (lds_bytes > 160 * 1024) { fprintf(stderr, "emu: %zu bytes of dynamic LDS requested\n", lds_bytes); abort(); }
    g_body = &body;
    g_bdim = {block.x, block.y, block.z}; g_gdim = {grid.x, grid.y, grid.z};
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_bid = {bx, by, bz};
                for (auto &w : g_waves) { w.arrived = 0; w.gen = 0; w.active = 0; w.mask = 0; }
                g_alive = nthr; g_bar_count = 0; g_bar_gen = 0;
                for (int t = 0; t < nthr; ++t) {
                    Fiber &f = g_fibers[t];
                    f.done = false; f.lin = t;
                    f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
                    f.wave = &g_waves[t / 64];
                    f.wave->active++; f.wave->mask |= 1ull << (t & 63);
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                while (g_alive > 0) {
                    const unsigned long long before = g_progress;
                    for (int t = 0; t < nthr; ++t) {
                        Fiber &f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = &f; g_tid = f.tid;
                        swapcontext(&g_sched, &f.ctx);
                    }
                    if (g_progress == before && g_alive > 0) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): %d threads alive, %d at the barrier (divergent __syncthreads / wave exchange?)\n",
                                bx, by, bz, g_alive, g_bar_count);
                        abort();
                    }
                }
            }
    g_body = nullptr; g_cur = nullptr;
}
}  // namespace emu

// PN_EMU_BACKTRACE=1: print the native frames of a crash inside an emulated kernel (fibers hide them from Python's faulthandler)
static void pn_emu_segv(int sig) {
    void *frames[48];
    const int n = backtrace(frames, 48);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}
__attribute__((constructor)) static void pn_emu_install_segv() {
    if (getenv("PN_EMU_BACKTRACE")) {
        static char alt[1 << 16];
        stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
        sigaltstack(&ss, nullptr);
        struct sigaction sa = {};
        sa.sa_handler = pn_emu_segv; sa.sa_flags = SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
    }
}
