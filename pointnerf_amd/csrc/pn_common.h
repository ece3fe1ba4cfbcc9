// pn_common.h -- shared host/device helpers of libpnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "pnerf.h"

#define PN_WAVE 64

#define PN_CHECK_LAUNCH()                                         \
    do {                                                          \
        if (hipGetLastError() != hipSuccess) return PNERF_E_LAUNCH; \
    } while (0)

static inline size_t pn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int pn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// simple bump carver over a caller-provided workspace
struct PnCarver {
    char *base; size_t off, cap;
    PnCarver(void *p, size_t c) : base((char *)p), off(0), cap(c) {}
    template <class T> T *take(size_t n) {
        T *r = (T *)(base + off);
        off += pn_align(n * sizeof(T));
        return r;
    }
    bool ok() const { return off <= cap; }
};

// ---- optional per-kernel timing with HIP events on the launch stream (prof.hip) ----------------
enum PnKernelId { PNK_GRID = 0, PNK_PROBE, PNK_NEIGHBORS, PNK_COMPACT, PNK_PACK, PNK_AGG_FWD, PNK_COLOR_FWD, PNK_RAYMARCH_FWD,
                  PNK_RAYMARCH_BWD, PNK_COLOR_BWD, PNK_AGG_BWD, PNK_WGRAD, PNK_WGRAD_REDUCE, PNK_GATHER, PNK_ADAM, PNK_COUNT };
extern int pn_prof_enabled;
void pn_prof_mark(int id, bool begin, hipStream_t s);
struct PnProfScope {       // RAII: records an event pair around the launches issued while it lives
    int id; hipStream_t s;
    PnProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { if (pn_prof_enabled) pn_prof_mark(id, true, s); }
    ~PnProfScope() { if (pn_prof_enabled) pn_prof_mark(id, false, s); }
};

// ---- scan / compaction primitives (scan.hip) ----------------------------------------------
// scratch needed (ints) for n elements
size_t pn_scan_scratch_ints(long long n);
// out[i] = sum_{j<i} in[j]; out[n] = total (out has n+1 entries; may not alias in)
int pn_exclusive_scan_i32(const int *in, int *out, long long n, int *scratch, hipStream_t s);
// list = ascending indices i with in[i] > 0 ; *count_out = their number
int pn_compact_gt0_i32(const int *in, long long n, int *list, int *count_out, int *scratch, hipStream_t s);

// ---- device view of the voxel grid (grid.hip builds it, query.hip walks it) ----------------
struct PnGridDev {
    float ox, oy, oz;        // grid origin (ranges[0..2])
    float vx, vy, vz;        // scaled voxel size
    int gx, gy, gz;          // dims in cells
    int by, bz;              // dims in 4 x 4 x 4 bricks along y and z
    int P;
    const char *base;        // the grid workspace (ostart sits at base + 256 * info[PNERF_GI_OSTART_OFF])
    const int *info;         // PNERF_GI_* words
    const uint4 *bricks;     // [NB] (bits 0..31, bits 32..63, occupied cells before the brick, points before the brick)
    const uint32_t *occ;     // [(G+31)/32] dilated occupancy bits, (x, y, z) order
    const float4 *pts;       // [n_in_grid] (x,y,z,bitcast idx) sorted by (brick, cell, idx)
};

struct PnGridLayout {       // byte offsets inside the grid workspace
    size_t info, bricks, occ, pts, ostart, keys, cnt, tmp_idx, ocell, bocc, bpts, brank, bbase, scan, total;
    long long G, NB;
};
PnGridLayout pn_grid_layout(const pnerf_grid_params *gp, int n);
PnGridDev pn_grid_dev(const pnerf_grid_params *gp, const void *ws, int n_unused);

// cell coordinate exactly as the reference computes it (query_worldcoords.cu:38-40):
// fp32 subtract, IEEE fp32 divide (hipcc's default correctly-rounded division), floor.
// Files using this are compiled with -ffp-contract=off.
__device__ __forceinline__ int pn_cell(float p, float o, float v) {
    return (int)floorf((p - o) / v);
}
// brick map addressing (grid.hip): brick of a cell, the cell's bit inside its brick
__host__ __device__ __forceinline__ int pn_brick_of(int x, int y, int z, int by, int bz) { return ((x >> 2) * by + (y >> 2)) * bz + (z >> 2); }
__host__ __device__ __forceinline__ int pn_cell_in_brick(int x, int y, int z) { return ((x & 3) << 4) | ((y & 3) << 2) | (z & 3); }
__device__ __forceinline__ const int *pn_grid_ostart(const PnGridDev &g) { return (const int *)(g.base + (size_t)g.info[PNERF_GI_OSTART_OFF] * 256); }
// candidates of the in-range cell (x, y, z): how many (0 for an empty cell and for the reference's voxel id 0), and where the first one is
__device__ __forceinline__ int pn_cell_points(const PnGridDev &g, const int *__restrict__ ostart, int cell0, int x, int y, int z, int &st) {
    const int brick = pn_brick_of(x, y, z, g.by, g.bz), local = pn_cell_in_brick(x, y, z);
    const uint4 rec = g.bricks[brick];
    const unsigned long long bits = ((unsigned long long)rec.y << 32) | rec.x;
    if (!((bits >> local) & 1ull) || brick * 64 + local == cell0) return 0;
    const int o = (int)rec.z + __popcll(bits & ((1ull << local) - 1ull));
    st = ostart[o];
    return min(g.P, ostart[o + 1] - st);
}
