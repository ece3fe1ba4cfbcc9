// grid.hip -- deterministic voxel grid over the neural point cloud.
//
// Replaces the reference's atomics-ordered hash build (claim_occ / map_coor2occ / fill_occ2pnts,
// models/neural_points/cuda/query_worldcoords.cu:18-162 and the five G-sized int32 tables the host
// allocates per call, :314-318,:337) with a brick map:
//     bricks[NB] uint4  one record per 4 x 4 x 4 block of cells: 64 "cell holds points" bits, the number of occupied cells in
//                       the bricks before it (rank), the number of points before it
//     ostart[n_occ+1]   offsets into pts of the occupied cells, in (brick, cell-in-brick) order
//     pts[M] float4     (x, y, z, bitcast point index), sorted by (brick, cell, point index)
//     occ bits [G/32]   occupancy dilated by query_size, plain (x, y, z) bit order (the ray probe's field)
// A cell lookup is one 16-byte brick record (4 bytes per 16 cells: the whole table is 2.2 MB at lego size and lives in L2 -- the
// dense CSR offset array it replaces was 35.5 MB and cost one HBM sector per visited cell: 5x the algorithmic traffic of the
// neighbor query, profiles/traffic.json of round 2), and only for cells that hold points a rank (popcount) -> ostart -> records
// walk.  The 27 cells of a sample's neighbourhood fall into <= 8 bricks, and their records are adjacent in memory.
// Points of a cell are contiguous and in ascending point index, which IS the reference's canonical serial order;
// the first P of them are the ones the reference keeps.  The reference's quirk that the first claimed voxel (id 0) never
// receives points (.cu:147) is carried as info[PNERF_GI_CELL0], the (brick * 64 + cell-in-brick) id of that cell.
//
// Compiled with -ffp-contract=off: the cell arithmetic must round exactly like the reference's.
#include <limits.h>
#include "pn_common.h"

PnGridLayout pn_grid_layout(const pnerf_grid_params *gp, int n) {
    PnGridLayout L;
    L.G = (long long)gp->vdim[0] * gp->vdim[1] * gp->vdim[2];
    L.NB = (long long)pn_cdiv(gp->vdim[0], 4) * pn_cdiv(gp->vdim[1], 4) * pn_cdiv(gp->vdim[2], 4);
    const size_t np = (size_t)(n > 0 ? n : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pn_align(bytes); return o; };
    // persistent part; everything up to pts sits at offsets that do not depend on n, the offset of ostart is kept in info
    L.info = take(PNERF_GI_LEN * sizeof(int));
    L.bricks = take((size_t)L.NB * sizeof(uint4));
    L.occ = take((size_t)((L.G + 31) / 32) * sizeof(uint32_t));
    L.pts = take(np * sizeof(float4));
    L.ostart = take((np + 2) * sizeof(int));
    // build-time scratch
    L.keys = take(np * sizeof(int));
    L.cnt = take((size_t)L.NB * 64 * sizeof(int));
    L.tmp_idx = take(np * sizeof(int));
    L.ocell = take((np + 1) * sizeof(int));
    L.bocc = take((size_t)(L.NB + 1) * sizeof(int));
    L.bpts = take((size_t)(L.NB + 1) * sizeof(int));
    L.brank = take((size_t)(L.NB + 1) * sizeof(int));
    L.bbase = take((size_t)(L.NB + 1) * sizeof(int));
    L.scan = take(pn_scan_scratch_ints(L.NB) * sizeof(int));
    L.total = off;
    return L;
}

PnGridDev pn_grid_dev(const pnerf_grid_params *gp, const void *ws, int) {
    PnGridLayout L = pn_grid_layout(gp, 1);   // offsets of the part used here do not depend on n
    const char *b = (const char *)ws;
    PnGridDev g;
    g.ox = gp->ranges[0]; g.oy = gp->ranges[1]; g.oz = gp->ranges[2];
    g.vx = gp->vsize[0]; g.vy = gp->vsize[1]; g.vz = gp->vsize[2];
    g.gx = gp->vdim[0]; g.gy = gp->vdim[1]; g.gz = gp->vdim[2];
    g.by = pn_cdiv(gp->vdim[1], 4); g.bz = pn_cdiv(gp->vdim[2], 4);
    g.P = gp->P;
    g.base = b;
    g.info = (const int *)(b + L.info);
    g.bricks = (const uint4 *)(b + L.bricks);
    g.occ = (const uint32_t *)(b + L.occ);
    g.pts = (const float4 *)(b + L.pts);
    return g;
}

namespace {
constexpr int TPB = 256;

struct GP {   // by-value kernel argument
    float ox, oy, oz, vx, vy, vz;
    int gx, gy, gz, qx, qy, qz;
    int by, bz;
};

__global__ __launch_bounds__(TPB) void k_grid_init(int *info, int ostart_off) {
    if (threadIdx.x < PNERF_GI_LEN) info[threadIdx.x] = (threadIdx.x == PNERF_GI_FIRST_IDX) ? INT_MAX :
                                                        (threadIdx.x == PNERF_GI_CELL0 ? -1 : (threadIdx.x == PNERF_GI_OSTART_OFF ? ostart_off : 0));
}

// one thread per point: cell id (brick * 64 + cell-in-brick, or -1), per-cell counts, first in-grid point index
__global__ __launch_bounds__(TPB) void k_grid_count(GP g, const float *__restrict__ xyz, int n,
                                                    int *__restrict__ keys, int *__restrict__ cnt, int *info) {
    int i = blockIdx.x * TPB + threadIdx.x;
    int key = -1;
    if (i < n) {
        int cx = pn_cell(xyz[3 * i], g.ox, g.vx);
        int cy = pn_cell(xyz[3 * i + 1], g.oy, g.vy);
        int cz = pn_cell(xyz[3 * i + 2], g.oz, g.vz);
        if (cx >= 0 && cx < g.gx && cy >= 0 && cy < g.gy && cz >= 0 && cz < g.gz)
            key = pn_brick_of(cx, cy, cz, g.by, g.bz) * 64 + pn_cell_in_brick(cx, cy, cz);
        keys[i] = key;
        if (key >= 0) atomicAdd(&cnt[key], 1);
    }
    // wave-aggregated bookkeeping: number of in-grid points, smallest in-grid index
    unsigned long long b = __ballot(key >= 0);
    if (b && (threadIdx.x & 63) == __ffsll((long long)b) - 1) {
        atomicAdd(&info[PNERF_GI_N_IN_GRID], __popcll(b));
        atomicMin(&info[PNERF_GI_FIRST_IDX], i);   // lowest active lane holds the wave's smallest index
    }
}

__global__ void k_grid_cell0(const int *keys, int n, int *info) {
    int f = info[PNERF_GI_FIRST_IDX];
    info[PNERF_GI_CELL0] = (f >= 0 && f < n) ? keys[f] : -1;
}

// one wavefront per brick, one lane per cell: occupancy bits, occupied cells and points of the brick
__global__ __launch_bounds__(TPB) void k_brick_bits(long long NB, const int *__restrict__ cnt, int *__restrict__ bocc, int *__restrict__ bpts,
                                                    int *info) {
    const long long brick = (long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (brick >= NB) return;
    const int c = cnt[brick * 64 + (threadIdx.x & 63)];
    const unsigned long long bits = __ballot(c > 0);
    int sum = c, mx = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off, 64); mx = max(mx, __shfl_xor(mx, off, 64)); }
    if ((threadIdx.x & 63) == 0) {
        bocc[brick] = __popcll(bits);
        bpts[brick] = sum;
        if (bits) atomicMax(&info[PNERF_GI_MAX_CNT], mx);
    }
}

// after the two scans over the bricks: brick records, the offsets of the occupied cells, the scatter cursors
__global__ __launch_bounds__(TPB) void k_brick_fill(long long NB, int *__restrict__ cnt, const int *__restrict__ brank, const int *__restrict__ bbase,
                                                    uint4 *__restrict__ bricks, int *__restrict__ ostart, int *__restrict__ ocell, int *info) {
    const long long brick = (long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (brick >= NB) return;
    const int lane = threadIdx.x & 63;
    const int key = (int)(brick * 64 + lane);
    const int c = cnt[key];
    const unsigned long long bits = __ballot(c > 0);
    const int rank0 = brank[brick], base0 = bbase[brick];
    if (lane == 0) bricks[brick] = make_uint4((unsigned)bits, (unsigned)(bits >> 32), (unsigned)rank0, (unsigned)base0);
    if (brick == NB - 1 && lane == 0) {              // totals: one past the last brick
        info[PNERF_GI_N_OCC] = brank[NB];
        ostart[brank[NB]] = bbase[NB];
    }
    if (!bits) return;
    int pre = c;                                     // inclusive prefix of the cell counts inside the brick
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(pre, d, 64); if (lane >= d) pre += v; }
    if (c > 0) {
        const int o = rank0 + __popcll(bits & ((1ull << lane) - 1ull));
        const int st = base0 + pre - c;
        ostart[o] = st;
        ocell[o] = key;
        cnt[key] = st;                               // from here on the cell's scatter cursor
    }
}

__global__ __launch_bounds__(TPB) void k_grid_scatter(const int *__restrict__ keys, int n, int *__restrict__ cursor, int *__restrict__ tmp_idx) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int key = keys[i];
    if (key < 0) return;
    tmp_idx[atomicAdd(&cursor[key], 1)] = i;
}

// one thread per occupied cell: order the cell's points by ascending index (insertion sort of a short
// segment), emit the packed point records, dilate the occupancy bits.
__global__ __launch_bounds__(TPB) void k_grid_finalize(GP g, int n, const float *__restrict__ xyz, const int *__restrict__ ostart,
                                                       const int *__restrict__ ocell, int *__restrict__ tmp_idx,
                                                       float4 *__restrict__ pts, uint32_t *__restrict__ occ, const int *info) {
    const int o = blockIdx.x * TPB + threadIdx.x;
    if (o >= n || o >= info[PNERF_GI_N_OCC]) return;
    const int s = ostart[o], m = ostart[o + 1] - s;
    for (int a = 1; a < m; ++a) {
        int v = tmp_idx[s + a];
        int b = a - 1;
        while (b >= 0 && tmp_idx[s + b] > v) { tmp_idx[s + b + 1] = tmp_idx[s + b]; --b; }
        tmp_idx[s + b + 1] = v;
    }
    for (int a = 0; a < m; ++a) {
        int idx = tmp_idx[s + a];
        pts[s + a] = make_float4(xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2], __int_as_float(idx));
    }
    // map_coor2occ dilation (.cu:104-113; the host passes query_size, .cu:342)
    const int key = ocell[o], brick = key >> 6, local = key & 63;
    const int bzi = brick % g.bz, byi = (brick / g.bz) % g.by, bxi = brick / (g.bz * g.by);
    const int cx = bxi * 4 + (local >> 4), cy = byi * 4 + ((local >> 2) & 3), cz = bzi * 4 + (local & 3);
    int x0 = max(0, cx - g.qx / 2), x1 = min(g.gx, cx + (g.qx + 1) / 2);
    int y0 = max(0, cy - g.qy / 2), y1 = min(g.gy, cy + (g.qy + 1) / 2);
    int z0 = max(0, cz - g.qz / 2), z1 = min(g.gz, cz + (g.qz + 1) / 2);
    for (int x = x0; x < x1; ++x)
        for (int y = y0; y < y1; ++y)
            for (int z = z0; z < z1; ++z) {
                long long lin = (long long)x * (g.gy * g.gz) + (long long)y * g.gz + z;
                atomicOr(&occ[lin >> 5], 1u << (lin & 31));
            }
}
}  // namespace

extern "C" size_t pnerf_grid_workspace_bytes(const pnerf_grid_params *gp, int n_points) {
    if (!gp) return 0;
    return pn_grid_layout(gp, n_points).total;
}

extern "C" int pnerf_grid_build(const pnerf_grid_params *gp, const float *d_xyz, int n, void *ws, size_t ws_bytes,
                                void *stream) {
    if (!gp || !ws || (n > 0 && !d_xyz) || n < 0) return PNERF_E_INVAL;
    for (int a = 0; a < 3; ++a)
        if (gp->vdim[a] <= 0 || !(gp->vsize[a] > 0.f) || gp->query_size[a] <= 0 || gp->kernel_size[a] <= 0) return PNERF_E_INVAL;
    PnGridLayout L = pn_grid_layout(gp, n);
    if (L.G <= 0 || L.G >= (1LL << 31) - 64 || L.NB * 64 >= (1LL << 31) - 64) return PNERF_E_UNSUP;
    if (ws_bytes < L.total) return PNERF_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *b = (char *)ws;
    int *info = (int *)(b + L.info), *ostart = (int *)(b + L.ostart), *keys = (int *)(b + L.keys), *cnt = (int *)(b + L.cnt);
    int *tmp_idx = (int *)(b + L.tmp_idx), *ocell = (int *)(b + L.ocell), *scan = (int *)(b + L.scan);
    int *bocc = (int *)(b + L.bocc), *bpts = (int *)(b + L.bpts), *brank = (int *)(b + L.brank), *bbase = (int *)(b + L.bbase);
    uint4 *bricks = (uint4 *)(b + L.bricks);
    uint32_t *occ = (uint32_t *)(b + L.occ);
    float4 *pts = (float4 *)(b + L.pts);
    GP g = {gp->ranges[0], gp->ranges[1], gp->ranges[2], gp->vsize[0], gp->vsize[1], gp->vsize[2],
            gp->vdim[0], gp->vdim[1], gp->vdim[2], gp->query_size[0], gp->query_size[1], gp->query_size[2],
            pn_cdiv(gp->vdim[1], 4), pn_cdiv(gp->vdim[2], 4)};

    PnProfScope prof(PNK_GRID, s);
    hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(TPB), 0, s, info, (int)(L.ostart / 256));
    if (hipMemsetAsync(cnt, 0, (size_t)L.NB * 64 * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(occ, 0, (size_t)((L.G + 31) / 32) * sizeof(uint32_t), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n > 0) hipLaunchKernelGGL(k_grid_count, dim3(pn_cdiv(n, TPB)), dim3(TPB), 0, s, g, d_xyz, n, keys, cnt, info);
    hipLaunchKernelGGL(k_grid_cell0, dim3(1), dim3(1), 0, s, keys, n, info);
    const int nbw = (int)pn_cdiv(L.NB, TPB / 64);
    hipLaunchKernelGGL(k_brick_bits, dim3(nbw), dim3(TPB), 0, s, L.NB, cnt, bocc, bpts, info);
    int rc = pn_exclusive_scan_i32(bocc, brank, L.NB, scan, s);
    if (rc) return rc;
    rc = pn_exclusive_scan_i32(bpts, bbase, L.NB, scan, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_brick_fill, dim3(nbw), dim3(TPB), 0, s, L.NB, cnt, brank, bbase, bricks, ostart, ocell, info);
    if (n > 0) {
        hipLaunchKernelGGL(k_grid_scatter, dim3(pn_cdiv(n, TPB)), dim3(TPB), 0, s, keys, n, cnt, tmp_idx);
        hipLaunchKernelGGL(k_grid_finalize, dim3(pn_cdiv(n, TPB)), dim3(TPB), 0, s, g, n, d_xyz, ostart, ocell, tmp_idx, pts, occ, info);
    }
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_grid_info(const void *ws, int32_t *host_info, void *stream) {
    if (!ws || !host_info) return PNERF_E_INVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(host_info, ws, PNERF_GI_LEN * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return PNERF_E_LAUNCH;
    return 0;
}

// ---- a1: elementwise min / max over all points (lighting_fast_querier.get_hyperparameters, models/neural_points/point_query.py:51-52: the grid's
// extent before it is clamped to opt.ranges and padded): one pass over xyz -- 24 MB at 2 M points, ~10 us at the HBM roof -- instead of two
// ATen reductions of 1.1 .. 1.5 ms each (profiles/r05_kernel_stats.csv: as long as the whole grid build).  Floats are ordered as unsigned keys
// (sign bit flipped for x >= 0, all bits for x < 0), per-wave shuffle reduction, one atomicMin / atomicMax per wave and component.
namespace {
__device__ __forceinline__ unsigned pn_fkey(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float pn_fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ void k_minmax_init(unsigned *__restrict__ key) { if (threadIdx.x < 6) key[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u; }
__global__ __launch_bounds__(256) void k_minmax(const float *__restrict__ xyz, long long n, unsigned *__restrict__ key) {
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned k = pn_fkey(xyz[3 * i + c]);
            mn[c] = min(mn[c], k); mx[c] = max(mx[c], k);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[c] = min(mn[c], __shfl_xor(mn[c], off, 64)); mx[c] = max(mx[c], __shfl_xor(mx[c], off, 64)); }
    }
    // one atomic per BLOCK and component (six addresses take every atomic of the launch: 2048 blocks x 4 waves of them serialised in L2 for 0.5 ms
    // in the first form of this kernel)
    __shared__ unsigned red[4][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { red[threadIdx.x >> 6][c] = mn[c]; red[threadIdx.x >> 6][3 + c] = mx[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned v = red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&key[threadIdx.x], v); else atomicMax(&key[threadIdx.x], v);
    }
}
__global__ void k_minmax_decode(unsigned *__restrict__ key) {
    if (threadIdx.x < 6) { const float f = pn_fkey_inv(key[threadIdx.x]); key[threadIdx.x] = __float_as_uint(f); }
}
}  // namespace

extern "C" int pnerf_points_minmax(const float *d_xyz, int64_t n, float *d_out6, void *stream) {
    if (!d_xyz || !d_out6 || n <= 0) return PNERF_E_INVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned *key = reinterpret_cast<unsigned *>(d_out6);
    PnProfScope prof(PNK_GRID, s);
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(64), 0, s, key);
    const int blocks = (int)(n / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, s, d_xyz, (long long)n, key);
    hipLaunchKernelGGL(k_minmax_decode, dim3(1), dim3(64), 0, s, key);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_version(void) { return 1000; }
extern "C" const char *pnerf_arch(void) { return "gfx950"; }
