// grid.hip -- deterministic voxel grid over the neural point cloud.
//
// Replaces the reference's atomics-ordered hash build (claim_occ / map_coor2occ / fill_occ2pnts,
// models/neural_points/cuda/query_worldcoords.cu:18-162 and the five G-sized int32 tables the host
// allocates per call, :314-318,:337) with a CSR grid:
//     cell_start[G+1]   offsets into pts
//     pts[M] float4     (x, y, z, bitcast point index), sorted by (cell, point index)
//     occ bits [G/32]   occupancy dilated by query_size (35 MB int32 -> 1.1 MB bit field at lego size)
// Points of a cell are contiguous (one coalesced walk per candidate cell) and in ascending point
// index, which IS the reference's canonical serial order; the first P of them are the
// ones the reference keeps.  The reference's quirk that the first claimed voxel (id 0) never receives
// points (.cu:147) is carried as info[PNERF_GI_CELL0], the linear id of that cell.
//
// Compiled with -ffp-contract=off: the cell arithmetic must round exactly like the reference's.
#include <limits.h>
#include "pn_common.h"

PnGridLayout pn_grid_layout(const pnerf_grid_params *gp, int n) {
    PnGridLayout L;
    L.G = (long long)gp->vdim[0] * gp->vdim[1] * gp->vdim[2];
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pn_align(bytes); return o; };
    L.info = take(PNERF_GI_LEN * sizeof(int));
    L.cell_start = take((size_t)(L.G + 1) * sizeof(int));
    L.occ = take((size_t)((L.G + 31) / 32) * sizeof(uint32_t));
    L.pts = take((size_t)(n > 0 ? n : 1) * sizeof(float4));
    L.keys = take((size_t)(n > 0 ? n : 1) * sizeof(int));
    L.cursor = take((size_t)L.G * sizeof(int));
    L.tmp_idx = take((size_t)(n > 0 ? n : 1) * sizeof(int));
    L.scan = take(pn_scan_scratch_ints(L.G) * sizeof(int));
    L.total = off;
    return L;
}

PnGridDev pn_grid_dev(const pnerf_grid_params *gp, const void *ws, int) {
    PnGridLayout L = pn_grid_layout(gp, 1);   // offsets of the persistent part do not depend on n
    const char *b = (const char *)ws;
    PnGridDev g;
    g.ox = gp->ranges[0]; g.oy = gp->ranges[1]; g.oz = gp->ranges[2];
    g.vx = gp->vsize[0]; g.vy = gp->vsize[1]; g.vz = gp->vsize[2];
    g.gx = gp->vdim[0]; g.gy = gp->vdim[1]; g.gz = gp->vdim[2];
    g.P = gp->P;
    g.info = (const int *)(b + L.info);
    g.cell_start = (const int *)(b + L.cell_start);
    g.occ = (const uint32_t *)(b + L.occ);
    g.pts = (const float4 *)(b + L.pts);
    return g;
}

namespace {
constexpr int TPB = 256;

struct GP {   // by-value kernel argument
    float ox, oy, oz, vx, vy, vz;
    int gx, gy, gz, qx, qy, qz;
};

__global__ __launch_bounds__(TPB) void k_grid_init(int *info) {
    if (threadIdx.x < PNERF_GI_LEN) info[threadIdx.x] = (threadIdx.x == PNERF_GI_FIRST_IDX) ? INT_MAX :
                                                        (threadIdx.x == PNERF_GI_CELL0 ? -1 : 0);
}

// one thread per point: linear cell id (or -1), per-cell counts, first in-grid point index
__global__ __launch_bounds__(TPB) void k_grid_count(GP g, const float *__restrict__ xyz, int n,
                                                    int *__restrict__ keys, int *__restrict__ cnt, int *info) {
    int i = blockIdx.x * TPB + threadIdx.x;
    int key = -1;
    if (i < n) {
        int cx = pn_cell(xyz[3 * i], g.ox, g.vx);
        int cy = pn_cell(xyz[3 * i + 1], g.oy, g.vy);
        int cz = pn_cell(xyz[3 * i + 2], g.oz, g.vz);
        if (cx >= 0 && cx < g.gx && cy >= 0 && cy < g.gy && cz >= 0 && cz < g.gz)
            key = cx * (g.gy * g.gz) + cy * g.gz + cz;
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

__global__ __launch_bounds__(TPB) void k_grid_scatter(const int *__restrict__ keys, int n, const int *__restrict__ cell_start,
                                                      int *__restrict__ cursor, int *__restrict__ tmp_idx) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int key = keys[i];
    if (key < 0) return;
    int pos = cell_start[key] + atomicAdd(&cursor[key], 1);
    tmp_idx[pos] = i;
}

// one thread per cell: order the cell's points by ascending index (insertion sort of a short
// segment), emit the packed point records, dilate the occupancy bits.
__global__ __launch_bounds__(TPB) void k_grid_finalize(GP g, long long G, const float *__restrict__ xyz,
                                                       const int *__restrict__ cell_start, int *__restrict__ tmp_idx,
                                                       float4 *__restrict__ pts, uint32_t *__restrict__ occ, int *info) {
    long long c = (long long)blockIdx.x * TPB + threadIdx.x;
    int n = 0;
    if (c < G) {
        int s = cell_start[c];
        n = cell_start[c + 1] - s;
        if (n > 0) {
            for (int a = 1; a < n; ++a) {
                int v = tmp_idx[s + a];
                int b = a - 1;
                while (b >= 0 && tmp_idx[s + b] > v) { tmp_idx[s + b + 1] = tmp_idx[s + b]; --b; }
                tmp_idx[s + b + 1] = v;
            }
            for (int a = 0; a < n; ++a) {
                int idx = tmp_idx[s + a];
                pts[s + a] = make_float4(xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2], __int_as_float(idx));
            }
            // map_coor2occ dilation (.cu:104-113; the host passes query_size, .cu:342)
            int cz = (int)(c % g.gz), cy = (int)((c / g.gz) % g.gy), cx = (int)(c / ((long long)g.gy * g.gz));
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
    }
    unsigned long long b = __ballot(n > 0);
    if (b) {
        int m = n;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&info[PNERF_GI_N_OCC], __popcll(b));
            atomicMax(&info[PNERF_GI_MAX_CNT], m);
        }
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
    if (L.G <= 0 || L.G >= (1LL << 31) - 64) return PNERF_E_UNSUP;
    if (ws_bytes < L.total) return PNERF_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *b = (char *)ws;
    int *info = (int *)(b + L.info), *cell_start = (int *)(b + L.cell_start), *keys = (int *)(b + L.keys);
    int *cursor = (int *)(b + L.cursor), *tmp_idx = (int *)(b + L.tmp_idx), *scan = (int *)(b + L.scan);
    uint32_t *occ = (uint32_t *)(b + L.occ);
    float4 *pts = (float4 *)(b + L.pts);
    GP g = {gp->ranges[0], gp->ranges[1], gp->ranges[2], gp->vsize[0], gp->vsize[1], gp->vsize[2],
            gp->vdim[0], gp->vdim[1], gp->vdim[2], gp->query_size[0], gp->query_size[1], gp->query_size[2]};

    PnProfScope prof(PNK_GRID, s);
    hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(TPB), 0, s, info);
    if (hipMemsetAsync(cursor, 0, (size_t)L.G * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(occ, 0, (size_t)((L.G + 31) / 32) * sizeof(uint32_t), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n > 0) hipLaunchKernelGGL(k_grid_count, dim3(pn_cdiv(n, TPB)), dim3(TPB), 0, s, g, d_xyz, n, keys, cursor, info);
    hipLaunchKernelGGL(k_grid_cell0, dim3(1), dim3(1), 0, s, keys, n, info);
    int rc = pn_exclusive_scan_i32(cursor, cell_start, L.G, scan, s);
    if (rc) return rc;
    if (hipMemsetAsync(cursor, 0, (size_t)L.G * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n > 0) hipLaunchKernelGGL(k_grid_scatter, dim3(pn_cdiv(n, TPB)), dim3(TPB), 0, s, keys, n, cell_start, cursor, tmp_idx);
    hipLaunchKernelGGL(k_grid_finalize, dim3(pn_cdiv(L.G, TPB)), dim3(TPB), 0, s, g, L.G, d_xyz, cell_start, tmp_idx, pts, occ, info);
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

extern "C" int pnerf_version(void) { return 1000; }
extern "C" const char *pnerf_arch(void) { return "gfx950"; }
