// pointinit.hip -- voxel down-sampling of a raw point cloud into neural-point positions: the step that feeds the hot path
// (run/train_ft.py:138-139 -> models/mvs/mvs_utils.py:537-561 construct_vox_points_closest).  The reference does it with
// torch.unique(dim=0) + torch_scatter.scatter_mean / scatter_min (atomics: the centroid sums have no defined order).  Here:
// counting sort of the points by voxel (ascending voxel = torch.unique's lexicographic (x,y,z) order, ascending point index
// inside a voxel), then one thread per occupied voxel: centroid in index order, the member closest to it (ties: lowest index).
// Deterministic; HBM-bound (a few passes over N points + one over vox_res^3 counters).
#include "pn_common.h"

namespace {
struct VoxArgs {
    const float *xyz; long long n;
    float mnx, mny, mnz, vsx, vsy, vsz;
    int rx, ry, rz;
};

__device__ __forceinline__ long long vox_key(const VoxArgs &a, long long i, int &cx, int &cy, int &cz) {
    cx = pn_cell(a.xyz[3 * i], a.mnx, a.vsx); cy = pn_cell(a.xyz[3 * i + 1], a.mny, a.vsy); cz = pn_cell(a.xyz[3 * i + 2], a.mnz, a.vsz);
    if (cx < 0 || cy < 0 || cz < 0 || cx >= a.rx || cy >= a.ry || cz >= a.rz) return -1;
    return ((long long)cx * a.ry + cy) * a.rz + cz;
}

__global__ void k_vox_count(VoxArgs a, int *__restrict__ count, int *__restrict__ n_outside) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int cx, cy, cz;
    const long long k = vox_key(a, i, cx, cy, cz);
    if (k < 0) atomicAdd(n_outside, 1);
    else atomicAdd(&count[k], 1);
}

__global__ void k_vox_scatter(VoxArgs a, const int *__restrict__ start, int *__restrict__ cursor, int *__restrict__ members) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int cx, cy, cz;
    const long long k = vox_key(a, i, cx, cy, cz);
    if (k >= 0) members[start[k] + atomicAdd(&cursor[k], 1)] = (int)i;
}

// one thread per occupied voxel: order its members by index, centroid, closest member
__global__ void k_vox_finalize(VoxArgs a, const int *__restrict__ occ_list, const int *__restrict__ n_occ, const int *__restrict__ start,
                               int *__restrict__ members, float *__restrict__ centroid, int *__restrict__ grid_idx, long long *__restrict__ min_idx) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *n_occ) return;
    const int cell = occ_list[v];
    const int s = start[cell], e = start[cell + 1];
    for (int i = s + 1; i < e; ++i) {             // insertion sort by point index (the scatter order is arbitrary)
        const int x = members[i];
        int j = i - 1;
        while (j >= s && members[j] > x) { members[j + 1] = members[j]; --j; }
        members[j + 1] = x;
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = s; i < e; ++i) { const long long p = members[i]; sx += a.xyz[3 * p]; sy += a.xyz[3 * p + 1]; sz += a.xyz[3 * p + 2]; }
    const float cnt = (float)(e - s);
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    float best = 3.402823466e38f; long long arg = -1;
    for (int i = s; i < e; ++i) {
        const long long p = members[i];
        const float dx = a.xyz[3 * p] - mx, dy = a.xyz[3 * p + 1] - my, dz = a.xyz[3 * p + 2] - mz;
        const float r = sqrtf(dx * dx + dy * dy + dz * dz);
        if (r < best) { best = r; arg = p; }
    }
    centroid[3 * v] = mx; centroid[3 * v + 1] = my; centroid[3 * v + 2] = mz;
    const int cz = cell % a.rz, cy = (cell / a.rz) % a.ry, cx = cell / (a.rz * a.ry);
    grid_idx[3 * v] = cx; grid_idx[3 * v + 1] = cy; grid_idx[3 * v + 2] = cz;
    min_idx[v] = arg;
}

struct VoxLayout { size_t count, start, cursor, members, occ, info, scan, total; long long cells; };
VoxLayout vox_layout(long long n, int rx, int ry, int rz) {
    VoxLayout L;
    L.cells = (long long)rx * ry * rz;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += pn_align(bytes); return r; };
    L.count = take((size_t)(L.cells + 1) * 4); L.start = take((size_t)(L.cells + 2) * 4); L.cursor = take((size_t)(L.cells + 1) * 4);
    L.members = take((size_t)(n > 0 ? n : 1) * 4); L.occ = take((size_t)(n > 0 ? n : 1) * 4); L.info = take(64);
    L.scan = take(pn_scan_scratch_ints(L.cells + 1) * 4);
    L.total = off;
    return L;
}
}  // namespace

extern "C" size_t pnerf_voxel_downsample_workspace_bytes(int64_t n_points, int rx, int ry, int rz) {
    if (n_points < 0 || rx <= 0 || ry <= 0 || rz <= 0 || (long long)rx * ry * rz > 0x7fffff00LL) return 0;
    return vox_layout(n_points, rx, ry, rz).total;
}

extern "C" int pnerf_voxel_downsample(const float *d_xyz, int64_t n_points, const float *space_min3_host, const float *vox_size3_host,
                                      int rx, int ry, int rz, float *d_centroid, int32_t *d_grid_idx, int64_t *d_min_idx, int32_t *d_counts,
                                      void *d_ws, size_t ws_bytes, void *stream) {
    if (!d_xyz || !space_min3_host || !vox_size3_host || !d_centroid || !d_grid_idx || !d_min_idx || !d_counts || !d_ws) return PNERF_E_INVAL;
    if (n_points <= 0 || n_points > 0x7fffff00LL || rx <= 0 || ry <= 0 || rz <= 0 || (long long)rx * ry * rz > 0x7fffff00LL) return PNERF_E_INVAL;
    const VoxLayout L = vox_layout(n_points, rx, ry, rz);
    if (ws_bytes < L.total) return PNERF_E_WS;
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)d_ws;
    int *count = (int *)(ws + L.count), *start = (int *)(ws + L.start), *cursor = (int *)(ws + L.cursor), *members = (int *)(ws + L.members);
    int *occ = (int *)(ws + L.occ), *scan = (int *)(ws + L.scan);
    VoxArgs a;
    a.xyz = d_xyz; a.n = n_points;
    a.mnx = space_min3_host[0]; a.mny = space_min3_host[1]; a.mnz = space_min3_host[2];
    a.vsx = vox_size3_host[0]; a.vsy = vox_size3_host[1]; a.vsz = vox_size3_host[2];
    a.rx = rx; a.ry = ry; a.rz = rz;
    if (hipMemsetAsync(count, 0, (size_t)(L.cells + 1) * 4, s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(cursor, 0, (size_t)(L.cells + 1) * 4, s) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipMemsetAsync(d_counts, 0, 2 * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    PnProfScope prof(PNK_GRID, s);
    const int nb = pn_cdiv(n_points, 256);
    hipLaunchKernelGGL(k_vox_count, dim3(nb), dim3(256), 0, s, a, count, d_counts + 1);
    int rc = pn_exclusive_scan_i32(count, start, L.cells, scan, s);          // start[cells] = points inside the grid
    if (rc) return rc;
    hipLaunchKernelGGL(k_vox_scatter, dim3(nb), dim3(256), 0, s, a, start, cursor, members);
    rc = pn_compact_gt0_i32(count, L.cells, occ, d_counts, scan, s);          // ascending voxel key = lexicographic (x, y, z); d_counts[0] = occupied voxels
    if (rc) return rc;
    hipLaunchKernelGGL(k_vox_finalize, dim3(pn_cdiv(n_points, 128)), dim3(128), 0, s, a, occ, d_counts, start, members, d_centroid, d_grid_idx, (long long *)d_min_idx);
    PN_CHECK_LAUNCH();
    return 0;
}
