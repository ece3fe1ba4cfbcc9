// query.hip -- ray probe + shading-sample selection + radius-limited layered K-neighbor query.
//
// Replaces mask_raypos, the two ATen compactions, get_shadingloc and
// query_neigh_along_ray_layered (models/neural_points/cuda/query_worldcoords.cu:165-302, host
// :367-431).  Differences in HOW (results are bit-identical to the reference's canonical serial
// execution):
//   * k_probe: one 64-lane wavefront per ray walks the D depth samples in 64-wide strides; the
//     occupancy test is one bit in a ~1 MB L2-resident field; __ballot + popcount gives each hit its
//     slot, so the [R,D,3] sample array, the [R,D] mask, its cumsum and both masked_select copies of
//     the reference never exist, and the walk stops as soon as SR samples are found.
//   * k_neighbors: one thread per selected sample, the reference's exact sequential top-K insertion
//     (needed for bit-exact slot order), but a cell lookup is one 16-byte record of the brick map
//     (grid.hip; L2-resident) and candidates come from contiguous float4 records, four in flight
//     per lane (one 16 B load per candidate instead of cell->occ->count->pidx->xyz chains), the
//     K-buffer lives in registers, and all 64 lanes of a wave belong to the same ray.
//     Measured and dropped in round 2 (profiles/r02_neighbors_variants.json): one wavefront per
//     64 samples staging the candidates' float4 records in LDS (global_load_lds, 32 KB per wave)
//     ran 2.2x slower, staging only their indices in LDS (flat candidate sequence, 4 loads in
//     flight) ran 1.0x / 2.1x / 1.8x slower at configs[1] / [3] / [4]: the kernel is bound by its
//     dependent steps per wave, not by bytes, and LDS per wave costs the occupancy that hides them.
//   * no device->host sync anywhere: ray compaction is replaced by dense [R,...] outputs plus a
//     device-built work list of the valid samples.
// Compiled with -ffp-contract=off (cell arithmetic and squared distances must round like the
// reference: left-to-right fp32, no FMA).
#include "pn_common.h"

#ifndef PN_NB_BATCH
#define PN_NB_BATCH 4
#endif
namespace {
constexpr int TPB = 256;

struct RayGen {            // how sample positions are produced
    const float *raypos;   // [R,D,3] or null
    const float *raydir;   // [R,3]
    const float *mid;      // [D] mid depths (jitter==0) or base segment lengths (jitter>0)
    float cx, cy, cz;      // campos
    float near_d, jitter;
    unsigned long long seed;
};

__device__ __forceinline__ float pn_uniform(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ bool pn_occupied(const PnGridDev &g, float x, float y, float z) {
    int cx = pn_cell(x, g.ox, g.vx), cy = pn_cell(y, g.oy, g.vy), cz = pn_cell(z, g.oz, g.vz);
    if (cx < 0 || cx >= g.gx || cy < 0 || cy >= g.gy || cz < 0 || cz >= g.gz) return false;
    int lin = cx * (g.gy * g.gz) + cy * g.gz + cz;
    return (g.occ[lin >> 5] >> (lin & 31)) & 1u;
}

__global__ void k_debug_uniform(unsigned long long seed, unsigned long long first, long long n, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = pn_uniform(seed, first + i);
}

// one wavefront per ray
template <bool FROM_RAYPOS, bool JITTER>
__global__ __launch_bounds__(TPB) void k_probe(PnGridDev g, RayGen rg, int R, int D, int SR,
                                               float *__restrict__ sample_loc, int *__restrict__ sel_cnt) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (r >= R) return;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (!FROM_RAYPOS) { dx = rg.raydir[3 * r]; dy = rg.raydir[3 * r + 1]; dz = rg.raydir[3 * r + 2]; }
    float *out = sample_loc + (size_t)r * SR * 3;
    int count = 0;
    double carry = 0.0;  // JITTER: running sum of segment lengths
    for (int base = 0; base < D && count < SR; base += 64) {
        const int d = base + lane;
        float px = 0.f, py = 0.f, pz = 0.f;
        bool hit = false;
        if (JITTER) {
            float seg = 0.f;
            if (d < D) seg = rg.mid[d] * (1.0f + rg.jitter * (pn_uniform(rg.seed, (unsigned long long)r * D + d) - 0.5f));
            // end points = near + running sum of the segment lengths, accumulated in double and rounded to fp32 per element: bit
            // for bit torch.cumsum of the reference's CPU path (diff_ray_marching.py:376-383; ATen's CPU cumsum accumulates floats
            // in double), so that with the same uniforms the jittered samples are the reference's.  The double sums are EXACT
            // (D <= a few thousand fp32 terms within a factor 1.35 of each other need 24 + 12 + 1 < 53 significand bits), so the
            // order of the additions is free: a 6-step wave scan gives the sequential sum's bits (tests: the jitter parity cases of
            // tests/test_gpu_query.py; the 64-step serial chain this replaces cost 0.34 ms per batch).
            double run = (double)seg;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const double up = __shfl_up(run, dlt, 64);
                if (lane >= dlt) run += up;
            }
            run += carry;
            const float inc = (float)run, prev = (float)(run - (double)seg);
            carry = __shfl(run, 63, 64);
            const float e1 = rg.near_d + inc, e0 = rg.near_d + prev;
            if (d < D) {
                float t = (e0 + e1) * 0.5f;
                px = rg.cx + dx * t; py = rg.cy + dy * t; pz = rg.cz + dz * t;
                hit = pn_occupied(g, px, py, pz);
            }
        } else if (d < D) {
            if (FROM_RAYPOS) {
                const float *p = rg.raypos + ((size_t)r * D + d) * 3;
                px = p[0]; py = p[1]; pz = p[2];
            } else {
                // raypos = campos + raydir * mid: separate multiply and add (diff_ray_marching.py:385)
                float t = rg.mid[d];
                px = rg.cx + dx * t; py = rg.cy + dy * t; pz = rg.cz + dz * t;
            }
            hit = pn_occupied(g, px, py, pz);
        }
        unsigned long long b = __ballot(hit);
        int slot = count + __popcll(b & ((1ull << lane) - 1ull));
        if (hit && slot < SR) { out[3 * slot] = px; out[3 * slot + 1] = py; out[3 * slot + 2] = pz; }
        count += __popcll(b);
    }
    if (count > SR) count = SR;
    for (int s = count + lane; s < SR; s += 64) { out[3 * s] = 0.f; out[3 * s + 1] = 0.f; out[3 * s + 2] = 0.f; }
    if (lane == 0) sel_cnt[r] = count;
}

// one thread per (ray, slot)
// One thread per SELECTED sample: sel_off = exclusive scan of the rays' sample counts, thread t finds its ray by bisection (17 steps in
// an L2-resident table) and its slot s = t - sel_off[r].  The dense [R, SR] outputs are pre-filled (-1 / 0) by memsets, so the 70 % of
// the rays that miss the scene and the unused tail slots of the others cost no lanes (the dense launch kept ~25 % of its lanes busy).
template <int KMAX>
__global__ __launch_bounds__(TPB) void k_neighbors(PnGridDev g, int ks0, float radius2, int R, int SR, int K,
                                                   const float *__restrict__ sample_loc, const int *__restrict__ sel_off,
                                                   int *__restrict__ sample_pidx, int *__restrict__ sample_nn) {
    const long long total_sel = sel_off[R];
    for (long long t = (long long)blockIdx.x * TPB + threadIdx.x; t < total_sel; t += (long long)gridDim.x * TPB) {
    int lo = 0, hi = R;                                  // invariant: sel_off[lo] <= t < sel_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sel_off[mid] <= (int)t) lo = mid; else hi = mid;
    }
    const int r = lo, s = (int)t - sel_off[lo];
    const long long index = (long long)r * SR + s;
    int out[KMAX];
    float buf[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { out[j] = -1; buf[j] = 0.f; }
    int kid = 0;
    {
        const float cx = sample_loc[index * 3], cy = sample_loc[index * 3 + 1], cz = sample_loc[index * 3 + 2];
        const int fx = pn_cell(cx, g.ox, g.vx), fy = pn_cell(cy, g.oy, g.vy), fz = pn_cell(cz, g.oz, g.vz);
        const int cell0 = g.info[PNERF_GI_CELL0];
        const int *ostart = pn_grid_ostart(g);
        int far_ind = 0;
        float far2 = 0.f;
        const int nlayer = (ks0 + 1) / 2;
        for (int layer = 0; layer < nlayer; ++layer) {
            for (int x = max(-fx, -layer); x < min(g.gx - fx, layer + 1); ++x) {
                for (int y = max(-fy, -layer); y < min(g.gy - fy, layer + 1); ++y) {
                    for (int z = max(-fz, -layer); z < min(g.gz - fz, layer + 1); ++z) {
                        if (max(abs(z), max(abs(x), abs(y))) != layer) continue;
                        int st = 0;                            // reference: voxel id 0 holds no points (.cu:147)
                        const int n = pn_cell_points(g, ostart, cell0, fx + x, fy + y, fz + z, st);
                        // the insertion is sequential, the loads are not: PN_NB_BATCH records of the cell in flight per lane
                        for (int g0 = 0; g0 < n; g0 += PN_NB_BATCH) {
                            float4 pb[PN_NB_BATCH];
#pragma unroll
                            for (int u = 0; u < PN_NB_BATCH; ++u) pb[u] = g.pts[st + min(g0 + u, n - 1)];
#pragma unroll
                            for (int u = 0; u < PN_NB_BATCH; ++u) {
                                if (g0 + u >= n) break;
                                const float4 p = pb[u];
                                const float xv = p.x - cx, yv = p.y - cy, zv = p.z - cz;
                                const float d2 = xv * xv + yv * yv + zv * zv;    // contract=off: ((xx+yy)+zz)
                                if (radius2 == 0.f || d2 <= radius2) {
                                    const int pid = __float_as_int(p.w);
                                    if (kid < K) {
#pragma unroll
                                        for (int j = 0; j < KMAX; ++j) if (j == kid) { out[j] = pid; buf[j] = d2; }
                                        if (d2 > far2) { far2 = d2; far_ind = kid; }
                                    } else if (d2 < far2) {
#pragma unroll
                                        for (int j = 0; j < KMAX; ++j) if (j == far_ind) { out[j] = pid; buf[j] = d2; }
                                        far2 = d2;
#pragma unroll
                                        for (int j = 0; j < KMAX; ++j) if (j < K && buf[j] > far2) { far2 = buf[j]; far_ind = j; }
                                    }
                                    ++kid;
                                }
                            }
                        }
                    }
                }
            }
            if (kid >= K) break;
        }
    }
    int *o = sample_pidx + index * K;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) if (j < K) o[j] = out[j];
    sample_nn[index] = min(kid, K);
    }
}

// does the ray have any sample with a neighbor?  + global tallies.  One wavefront walks RAYS_PER_WAVE rays and the
// block folds its tallies in LDS, so the three counters see 3 atomics per 64 rays instead of 3 per ray (65 536 rays
// hammering one address cost 1.6 ms).
constexpr int RAYS_PER_WAVE = 16;
__global__ __launch_bounds__(TPB) void k_ray_hit(int R, int SR, const int *__restrict__ sel_cnt, const int *__restrict__ sample_nn,
                                                 int *__restrict__ ray_hit, int *__restrict__ counters) {
    __shared__ int tally[3];
    if (threadIdx.x < 3) tally[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * RAYS_PER_WAVE;
    int hits = 0, sel = 0, nbs = 0;
    for (int r = r0; r < r0 + RAYS_PER_WAVE && r < R; ++r) {
        int nb = 0;
        for (int s = lane; s < SR; s += 64) nb += sample_nn[(size_t)r * SR + s];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nb += __shfl_xor(nb, off, 64);
        if (lane == 0) {
            ray_hit[r] = nb > 0;
            hits += nb > 0; sel += sel_cnt[r]; nbs += nb;
        }
    }
    if (lane == 0) { atomicAdd(&tally[0], hits); atomicAdd(&tally[1], sel); atomicAdd(&tally[2], nbs); }
    __syncthreads();
    if (threadIdx.x < 3 && tally[threadIdx.x]) atomicAdd(&counters[1 + threadIdx.x], tally[threadIdx.x]);
}
}  // namespace

extern "C" int pnerf_debug_uniform(uint64_t seed, uint64_t first, int64_t n, float *d_out, void *stream) {
    if (!d_out || n < 0) return PNERF_E_INVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_debug_uniform, dim3(256), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, (unsigned long long)first, (long long)n, d_out);
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t pnerf_query_workspace_bytes(int R, int SR) {
    return pn_align((size_t)(R > 0 ? R : 1) * sizeof(int)) + pn_align((size_t)(R > 0 ? R + 1 : 2) * sizeof(int)) +
           pn_align(pn_scan_scratch_ints((long long)R * SR) * sizeof(int));
}

extern "C" int pnerf_query(const pnerf_grid_params *gp, const void *d_grid_ws, const float *d_raypos,
                           const float *campos3_host, const float *d_raydir, const float *d_mid,
                           float near_depth, float far_depth, float jitter, uint64_t seed,
                           int R, int D, int SR, int K,
                           float *d_sample_loc, int32_t *d_sample_pidx, int32_t *d_sample_nn,
                           int32_t *d_ray_hit, int32_t *d_valid_list, int32_t *d_counters,
                           void *d_query_ws, size_t ws_bytes, void *stream) {
    (void)far_depth;
    if (!gp || !d_grid_ws || R < 0 || D <= 0 || SR <= 0 || K <= 0 || K > PNERF_MAX_K) return PNERF_E_INVAL;
    if (!d_sample_loc || !d_sample_pidx || !d_sample_nn || !d_ray_hit || !d_valid_list || !d_counters || !d_query_ws) return PNERF_E_INVAL;
    if (!d_raypos && (!campos3_host || !d_raydir || !d_mid)) return PNERF_E_INVAL;
    if (ws_bytes < pnerf_query_workspace_bytes(R, SR)) return PNERF_E_WS;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_counters, 0, 8 * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (R == 0) return 0;
    PnCarver cv(d_query_ws, ws_bytes);
    int *sel_cnt = cv.take<int>(R);
    int *sel_off = cv.take<int>(R + 1);
    int *scan = cv.take<int>(pn_scan_scratch_ints((long long)R * SR));
    PnGridDev g = pn_grid_dev(gp, d_grid_ws, 0);
    RayGen rg;
    rg.raypos = d_raypos; rg.raydir = d_raydir; rg.mid = d_mid;
    rg.cx = campos3_host ? campos3_host[0] : 0.f; rg.cy = campos3_host ? campos3_host[1] : 0.f; rg.cz = campos3_host ? campos3_host[2] : 0.f;
    rg.near_d = near_depth; rg.jitter = jitter; rg.seed = seed;
    const int wb = pn_cdiv(R, TPB / 64);
    { PnProfScope prof(PNK_PROBE, s);
    if (d_raypos) hipLaunchKernelGGL((k_probe<true, false>), dim3(wb), dim3(TPB), 0, s, g, rg, R, D, SR, d_sample_loc, sel_cnt);
    else if (jitter > 0.f) hipLaunchKernelGGL((k_probe<false, true>), dim3(wb), dim3(TPB), 0, s, g, rg, R, D, SR, d_sample_loc, sel_cnt);
    else hipLaunchKernelGGL((k_probe<false, false>), dim3(wb), dim3(TPB), 0, s, g, rg, R, D, SR, d_sample_loc, sel_cnt); }
    const long long total = (long long)R * SR;
    const float radius2 = gp->radius * gp->radius;     // fp32 product, as .cu:410
    const int nbk = pn_cdiv(total, TPB);
    { PnProfScope prof(PNK_NEIGHBORS, s);
    int rc = pn_exclusive_scan_i32(sel_cnt, sel_off, R, scan, s);
    if (rc) return rc;
    if (hipMemsetAsync(d_sample_pidx, 0xFF, (size_t)total * K * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;      // -1 everywhere
    if (hipMemsetAsync(d_sample_nn, 0, (size_t)total * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const int grid = nbk < 16 * ncu ? nbk : 16 * ncu;
    if (K <= 4) hipLaunchKernelGGL(k_neighbors<4>, dim3(grid), dim3(TPB), 0, s, g, gp->kernel_size[0], radius2, R, SR, K, d_sample_loc, sel_off, d_sample_pidx, d_sample_nn);
    else if (K <= 8) hipLaunchKernelGGL(k_neighbors<8>, dim3(grid), dim3(TPB), 0, s, g, gp->kernel_size[0], radius2, R, SR, K, d_sample_loc, sel_off, d_sample_pidx, d_sample_nn);
    else if (K <= 12) hipLaunchKernelGGL(k_neighbors<12>, dim3(grid), dim3(TPB), 0, s, g, gp->kernel_size[0], radius2, R, SR, K, d_sample_loc, sel_off, d_sample_pidx, d_sample_nn);      // (configs[4]: K = 12)
    else hipLaunchKernelGGL(k_neighbors<16>, dim3(grid), dim3(TPB), 0, s, g, gp->kernel_size[0], radius2, R, SR, K, d_sample_loc, sel_off, d_sample_pidx, d_sample_nn);
    }
    PnProfScope prof(PNK_COMPACT, s);
    hipLaunchKernelGGL(k_ray_hit, dim3(pn_cdiv(R, (TPB / 64) * RAYS_PER_WAVE)), dim3(TPB), 0, s, R, SR, sel_cnt, d_sample_nn, d_ray_hit, d_counters);
    PN_CHECK_LAUNCH();
    return pn_compact_gt0_i32(d_sample_nn, total, d_valid_list, d_counters, scan, s);
}
