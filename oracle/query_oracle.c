/*
 * oracle/query_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's world-coordinate neural-point
 * query, executed in the CANONICAL SERIAL ORDER defined in SURVEY.md section 8c:
 * every reference kernel behaves as if its threads ran one after another in
 * ascending global thread index.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared library.
 *
 * Reference followed (all under /root/reference):
 *   models/neural_points/cuda/query_worldcoords.cu
 *     claim_occ                      :18-78
 *     map_coor2occ                   :80-115
 *     fill_occ2pnts                  :117-162   (incl. the `voxel_idx > 0` test at :147)
 *     mask_raypos                    :165-189
 *     get_shadingloc                 :192-214
 *     query_neigh_along_ray_layered  :217-302   (K-buffer sized K, as the pycuda twin does)
 *     host orchestration             :305-433   (the ATen compactions at :381-391 and :425-429)
 *
 * Parity pin: the reference ships no tests or golden vectors for this path
 * (SURVEY.md section 4).  This restatement is pinned instead against the
 * reference's OWN kernel source compiled for the CPU and run serially
 * (oracle/_ref, built by oracle/Makefile from the .cu where it lies); see
 * tests/test_oracle_vs_ref.py.
 *
 * Arithmetic: fp32, no FMA contraction (build with -ffp-contract=off), IEEE
 * division.  Overflow of max_o or P is where the reference switches to a
 * wall-clock seeded curand reservoir (parity undefined); here it is detected and
 * reported in info[] and the overflowing item is dropped.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* info[] slots */
enum { INFO_N_OCC = 0, INFO_MAX_CNT = 1, INFO_OVF_MAXO = 2, INFO_OVF_P = 3,
       INFO_R1 = 4, INFO_R2 = 5, INFO_NSEL = 6, INFO_NVALID_NEIGH = 7, INFO_LEN = 8 };

static inline int cell_of(float p, float shift, float vs) {
    /* query_worldcoords.cu:38-40: (int) floor((p - shift) / vsize) in fp32 */
    float q = (p - shift) / vs;
    return (int)floorf(q);
}

typedef struct {
    int gx, gy, gz;
    long long G;
    int *coor_occ;      /* [G] dilated occupancy, 0/1            (.cu:314) */
    int *coor_2_occ;    /* [G] cell -> occupied-voxel id or -1   (.cu:318,337) */
    int *occ_2_coor;    /* [max_o*3]                             (.cu:316) */
    int *occ_numpnts;   /* [max_o]                               (.cu:317) */
    int *occ_2_pnts;    /* [max_o*P]                             (.cu:315) */
    int n_occ;
} grid_t;

static void grid_free(grid_t *g) {
    free(g->coor_occ); free(g->coor_2_occ); free(g->occ_2_coor);
    free(g->occ_numpnts); free(g->occ_2_pnts);
}

/* .cu:308-365 in serial order.  Returns 0, or -1 on allocation failure. */
static int grid_build(grid_t *g, const float *xyz, int n_actual,
                      const int *query_size, const int *vdim, int max_o, int P,
                      const float *ranges, const float *vsize, int *info)
{
    g->gx = vdim[0]; g->gy = vdim[1]; g->gz = vdim[2];
    g->G = (long long)g->gx * g->gy * g->gz;
    g->coor_occ = (int *)calloc((size_t)g->G, sizeof(int));
    g->coor_2_occ = (int *)malloc((size_t)g->G * sizeof(int));
    g->occ_2_coor = (int *)malloc((size_t)max_o * 3 * sizeof(int));
    g->occ_numpnts = (int *)calloc((size_t)max_o, sizeof(int));
    g->occ_2_pnts = (int *)malloc((size_t)max_o * (size_t)P * sizeof(int));
    if (!g->coor_occ || !g->coor_2_occ || !g->occ_2_coor || !g->occ_numpnts || !g->occ_2_pnts)
        return -1;
    memset(g->coor_2_occ, 0xff, (size_t)g->G * sizeof(int));
    memset(g->occ_2_coor, 0xff, (size_t)max_o * 3 * sizeof(int));
    memset(g->occ_2_pnts, 0xff, (size_t)max_o * (size_t)P * sizeof(int));
    const int gy = g->gy, gz = g->gz;
    int occ_idx = 0;

    /* claim_occ (.cu:18-78): first point (ascending index) to land in a cell claims it */
    for (int i = 0; i < n_actual; ++i) {
        int cx = cell_of(xyz[3 * i], ranges[0], vsize[0]);
        int cy = cell_of(xyz[3 * i + 1], ranges[1], vsize[1]);
        int cz = cell_of(xyz[3 * i + 2], ranges[2], vsize[2]);
        if (cx < 0 || cx >= g->gx || cy < 0 || cy >= gy || cz < 0 || cz >= gz) continue;
        long long lin = (long long)cx * (gy * gz) + (long long)cy * gz + cz;
        if (g->coor_2_occ[lin] == -1) {
            g->coor_2_occ[lin] = 0;
            int tmp = occ_idx++;
            if (tmp < max_o) {
                g->occ_2_coor[3 * tmp] = cx; g->occ_2_coor[3 * tmp + 1] = cy; g->occ_2_coor[3 * tmp + 2] = cz;
            } else {
                info[INFO_OVF_MAXO] = 1;   /* .cu:64-73 curand path: parity undefined */
            }
        }
    }
    g->n_occ = occ_idx;
    info[INFO_N_OCC] = occ_idx;
    /* host refill (.cu:337) */
    memset(g->coor_2_occ, 0xff, (size_t)g->G * sizeof(int));

    /* map_coor2occ (.cu:80-115); the host passes query_size as `kernel_size` (.cu:342) */
    for (int id = 0; id < max_o; ++id) {
        if (!(id < occ_idx)) break;
        int cx = g->occ_2_coor[3 * id];
        if (cx < 0) continue;
        int cy = g->occ_2_coor[3 * id + 1], cz = g->occ_2_coor[3 * id + 2];
        g->coor_2_occ[(long long)cx * (gy * gz) + (long long)cy * gz + cz] = id;
        int x0 = cx - query_size[0] / 2; if (x0 < 0) x0 = 0;
        int x1 = cx + (query_size[0] + 1) / 2; if (x1 > g->gx) x1 = g->gx;
        int y0 = cy - query_size[1] / 2; if (y0 < 0) y0 = 0;
        int y1 = cy + (query_size[1] + 1) / 2; if (y1 > gy) y1 = gy;
        int z0 = cz - query_size[2] / 2; if (z0 < 0) z0 = 0;
        int z1 = cz + (query_size[2] + 1) / 2; if (z1 > gz) z1 = gz;
        for (int x = x0; x < x1; ++x)
            for (int y = y0; y < y1; ++y)
                for (int z = z0; z < z1; ++z)
                    g->coor_occ[(long long)x * (gy * gz) + (long long)y * gz + z] = 1;
    }

    /* fill_occ2pnts (.cu:117-162) */
    int max_cnt = 0;
    for (int i = 0; i < n_actual; ++i) {
        int cx = cell_of(xyz[3 * i], ranges[0], vsize[0]);
        int cy = cell_of(xyz[3 * i + 1], ranges[1], vsize[1]);
        int cz = cell_of(xyz[3 * i + 2], ranges[2], vsize[2]);
        if (cx < 0 || cx >= g->gx || cy < 0 || cy >= gy || cz < 0 || cz >= gz) continue;
        int v = g->coor_2_occ[(long long)cx * (gy * gz) + (long long)cy * gz + cz];
        if (v > 0) {                               /* .cu:147 -- voxel id 0 is skipped */
            int tmp = g->occ_numpnts[v]++;
            if (tmp < P) g->occ_2_pnts[(size_t)v * P + tmp] = i;
            else info[INFO_OVF_P] = 1;             /* .cu:152-158 curand path: parity undefined */
            if (tmp + 1 > max_cnt) max_cnt = tmp + 1;
        }
    }
    info[INFO_MAX_CNT] = max_cnt;
    return 0;
}

/* query_neigh_along_ray_layered (.cu:239-301) for ONE shading sample */
static int query_one(const grid_t *g, const float *xyz, const float *c, const int *kernel_size,
                     int max_o, int P, int K, float radius2, const float *ranges,
                     const float *vsize, int *out /*[K], preset -1*/, float *buf /*[K]*/)
{
    (void)max_o;
    const int gx = g->gx, gy = g->gy, gz = g->gz;
    int fx = cell_of(c[0], ranges[0], vsize[0]);
    int fy = cell_of(c[1], ranges[1], vsize[1]);
    int fz = cell_of(c[2], ranges[2], vsize[2]);
    int kid = 0, far_ind = 0;
    float far2 = 0.0f;
    int nlayer = (kernel_size[0] + 1) / 2;
    for (int layer = 0; layer < nlayer; ++layer) {
        int xlo = -fx > -layer ? -fx : -layer, xhi = gx - fx < layer + 1 ? gx - fx : layer + 1;
        for (int x = xlo; x < xhi; ++x) {
            int ylo = -fy > -layer ? -fy : -layer, yhi = gy - fy < layer + 1 ? gy - fy : layer + 1;
            for (int y = ylo; y < yhi; ++y) {
                int zlo = -fz > -layer ? -fz : -layer, zhi = gz - fz < layer + 1 ? gz - fz : layer + 1;
                for (int z = zlo; z < zhi; ++z) {
                    int ax = abs(x), ay = abs(y), az = abs(z);
                    int m = ax > ay ? ax : ay; if (az > m) m = az;
                    if (m != layer) continue;
                    long long lin = (long long)(fx + x) * (gy * gz) + (long long)(fy + y) * gz + (fz + z);
                    int v = g->coor_2_occ[lin];
                    if (v < 0) continue;
                    int n = g->occ_numpnts[v] < P ? g->occ_numpnts[v] : P;
                    for (int gi = 0; gi < n; ++gi) {
                        int pidx = g->occ_2_pnts[(size_t)v * P + gi];
                        float xv = xyz[3 * pidx] - c[0];
                        float yv = xyz[3 * pidx + 1] - c[1];
                        float zv = xyz[3 * pidx + 2] - c[2];
                        float d2 = xv * xv + yv * yv + zv * zv;     /* left-to-right, no FMA */
                        if (radius2 == 0.0f || d2 <= radius2) {
                            if (kid++ < K) {
                                out[kid - 1] = pidx; buf[kid - 1] = d2;
                                if (d2 > far2) { far2 = d2; far_ind = kid - 1; }
                            } else if (d2 < far2) {
                                out[far_ind] = pidx; buf[far_ind] = d2; far2 = d2;
                                for (int j = 0; j < K; ++j)
                                    if (buf[j] > far2) { far2 = buf[j]; far_ind = j; }
                            }
                        }
                    }
                }
            }
        }
        if (kid >= K) break;
    }
    return kid;
}

/*
 * The whole native op (query_worldcoords.cpp:34-82 -> .cu:305-433).
 * Outputs are sized for the worst case R''=R; the first *out_R2 rays are valid.
 *   sample_pidx [R*SR*K] i32, sample_loc [R*SR*3] f32, ray_mask [R] i8.
 * nthreads>1 parallelises the per-sample stages only (their results do not
 * depend on execution order); the grid build always runs serially.
 */
int pnerf_oracle_query(const float *raypos, const float *xyz, int N, int n_actual,
                       const int *kernel_size, const int *query_size, int SR, int K,
                       int R, int D, const int *vdim, int max_o, int P, float radius_limit,
                       const float *ranges, const float *vsize, int nthreads,
                       int *sample_pidx, float *sample_loc, signed char *ray_mask,
                       int *out_R2, int *info)
{
    (void)N;
    if (K <= 0 || K > 64 || SR <= 0 || D <= 0 || P <= 0 || max_o <= 0) return -2;
    memset(info, 0, INFO_LEN * sizeof(int));
    grid_t g;
    if (grid_build(&g, xyz, n_actual, query_size, vdim, max_o, P, ranges, vsize, info) != 0) {
        grid_free(&g);
        return -1;
    }
    const int gy = g.gy, gz = g.gz;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif

    /* mask_raypos (.cu:165-189) + ray_mask = max_D > 0 (.cu:381) */
    int *mask = (int *)calloc((size_t)R * D, sizeof(int));
    int *keep1 = (int *)calloc((size_t)R, sizeof(int));
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        int any = 0;
        for (int d = 0; d < D; ++d) {
            const float *p = raypos + ((size_t)r * D + d) * 3;
            int cx = cell_of(p[0], ranges[0], vsize[0]);
            int cy = cell_of(p[1], ranges[1], vsize[1]);
            int cz = cell_of(p[2], ranges[2], vsize[2]);
            if (cx >= 0 && cx < g.gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz) {
                int m = g.coor_occ[(long long)cx * (gy * gz) + (long long)cy * gz + cz];
                mask[(size_t)r * D + d] = m;
                any |= (m > 0);
            }
        }
        keep1[r] = any;
    }
    /* compaction #1 (.cu:382-391) */
    int *r1_to_r = (int *)malloc((size_t)(R > 0 ? R : 1) * sizeof(int));
    int R1 = 0;
    for (int r = 0; r < R; ++r) if (keep1[r]) r1_to_r[R1++] = r;
    info[INFO_R1] = R1;

    float *loc1 = (float *)calloc((size_t)(R1 > 0 ? R1 : 1) * SR * 3, sizeof(float));
    int *lmask1 = (int *)calloc((size_t)(R1 > 0 ? R1 : 1) * SR, sizeof(int));
    int *pidx1 = (int *)malloc((size_t)(R1 > 0 ? R1 : 1) * SR * K * sizeof(int));
    memset(pidx1, 0xff, (size_t)(R1 > 0 ? R1 : 1) * SR * K * sizeof(int));
    int *valid1 = (int *)calloc((size_t)(R1 > 0 ? R1 : 1), sizeof(int));
    long long nsel = 0, nneigh = 0;

    /* cumsum / slot assignment (.cu:389-390) + get_shadingloc (.cu:192-214) */
#pragma omp parallel for schedule(static) reduction(+:nsel)
    for (int r1 = 0; r1 < R1; ++r1) {
        int r = r1_to_r[r1];
        int cum = 0;
        for (int d = 0; d < D; ++d) {
            int m = mask[(size_t)r * D + d];
            cum += m;
            int slot = m * cum * (cum <= SR ? 1 : 0) - 1;
            if (slot >= 0) {
                const float *p = raypos + ((size_t)r * D + d) * 3;
                float *q = loc1 + ((size_t)r1 * SR + slot) * 3;
                q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
                lmask1[(size_t)r1 * SR + slot] = 1;
                nsel++;
            }
        }
    }
    info[INFO_NSEL] = (int)nsel;

    /* query_neigh_along_ray_layered (.cu:217-302) */
    const float radius2 = radius_limit * radius_limit;      /* .cu:410, fp32 */
#pragma omp parallel for schedule(dynamic, 16) reduction(+:nneigh)
    for (int r1 = 0; r1 < R1; ++r1) {
        float buf[64];
        int any = 0;
        for (int s = 0; s < SR; ++s) {
            if (lmask1[(size_t)r1 * SR + s] <= 0) continue;
            int kid = query_one(&g, xyz, loc1 + ((size_t)r1 * SR + s) * 3, kernel_size, max_o, P, K,
                                radius2, ranges, vsize, pidx1 + ((size_t)r1 * SR + s) * K, buf);
            if (kid > 0) any = 1;
            nneigh += kid < K ? kid : K;
        }
        valid1[r1] = any;
    }
    info[INFO_NVALID_NEIGH] = (int)(nneigh > 2147483647LL ? 2147483647LL : nneigh);

    /* compaction #2 (.cu:425-429) */
    memset(ray_mask, 0, (size_t)R);
    int R2 = 0;
    for (int r1 = 0; r1 < R1; ++r1) {
        if (!valid1[r1]) continue;
        ray_mask[r1_to_r[r1]] = 1;
        memcpy(sample_pidx + (size_t)R2 * SR * K, pidx1 + (size_t)r1 * SR * K, (size_t)SR * K * sizeof(int));
        memcpy(sample_loc + (size_t)R2 * SR * 3, loc1 + (size_t)r1 * SR * 3, (size_t)SR * 3 * sizeof(float));
        R2++;
    }
    info[INFO_R2] = R2;
    *out_R2 = R2;

    free(mask); free(keep1); free(r1_to_r); free(loc1); free(lmask1); free(pidx1); free(valid1);
    grid_free(&g);
    return 0;
}

/* Brute-force cross-check used by the property tests: all points within radius of c,
 * written as (d2, idx) pairs sorted by nothing -- the test sorts.  Returns the count. */
int pnerf_oracle_bruteforce(const float *xyz, int n, const float *c, float radius_limit,
                            int cap, int *out_idx, float *out_d2)
{
    const float r2 = radius_limit * radius_limit;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        float xv = xyz[3 * i] - c[0], yv = xyz[3 * i + 1] - c[1], zv = xyz[3 * i + 2] - c[2];
        float d2 = xv * xv + yv * yv + zv * zv;
        if (d2 <= r2) {
            if (cnt < cap) { out_idx[cnt] = i; out_d2[cnt] = d2; }
            cnt++;
        }
    }
    return cnt;
}
