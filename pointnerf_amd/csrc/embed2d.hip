// embed2d.hip -- initial per-point appearance from GIVEN 2-D feature maps: project the candidate points into the
// source views, sample every view's image / feature pyramid bilinearly, and the per-view unit directions.  Replaces
//   MvsPointsModel.extract_2d        models/mvs/mvs_points_model.py:198-218
//   homo_warp_nongrid(_occ)          models/mvs/mvs_utils.py:299-315, 333-369   (projection, in-image mask, optional z-buffer test)
//   extract_from_2d_grid             models/mvs/mvs_utils.py:411-421            (F.grid_sample bilinear, zeros padding, align_corners)
//   the "dir" block of query_embedding  models/mvs/mvs_points_model.py:239-251
// The reference compacts the in-image points (masked_select), samples them, and scatters the rows back into a zero tensor; here nothing is
// compacted: pass 1 = one thread per (point, view) (projection, mask, z-buffer atomicMin); pass 2 (round 5) = one workgroup per 64 points: lanes
// are POINTS (neighbouring candidates project to neighbouring texels: a wave's four texel loads of a channel touch a few lines of ONE plane --
// the first form, one thread per output element with lanes = channels, sent every lane to its own plane: 64 lines per load instruction),
// the items (map, chunk of 8 channels) are dealt over the four waves, results go to an LDS tile [64][columns] and leave as whole rows.
// Same arithmetic per element as before (bit-identical outputs).  Column counts whose tile exceeds 64 KB of LDS take the first form.
// HBM: N * (12 + 16 V + 4 (F + 3 V)) B.  fp32 with the reference's operation order, no FMA contraction (built with -ffp-contract=off):
// the in-image mask and the z-buffer cell are threshold decisions on the projected pixel.
#include "pn_common.h"

namespace {
struct Ex2dArgs {
    const float *xyz; long long n;
    int n_views, n_maps, HD, WD, occ;
    float tolerate;
    pnerf_view_desc views[PNERF_EX2D_MAX_VIEWS];
    pnerf_map_desc maps[PNERF_EX2D_MAX_MAPS];
    int feat_cols, color_cols;
};

// order-preserving map of a float onto unsigned integers (atomicMin over depths of either sign)
__device__ __forceinline__ unsigned ord_of(float z) { const unsigned b = __float_as_uint(z); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ord_back(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// row vector [x y z w] @ M^T for a row-major 4 x 4 M: out[j] = sum_k in[k] M[j][k], k ascending
__device__ __forceinline__ void xform4(const float *M, const float in[4], float out[4]) {
    for (int j = 0; j < 4; ++j) out[j] = ((in[0] * M[4 * j] + in[1] * M[4 * j + 1]) + in[2] * M[4 * j + 2]) + in[3] * M[4 * j + 3];
}

// pass 1: proj[v][i] = (u, v, z of the source camera, in-image flag); z-buffer of the occ variant
__global__ void k_ex2d_project(Ex2dArgs a, float4 *__restrict__ proj, unsigned *__restrict__ zbuf) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n * a.n_views) return;
    const int v = (int)(t / a.n);
    const long long i = t - (long long)v * a.n;
    const pnerf_view_desc &V = a.views[v];
    float p[4] = {a.xyz[3 * i], a.xyz[3 * i + 1], a.xyz[3 * i + 2], 1.f};
    if (V.has_w2c) {                                            // [xyz 1] @ c2w^T @ w2c^T (:303), left to right
        float w[4], s[4];
        xform4(V.c2w, p, w);
        xform4(V.w2c, w, s);
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }
    const float qx = p[0] / p[2], qy = p[1] / p[2], qz = p[2] / p[2];
    const float px = (qx * V.intrinsic[0] + qy * V.intrinsic[1]) + qz * V.intrinsic[2];
    const float py = (qx * V.intrinsic[3] + qy * V.intrinsic[4]) + qz * V.intrinsic[5];
    bool in;
    if (a.occ) in = px >= 0.f && py >= 0.f && ceilf(px) <= (float)(a.WD - 1) && ceilf(py) <= (float)(a.HD - 1);     // :343
    else in = px >= 0.f && py >= 0.f && px <= (float)(a.WD - 1) && py <= (float)(a.HD - 1);                           // :308
    proj[t] = make_float4(px, py, p[2], in ? 1.f : 0.f);
    if (a.occ && in) {
        const long long cell = (long long)ceilf(px) * a.HD + (long long)ceilf(py);                                    // :355 (x * HD + y)
        atomicMin(&zbuf[(long long)v * a.WD * a.HD + cell], ord_of(p[2]));
    }
}

// F.grid_sample(mode bilinear, padding zeros, align_corners=True) of one channel plane at the normalised position (gx, gy)
__device__ __forceinline__ float bilinear(const float *__restrict__ plane, int H, int W, float gx, float gy) {
    const float x = (gx + 1.f) * ((float)(W - 1) / 2.f), y = (gy + 1.f) * ((float)(H - 1) / 2.f);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.f - w, n = y - yn, s = 1.f - n;
    const int ix = (int)xw, iy = (int)yn;
    const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
    const float nw = (x0 && y0) ? plane[(long long)iy * W + ix] : 0.f;
    const float ne = (x1 && y0) ? plane[(long long)iy * W + ix + 1] : 0.f;
    const float sw = (x0 && y1) ? plane[(long long)(iy + 1) * W + ix] : 0.f;
    const float se = (x1 && y1) ? plane[(long long)(iy + 1) * W + ix + 1] : 0.f;
    return ((nw * (s * e) + ne * (s * w)) + sw * (n * e)) + se * (n * w);
}

// pass 2: one thread per output element; column c < feat_cols belongs to the feature tensor, the rest to the colours
__global__ void k_ex2d_sample(Ex2dArgs a, const float4 *__restrict__ proj, const unsigned *__restrict__ zbuf, float *__restrict__ feats,
                              float *__restrict__ colors, unsigned char *__restrict__ mask_out) {
    const int cols = a.feat_cols + a.color_cols;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n * cols) return;
    const long long i = t / cols;
    const int c = (int)(t - i * cols);
    const bool is_color = c >= a.feat_cols;
    const int cc = is_color ? c - a.feat_cols : c;
    int m = 0;
    for (; m < a.n_maps; ++m)
        if ((a.maps[m].is_color != 0) == is_color && cc >= a.maps[m].out_col && cc < a.maps[m].out_col + a.maps[m].C) break;
    float val = 0.f;
    if (m < a.n_maps) {
        const pnerf_map_desc &M = a.maps[m];
        const float4 q = proj[(long long)M.view * a.n + i];
        bool in = q.w > 0.f;
        if (in && a.occ) {
            const long long cell = (long long)ceilf(q.x) * a.HD + (long long)ceilf(q.y);
            in = q.z <= ord_back(zbuf[(long long)M.view * a.WD * a.HD + cell]) + a.tolerate;                           // :361
        }
        if (in) {
            const float gx = q.x / (((float)a.WD - 1.f) / 2.f) - 1.f, gy = q.y / (((float)a.HD - 1.f) / 2.f) - 1.f;  // :313-314, :349-350
            val = bilinear((const float *)M.d_map + (long long)(cc - M.out_col) * M.H * M.W, M.H, M.W, gx, gy);
        }
        if (mask_out && cc == M.out_col && M.first_of_view) mask_out[(long long)M.view * a.n + i] = in ? 1 : 0;
    }
    (is_color ? colors[i * a.color_cols + cc] : feats[i * a.feat_cols + cc]) = val;
}

// pass 2, tiled: see the header.  LDS: tile[64][cols + 1] floats (the + 1 keeps the lanes' row-strided writes off one bank)
constexpr int EX_PTS = 64, EX_CH = 8;
__global__ __launch_bounds__(256) void k_ex2d_sample_tiled(Ex2dArgs a, const float4 *__restrict__ proj, const unsigned *__restrict__ zbuf, float *__restrict__ feats,
                                                           float *__restrict__ colors, unsigned char *__restrict__ mask_out) {
    extern __shared__ float ex_tile[];
    const int cols = a.feat_cols + a.color_cols, ld = cols + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long i0 = (long long)blockIdx.x * EX_PTS, i = i0 + lane;
    const int npts = (int)((a.n - i0) < EX_PTS ? (a.n - i0) : EX_PTS);
    for (int e = tid; e < EX_PTS * ld; e += 256) ex_tile[e] = 0.f;            // columns no map writes stay zero
    __syncthreads();
    // items: (map m, channel chunk k) in map order, dealt round-robin over the waves
    int item = 0;
    for (int m = 0; m < a.n_maps; ++m) {
        const pnerf_map_desc &M = a.maps[m];
        const int chunks = (M.C + EX_CH - 1) / EX_CH;
        for (int k = 0; k < chunks; ++k, ++item) {
            if ((item & 3) != wave || lane >= npts) continue;
            const float4 q = proj[(long long)M.view * a.n + i];
            bool in = q.w > 0.f;
            if (in && a.occ) {
                const long long cell = (long long)ceilf(q.x) * a.HD + (long long)ceilf(q.y);
                in = q.z <= ord_back(zbuf[(long long)M.view * a.WD * a.HD + cell]) + a.tolerate;                       // :361
            }
            if (k == 0 && mask_out && M.first_of_view) mask_out[(long long)M.view * a.n + i] = in ? 1 : 0;
            if (!in) continue;
            const float gx = q.x / (((float)a.WD - 1.f) / 2.f) - 1.f, gy = q.y / (((float)a.HD - 1.f) / 2.f) - 1.f;  // :313-314, :349-350
            const int c0 = k * EX_CH, c1 = c0 + EX_CH < M.C ? c0 + EX_CH : M.C;
            float *dst = ex_tile + lane * ld + (M.is_color ? a.feat_cols : 0) + M.out_col;
            for (int c = c0; c < c1; ++c) dst[c] = bilinear((const float *)M.d_map + (long long)c * M.H * M.W, M.H, M.W, gx, gy);
        }
    }
    __syncthreads();
    // the tile's feature rows / colour rows are contiguous in the outputs: whole-row stores
    for (int e = tid; e < npts * a.feat_cols; e += 256) feats[i0 * a.feat_cols + e] = ex_tile[(e / a.feat_cols) * ld + e % a.feat_cols];
    for (int e = tid; e < npts * a.color_cols; e += 256) colors[i0 * a.color_cols + e] = ex_tile[(e / a.color_cols) * ld + a.feat_cols + e % a.color_cols];
}

struct DirArgs {
    const float *xyz; long long n; int n_views;
    float cam_pos[PNERF_EX2D_MAX_VIEWS][3];     // the views' camera centres in the CURRENT camera's frame
    float r1[9], r2[9]; int has_r2;             // d @ r1^T (@ r2^T)
};

// the "dir" block (:241-251): unit vector from each view's centre to the point, rotated
__global__ void k_point_dirs(DirArgs a, float *__restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n * a.n_views) return;
    const long long i = t / a.n_views;
    const int v = (int)(t - i * a.n_views);
    float d[3];
    for (int k = 0; k < 3; ++k) d[k] = a.xyz[3 * i + k] - a.cam_pos[v][k];
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + 1e-6f;
    for (int k = 0; k < 3; ++k) d[k] = d[k] / nrm;
    float w[3];
    for (int j = 0; j < 3; ++j) w[j] = (d[0] * a.r1[3 * j] + d[1] * a.r1[3 * j + 1]) + d[2] * a.r1[3 * j + 2];
    if (a.has_r2) {
        float r[3];
        for (int j = 0; j < 3; ++j) r[j] = (w[0] * a.r2[3 * j] + w[1] * a.r2[3 * j + 1]) + w[2] * a.r2[3 * j + 2];
        for (int j = 0; j < 3; ++j) w[j] = r[j];
    }
    for (int j = 0; j < 3; ++j) out[t * 3 + j] = w[j];
}
}  // namespace

extern "C" size_t pnerf_extract_2d_workspace_bytes(int64_t n_points, int n_views, int HD, int WD, int depth_occ) {
    if (n_points < 0 || n_views < 1 || n_views > PNERF_EX2D_MAX_VIEWS || HD < 1 || WD < 1) return 0;
    size_t b = pn_align((size_t)n_points * n_views * sizeof(float4));
    if (depth_occ) b += pn_align((size_t)n_views * HD * WD * sizeof(unsigned));
    return b + 256;
}

extern "C" int pnerf_extract_2d(const float *d_cam_xyz, int64_t n_points, const pnerf_view_desc *views, int n_views, const pnerf_map_desc *maps,
                                int n_maps, int HD, int WD, int depth_occ, float tolerate, float *d_feats, int feat_cols, float *d_colors,
                                int color_cols, uint8_t *d_mask, void *d_ws, size_t ws_bytes, void *stream) {
    if (!views || !maps || n_views < 1 || n_views > PNERF_EX2D_MAX_VIEWS || n_maps < 1 || n_maps > PNERF_EX2D_MAX_MAPS || HD < 2 || WD < 2 ||
        n_points < 0 || feat_cols < 0 || color_cols < 0 || feat_cols + color_cols == 0)
        return PNERF_E_INVAL;
    if (n_points == 0) return 0;                 // (nothing to write: the output pointers of an empty cloud may be null)
    if ((feat_cols && !d_feats) || (color_cols && !d_colors) || !d_cam_xyz || !d_ws) return PNERF_E_INVAL;
    if (ws_bytes < pnerf_extract_2d_workspace_bytes(n_points, n_views, HD, WD, depth_occ)) return PNERF_E_WS;
    if ((long long)n_points * (feat_cols + color_cols) / 256 >= 0x7fffffffLL) return PNERF_E_UNSUP;
    Ex2dArgs a;
    a.xyz = d_cam_xyz; a.n = n_points; a.n_views = n_views; a.n_maps = n_maps; a.HD = HD; a.WD = WD; a.occ = depth_occ ? 1 : 0;
    a.tolerate = tolerate; a.feat_cols = feat_cols; a.color_cols = color_cols;
    for (int v = 0; v < n_views; ++v) a.views[v] = views[v];
    bool seen[PNERF_EX2D_MAX_VIEWS] = {};
    for (int m = 0; m < n_maps; ++m) {
        a.maps[m] = maps[m];
        const pnerf_map_desc &M = maps[m];
        if (!M.d_map || M.view < 0 || M.view >= n_views || M.C < 1 || M.H < 2 || M.W < 2 || M.out_col < 0 ||
            M.out_col + M.C > (M.is_color ? color_cols : feat_cols)) return PNERF_E_INVAL;
        a.maps[m].first_of_view = seen[M.view] ? 0 : 1;
        seen[M.view] = true;
    }
    hipStream_t s = (hipStream_t)stream;
    PnCarver cv(d_ws, ws_bytes);
    float4 *proj = cv.take<float4>((size_t)n_points * n_views);
    unsigned *zbuf = nullptr;
    if (a.occ) {
        zbuf = cv.take<unsigned>((size_t)n_views * HD * WD);
        if (hipMemsetAsync(zbuf, 0xff, (size_t)n_views * HD * WD * sizeof(unsigned), s) != hipSuccess) return PNERF_E_LAUNCH;
    }
    if (d_mask && hipMemsetAsync(d_mask, 0, (size_t)n_views * n_points, s) != hipSuccess) return PNERF_E_LAUNCH;
    hipLaunchKernelGGL(k_ex2d_project, dim3(pn_cdiv((long long)n_points * n_views, 256)), dim3(256), 0, s, a, proj, zbuf);
    PN_CHECK_LAUNCH();
    // descriptors whose output columns overlap: the first-form kernel lets the FIRST matching map win per column; the tiled one would let
    // whichever wave writes the LDS cell last win (a race) -- such calls take the first form
    bool overlap = false;
    for (int m = 0; m < n_maps && !overlap; ++m)
        for (int k = 0; k < m; ++k)
            if (maps[m].is_color == maps[k].is_color && maps[m].out_col < maps[k].out_col + maps[k].C && maps[k].out_col < maps[m].out_col + maps[m].C) { overlap = true; break; }
    const size_t tile_bytes = (size_t)EX_PTS * (feat_cols + color_cols + 1) * sizeof(float);
    if (tile_bytes <= 64 * 1024 && !overlap) {
        if (hipFuncSetAttribute((const void *)k_ex2d_sample_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes) != hipSuccess) return PNERF_E_LAUNCH;
        hipLaunchKernelGGL(k_ex2d_sample_tiled, dim3(pn_cdiv((long long)n_points, EX_PTS)), dim3(256), tile_bytes, s, a, proj, zbuf, d_feats, d_colors, d_mask);
    } else {
        hipLaunchKernelGGL(k_ex2d_sample, dim3(pn_cdiv((long long)n_points * (feat_cols + color_cols), 256)), dim3(256), 0, s, a, proj, zbuf, d_feats,
                           d_colors, d_mask);
    }
    PN_CHECK_LAUNCH();
    return 0;
}

extern "C" int pnerf_point_dirs(const float *d_cam_xyz, int64_t n_points, const float *cam_pos_cam_host, int n_views, const float *rot1_host9,
                                const float *rot2_host9, float *d_dirs, void *stream) {
    if (!cam_pos_cam_host || !rot1_host9 || n_views < 1 || n_views > PNERF_EX2D_MAX_VIEWS || n_points < 0) return PNERF_E_INVAL;
    if (n_points == 0) return 0;
    if (!d_cam_xyz || !d_dirs) return PNERF_E_INVAL;
    DirArgs a;
    a.xyz = d_cam_xyz; a.n = n_points; a.n_views = n_views; a.has_r2 = rot2_host9 ? 1 : 0;
    for (int v = 0; v < n_views; ++v) for (int k = 0; k < 3; ++k) a.cam_pos[v][k] = cam_pos_cam_host[3 * v + k];
    for (int k = 0; k < 9; ++k) { a.r1[k] = rot1_host9[k]; a.r2[k] = rot2_host9 ? rot2_host9[k] : 0.f; }
    hipLaunchKernelGGL(k_point_dirs, dim3(pn_cdiv((long long)n_points * n_views, 256)), dim3(256), 0, (hipStream_t)stream, a, d_dirs);
    PN_CHECK_LAUNCH();
    return 0;
}
