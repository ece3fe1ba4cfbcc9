// aggregate.hip -- per-sample feature aggregator (forward): gather + feature build + MFMA MLP chain +
// K-weighted reduction, then the colour MLP, for the lego-script architecture.
//
// Replaces NeuralPoints.forward's index_select gathers (models/neural_points/neural_points.py:706-717),
// PointAggregator.forward / viewmlp (models/aggregators/point_aggregators.py:727-814, 488-644) and
// positional_encoding (models/helpers/networks.py:175-190).  The reference runs this as ~60 ATen kernels
// that move every [Nv,256] activation through HBM between layers and compact/scatter rows by boolean
// masks; here one persistent workgroup per CU keeps a 64-row tile (TS samples x K neighbor slots) in LDS
// across the whole chain:
//   gather (embedding 128 B + xyz/dir/colour/conf) -> X0[64x284] in LDS (sin/cos PE computed in place)
//   -> 284->256->256 -> (+7) ->256->256 on v_mfma_f32_32x32x2_f32, weights streamed from an L2-resident
//   fragment-ordered image -> alpha head + K-weighted sums (sigma, f[256]) -> f to HBM
//   -> colour kernel: 64 samples per tile, 280->128->128->128->3.
// In training mode the activations needed by the backward pass are written once (coalesced) to HBM.
#include "mlp_common.h"

// ------------------------------------------------------------------------------ layout / packing
extern "C" int pnerf_mlp_layout(int feat_dim, int64_t *offsets) {
    if (feat_dim != PN_F || !offsets) return PNERF_E_UNSUP;
    const int64_t o[PNERF_MLP_NTENSORS + 1] = {PO_W1, PO_B1, PO_W2, PO_B2, PO_W3, PO_B3, PO_W4, PO_B4, PO_W5, PO_B5,
                                               PO_WC1, PO_BC1, PO_WC2, PO_BC2, PO_WC3, PO_BC3, PO_WC4, PO_BC4, PO_TOTAL};
    for (int i = 0; i <= PNERF_MLP_NTENSORS; ++i) offsets[i] = o[i];
    return 0;
}
extern "C" size_t pnerf_mlp_packed_bytes(void) { return (size_t)PK_TOTAL * sizeof(float); }

namespace {
struct PackDesc { int src, ld, trans, Kreal, Nreal, Kpad, N, NT, dst; };
struct PackTable { PackDesc d[14]; };

__global__ __launch_bounds__(256) void k_pack(PackTable t, const float *__restrict__ params, float *__restrict__ packed) {
    const PackDesc d = t.d[blockIdx.y];
    const int total = d.Kpad * d.N;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e & 3, lane = (e >> 2) & 63;
        int rest = e >> 8;                       // (c*4 + w)*NT + ct
        const int ct = rest % d.NT; rest /= d.NT;
        const int w = rest & 3, c = rest >> 2;
        const int k = 8 * c + 4 * (lane >> 5) + i;
        const int n = w * d.NT * 32 + ct * 32 + (lane & 31);
        float v = 0.f;
        if (k < d.Kreal && n < d.Nreal) v = d.trans ? params[d.src + n * d.ld + k] : params[d.src + k * d.ld + n];
        packed[d.dst + e] = v;
    }
}
}  // namespace

extern "C" int pnerf_mlp_pack(const float *d_params, void *d_packed, void *stream) {
    if (!d_params || !d_packed) return PNERF_E_INVAL;
    PackTable t = {{
        // forward images: B[k][n] = W[n][k]
        {PO_W1, PN_IN1, 1, PN_IN1, PN_H, PN_IN1P, PN_H, 2, PK_F1},
        {PO_W2, PN_H, 1, PN_H, PN_H, PN_H, PN_H, 2, PK_F2},
        {PO_W3, PN_IN3, 1, PN_IN3, PN_H, PN_H + 8, PN_H, 2, PK_F3},
        {PO_W4, PN_H, 1, PN_H, PN_H, PN_H, PN_H, 2, PK_F4},
        {PO_WC1, PN_INC, 1, PN_INC, PN_HC, PN_INC, PN_HC, 1, PK_C1},
        {PO_WC2, PN_HC, 1, PN_HC, PN_HC, PN_HC, PN_HC, 1, PK_C2},
        {PO_WC3, PN_HC, 1, PN_HC, PN_HC, PN_HC, PN_HC, 1, PK_C3},
        // dgrad images: B[k][n] = W[k][n] (k = output unit, n = input unit, first N inputs only)
        {PO_W4, PN_H, 0, PN_H, PN_H, PN_H, PN_H, 2, PK_D4},
        {PO_W3, PN_IN3, 0, PN_H, PN_H, PN_H, PN_H, 2, PK_D3},
        {PO_W2, PN_H, 0, PN_H, PN_H, PN_H, PN_H, 2, PK_D2},
        {PO_W1, PN_IN1, 0, PN_H, PN_H, PN_H, PN_H, 2, PK_D1},
        {PO_WC3, PN_HC, 0, PN_HC, PN_HC, PN_HC, PN_HC, 1, PK_DC3},
        {PO_WC2, PN_HC, 0, PN_HC, PN_HC, PN_HC, PN_HC, 1, PK_DC2},
        {PO_WC1, PN_INC, 0, PN_HC, PN_H, PN_HC, PN_H, 2, PK_DC1},
    }};
    PnProfScope prof(PNK_PACK, (hipStream_t)stream);
    hipLaunchKernelGGL(k_pack, dim3(64, 14), dim3(256), 0, (hipStream_t)stream, t, d_params, (float *)d_packed);
    PN_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ saved activations
size_t pn_saved_bytes(long long n_valid, int K, long long *rows_out, long long *samples_out) {
    const int TS = pn_tile_samples(K);
    // + PN_NCLS tiles of rounding (each class ends in a partial tile) + PN_NCLS gap tiles (every class owns a padding tile) + 1
    const long long tiles = (n_valid + TS - 1) / TS + 2 * PN_NCLS + 1;
    const long long rows = tiles * PN_TILE;
    const long long samples = ((tiles * TS + PN_CTILE - 1) / PN_CTILE + 1) * PN_CTILE;
    if (rows_out) *rows_out = rows;
    if (samples_out) *samples_out = samples;
    size_t b = 0;
    b += pn_align((size_t)rows * PN_IN1P * 4) + 8 * pn_align((size_t)rows * PN_H * 4) + pn_align((size_t)rows * 8 * 4) + pn_align((size_t)rows * 16);
    b += pn_align((size_t)tiles * 3 * PN_NTHR * 8);
    b += pn_cls_bytes(samples);
    b += 2 * pn_align((size_t)samples * PN_H * 4) + pn_align((size_t)samples * 32 * 4) + 6 * pn_align((size_t)samples * PN_HC * 4);
    return b;
}

PnSaved pn_saved_carve(void *base, long long n_valid, int K) {
    PnSaved s;
    size_t total = pn_saved_bytes(n_valid, K, &s.rows, &s.samples);
    PnCarver cv(base, total);
    s.x0 = cv.take<float>((size_t)s.rows * PN_IN1P);
    s.h1 = cv.take<float>((size_t)s.rows * PN_H); s.h2 = cv.take<float>((size_t)s.rows * PN_H);
    s.h3 = cv.take<float>((size_t)s.rows * PN_H); s.h4 = cv.take<float>((size_t)s.rows * PN_H);
    s.dy1 = cv.take<float>((size_t)s.rows * PN_H); s.dy2 = cv.take<float>((size_t)s.rows * PN_H);
    s.dy3 = cv.take<float>((size_t)s.rows * PN_H); s.dy4 = cv.take<float>((size_t)s.rows * PN_H);
    s.ex = cv.take<float>((size_t)s.rows * 8); s.rmeta = cv.take<int4>((size_t)s.rows);
    s.lmask = cv.take<unsigned long long>((size_t)(s.rows / PN_TILE) * 3 * PN_NTHR);
    s.fs = cv.take<float>((size_t)s.samples * PN_H); s.dfs = cv.take<float>((size_t)s.samples * PN_H);
    s.pe = cv.take<float>((size_t)s.samples * 32);
    s.c1 = cv.take<float>((size_t)s.samples * PN_HC); s.c2 = cv.take<float>((size_t)s.samples * PN_HC);
    s.c3 = cv.take<float>((size_t)s.samples * PN_HC); s.dc1 = cv.take<float>((size_t)s.samples * PN_HC);
    s.dc2 = cv.take<float>((size_t)s.samples * PN_HC); s.dc3 = cv.take<float>((size_t)s.samples * PN_HC);
    pn_cls_carve(cv.take<char>(pn_cls_bytes(s.samples)), s.samples, s);
    return s;
}

size_t pn_cls_bytes(long long samples) {
    return pn_align((size_t)samples * 4) + pn_align(PN_CI_WORDS * 4) + pn_align(((size_t)2 * samples + pn_scan_scratch_ints(samples) + 8) * 4);
}
void pn_cls_carve(void *base, long long samples, PnSaved &s) {
    PnCarver cv(base, pn_cls_bytes(samples));
    s.cls_list = cv.take<int>((size_t)samples);
    s.cls_info = cv.take<int>(PN_CI_WORDS);
    s.cls_tmp = cv.take<int>((size_t)2 * samples + pn_scan_scratch_ints(samples) + 8);
}

extern "C" size_t pnerf_agg_saved_bytes(int64_t n_valid_samples, int K) {
    if (K <= 0 || K > PNERF_MAX_K || n_valid_samples < 0) return 0;
    return pn_saved_bytes(n_valid_samples, K, nullptr, nullptr);
}

// ------------------------------------------------------------------------------ forward kernels
namespace {
constexpr int LDX = 292;    // X0 / colour input row stride in LDS (odd multiple of 4 floats: conflict-free b128 reads)
constexpr int LDC = 132;    // colour hidden row stride
constexpr int TPR = PN_TPR; // threads per tile row in the element-wise phases
constexpr int EPT = PN_F / TPR;              // embedding dims per thread in the feature build

struct FwdArgs {
    pnerf_camera cam;
    const float *xyz, *emb, *conf, *dir, *color;
    const float *params;
    const float4 *packed;
    const float *raydir, *sample_loc;
    const float *xyz_pers, *loc_pers;   // optional: perspective coords supplied by the caller (stand-alone aggregator)
    const int *pidx, *valid_list, *counters;
    const int *cls_list, *cls_info;     // sample classes (pn_classify); cls = the class this launch processes
    int cls, Kstride;                   // K = neighbor slots PROCESSED per sample of this class, Kstride = slots per sample in pidx / weight
    int R, SR, K, TS;
    long long cap_samples;      // capacity (in valid samples) of fs / saved buffers
    float *decoded, *weight;
    PnSaved sv;                 // fs always valid; the rest only when TRAIN
};

__device__ __forceinline__ void rot3(const float *M /*row-major*/, float x, float y, float z, bool transpose, float &ox, float &oy, float &oz) {
    // transpose=false: out_j = sum_i v_i M[i][j] (v @ M);  true: out_j = sum_i v_i M[j][i] (v @ M^T)
    if (!transpose) { ox = x * M[0] + y * M[3] + z * M[6]; oy = x * M[1] + y * M[4] + z * M[7]; oz = x * M[2] + y * M[5] + z * M[8]; }
    else { ox = x * M[0] + y * M[1] + z * M[2]; oy = x * M[3] + y * M[4] + z * M[5]; oz = x * M[6] + y * M[7] + z * M[8]; }
}

template <int N> __device__ __forceinline__ float group_sum(float v) {      // sum over N adjacent lanes (N = 4 or 8)
#pragma unroll
    for (int off = 1; off < N; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------ sample classes
// 6 % of the neighbor rows of valid samples are empty slots (a sample near the surface's edge has 1..K-1 neighbors); the
// reference drops them by boolean masking (point_aggregators.py:523-538).  Tiles need whole samples of equal row count, so the
// valid samples are partitioned by the position of their LAST occupied slot into up to three classes that are processed with
// K, K/2 and K/4 rows per sample (K % 4 == 0; the query fills slots front to back, and for a caller-supplied mask the last
// occupied slot is what counts).  The partition is stable (ascending sample id inside a class): results are identical to the
// one-class launch, rows processed drop from 7.49 M to 7.1 M at the bench configuration.  Every class owns a run of tiles
// followed by one padding tile (the partner slot of an odd tile count works on it).
namespace {
struct ClsArgs { const int *valid_list, *counters, *pidx; int K, n, ncls; int kc[PN_NCLS], ts[PN_NCLS]; };

__global__ void k_cls_flags(ClsArgs c, int which, int *__restrict__ flags) {
    const int vs = blockIdx.x * blockDim.x + threadIdx.x;
    if (vs >= c.n) return;
    const int Ns = c.counters[0] < c.n ? c.counters[0] : c.n;
    int f = 0;
    if (vs < Ns) {
        const long long si = c.valid_list[vs];
        int hv = 0;
        for (int k = 0; k < c.K; ++k)
            if (c.pidx[si * c.K + k] >= 0) hv = k + 1;
        int cl = 0;
        for (int j = 1; j < c.ncls; ++j)
            if (hv <= c.kc[j]) cl = j;
        f = cl == which;
    }
    flags[vs] = f;
}

__global__ void k_cls_gather(ClsArgs c, int which, const int *__restrict__ pos, const int *__restrict__ count, int *__restrict__ cls_list,
                             int *__restrict__ info) {
    const int n = *count, base = info[PN_CI_VBASE + which];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cls_list[base + i] = c.valid_list[pos[i]];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int tiles = (n + c.ts[which] - 1) / c.ts[which], tb = info[PN_CI_TBASE + which];
        info[PN_CI_COUNT + which] = n;
        if (which + 1 < PN_NCLS) { info[PN_CI_VBASE + which + 1] = base + n; info[PN_CI_TBASE + which + 1] = tb + tiles + 1; }
        info[PN_CI_TILES] = tb + tiles + 1;
    }
}

// the padding tile of every class: finite (zero) rows in everything the weight-gradient GEMMs read
__global__ void k_cls_zero_gaps(PnSaved sv, int ncls) {
    const int c = blockIdx.y;
    if (c >= ncls) return;
    const int n = sv.cls_info[PN_CI_COUNT + c];
    (void)n;
    const long long gap = (c + 1 < PN_NCLS ? sv.cls_info[PN_CI_TBASE + c + 1] : sv.cls_info[PN_CI_TILES]) - 1;
    float *arrs[9] = {sv.x0, sv.h1, sv.h2, sv.h3, sv.h4, sv.dy1, sv.dy2, sv.dy3, sv.dy4};
    const int which = blockIdx.x;                      // 9 arrays
    const int w = which == 0 ? PN_IN1P : PN_H;
    float *p = arrs[which] + gap * PN_TILE * w;
    for (int i = threadIdx.x; i < PN_TILE * w; i += blockDim.x) p[i] = 0.f;
}
}  // namespace

}  // namespace

// class c processes kc[c] slots per sample
int pn_class_slots(int K, int kc[PN_NCLS]) {
    if (K % 4 == 0) { kc[0] = K; kc[1] = K / 2; kc[2] = K / 4; return 3; }
    kc[0] = K; kc[1] = kc[2] = 0;
    return 1;
}

int pn_classify(const PnSaved &sv, const int32_t *d_valid_list, const int32_t *d_counters, const int32_t *d_pidx, int K, long long n_valid,
                bool train, hipStream_t s) {
    ClsArgs c;
    c.valid_list = d_valid_list; c.counters = d_counters; c.pidx = d_pidx; c.K = K; c.n = (int)n_valid;
    c.ncls = pn_class_slots(K, c.kc);
    for (int j = 0; j < PN_NCLS; ++j) c.ts[j] = c.kc[j] > 0 ? pn_tile_samples(c.kc[j]) : 1;
    if (hipMemsetAsync(sv.cls_info, 0, PN_CI_WORDS * sizeof(int), s) != hipSuccess) return PNERF_E_LAUNCH;
    if (n_valid <= 0) return 0;
    int *flags = sv.cls_tmp, *pos = flags + n_valid, *cnt = pos + n_valid, *scratch = cnt + 8;
    PnProfScope prof(PNK_COMPACT, s);
    for (int j = 0; j < c.ncls; ++j) {
        hipLaunchKernelGGL(k_cls_flags, dim3(pn_cdiv(n_valid, 256)), dim3(256), 0, s, c, j, flags);
        int rc = pn_compact_gt0_i32(flags, n_valid, pos, cnt, scratch, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_cls_gather, dim3(256), dim3(256), 0, s, c, j, pos, cnt, sv.cls_list, sv.cls_info);
    }
    if (train) hipLaunchKernelGGL(k_cls_zero_gaps, dim3(9, c.ncls), dim3(256), 0, s, sv, c.ncls);
    PN_CHECK_LAUNCH();
    return 0;
}

namespace {

// ------------------------------------------------------------------------------ two-tile forward
// Same organisation as the backward (backward.hip): one workgroup (4 waves, one per SIMD) per CU, two tiles A / B in flight,
// their layer GEMMs alternating  G1(A) G1(B) G2(A) G2(B) G3(A) G3(B) G4(A) G4(B).  Everything element-wise is issued by the
// GEMM waves themselves in the shadow of their own MFMAs (pn_tile_gemm_side): the other tile's epilogue (bias, LeakyReLU,
// sign bits, accumulators -> LDS), the copy-out of the own tile's previous layer (training) and, in G1(A) / G4(B), the whole
// boundary of the partner buffer: last epilogue, alpha head, K-weighted sums of the finished tile; gather, positional
// encodings and weights of the tile that replaces it.  All rows of the tile use stride LDX, so that layer 3's input
// [h2 | colour, dir - view, dir . view] is one row (264 columns) and every layer is a single GEMM call.
constexpr int F2_TILE_FLOATS = PN_TILE * LDX + PN_TILE * 8 + 4 * PN_TILE;
constexpr int F2_LDS_FLOATS = 2 * F2_TILE_FLOATS + PN_H;
static_assert(F2_LDS_FLOATS * 4 <= 160 * 1024, "two forward tiles must fit the 160 KB LDS");

struct F2Tile {            // LDS of one in-flight tile
    float *buf;            // [64][LDX]  X0 -> h1 -> h2 (+ extras in columns 256..263) -> h3 -> h4
    float *exb;            // [64][8]    layer-3 extras until they move into buf
    float *wraw, *wrow, *wnrm;
    int *sidx;             // [64] sample id of each ROW (or -1)
};
struct F2State {           // registers of one buffer: indices of the tile being loaded next and of the one after it
    int si1, p1;           // sample / point id of this thread's row in the next tile (ready)
    int si2;               // sample id of this thread's row in the tile after that (ready)
    int p2, si3;           // requested during the current boundary
    int tile;              // tile whose activations live in the buffer
    int t1, t2, t3;        // tile indices behind si1, si2, si3
};
struct F2Bnd {             // registers of one hosted boundary program
    float4 e0, e1;                 // the thread's 8 embedding dims
    float px, py, pz, lx, ly, lz;  // point / sample position
    float ppx, ppy, ppz, lpx, lpy, lpz;   // optional caller-supplied perspective coordinates
    float cf, dxv, dyv, dzv, cx, cy, cz, rx, ry, rz;
    float da, db;                  // distance components q and q + 4 of this thread (PE5 input)
    float s, wn, w;
    float4 hv, wv, f, cpv;
    int pcur, sicur, m, k;
#ifdef PN_PHASE_TRACE
    int titer, trbase;
#endif
};

__device__ __forceinline__ F2Tile f2_carve(float *base) {
    F2Tile t;
    t.buf = base; t.exb = t.buf + PN_TILE * LDX;
    t.wraw = t.exb + PN_TILE * 8; t.wrow = t.wraw + PN_TILE; t.wnrm = t.wrow + PN_TILE;
    t.sidx = reinterpret_cast<int *>(t.wnrm + PN_TILE);
    return t;
}

// sample id of row `row` of tile `tile` (or -1)
__device__ __forceinline__ int f2_sample_of(const FwdArgs &a, int tile, int row, int Ns) {
    const int ls = row / a.K;
    const long long vs = (long long)tile * a.TS + ls;
    return (ls < a.TS && vs < Ns) ? a.valid_list[vs] : -1;
}

// requests of the boundary program: the point data of the next tile (indices are already in registers) and the indices of the two after it
template <bool PERS>
__device__ __forceinline__ void f2_request(const FwdArgs &a, F2State &S, F2Bnd &C, int tl, int Ns, int stride) {
    const int row = tl / TPR, q = tl % TPR, k = row % a.K;
    const int p = S.p1 > 0 ? S.p1 : 0, si = S.si1 > 0 ? S.si1 : 0;     // empty slots / rows read point 0 / sample 0 like the reference (neural_points.py:709); their weight is 0
    C.pcur = S.p1; C.sicur = S.si1;
    const float *ep = a.emb + (long long)p * PN_F + EPT * q;
    C.e0 = *reinterpret_cast<const float4 *>(ep); C.e1 = *reinterpret_cast<const float4 *>(ep + 4);
    C.px = a.xyz[3 * p]; C.py = a.xyz[3 * p + 1]; C.pz = a.xyz[3 * p + 2];
    C.lx = a.sample_loc[(long long)si * 3]; C.ly = a.sample_loc[(long long)si * 3 + 1]; C.lz = a.sample_loc[(long long)si * 3 + 2];
    if (PERS) {
        C.ppx = a.xyz_pers[3 * p]; C.ppy = a.xyz_pers[3 * p + 1]; C.ppz = a.xyz_pers[3 * p + 2];
        C.lpx = a.loc_pers[(long long)si * 3]; C.lpy = a.loc_pers[(long long)si * 3 + 1]; C.lpz = a.loc_pers[(long long)si * 3 + 2];
    }
    C.cf = a.conf[p];
    if (q == 0) {
        const int r = si / a.SR;
        C.dxv = a.dir[3 * p]; C.dyv = a.dir[3 * p + 1]; C.dzv = a.dir[3 * p + 2];
        C.cx = a.color[3 * p]; C.cy = a.color[3 * p + 1]; C.cz = a.color[3 * p + 2];
        C.rx = a.raydir[3 * r]; C.ry = a.raydir[3 * r + 1]; C.rz = a.raydir[3 * r + 2];
    }
    S.p2 = S.si2 >= 0 ? a.pidx[(long long)S.si2 * a.Kstride + k] : -1;
    S.t3 = S.t2 + stride;
    S.si3 = f2_sample_of(a, S.t3, row, Ns);
}

// geometry of the row: 6 distance components, raw weight, layer-3 extras (point_aggregators.py:773-784, :425-428, :506, :566)
template <bool PERS>
__device__ __forceinline__ void f2_geometry(const FwdArgs &a, const F2Tile &T, F2Bnd &C, int tl) {
    const int row = tl / TPR, q = tl % TPR;
    const float dwx = C.px - C.lx, dwy = C.py - C.ly, dwz = C.pz - C.lz;
    float ppx, ppy, pcz, spx, spy, scz;
    if (PERS) {
        ppx = C.ppx; ppy = C.ppy; pcz = C.ppz; spx = C.lpx; spy = C.lpy; scz = C.lpz;
    } else {
        float pcx, pcy, scx, scy;
        rot3(a.cam.camrot, C.px - a.cam.campos[0], C.py - a.cam.campos[1], C.pz - a.cam.campos[2], false, pcx, pcy, pcz);
        rot3(a.cam.camrot, C.lx - a.cam.campos[0], C.ly - a.cam.campos[1], C.lz - a.cam.campos[2], false, scx, scy, scz);
        ppx = pcx / pcz; ppy = pcy / pcz; spx = scx / scz; spy = scy / scz;
    }
    float d0, d1, d2;
    rot3(a.cam.rw2c, dwx, dwy, dwz, true, d0, d1, d2);
    const float d3 = ppx * pcz - spx * scz, d4 = ppy * pcz - spy * scz, d5 = pcz - scz;
    C.da = q == 0 ? d0 : q == 1 ? d1 : q == 2 ? d2 : d3;      // (selects of values, not of struct members: a select of addresses would pin the struct in scratch)
    C.db = q == 0 ? d4 : d5;
    if (q == 0) {
        float vx, vy, vz, qx, qy, qz;
        rot3(a.cam.rw2c, C.rx, C.ry, C.rz, true, vx, vy, vz);
        rot3(a.cam.rw2c, C.dxv, C.dyv, C.dzv, true, qx, qy, qz);
        float *ex = T.exb + row * 8;
        *reinterpret_cast<float4 *>(ex) = make_float4(C.cx, C.cy, C.cz, qx - vx);
        *reinterpret_cast<float4 *>(ex + 4) = make_float4(qy - vy, qz - vz, qx * vx + qy * vy + qz * vz, 0.f);
        T.wraw[row] = C.pcur >= 0 ? 1.0f / fmaxf(sqrtf(dwx * dwx + dwy * dwy + dwz * dwz), 1e-6f) : 0.f;
        T.sidx[row] = C.sicur;
    }
}

// [e | PE3(e)] of embedding dim i of this thread (one accurate sincosf, exact double-angle steps for the octaves)
template <int I>
__device__ __forceinline__ void f2_pe_emb(const F2Tile &T, const F2Bnd &C, int tl) {
    const int row = tl / TPR, q = tl % TPR, dd = EPT * q + I;
    const float e = I == 0 ? C.e0.x : I == 1 ? C.e0.y : I == 2 ? C.e0.z : I == 3 ? C.e0.w : I == 4 ? C.e1.x : I == 5 ? C.e1.y : I == 6 ? C.e1.z : C.e1.w;
    float *xa = T.buf + row * LDX;
    float s, c;
    sincosf(e, &s, &c);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        *reinterpret_cast<float2 *>(xa + PN_F + (dd * 3 + f) * 2) = make_float2(s, c);
        const float s2 = 2.f * s * c;
        c = 1.f - 2.f * s * s; s = s2;
    }
}

// PE5 of distance component q + 4*J of this row (thread q takes components q and q + 4)
template <int J>
__device__ __forceinline__ void f2_pe_dist(const F2Tile &T, const F2Bnd &C, int tl) {
    const int row = tl / TPR, q = tl % TPR, comp = q + 4 * J;
    if (comp < 6) {
        float *xa = T.buf + row * LDX;
        float s, c;
        sincosf(J == 0 ? C.da : C.db, &s, &c);
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            *reinterpret_cast<float2 *>(xa + PN_F * 7 + (comp * 5 + f) * 2) = make_float2(s, c);
            const float s2 = 2.f * s * c;
            c = 1.f - 2.f * s * s; s = s2;
        }
    }
}

// weights of the row (q == 0 threads): normalise over the K slots, multiply by the clamped confidence (:801-811)
template <bool TRAIN>
__device__ __forceinline__ void f2_weights(const FwdArgs &a, const F2Tile &T, const F2Bnd &C, int tile, int tl) {
    const int row = tl / TPR, q = tl % TPR;
    if (q == 0) {
        const int ls = row / a.K, k = row - ls * a.K;
        float wn = 0.f, w = 0.f;
        if (C.sicur >= 0) {
            float sum = 0.f;
            for (int kk = 0; kk < a.K; ++kk) sum += T.wraw[ls * a.K + kk];
            wn = T.wraw[row] / fmaxf(sum, 1e-8f);
            w = wn * fminf(fmaxf(C.cf, 1e-4f), 1.0f);
            a.weight[(long long)C.sicur * a.Kstride + k] = wn;
        }
        T.wnrm[row] = wn; T.wrow[row] = w;
        if (TRAIN) a.sv.rmeta[(long long)tile * PN_TILE + row] = make_int4(C.sicur, C.sicur >= 0 ? C.pcur : -1, __float_as_int(wn), __float_as_int(w));
    }
}

// epilogue piece R of a layer: bias + LeakyReLU + sign bit of accumulator element R -> LDS (accumulator layout)
template <int R, bool BITS>
__device__ __forceinline__ void f2_epi_piece(const f32x16 (&acc)[2][2], const float (&bias)[2], float *wy, unsigned &mlo, unsigned &mhi) {
    constexpr int mt = R >> 5, ct = (R >> 4) & 1, reg = R & 15;
    const float v = acc[mt][ct][reg] + bias[ct];
    if (BITS) {
        if (R < 32) { mlo |= (v > 0.f ? 1u : 0u) << (R & 31); asm volatile("" : "+v"(mlo)); }
        else { mhi |= (v > 0.f ? 1u : 0u) << (R & 31); asm volatile("" : "+v"(mhi)); }
    }
    wy[(mt * 32 + (reg & 3) + 8 * (reg >> 2)) * LDX + ct * 32] = fmaxf(v, 0.01f * v);
}

#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_fwd);
#endif
// ---- the boundary program of one buffer, slot by slot (see backward.hip for the rules: no value is consumed in the slot that
// requested it, pieces stay under ~a dozen instructions, one burst of requests).  Slot map:
//   0..2    requests for the next tile        4..67  E4: accumulator element s-4 (+ bias, LeakyReLU) -> LDS      68: barrier
//   70..133 alpha head of the finished tile (column group j: read 70+4j, use 72+4j)   136, 138: reduce, softplus   142: barrier
//   144..190 K-weighted sums (<= 24 (item, term) pieces, item-major, stores inline), h4 copy-out (training) interleaved; 214 sigma   216: barrier
//   218..   next tile: geometry (218..221), embedding PE (224 + 6 i), distance PE (272, 280), pad (288)           296: barrier
//   298     weights;  374: index shift          (X0 / extras go to HBM during the tile's own first GEMM, like every other layer's input)
template <int SLOT, bool TRAIN, bool PERS>
__device__ __forceinline__ void f2_boundary_slot(const FwdArgs &a, const F2Tile &T, F2State &S, const f32x16 (&acc)[2][2], const float (&bias4)[2],
                                                 float *wy, F2Bnd &C, const float *w5s, float b5, int tl, int Ns, int stride) {
    const int rrow = tl / TPR, rq = tl % TPR;
    const int K = a.K, TS = a.TS;
    if constexpr (SLOT == 0) f2_request<PERS>(a, S, C, tl, Ns, stride);
    if constexpr (SLOT >= 4 && SLOT < 68) {
        unsigned d0 = 0u, d1 = 0u;
        f2_epi_piece<SLOT - 4, false>(acc, bias4, wy, d0, d1);
    }
    if constexpr (SLOT == 68 || SLOT == 142 || SLOT == 216 || SLOT == 296) __syncthreads();
#ifdef PN_PHASE_TRACE
    if constexpr (SLOT == 1 || SLOT == 3 || SLOT == 67 || SLOT == 141 || SLOT == 215 || SLOT == 223 || SLOT == 271 || SLOT == 295 || SLOT == 299 || SLOT == 375) {
        constexpr int k = SLOT == 1 ? 0 : SLOT == 3 ? 1 : SLOT == 67 ? 2 : SLOT == 141 ? 3 : SLOT == 215 ? 4 : SLOT == 223 ? 5 : SLOT == 271 ? 6 : SLOT == 295 ? 7 : SLOT == 299 ? 8 : 9;
        const int tid = threadIdx.x, titer = C.titer;
        if (C.trbase >= 0) PN_TR(pn_trace_fwd, C.trbase + k);
    }
#endif
    // ---- alpha head of the finished tile (256 -> 1, softplus(x - 1), raw2out_density :262-265)
    if constexpr (SLOT == 69) C.s = 0.f;
    if constexpr (SLOT >= 70 && SLOT < 134 && (SLOT - 70) % 4 == 0) {
        constexpr int j = (SLOT - 70) / 4;
        C.hv = *reinterpret_cast<const float4 *>(T.buf + rrow * LDX + rq * 4 + 16 * j);
        C.wv = *reinterpret_cast<const float4 *>(w5s + rq * 4 + 16 * j);
    }
    if constexpr (SLOT >= 70 && SLOT < 136 && (SLOT - 70) % 4 == 2) {
        C.s += C.hv.x * C.wv.x + C.hv.y * C.wv.y + C.hv.z * C.wv.z + C.hv.w * C.wv.w;
        asm volatile("" : "+v"(C.s));
    }
    if constexpr (SLOT == 136) C.s = group_sum<TPR>(C.s);
    if constexpr (SLOT == 138) {
        if (rq == 0) {
            const float x = C.s + b5 - 1.0f;
            const float alpha = x > 20.f ? x : log1pf(expf(x));
            T.wraw[rrow] = alpha * T.wrow[rrow];
        }
    }
    // ---- K-weighted sums of the finished tile -> f[256] per sample (HBM), sigma; h4 copy-out
    // (items of 64 float4 columns x TS samples are dealt to the 256 threads; item m of a thread is sample (tl >> 6) + 4 m.  For every K
    //  there are at most 24 (item, term) pieces per thread: they run in item-major order through one accumulator.)
    if constexpr (SLOT == 143) { C.f = make_float4(0.f, 0.f, 0.f, 0.f); C.m = 0; C.k = 0; }
    if constexpr (SLOT >= 144 && SLOT < 192 && (SLOT - 144) % 2 == 0) {
        const int ls = (tl >> 6) + 4 * C.m, c4 = tl & 63;
        if (ls < TS) {
            const float w = T.wrow[ls * K + C.k];
            const float4 v = *reinterpret_cast<const float4 *>(T.buf + (ls * K + C.k) * LDX + c4 * 4);
            C.f.x += w * v.x; C.f.y += w * v.y; C.f.z += w * v.z; C.f.w += w * v.w;
            if (C.k == K - 1) {
                const long long vs = (long long)S.tile * TS + ls;
                if (vs < a.cap_samples) *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + c4 * 4) = C.f;
                C.f = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        C.k += 1;
        if (C.k == K) { C.k = 0; C.m += 1; }
        asm volatile("" : "+v"(C.f.x), "+v"(C.f.y), "+v"(C.f.z), "+v"(C.f.w));
    }
    if constexpr (TRAIN && SLOT >= 145 && SLOT < 209 && (SLOT - 145) % 4 == 0) {
        constexpr int i = (SLOT - 145) / 4;
        C.cpv = *reinterpret_cast<const float4 *>(T.buf + ((tl >> 6) + 4 * i) * LDX + (tl & 63) * 4);
    }
    if constexpr (TRAIN && SLOT >= 145 && SLOT < 209 && (SLOT - 145) % 4 == 2) {
        constexpr int i = (SLOT - 145) / 4;
        pn_store_stream(a.sv.h4 + ((long long)S.tile * PN_TILE + (tl >> 6) + 4 * i) * PN_H + (tl & 63) * 4, C.cpv);
    }
    if constexpr (SLOT == 214) {
        if (tl < TS) {
            const int si = T.sidx[tl * K];
            if (si >= 0) {
                float sg = 0.f;
                for (int k = 0; k < K; ++k) sg += T.wraw[tl * K + k];
                a.decoded[(long long)si * 4] = sg;
            }
        }
    }
    // ---- the next tile takes over the buffer
    if constexpr (SLOT == 218) {
        const int ntiles = (Ns + TS - 1) / TS;
        S.tile = S.t1 < ntiles ? S.t1 : ntiles;              // a tile past the end lives on the padding tile's storage (all rows empty)
        f2_geometry<PERS>(a, T, C, tl);
        float *xa = T.buf + rrow * LDX;
        *reinterpret_cast<float4 *>(xa + EPT * rq) = C.e0; *reinterpret_cast<float4 *>(xa + EPT * rq + 4) = C.e1;
    }
    if constexpr (SLOT >= 224 && SLOT < 272 && (SLOT - 224) % 6 == 0) f2_pe_emb<(SLOT - 224) / 6>(T, C, tl);
    if constexpr (SLOT == 272) f2_pe_dist<0>(T, C, tl);
    if constexpr (SLOT == 280) f2_pe_dist<1>(T, C, tl);
    if constexpr (SLOT == 288) {
        if (rq == TPR - 1) {
            float *xa = T.buf + rrow * LDX;
            *reinterpret_cast<float4 *>(xa + PN_IN1) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(xa + PN_IN1 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if constexpr (SLOT == 298) f2_weights<TRAIN>(a, T, C, S.tile, tl);
    if constexpr (SLOT == 374) {
        S.si1 = S.si2; S.p1 = S.p2; S.t1 = S.t2; S.si2 = S.si3; S.t2 = S.t3;
    }
}

template <bool TRAIN, bool PERS>
__global__ __launch_bounds__(PN_NTHR, 1) void k_agg_forward2(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const F2Tile TA = f2_carve(smem), TB = f2_carve(smem + F2_TILE_FLOATS);
    float *w5s = smem + 2 * F2_TILE_FLOATS;
    const int tid = threadIdx.x;
    const int K = a.K, TS = a.TS;
    // this launch processes one sample class: its list, its run of tiles, its range of per-sample rows
    const int Ns = a.cls_info[PN_CI_COUNT + a.cls];
    {
        const long long vb = a.cls_info[PN_CI_VBASE + a.cls], tb = a.cls_info[PN_CI_TBASE + a.cls];
        a.valid_list = a.cls_list + vb; a.cap_samples = Ns;
        a.sv.fs += vb * PN_H;
        if (TRAIN) {
            a.sv.x0 += tb * PN_TILE * PN_IN1P; a.sv.ex += tb * PN_TILE * 8; a.sv.rmeta += tb * PN_TILE; a.sv.lmask += tb * 3 * PN_NTHR;
            a.sv.h1 += tb * PN_TILE * PN_H; a.sv.h2 += tb * PN_TILE * PN_H; a.sv.h3 += tb * PN_TILE * PN_H; a.sv.h4 += tb * PN_TILE * PN_H;
        }
    }
    const int ntiles = (Ns + TS - 1) / TS;
    const float *P = a.params;
    if (tid < PN_H) w5s[tid] = P[PO_W5 + tid];
    const float b5 = P[PO_B5];
    const int stride = 2 * (int)gridDim.x;
    if ((int)blockIdx.x * 2 >= ntiles) return;

    f32x16 accA[2][2], accB[2][2];
    pn_acc_zero(accA); pn_acc_zero(accB);
    F2State SA, SB;
    F2Bnd CB;
    float4 bpre[2];
    {   // prologue: index pipelines of both buffers; the first tile of buffer A is built plainly
        const int row = tid / TPR, k = row % K;
        SA.t1 = 2 * (int)blockIdx.x; SA.t2 = SA.t1 + stride;
        SB.t1 = SA.t1 + 1; SB.t2 = SB.t1 + stride;
        SA.si1 = f2_sample_of(a, SA.t1, row, Ns); SA.si2 = f2_sample_of(a, SA.t2, row, Ns);
        SB.si1 = f2_sample_of(a, SB.t1, row, Ns); SB.si2 = f2_sample_of(a, SB.t2, row, Ns);
        SA.p1 = SA.si1 >= 0 ? a.pidx[(long long)SA.si1 * a.Kstride + k] : -1;
        SB.p1 = SB.si1 >= 0 ? a.pidx[(long long)SB.si1 * a.Kstride + k] : -1;
        SA.tile = SA.t1; SB.tile = ntiles;              // buffer B starts as an empty finished tile on the padding tile's storage
        f2_request<PERS>(a, SA, CB, tid, Ns, stride);
        SA.tile = SA.t1;
        f2_geometry<PERS>(a, TA, CB, tid);
        float *xa = TA.buf + row * LDX;
        *reinterpret_cast<float4 *>(xa + EPT * (tid % TPR)) = CB.e0; *reinterpret_cast<float4 *>(xa + EPT * (tid % TPR) + 4) = CB.e1;
        pn_static_for<8>([&](auto ii) { f2_pe_emb<decltype(ii)::value>(TA, CB, tid); });
        f2_pe_dist<0>(TA, CB, tid); f2_pe_dist<1>(TA, CB, tid);
        if (tid % TPR == TPR - 1) {
            *reinterpret_cast<float4 *>(xa + PN_IN1) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(xa + PN_IN1 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < PN_TILE) { TB.sidx[tid] = -1; TB.wrow[tid] = 0.f; TB.wraw[tid] = 0.f; }
        __syncthreads();
        f2_weights<TRAIN>(a, TA, CB, SA.tile, tid);
        SA.si1 = SA.si2; SA.p1 = SA.p2; SA.t1 = SA.t2; SA.si2 = SA.si3; SA.t2 = SA.t3;
        pn_gemm_prefetch_b0(a.packed + PK_F1 / 4, tid >> 6, tid & 63, bpre);
    }
#ifdef PN_PHASE_TRACE
    int titer = -1;
#endif
    for (int pair = blockIdx.x; pair * 2 < ntiles; pair += gridDim.x) {
#ifdef PN_PHASE_TRACE
        ++titer;
#endif
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int lane = tl & 63, wave = tl >> 6;
        __syncthreads();
        PN_TR(pn_trace_fwd, 0);
        float *wyA = TA.buf + (4 * (lane >> 5)) * LDX + wave * 64 + (lane & 31);     // accumulator-layout write base
        float *wyB = TB.buf + (4 * (lane >> 5)) * LDX + wave * 64 + (lane & 31);
        const float *rxA = TA.buf + wave * LDX + lane * 4, *rxB = TB.buf + wave * LDX + lane * 4;   // copy-out read base (+ 4*i rows)
        const int bcol = wave * 64 + (lane & 31);
        float4 cpv = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned mlo = 0u, mhi = 0u;
        float bias[2] = {0.f, 0.f};
#ifdef PN_PHASE_TRACE
        CB.titer = titer; CB.trbase = 9;
#endif
        const float bias4[2] = {P[PO_B4 + bcol], P[PO_B4 + bcol + 32]};

        // one G step: GEMM of tile X (NCH chunks of LDS XB against weight image PK) with, in the MFMA shadows,
        //   E (bias PB, LeakyReLU, sign bits -> mask word ML of tile YT) of the other tile's accumulators ACCY -> WY,
        //   EXTRAS: that tile's layer-3 extras move next to its h2,
        //   the copy-out of X's own previous layer RX -> DST (training), and optionally a boundary program BND
#define F2_NOBND(s_) (void)0
#define F2_BND_B(s_) f2_boundary_slot<s_, TRAIN, PERS>(a, TB, SB, accB, bias4, wyB, CB, w5s, b5, tl, Ns, stride)
#define F2_BND_A(s_) f2_boundary_slot<s_, TRAIN, PERS>(a, TA, SA, accA, bias4, wyA, CB, w5s, b5, tl, Ns, stride)
#define F2_STEP(NCH, XB, ACCX, PK, PKNEXT, ACCY, PB, WY, YT, YTILE, ML, EPI, EXTRAS, COPY, COPY0, XT, RX, DST, XTILE, BND)             \
        {                                                                                                                           \
            if (EPI) { bias[0] = P[(PB) + bcol]; bias[1] = P[(PB) + bcol + 32]; mlo = 0u; mhi = 0u; }                               \
            pn_acc_zero(ACCX);                                                                                                      \
            pn_tile_gemm_side<NCH>(XB, LDX, a.packed + (PK) / 4, wave, lane, ACCX, bpre, a.packed + (PKNEXT) / 4, [&](auto ss) {    \
                constexpr int s = decltype(ss)::value;                                                                              \
                if constexpr (EPI && s % 8 == 0 && s < 512) f2_epi_piece<s / 8, TRAIN>(ACCY, bias, WY, mlo, mhi);                   \
                if constexpr (EPI && EXTRAS && s == 509) {                                                                          \
                    if (tl < PN_TILE * 2) *reinterpret_cast<float4 *>((YT).buf + (tl >> 1) * LDX + PN_H + (tl & 1) * 4) = *reinterpret_cast<const float4 *>((YT).exb + (tl >> 1) * 8 + (tl & 1) * 4); \
                }                                                                                                                   \
                if constexpr (EPI && TRAIN && s == 510) a.sv.lmask[((long long)(YTILE) * 3 + (ML)) * PN_NTHR + tl] = ((unsigned long long)mhi << 32) | mlo; \
                if constexpr (COPY0 && TRAIN && s % 32 == 8) {             /* X0 [64][288]: float4 number tl + 256 i, i < 18 */            \
                    const int e_ = tl + (s / 32) * PN_NTHR, row_ = e_ / (PN_IN1P / 4), c4_ = e_ - row_ * (PN_IN1P / 4);                 \
                    cpv = *reinterpret_cast<const float4 *>((XT).buf + row_ * LDX + c4_ * 4);                                          \
                }                                                                                                                   \
                if constexpr (COPY0 && TRAIN && s % 32 == 24) {                                                                     \
                    const int e_ = tl + (s / 32) * PN_NTHR, row_ = e_ / (PN_IN1P / 4), c4_ = e_ - row_ * (PN_IN1P / 4);                 \
                    pn_store_stream(a.sv.x0 + ((long long)(XTILE) * PN_TILE + row_) * PN_IN1P + c4_ * 4, cpv);                        \
                }                                                                                                                   \
                if constexpr (COPY0 && TRAIN && s == 30) {                                                                          \
                    if (tl < PN_TILE * 2) *reinterpret_cast<float4 *>(a.sv.ex + ((long long)(XTILE) * PN_TILE + (tl >> 1)) * 8 + (tl & 1) * 4) = *reinterpret_cast<const float4 *>((XT).exb + (tl >> 1) * 8 + (tl & 1) * 4); \
                }                                                                                                                   \
                if constexpr (COPY && TRAIN && s % 32 == 4 && s < 512) cpv = *reinterpret_cast<const float4 *>((RX) + 4 * (s / 32) * LDX); \
                if constexpr (COPY && TRAIN && s % 32 == 20 && s < 512) pn_store_stream((DST) + ((long long)(XTILE) * PN_TILE + wave + 4 * (s / 32)) * PN_H + lane * 4, cpv); \
                BND(s);                                                                                                             \
            });                                                                                                                     \
            __syncthreads();                                                                                                        \
        }
        //      chunks       X-tile  accX  image  next   accY  bias   writeY Y  Y-tile   mask EPI    EXTRAS COPY   COPY0 X   readX dst       X-tile   boundary
        F2_STEP(PN_IN1P / 8, TA.buf, accA, PK_F1, PK_F1, accB, PO_B4, wyB, TB, SB.tile, 0, false, false, false, true, TA, rxA, a.sv.h1, SA.tile, F2_BND_B)
        PN_TR(pn_trace_fwd, 1);
        F2_STEP(PN_IN1P / 8, TB.buf, accB, PK_F1, PK_F2, accA, PO_B1, wyA, TA, SA.tile, 0, true, false, false, true, TB, rxB, a.sv.h1, SB.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 2);
        F2_STEP(PN_H / 8, TA.buf, accA, PK_F2, PK_F2, accB, PO_B1, wyB, TB, SB.tile, 0, true, false, true, false, TA, rxA, a.sv.h1, SA.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 3);
        F2_STEP(PN_H / 8, TB.buf, accB, PK_F2, PK_F3, accA, PO_B2, wyA, TA, SA.tile, 1, true, true, true, false, TB, rxB, a.sv.h1, SB.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 4);
        F2_STEP(PN_H / 8 + 1, TA.buf, accA, PK_F3, PK_F3, accB, PO_B2, wyB, TB, SB.tile, 1, true, true, true, false, TA, rxA, a.sv.h2, SA.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 5);
        F2_STEP(PN_H / 8 + 1, TB.buf, accB, PK_F3, PK_F4, accA, PO_B3, wyA, TA, SA.tile, 2, true, false, true, false, TB, rxB, a.sv.h2, SB.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 6);
        F2_STEP(PN_H / 8, TA.buf, accA, PK_F4, PK_F4, accB, PO_B3, wyB, TB, SB.tile, 2, true, false, true, false, TA, rxA, a.sv.h3, SA.tile, F2_NOBND)
        PN_TR(pn_trace_fwd, 7);
#ifdef PN_PHASE_TRACE
        CB.trbase = -1;
#endif
        F2_STEP(PN_H / 8, TB.buf, accB, PK_F4, PK_F1, accA, PO_B4, wyA, TA, SA.tile, 0, false, false, true, false, TB, rxB, a.sv.h3, SB.tile, F2_BND_A)
        PN_TR(pn_trace_fwd, 8);
#undef F2_STEP
#undef F2_BND_A
#undef F2_BND_B
#undef F2_NOBND
    }
    {   // epilogue: the last tile of buffer B: last layer's epilogue, alpha head, K-weighted sums (plain)
        const int lane = tid & 63, wave = tid >> 6, bcol = wave * 64 + (lane & 31);
        const float bias4[2] = {P[PO_B4 + bcol], P[PO_B4 + bcol + 32]};
        float *wyB = TB.buf + (4 * (lane >> 5)) * LDX + wave * 64 + (lane & 31);
        unsigned d0 = 0u, d1 = 0u;
        pn_static_for<64>([&](auto rr) { f2_epi_piece<decltype(rr)::value, false>(accB, bias4, wyB, d0, d1); });
        __syncthreads();
        if (TRAIN) pn_tile_copy_out<PN_TILE, PN_H, PN_NTHR>(TB.buf, LDX, a.sv.h4, PN_H, (long long)SB.tile * PN_TILE, tid);
        {
            const int row = tid / TPR, q = tid % TPR;
            const float *h = TB.buf + row * LDX + q * 4;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(h + 16 * j);
                const float4 w = *reinterpret_cast<const float4 *>(w5s + q * 4 + 16 * j);
                s += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
            }
            s = group_sum<TPR>(s);
            if (q == 0) {
                const float x = s + b5 - 1.0f;
                TB.wraw[row] = (x > 20.f ? x : log1pf(expf(x))) * TB.wrow[row];
            }
        }
        __syncthreads();
        for (int e = tid; e < TS * 64; e += PN_NTHR) {
            const int ls = e >> 6, c4 = e & 63;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < K; ++k) {
                const float w = TB.wrow[ls * K + k];
                const float4 v = *reinterpret_cast<const float4 *>(TB.buf + (ls * K + k) * LDX + c4 * 4);
                f.x += w * v.x; f.y += w * v.y; f.z += w * v.z; f.w += w * v.w;
            }
            const long long vs = (long long)SB.tile * TS + ls;
            if (vs < a.cap_samples) *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + c4 * 4) = f;
        }
        if (tid < TS) {
            const int si = TB.sidx[tid * K];
            if (si >= 0) {
                float sg = 0.f;
                for (int k = 0; k < K; ++k) sg += TB.wraw[tid * K + k];
                a.decoded[(long long)si * 4] = sg;
            }
        }
    }
}

constexpr int COL_LDS_FLOATS = PN_CTILE * LDX + 32;      // 75 KB: two workgroups per CU (the hidden layers reuse the input tile's space)

template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_color_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *X = smem;                            // [64][LDX]
    float *H1 = X;                              // [64][LDC]  (over X once layer 1 has read it)
    float *H2 = X + PN_CTILE * LDC;             // [64][LDC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;

    for (long long tile = blockIdx.x; tile * PN_CTILE < Ns; tile += gridDim.x) {
        const long long grow0 = tile * PN_CTILE;
        __syncthreads();
        {
            const int row = tid >> 2, q = tid & 3;
            const long long vs = grow0 + row;
            float *xr = X + row * LDX;
            const int si = vs < Ns ? a.valid_list[vs] : -1;
            if (si >= 0) {
                const float *f = a.sv.fs + vs * PN_H + q * 64;
#pragma unroll
                for (int c = 0; c < 64; c += 4) *reinterpret_cast<float4 *>(xr + q * 64 + c) = *reinterpret_cast<const float4 *>(f + c);
                if (q == 0) {
                    const int r = si / a.SR;
                    float v[3];
                    rot3(a.cam.rw2c, a.raydir[3 * r], a.raydir[3 * r + 1], a.raydir[3 * r + 2], true, v[0], v[1], v[2]);
                    // positional_encoding(viewdirs, 4, ori=True)[..., 3:] = [sin(v_d 2^f) (d-major) | cos(...)]   networks.py:185-187
#pragma unroll
                    for (int dd = 0; dd < 3; ++dd) {
                        float fr = 1.f;
#pragma unroll
                        for (int f2 = 0; f2 < 4; ++f2) {
                            float s, c;
                            sincosf(v[dd] * fr, &s, &c);
                            xr[PN_H + dd * 4 + f2] = s;
                            xr[PN_H + 12 + dd * 4 + f2] = c;
                            fr *= 2.f;
                        }
                    }
#pragma unroll
                    for (int j = PN_INC; j < LDX; ++j) xr[j] = 0.f;
                    if (TRAIN) {
#pragma unroll
                        for (int j = 0; j < 32; ++j) a.sv.pe[vs * 32 + j] = j < 24 ? xr[PN_H + j] : 0.f;
                    }
                }
            } else {
                for (int j = q; j < LDX; j += 4) xr[j] = 0.f;
            }
        }
        __syncthreads();
        f32x16 acc[2][1];
        pn_acc_init_bias<2, 1>(acc, P + PO_BC1, wave, lane);
        pn_tile_gemm<2, 1>(X, LDX, PN_INC / 8, a.packed + PK_C1 / 4, wave, lane, acc);
        __syncthreads();                        // every wave is done reading X before H1 takes its place
        pn_acc_to_lds<2, 1, true>(acc, H1, LDC, wave, lane);
        __syncthreads();
        if (TRAIN) pn_tile_copy_out<PN_CTILE, PN_HC>(H1, LDC, a.sv.c1, PN_HC, grow0, tid);
        pn_acc_init_bias<2, 1>(acc, P + PO_BC2, wave, lane);
        pn_tile_gemm<2, 1>(H1, LDC, PN_HC / 8, a.packed + PK_C2 / 4, wave, lane, acc);
        pn_acc_to_lds<2, 1, true>(acc, H2, LDC, wave, lane);
        __syncthreads();
        if (TRAIN) pn_tile_copy_out<PN_CTILE, PN_HC>(H2, LDC, a.sv.c2, PN_HC, grow0, tid);
        pn_acc_init_bias<2, 1>(acc, P + PO_BC3, wave, lane);
        pn_tile_gemm<2, 1>(H2, LDC, PN_HC / 8, a.packed + PK_C3 / 4, wave, lane, acc);
        pn_acc_to_lds<2, 1, true>(acc, H1, LDC, wave, lane);
        __syncthreads();
        if (TRAIN) pn_tile_copy_out<PN_CTILE, PN_HC>(H1, LDC, a.sv.c3, PN_HC, grow0, tid);
        {
            const int row = tid >> 2, q = tid & 3;
            const float *h = H1 + row * LDC + q * 32;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float hv = h[c];
                o0 += hv * P[PO_WC4 + q * 32 + c];
                o1 += hv * P[PO_WC4 + PN_HC + q * 32 + c];
                o2 += hv * P[PO_WC4 + 2 * PN_HC + q * 32 + c];
            }
            o0 = group_sum<4>(o0); o1 = group_sum<4>(o1); o2 = group_sum<4>(o2);
            const long long vs = grow0 + row;
            if (q == 0 && vs < Ns) {
                const int si = a.valid_list[vs];
                float *o = a.decoded + (long long)si * 4;
                o[1] = 1.0f / (1.0f + expf(-(o0 + P[PO_BC4]))) * 1.002f - 0.001f;                                     // raw2out_color :269-273
                o[2] = 1.0f / (1.0f + expf(-(o1 + P[PO_BC4 + 1]))) * 1.002f - 0.001f;
                o[3] = 1.0f / (1.0f + expf(-(o2 + P[PO_BC4 + 2]))) * 1.002f - 0.001f;
            }
        }
    }
}
}  // namespace

// shared with render.hip
int pn_agg_forward_launch(const pnerf_camera *cam, const pnerf_points *pts, const float *d_params, const void *d_packed,
                          const float *d_raydir, const float *d_sample_loc, const float *d_xyz_pers, const float *d_loc_pers,
                          const int32_t *d_sample_pidx,
                          const int32_t *d_valid_list, const int32_t *d_counters, int R, int SR, int K,
                          float *d_decoded, float *d_weight, const PnSaved &sv, long long cap_samples, bool train,
                          hipStream_t s) {
    FwdArgs a;
    a.cam = *cam;
    a.xyz = pts->xyz; a.emb = pts->embedding; a.conf = pts->conf; a.dir = pts->dir; a.color = pts->color;
    a.params = d_params; a.packed = (const float4 *)d_packed;
    a.raydir = d_raydir; a.sample_loc = d_sample_loc; a.xyz_pers = d_xyz_pers; a.loc_pers = d_loc_pers; a.pidx = d_sample_pidx; a.valid_list = d_valid_list; a.counters = d_counters;
    a.R = R; a.SR = SR; a.K = K; a.TS = pn_tile_samples(K);
    a.cap_samples = cap_samples;
    a.decoded = d_decoded; a.weight = d_weight; a.sv = sv;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const long long ctiles = (cap_samples + PN_CTILE - 1) / PN_CTILE;
    const int grid_c = (int)(ctiles < 2 * ncu ? (ctiles > 0 ? ctiles : 1) : 2 * ncu);       // two workgroups per CU
    const size_t lds_a = F2_LDS_FLOATS * sizeof(float), lds_c = COL_LDS_FLOATS * sizeof(float);
    const bool pers = d_xyz_pers != nullptr;
    const void *kfn = train ? (pers ? (const void *)k_agg_forward2<true, true> : (const void *)k_agg_forward2<true, false>)
                            : (pers ? (const void *)k_agg_forward2<false, true> : (const void *)k_agg_forward2<false, false>);
    const void *cfn = train ? (const void *)k_color_forward<true> : (const void *)k_color_forward<false>;
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
    if (hipFuncSetAttribute(cfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
    int rc = pn_classify(sv, d_valid_list, d_counters, d_sample_pidx, K, cap_samples, train, s);
    if (rc) return rc;
    a.cls_list = sv.cls_list; a.cls_info = sv.cls_info; a.Kstride = K;
    int kc[PN_NCLS];
    const int ncls = pn_class_slots(K, kc);
    {
        PnProfScope prof(PNK_AGG_FWD, s);
        for (int j = 0; j < ncls; ++j) {            // class sizes are only known on the device: the grid covers the worst case, surplus workgroups return at once
            a.cls = j; a.K = kc[j]; a.TS = pn_tile_samples(kc[j]);
            const long long pairs = ((cap_samples + a.TS - 1) / a.TS + 1) / 2;
            const int grid_a = (int)(pairs < (long long)ncu ? (pairs > 0 ? pairs : 1) : ncu);
            if (train && pers) hipLaunchKernelGGL((k_agg_forward2<true, true>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (train) hipLaunchKernelGGL((k_agg_forward2<true, false>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else if (pers) hipLaunchKernelGGL((k_agg_forward2<false, true>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
            else hipLaunchKernelGGL((k_agg_forward2<false, false>), dim3(grid_a), dim3(PN_NTHR), lds_a, s, a);
        }
    }
    a.K = K; a.TS = pn_tile_samples(K); a.valid_list = sv.cls_list;      // the colour MLP walks the class-ordered list: f rows are in that order
    {
        PnProfScope prof(PNK_COLOR_FWD, s);
        if (train) hipLaunchKernelGGL(k_color_forward<true>, dim3(grid_c), dim3(256), lds_c, s, a);
        else hipLaunchKernelGGL(k_color_forward<false>, dim3(grid_c), dim3(256), lds_c, s, a);
    }
    PN_CHECK_LAUNCH();
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_fwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_fwd), bytes < sizeof(pn_trace_fwd) ? bytes : sizeof(pn_trace_fwd)) == hipSuccess ? 0 : -1;
}
#endif
