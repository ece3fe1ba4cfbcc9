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
    const long long tiles = (n_valid + TS - 1) / TS + 1;
    const long long rows = tiles * PN_TILE;
    const long long samples = ((tiles * TS + PN_CTILE - 1) / PN_CTILE + 1) * PN_CTILE;
    if (rows_out) *rows_out = rows;
    if (samples_out) *samples_out = samples;
    size_t b = 0;
    b += pn_align((size_t)rows * PN_IN1P * 4) + 8 * pn_align((size_t)rows * PN_H * 4) + pn_align((size_t)rows * 8 * 4) + pn_align((size_t)rows * 16);
    b += pn_align((size_t)tiles * 3 * PN_NTHR * 8);
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
    return s;
}

extern "C" size_t pnerf_agg_saved_bytes(int64_t n_valid_samples, int K) {
    if (K <= 0 || K > PNERF_MAX_K || n_valid_samples < 0) return 0;
    return pn_saved_bytes(n_valid_samples, K, nullptr, nullptr);
}

// ------------------------------------------------------------------------------ forward kernels
namespace {
constexpr int LDX = 292;    // X0 / colour input row stride in LDS (odd multiple of 4 floats: conflict-free b128 reads)
constexpr int LDH = 260;    // hidden row stride
constexpr int LDC = 132;    // colour hidden row stride
constexpr int TPR = PN_TPR; // threads per tile row in the element-wise phases
constexpr int EPT = PN_F / TPR;              // embedding dims per thread in the feature build
constexpr int CPT = PN_H / TPR;              // hidden columns per thread in the row-wise dot products
constexpr int AGG_LDS_FLOATS = PN_TILE * LDX + PN_TILE * 8 + PN_TILE * 8 + 4 * PN_TILE + PN_H + PN_TILE;
constexpr int AGG_WG_PER_CU = (160 * 1024) / (AGG_LDS_FLOATS * 4);

struct FwdArgs {
    pnerf_camera cam;
    const float *xyz, *emb, *conf, *dir, *color;
    const float *params;
    const float4 *packed;
    const float *raydir, *sample_loc;
    const float *xyz_pers, *loc_pers;   // optional: perspective coords supplied by the caller (stand-alone aggregator)
    const int *pidx, *valid_list, *counters;
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

// One LDS activation buffer, updated in place (GEMM -> barrier -> epilogue -> barrier): with 32-row tiles 46 KB per
// workgroup, so three workgroups share a CU and one's gather / epilogue latency hides under the others' MFMA phases.
#ifdef PN_PHASE_TRACE
PN_TR_DECL(pn_trace_fwd);
#endif
template <bool TRAIN>
__global__ __launch_bounds__(PN_NTHR, PN_NTHR == 512 ? 4 : (PN_TILE == 32 ? 3 : 2)) void k_agg_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufA = smem;                         // [PN_TILE][LDX]  X0, then h1..h4 at stride LDH
    float *exb = bufA + PN_TILE * LDX;          // [PN_TILE][8]  layer-3 extras
    float *dst = exb + PN_TILE * 8;             // [PN_TILE][8]  the 6 distance components of each row
    float *wraw = dst + PN_TILE * 8;            // [PN_TILE] raw 1/dist weights, later alpha*w
    float *wrow = wraw + PN_TILE;               // final weight (normalised * clamped conf)
    float *wnrm = wrow + PN_TILE;               // normalised weight
    float *rawa = wnrm + PN_TILE;               // (spare)
    float *w5s = rawa + PN_TILE;                // [256]
    int *sidx = reinterpret_cast<int *>(w5s + PN_H);   // [<=PN_TILE] sample ids of this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, TS = a.TS;
    const int Ns = a.counters[0] < a.cap_samples ? a.counters[0] : (int)a.cap_samples;
    const float *P = a.params;
    if (tid < PN_H) w5s[tid] = P[PO_W5 + tid];
    const float b5 = P[PO_B5];

#ifdef PN_PHASE_TRACE
    int titer = -1;
#endif
    for (long long tile = blockIdx.x; tile * TS < Ns; tile += gridDim.x) {
        const long long grow0 = tile * PN_TILE;
#ifdef PN_PHASE_TRACE
        ++titer;
#endif
        __syncthreads();
        PN_TR(pn_trace_fwd, 0); PN_TR_HWID(pn_trace_fwd);
        if (tid < PN_TILE) {
            const long long vs = tile * TS + tid;
            sidx[tid] = (tid < TS && vs < Ns) ? a.valid_list[vs] : -1;
        }
        __syncthreads();
        // ---- P1: gather + feature build -----------------------------------------------------
        {
            const int row = tid / TPR, q = tid % TPR;
            const int ls = row / K, k = row - ls * K;
            const int si = ls < TS ? sidx[ls] : -1;
            int p = -1;
            if (si >= 0) p = a.pidx[(long long)si * K + k];
            float *xa = bufA + row * LDX;
            if (p >= 0) {
                const float lx = a.sample_loc[(long long)si * 3], ly = a.sample_loc[(long long)si * 3 + 1], lz = a.sample_loc[(long long)si * 3 + 2];
                const float px = a.xyz[3 * p], py = a.xyz[3 * p + 1], pz = a.xyz[3 * p + 2];
                const float dwx = px - lx, dwy = py - ly, dwz = pz - lz;
                float ppx, ppy, pcz, spx, spy, scz;
                if (a.xyz_pers) {                      // PointAggregator.forward(sampled_xyz_pers, sample_loc) inputs
                    ppx = a.xyz_pers[3 * p]; ppy = a.xyz_pers[3 * p + 1]; pcz = a.xyz_pers[3 * p + 2];
                    spx = a.loc_pers[(long long)si * 3]; spy = a.loc_pers[(long long)si * 3 + 1]; scz = a.loc_pers[(long long)si * 3 + 2];
                } else {                               // fused path: project in-kernel (neural_points.py:604-610)
                    float pcx, pcy, scx, scy;
                    rot3(a.cam.camrot, px - a.cam.campos[0], py - a.cam.campos[1], pz - a.cam.campos[2], false, pcx, pcy, pcz);
                    rot3(a.cam.camrot, lx - a.cam.campos[0], ly - a.cam.campos[1], lz - a.cam.campos[2], false, scx, scy, scz);
                    ppx = pcx / pcz; ppy = pcy / pcz; spx = scx / scz; spy = scy / scz;
                }
                float d[6];
                rot3(a.cam.rw2c, dwx, dwy, dwz, true, d[0], d[1], d[2]);            // dists[:3] @ Rw2c^T (point_aggregators.py:526)
                d[3] = ppx * pcz - spx * scz; d[4] = ppy * pcz - spy * scz; d[5] = pcz - scz;   // :775-777
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) dst[row * 8 + j] = d[j];
                }
                // embedding + PE3(embedding): EPT dims per thread
                const float *ep = a.emb + (long long)p * PN_F + EPT * q;
#pragma unroll
                for (int i = 0; i < EPT; i += 4) *reinterpret_cast<float4 *>(xa + EPT * q + i) = *reinterpret_cast<const float4 *>(ep + i);
                // sin/cos(e 2^f): one accurate sincosf at the base frequency, then exact double-angle steps
                // (sin 2x = 2 s c, cos 2x = 1 - 2 s^2): 38 instead of 126 sincosf per row; |error| grows ~2x per octave
                // from <= 1 ulp, i.e. <= 2e-6 at the 16x band, far inside the 1e-4 bar.
#pragma unroll 2
                for (int i = 0; i < EPT; ++i) {
                    const int dd = EPT * q + i;
                    float s, c;
                    sincosf(xa[dd], &s, &c);
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        *reinterpret_cast<float2 *>(xa + PN_F + (dd * 3 + f) * 2) = make_float2(s, c);
                        const float s2 = 2.f * s * c;
                        c = 1.f - 2.f * s * s; s = s2;
                    }
                }
                // PE5(dists6): 6 components x 5 octaves = 30 (sin,cos) pairs; thread q takes components q, q+TPR, ...
                __builtin_amdgcn_wave_barrier();
                for (int comp = q; comp < 6; comp += TPR) {
                    float s, c;
                    sincosf(dst[row * 8 + comp], &s, &c);
#pragma unroll
                    for (int f = 0; f < 5; ++f) {
                        *reinterpret_cast<float2 *>(xa + PN_F * 7 + (comp * 5 + f) * 2) = make_float2(s, c);
                        const float s2 = 2.f * s * c;
                        c = 1.f - 2.f * s * s; s = s2;
                    }
                }
                if (q == TPR - 1) {
#pragma unroll
                    for (int j = PN_IN1; j < LDX; ++j) xa[j] = 0.f;
                }
                if (q == 0) {
                    float vx, vy, vz, qx, qy, qz;
                    const int r = si / a.SR;
                    rot3(a.cam.rw2c, a.raydir[3 * r], a.raydir[3 * r + 1], a.raydir[3 * r + 2], true, vx, vy, vz);      // :506
                    rot3(a.cam.rw2c, a.dir[3 * p], a.dir[3 * p + 1], a.dir[3 * p + 2], true, qx, qy, qz);               // :566
                    float *ex = exb + row * 8;
                    ex[0] = a.color[3 * p]; ex[1] = a.color[3 * p + 1]; ex[2] = a.color[3 * p + 2];
                    ex[3] = qx - vx; ex[4] = qy - vy; ex[5] = qz - vz;
                    ex[6] = qx * vx + qy * vy + qz * vz; ex[7] = 0.f;
                    wraw[row] = 1.0f / fmaxf(sqrtf(dwx * dwx + dwy * dwy + dwz * dwz), 1e-6f);                            // linear :425-428
                }
            } else {
                for (int j = q; j < LDX; j += TPR) xa[j] = 0.f;
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) exb[row * 8 + j] = 0.f;
                    wraw[row] = 0.f;
                }
            }
        }
        __syncthreads();
        PN_TR(pn_trace_fwd, 1);
        // ---- P2: normalise weights over the K slots, multiply by the clamped confidence -------
        if (tid < PN_TILE) {
            const int row = tid, ls = row / K, k = row - ls * K;
            const int si = ls < TS ? sidx[ls] : -1;
            float wn = 0.f, w = 0.f;
            int p = -1;
            if (si >= 0) {
                float sum = 0.f;
                for (int kk = 0; kk < K; ++kk) sum += wraw[ls * K + kk];
                wn = wraw[row] / fmaxf(sum, 1e-8f);                                                                   // :801-802
                p = a.pidx[(long long)si * K + k];
                const float cf = a.conf[p >= 0 ? p : 0];
                w = wn * fminf(fmaxf(cf, 1e-4f), 1.0f);                                                               // :807-811
                a.weight[(long long)si * K + k] = wn;
            }
            wnrm[row] = wn; wrow[row] = w;
            if (TRAIN) a.sv.rmeta[grow0 + row] = make_int4(si, p, __float_as_int(wn), __float_as_int(w));
        }
        if (TRAIN) {
            for (int e = tid; e < PN_TILE * (PN_IN1P / 4); e += PN_NTHR) {
                const int row = e / (PN_IN1P / 4), c4 = e - row * (PN_IN1P / 4);
                *reinterpret_cast<float4 *>(a.sv.x0 + (grow0 + row) * PN_IN1P + c4 * 4) = *reinterpret_cast<const float4 *>(bufA + row * LDX + c4 * 4);
            }
            if (tid < PN_TILE * 2) {
                const int row = tid >> 1, h = tid & 1;
                *reinterpret_cast<float4 *>(a.sv.ex + (grow0 + row) * 8 + h * 4) = *reinterpret_cast<const float4 *>(exb + row * 8 + h * 4);
            }
        }
        // ---- layers (in place: all waves finish reading A before anyone overwrites it) ---------------
        f32x16 acc[PN_MT][PN_NT];
        PN_TR(pn_trace_fwd, 2);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, P + PO_B1, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(bufA, LDX, PN_IN1P / 8, a.packed + PK_F1 / 4, wave, lane, acc);
        PN_TR(pn_trace_fwd, 3);
        __syncthreads();
        {
            const unsigned long long mbits = pn_acc_to_lds_bits<true>(acc, bufA, LDH, wave, lane);
            if (TRAIN) a.sv.lmask[(tile * 3 + 0) * PN_NTHR + tid] = mbits;
        }
        __syncthreads();
        PN_TR(pn_trace_fwd, 4);
        if (TRAIN) pn_tile_copy_out<PN_TILE, PN_H, PN_NTHR>(bufA, LDH, a.sv.h1, PN_H, grow0, tid);
        PN_TR(pn_trace_fwd, 5);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, P + PO_B2, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(bufA, LDH, PN_H / 8, a.packed + PK_F2 / 4, wave, lane, acc);
        PN_TR(pn_trace_fwd, 6);
        __syncthreads();
        {
            const unsigned long long mbits = pn_acc_to_lds_bits<true>(acc, bufA, LDH, wave, lane);
            if (TRAIN) a.sv.lmask[(tile * 3 + 1) * PN_NTHR + tid] = mbits;
        }
        __syncthreads();
        PN_TR(pn_trace_fwd, 7);
        if (TRAIN) pn_tile_copy_out<PN_TILE, PN_H, PN_NTHR>(bufA, LDH, a.sv.h2, PN_H, grow0, tid);
        PN_TR(pn_trace_fwd, 8);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, P + PO_B3, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(bufA, LDH, PN_H / 8, a.packed + PK_F3 / 4, wave, lane, acc);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(exb, 8, 1, a.packed + PK_F3 / 4 + (PN_H / 8) * (PN_H / 32) * 64, wave, lane, acc);
        PN_TR(pn_trace_fwd, 9);
        __syncthreads();
        {
            const unsigned long long mbits = pn_acc_to_lds_bits<true>(acc, bufA, LDH, wave, lane);
            if (TRAIN) a.sv.lmask[(tile * 3 + 2) * PN_NTHR + tid] = mbits;
        }
        __syncthreads();
        PN_TR(pn_trace_fwd, 10);
        if (TRAIN) pn_tile_copy_out<PN_TILE, PN_H, PN_NTHR>(bufA, LDH, a.sv.h3, PN_H, grow0, tid);
        PN_TR(pn_trace_fwd, 11);
        pn_acc_init_bias<PN_MT, PN_NT>(acc, P + PO_B4, wave, lane);
        pn_tile_gemm<PN_MT, PN_NT, PN_NW>(bufA, LDH, PN_H / 8, a.packed + PK_F4 / 4, wave, lane, acc);
        PN_TR(pn_trace_fwd, 12);
        __syncthreads();
        pn_acc_to_lds<PN_MT, PN_NT, true>(acc, bufA, LDH, wave, lane);
        __syncthreads();
        PN_TR(pn_trace_fwd, 13);
        if (TRAIN) pn_tile_copy_out<PN_TILE, PN_H, PN_NTHR>(bufA, LDH, a.sv.h4, PN_H, grow0, tid);
        PN_TR(pn_trace_fwd, 14);
        // ---- P5: alpha head (256 -> 1, softplus(x - 1)) -----------------------------------------
        {
            const int row = tid / TPR, q = tid % TPR;
            const float *h = bufA + row * LDH + q * CPT;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(h + c);
                s += v.x * w5s[q * CPT + c] + v.y * w5s[q * CPT + c + 1] + v.z * w5s[q * CPT + c + 2] + v.w * w5s[q * CPT + c + 3];
            }
            s = group_sum<TPR>(s);
            if (q == 0) {
                const float x = s + b5 - 1.0f;
                const float alpha = x > 20.f ? x : log1pf(expf(x));                                                   // raw2out_density :262-265
                wraw[row] = alpha * wrow[row];
            }
        }
        __syncthreads();
        PN_TR(pn_trace_fwd, 15);
        // ---- P6: K-weighted sums -> sigma, f[256] -------------------------------------------------
        for (int e = tid; e < TS * 64; e += PN_NTHR) {
            const int ls = e >> 6, c4 = e & 63;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < K; ++k) {
                const float w = wrow[ls * K + k];
                const float4 v = *reinterpret_cast<const float4 *>(bufA + (ls * K + k) * LDH + c4 * 4);
                f.x += w * v.x; f.y += w * v.y; f.z += w * v.z; f.w += w * v.w;
            }
            const long long vs = tile * TS + ls;
            if (vs < a.cap_samples) *reinterpret_cast<float4 *>(a.sv.fs + vs * PN_H + c4 * 4) = f;
        }
        if (tid < TS) {
            const int si = sidx[tid];
            if (si >= 0) {
                float sg = 0.f;
                for (int k = 0; k < K; ++k) sg += wraw[tid * K + k];
                a.decoded[(long long)si * 4] = sg;
            }
        }
        PN_TR(pn_trace_fwd, 16);
    }
}

constexpr int COL_LDS_FLOATS = PN_CTILE * LDX + 2 * PN_CTILE * LDC + 32;

template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void k_color_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *X = smem;                            // [64][LDX]
    float *H1 = X + PN_CTILE * LDX;             // [64][LDC]
    float *H2 = H1 + PN_CTILE * LDC;            // [64][LDC]
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
    const long long tiles = (cap_samples + a.TS - 1) / a.TS;
    const int wgcu = AGG_WG_PER_CU < 1 ? 1 : (AGG_WG_PER_CU > 4 ? 4 : AGG_WG_PER_CU);
    const int grid_a = (int)(tiles < (long long)wgcu * ncu ? (tiles > 0 ? tiles : 1) : wgcu * ncu);   // as many workgroups per CU as the LDS admits
    const long long ctiles = (cap_samples + PN_CTILE - 1) / PN_CTILE;
    const int grid_c = (int)(ctiles < ncu ? (ctiles > 0 ? ctiles : 1) : ncu);
    const size_t lds_a = AGG_LDS_FLOATS * sizeof(float), lds_c = COL_LDS_FLOATS * sizeof(float);
    if (train) {
        if (hipFuncSetAttribute((const void *)k_agg_forward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
        if (hipFuncSetAttribute((const void *)k_color_forward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
        { PnProfScope prof(PNK_AGG_FWD, s); hipLaunchKernelGGL(k_agg_forward<true>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a); }
        { PnProfScope prof(PNK_COLOR_FWD, s); hipLaunchKernelGGL(k_color_forward<true>, dim3(grid_c), dim3(256), lds_c, s, a); }
    } else {
        if (hipFuncSetAttribute((const void *)k_agg_forward<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a) != hipSuccess) return PNERF_E_LAUNCH;
        if (hipFuncSetAttribute((const void *)k_color_forward<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c) != hipSuccess) return PNERF_E_LAUNCH;
        { PnProfScope prof(PNK_AGG_FWD, s); hipLaunchKernelGGL(k_agg_forward<false>, dim3(grid_a), dim3(PN_NTHR), lds_a, s, a); }
        { PnProfScope prof(PNK_COLOR_FWD, s); hipLaunchKernelGGL(k_color_forward<false>, dim3(grid_c), dim3(256), lds_c, s, a); }
    }
    PN_CHECK_LAUNCH();
    return 0;
}

#ifdef PN_PHASE_TRACE
extern "C" int pnerf_debug_trace_fwd(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pn_trace_fwd), bytes < sizeof(pn_trace_fwd) ? bytes : sizeof(pn_trace_fwd)) == hipSuccess ? 0 : -1;
}
#endif
