// b3core_probe.hip -- dev tool (not part of the product): how fast is the aggregator's 64-row layer GEMM (64 x 256 x 256 per
// tile, A = activation tile in LDS as fp32, B = weights streamed from L2) when the fp32 products run on the bf16 MFMA with the
// exact three-way split of pointnerf_amd/csrc/backward.hip (k_wgrad_b3)?  Every wave splits its own A fragments in registers
// (the LDS tile stays fp32), the weights are pre-split into three bf16 planes.  Prints the fp32-equivalent TFLOP/s for
//   pure      : 24 MFMAs per 16 columns of K, operands already in registers (the matrix-pipe ceiling at the clock it gets)
//   core      : + fp32 A fragments from LDS, split in the MFMA shadows, + B planes from L2
//   core+side : + SIDE dummy VALU operations per MFMA slot (stand-in for the hosted epilogues / tile-boundary work)
// hipcc --offload-arch=gfx950 -O3 b3core_probe.hip -o b3core_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDH = 260, K = 256, NCH = K / 16;

template <int... I, class F> __device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F &&f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ unsigned pack(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ float lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float up(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

struct Planes { uint4 h, m, l; };

// MODE 0: pure, 1: core, 2: core + SIDE VALU per slot
template <int MODE, int SIDE>
__global__ __launch_bounds__(256) void k_core(const uint4 *__restrict__ W, int iters, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * LDH; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) & 15) * 0.001f + 0.5f;
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap = smem + (lane & 31) * LDH + 8 * (lane >> 5);
    // weight image: [chunk][wave][ct][plane][64 lanes] uint4
    const uint4 *wp = W + (size_t)wave * 6 * 64 + lane;
    float side = (float)lane;
    float4 raw[2][2];               // next chunk's A: [mt][half]
    Planes an[2], ac[2];            // A planes of the next / current chunk per mt
    uint4 bn[2][3], bc[2][3];       // B planes [ct][plane]
    auto load_raw = [&](int c) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            raw[mt][0] = *reinterpret_cast<const float4 *>(ap + mt * 32 * LDH + 16 * c);
            raw[mt][1] = *reinterpret_cast<const float4 *>(ap + mt * 32 * LDH + 16 * c + 4);
        }
    };
    auto load_b = [&](int c, uint4 (&b)[2][3]) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[ct][p] = wp[((size_t)c * 4 * 6 + ct * 3 + p) * 64];
    };
    auto split_pair = [&](float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
        h = pack(x0, x1);
        const float r0 = x0 - lo(h), r1 = x1 - up(h);
        m = pack(r0, r1);
        l = pack(r0 - lo(m), r1 - up(m));
    };
    auto split_all = [&](Planes (&a)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            split_pair(raw[mt][0].x, raw[mt][0].y, a[mt].h.x, a[mt].m.x, a[mt].l.x);
            split_pair(raw[mt][0].z, raw[mt][0].w, a[mt].h.y, a[mt].m.y, a[mt].l.y);
            split_pair(raw[mt][1].x, raw[mt][1].y, a[mt].h.z, a[mt].m.z, a[mt].l.z);
            split_pair(raw[mt][1].z, raw[mt][1].w, a[mt].h.w, a[mt].m.w, a[mt].l.w);
        }
    };
    load_raw(0); split_all(ac); load_b(0, bc);
    for (int it = 0; it < iters; ++it) {
        sfor<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if (MODE > 0) { load_raw((c + 1) % NCH); load_b((c + 1) % NCH, bn); }
            // 24 MFMAs: term t = s / 4 in {hl, hm, hh, mm, mh, lh}, tile = s % 4
            sfor<24>([&](auto ss) {
                constexpr int s = decltype(ss)::value, t = s / 4, mt = s & 1, ct = (s >> 1) & 1;
                const uint4 av = t < 3 ? ac[mt].h : (t < 5 ? ac[mt].m : ac[mt].l);
                const uint4 bv = (t == 0) ? bc[ct][2] : ((t == 1 || t == 3) ? bc[ct][1] : bc[ct][0]);
                acc[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[mt][ct], 0, 0, 0);
                if constexpr (MODE > 0 && s >= 4 && s < 20 && s % 2 == 0) {       // 8 pair splits of the next chunk's A, one per other slot
                    constexpr int pi = (s - 4) / 2, mt2 = pi / 4, q = pi % 4;
                    const float x0 = q == 0 ? raw[mt2][0].x : q == 1 ? raw[mt2][0].z : q == 2 ? raw[mt2][1].x : raw[mt2][1].z;
                    const float x1 = q == 0 ? raw[mt2][0].y : q == 1 ? raw[mt2][0].w : q == 2 ? raw[mt2][1].y : raw[mt2][1].w;
                    unsigned h, m, l;
                    split_pair(x0, x1, h, m, l);
                    if (q == 0) { an[mt2].h.x = h; an[mt2].m.x = m; an[mt2].l.x = l; }
                    else if (q == 1) { an[mt2].h.y = h; an[mt2].m.y = m; an[mt2].l.y = l; }
                    else if (q == 2) { an[mt2].h.z = h; an[mt2].m.z = m; an[mt2].l.z = l; }
                    else { an[mt2].h.w = h; an[mt2].m.w = m; an[mt2].l.w = l; }
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < SIDE; ++k) side = side * 1.0001f + 0.5f;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (MODE > 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) ac[mt] = an[mt];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bc[ct][p] = bn[ct][p];
            }
        });
    }
    float s = side;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}


// MODE 3 ("lean"): the same GEMM with the register budget a production kernel can afford -- 80 registers for the core instead
// of the ~190 the compiler takes when left alone: only `ah` and `bh` are double-buffered, the m / l planes and bm / bl are
// reloaded in place as soon as their last MFMA of the chunk has issued.  Term order per 16 columns of K:
//   [0-3] am*bm  [4-7] ah*bm  [8-11] am*bh  [12-15] al*bh  [16-19] ah*bh  [20-23] ah*bl
// next chunk: bh' + raw A at slots 0-3, h-stage (ah', residual) 4-11, bm' 8-9, m-stage (am') 12-19, l-stage (al') 16-23, bl' after 23.
template <int SIDE>
__global__ __launch_bounds__(256) void k_lean(const uint4 *__restrict__ W, int iters, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * LDH; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) & 15) * 0.001f + 0.5f;
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap = smem + (lane & 31) * LDH + 8 * (lane >> 5);
    const uint4 *wp = W + (size_t)wave * 6 * 64 + lane;      // [chunk][wave][ct*3 + plane(0 h, 1 m, 2 l)][lane]
    float side = (float)lane;
    uint4 ah[2][2], bh[2][2];        // [generation][mt or ct]
    uint4 am[2], al[2], bm[2], bl[2];
    float4 raw[2][2];                // [mt][half]: raw values, then residuals in place
    auto get = [](const uint4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; };
    auto put = [](uint4 &v, int q, unsigned x) { if (q == 0) v.x = x; else if (q == 1) v.y = x; else if (q == 2) v.z = x; else v.w = x; };
    // prologue: chunk 0 completely
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        raw[mt][0] = *reinterpret_cast<const float4 *>(ap + mt * 32 * LDH);
        raw[mt][1] = *reinterpret_cast<const float4 *>(ap + mt * 32 * LDH + 4);
        float x[8] = {raw[mt][0].x, raw[mt][0].y, raw[mt][0].z, raw[mt][0].w, raw[mt][1].x, raw[mt][1].y, raw[mt][1].z, raw[mt][1].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned h = pack(x[2 * q], x[2 * q + 1]);
            const float r0 = x[2 * q] - lo(h), r1 = x[2 * q + 1] - up(h);
            const unsigned m = pack(r0, r1);
            put(ah[0][mt], q, h); put(am[mt], q, m); put(al[mt], q, pack(r0 - lo(m), r1 - up(m)));
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { bh[0][ct] = wp[(ct * 3 + 0) * 64]; bm[ct] = wp[(ct * 3 + 1) * 64]; bl[ct] = wp[(ct * 3 + 2) * 64]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int c2 = 0; c2 < NCH / 2; ++c2) {
        sfor<2>([&](auto gg) {
            constexpr int g = decltype(gg)::value, n = g ^ 1;
            const int c = 2 * c2 + g, cn = (c + 1) % NCH;
            const uint4 *wn = wp + (size_t)cn * 4 * 6 * 64;
            const float *an = ap + 16 * cn;
            sfor<24>([&](auto ss) {
                constexpr int s = decltype(ss)::value, t = s / 4, mt = s & 1, ct = (s >> 1) & 1;
                const uint4 av = (t == 0 || t == 2) ? am[mt] : (t == 3 ? al[mt] : ah[g][mt]);
                const uint4 bv = (t == 0 || t == 1) ? bm[ct] : (t == 5 ? bl[ct] : bh[g][ct]);
                acc[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[mt][ct], 0, 0, 0);
                // ---- next chunk's operands
                if constexpr (s < 2) bh[n][s] = wn[(s * 3 + 0) * 64];
                if constexpr (s < 4) raw[s >> 1][s & 1] = *reinterpret_cast<const float4 *>(an + (s >> 1) * 32 * LDH + 4 * (s & 1));
                if constexpr (s == 8 || s == 9) bm[s - 8] = wn[((s - 8) * 3 + 1) * 64];
                if constexpr (s >= 4 && s < 12) {                      // h-stage: pair p of the 8
                    constexpr int p = s - 4, m2 = p / 4, q = p % 4;
                    float4 &v = raw[m2][q >> 1];
                    float &x0 = (q & 1) ? v.z : v.x, &x1 = (q & 1) ? v.w : v.y;
                    const unsigned h = pack(x0, x1);
                    put(ah[n][m2], q, h);
                    x0 -= lo(h); x1 -= up(h);
                }
                if constexpr (s >= 12 && s < 20) {                     // m-stage (am is free: its last MFMA was slot 11)
                    constexpr int p = s - 12, m2 = p / 4, q = p % 4;
                    float4 &v = raw[m2][q >> 1];
                    float &x0 = (q & 1) ? v.z : v.x, &x1 = (q & 1) ? v.w : v.y;
                    const unsigned m = pack(x0, x1);
                    put(am[m2], q, m);
                    x0 -= lo(m); x1 -= up(m);
                }
                if constexpr (s >= 20) {                               // l-stage, two pairs per slot (al is free since slot 15)
                    constexpr int p0 = 2 * (s - 20);
                    sfor<2>([&](auto pp) {
                        constexpr int p = p0 + decltype(pp)::value, m2 = p / 4, q = p % 4;
                        const float4 &v = raw[m2][q >> 1];
                        put(al[m2], q, (q & 1) ? pack(v.z, v.w) : pack(v.x, v.y));
                    });
                }
                if constexpr (SIDE > 0) {
#pragma unroll
                    for (int k = 0; k < SIDE; ++k) side = side * 1.0001f + 0.5f;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // bl' after its last use (slot 23): needed at slot 20 of the next chunk
            bl[0] = wn[2 * 64]; bl[1] = wn[5 * 64];
        });
        }
    }
    float s = side;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}

template <int SIDE>
void run_lean(const uint4 *W, float *out, int iters, const char *name) {
    const size_t lds = 64 * LDH * 4;
    hipFuncSetAttribute((const void *)k_lean<SIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_lean<SIDE>), dim3(256), dim3(256), lds, 0, W, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 2.0 * 64 * 256 * 256 * (double)iters * 256;
    printf("{\"probe\": \"b3core\", \"variant\": \"%s\", \"ms\": %.3f, \"fp32_equiv_TFLOPs\": %.1f, \"us_per_tile_layer\": %.2f}\n", name, ms, flop / ms * 1e-9, ms * 1e3 / iters);
}

template <int MODE, int SIDE>
void run(const uint4 *W, float *out, int iters, const char *name) {
    const size_t lds = 64 * LDH * 4;
    hipFuncSetAttribute((const void *)k_core<MODE, SIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_core<MODE, SIDE>), dim3(256), dim3(256), lds, 0, W, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 2.0 * 64 * 256 * 256 * (double)iters * 256;
    printf("{\"probe\": \"b3core\", \"variant\": \"%s\", \"ms\": %.3f, \"fp32_equiv_TFLOPs\": %.1f, \"us_per_tile_layer\": %.2f}\n", name, ms, flop / ms * 1e-9, ms * 1e3 / iters);
}

int main() {
    const size_t nW = (size_t)NCH * 4 * 6 * 64;          // uint4 of one pre-split 256 x 256 image (393 KB)
    uint4 *W; float *out;
    hipMalloc(&W, nW * 16); hipMalloc(&out, 256 * 256 * 4);
    std::vector<unsigned> h(nW * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u + (unsigned)((i * 13) & 7) * 0x00010001u;      // small finite bf16 pairs
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    run<0, 0>(W, out, iters, "pure (operands in registers)");
    run<1, 0>(W, out, iters, "core (A fp32 from LDS split in registers, B planes from L2)");
    run<2, 2>(W, out, iters, "core + 2 side VALU per MFMA");
    run<2, 4>(W, out, iters, "core + 4 side VALU per MFMA");
    run<2, 8>(W, out, iters, "core + 8 side VALU per MFMA");
    run_lean<0>(W, out, iters, "lean core (80 registers: ah / bh double-buffered, the rest reloaded in place)");
    run_lean<2>(W, out, iters, "lean core + 2 side VALU per MFMA");
    run_lean<4>(W, out, iters, "lean core + 4 side VALU per MFMA");
    return 0;
}
