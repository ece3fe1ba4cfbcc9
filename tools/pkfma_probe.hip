// dev probe (round 4): is a packed-fp32 result safe to STORE right behind the instruction that wrote it?
//
// Round 3's faulty build of the forward tail (aggregate.hip f_tail, SLP-vectorised) ended in
//     v_pk_fma_f32 v[176:177], ...            ; chain A, last step
//     v_pk_fma_f32 v[178:179], ...            ; chain B, last step  (A and B interleaved, one instruction apart, no s_nop)
//     global_store_dwordx4 v[162:163], v[184:187], off
//     global_store_dwordx4 v[162:163], v[176:179], off offset:16
// and in 1.1 % of the launches ONE 16-lane pass of v176 / v178 (the LOW registers of the two pairs) reached memory wrong.  This probe
// replays that shape in inline asm on fixed physical registers (the compiler pads nothing inside an asm block) and compares what reaches
// memory with the scalar result:
//   mode 0  chains interleaved as the compiler emitted them, stores right behind                          (the faulty shape)
//   mode 1  same chains, s_nop 4 between the last pk_fma and the stores                                    (hazard = VALU -> VMEM data read?)
//   mode 2  dependent pk_fma back to back inside ONE chain (no interleave, no s_nop), s_nop 4, stores       (hazard = dependent chain?)
//   mode 3  like 0, but the results pass through v_mov_b32 (plain full-register VALU writes) right before the stores   (any VALU -> VMEM?)
//   mode 4  like 0 with s_nop 0 between the last pk_fma and the stores
//   modes 5 / 6 / 7  every pk_fma of the interleaved chains followed by s_nop 7 / by s_waitcnt vmcnt(0) lgkmcnt(0) / by both (the padding of the
//           replay's reference stream), stores right behind
// bg: what waves 4 .. 7 of every workgroup (one per SIMD, beside the probing waves) do: 0 nothing, 1 MFMA, 2 VALU, 3 LDS, 4 all in turn (background()).
// usage: pkfma_probe [iters = 100000] [bg = 1]   |   pkfma_probe replay [iters = 20000] [bg = 1]   (the verbatim block, see k_replay)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct Hit { int mode, block, tid, iter; float got[4], want[4]; };

// results: v[126:127] (chain A), v[128:129] (chain B); the first store's data: v[130:133]; mode 3's copies: v[134:137]
#define PK_A(i) "v_pk_fma_f32 v[126:127], %[a" #i "], %[w], v[126:127] op_sel_hi:[1,0,1]\n\t"
#define PK_B(i) "v_pk_fma_f32 v[128:129], %[b" #i "], %[w], v[128:129] op_sel_hi:[1,0,1]\n\t"
#define PK_INIT  "v_mov_b32 v126, 1.0\n\tv_mov_b32 v127, 2.0\n\tv_mov_b32 v128, 4.0\n\tv_mov_b32 v129, 0.5\n\t" \
                 "v_mov_b32 v130, %[o]\n\tv_mov_b32 v131, %[o]\n\tv_mov_b32 v132, %[o]\n\tv_mov_b32 v133, %[o]\n\ts_nop 4\n\t"
#define PK_INTERLEAVED PK_A(0) PK_B(0) PK_A(1) PK_B(1) PK_A(2) PK_B(2) PK_A(3) PK_B(3) PK_A(4) PK_B(4) PK_A(5) PK_B(5)
#define PK_BACK2BACK   PK_A(0) PK_A(1) PK_A(2) PK_A(3) PK_A(4) PK_A(5) PK_B(0) PK_B(1) PK_B(2) PK_B(3) PK_B(4) PK_B(5)
#define PK_STORES      "global_store_dwordx4 %[p], v[130:133], off\n\tglobal_store_dwordx4 %[p], v[126:129], off offset:16\n\t"
#define PK_STORES_MOV  "v_mov_b32 v134, v126\n\tv_mov_b32 v135, v127\n\tv_mov_b32 v136, v128\n\tv_mov_b32 v137, v129\n\t" \
                       "global_store_dwordx4 %[p], v[130:133], off\n\tglobal_store_dwordx4 %[p], v[134:137], off offset:16\n\t"
#define PK_END   "s_nop 4\n\ts_waitcnt vmcnt(0)"
// padded chains (modes 5 .. 7): every instruction of the interleaved chains is followed by PAD
#define PK_PADDED(PAD) PK_A(0) PAD PK_B(0) PAD PK_A(1) PAD PK_B(1) PAD PK_A(2) PAD PK_B(2) PAD PK_A(3) PAD PK_B(3) PAD PK_A(4) PAD PK_B(4) PAD PK_A(5) PAD PK_B(5) PAD
#define PAD_NOP   "s_nop 7\n\t"
#define PAD_WAIT  "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define PAD_BOTH  "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"

#define PK_OPERANDS  : : [a0] "v"(xa[0]), [a1] "v"(xa[1]), [a2] "v"(xa[2]), [a3] "v"(xa[3]), [a4] "v"(xa[4]), [a5] "v"(xa[5]),                       \
                         [b0] "v"(xb[0]), [b1] "v"(xb[1]), [b2] "v"(xb[2]), [b3] "v"(xb[3]), [b4] "v"(xb[4]), [b5] "v"(xb[5]),                       \
                         [w] "v"(wv), [p] "v"(slot), [o] "v"(oth)                                                                                  \
                     : "memory", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137"

// what the OTHER waves of the SIMD do while a wave probes (waves 4 .. 7 of the workgroup: one per SIMD): bg = 1 MFMA chain; 2 VALU mix
// (fma chains, conversions, integer ops); 3 LDS traffic (b128 reads, b64 writes); 4 all of them in turn + global loads -- the neighbours a
// tile kernel's wave really has (the CU's second workgroup is somewhere else in its tile: GEMM, epilogue, feature build)
__device__ __noinline__ void background(int bg, int iters, int tid, float *__restrict__ sink, float *lds_bg, const float *__restrict__ gin) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)(tid + i)); b[i] = (_Float16)(0.002f * (float)(tid - i)); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.1f * (float)(tid + i);
    unsigned u = (unsigned)tid * 2654435761u;
    v4f l = {0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int what = bg == 4 ? (it & 3) + 1 : bg;
        if (what == 1 || what == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (what == 2 || what == 4) {
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, v[(j + 3) & 7] * 1e-3f);
                u = u * 1664525u + 1013904223u;
                v[(j + 1) & 7] += (float)(_Float16)((float)(u >> 20) * 1e-3f);
            }
        }
        if (what == 3 || what == 4) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const v4f t = *reinterpret_cast<const v4f *>(lds_bg + ((tid * 4 + j * 260 + it * 4) & 4092));
                l += t;
                *reinterpret_cast<v2f *>(lds_bg + ((tid * 2 + j * 130 + it * 6) & 4094)) = v2f{l[0], l[1]};
            }
        }
        if (what == 4) gsum += gin[(tid * 16 + it * 64) & 4095];
    }
    if (acc[0] + v[0] + v[3] + l[2] + gsum == 12345.678f) sink[tid] = acc[1] + v[1];
}

template <int MODE>
__device__ __forceinline__ void run_mode(const float *__restrict__ in, float *slot, int iters, unsigned *count, Hit *hits, int tid) {
    v2f xa[6], xb[6];          // six (lo, hi) input pairs per chain, distinct per lane
    for (int i = 0; i < 6; ++i) {
        xa[i] = v2f{in[(tid * 24 + 4 * i) & 4095], in[(tid * 24 + 4 * i + 1) & 4095]};
        xb[i] = v2f{in[(tid * 24 + 4 * i + 2) & 4095], in[(tid * 24 + 4 * i + 3) & 4095]};
    }
    for (int it = 0; it < iters; ++it) {
        const float w = 0.5f + 1e-3f * (float)((it * 7 + tid) & 255);
        const v2f wv = v2f{w, 0.f};
        const float oth = (float)it;
        // scalar expectation (this file is built without the vectorisers: plain v_fma_f32)
        float e[4] = {1.f, 2.f, 4.f, 0.5f};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            e[0] = __builtin_fmaf(xa[i][0], w, e[0]); e[1] = __builtin_fmaf(xa[i][1], w, e[1]);
            e[2] = __builtin_fmaf(xb[i][0], w, e[2]); e[3] = __builtin_fmaf(xb[i][1], w, e[3]);
        }
        if (MODE == 0) asm volatile(PK_INIT PK_INTERLEAVED PK_STORES PK_END PK_OPERANDS);
        else if (MODE == 1) asm volatile(PK_INIT PK_INTERLEAVED "s_nop 4\n\t" PK_STORES PK_END PK_OPERANDS);
        else if (MODE == 2) asm volatile(PK_INIT PK_BACK2BACK "s_nop 4\n\t" PK_STORES PK_END PK_OPERANDS);
        else if (MODE == 3) asm volatile(PK_INIT PK_INTERLEAVED PK_STORES_MOV PK_END PK_OPERANDS);
        else if (MODE == 4) asm volatile(PK_INIT PK_INTERLEAVED "s_nop 0\n\t" PK_STORES PK_END PK_OPERANDS);
        else if (MODE == 5) asm volatile(PK_INIT PK_PADDED(PAD_NOP) PK_STORES PK_END PK_OPERANDS);
        else if (MODE == 6) asm volatile(PK_INIT PK_PADDED(PAD_WAIT) PK_STORES PK_END PK_OPERANDS);
        else asm volatile(PK_INIT PK_PADDED(PAD_BOTH) PK_STORES PK_END PK_OPERANDS);
        const v4f got = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(slot + 4));
        if (got[0] != e[0] || got[1] != e[1] || got[2] != e[2] || got[3] != e[3]) {
            const unsigned n = atomicAdd(count + MODE, 1u);
            if (n < 8) {
                Hit &h = hits[MODE * 8 + n];
                h.mode = MODE; h.block = blockIdx.x; h.tid = tid; h.iter = it;
                for (int i = 0; i < 4; ++i) { h.got[i] = got[i]; h.want[i] = e[i]; }
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_probe(const float *__restrict__ in, float *__restrict__ slots, int iters, int bg, unsigned *count, Hit *hits,
                                               float *__restrict__ sink) {
    const int tid = threadIdx.x;
    __shared__ __attribute__((aligned(16))) float lds_bg[4096];
    for (int i = tid; i < 4096; i += 512) lds_bg[i] = 1e-3f * (float)i;
    __syncthreads();
    if (tid >= 256) {           // background on the same SIMDs (waves 4 .. 7)
        if (bg) background(bg, iters * 5, tid, sink, lds_bg, in);
        return;
    }
    float *slot = slots + ((size_t)blockIdx.x * 256 + tid) * 8;
    run_mode<0>(in, slot, iters, count, hits, tid);
    run_mode<1>(in, slot, iters, count, hits, tid);
    run_mode<2>(in, slot, iters, count, hits, tid);
    run_mode<3>(in, slot, iters, count, hits, tid);
    run_mode<4>(in, slot, iters, count, hits, tid);
    run_mode<5>(in, slot, iters, count, hits, tid);
    run_mode<6>(in, slot, iters, count, hits, tid);
    run_mode<7>(in, slot, iters, count, hits, tid);
}

// ---- verbatim replay of the faulty build's tail (tools/pkfma_tail_block.s -> tools/pkfma_replay.inc): the block as the compiler scheduled it
// against the same instructions with every hazard padded away, same inputs; any lane whose eight stored values differ is a hit
#include "pkfma_replay.inc"
template <int V>
__global__ __launch_bounds__(512) void k_replay(float *__restrict__ fast, float *__restrict__ safe, int iters, int bg, unsigned *count, Hit *hits, float *__restrict__ sink, const float *__restrict__ gin, unsigned *colhist) {
    __shared__ __attribute__((aligned(16))) float wl[64];
    const int tid = threadIdx.x;
    if (tid < 64) wl[tid] = 0.25f + 0.01f * (float)tid;
    __syncthreads();
    __shared__ __attribute__((aligned(16))) float lds_bg[4096];
    for (int i = tid; i < 4096; i += 512) lds_bg[i] = 1e-3f * (float)i;
    __syncthreads();
    if (tid >= 256) {
        if (bg) background(bg, iters * 8, tid, sink, lds_bg, gin);
        return;
    }
    const unsigned slot = blockIdx.x * 256 + tid;
    const unsigned lds = (unsigned)(size_t)wl + (tid >> 5) * 32;          // lanes 0 .. 31 of a wave share 8 weights, lanes 32 .. 63 the next 8 (f_tail: r0 = 8 (tid >> 5))
    for (int it = 0; it < iters; ++it) {
        const float seed = 0.75f + 1e-3f * (float)((it * 13 + tid * 7) & 511);
        // (w0 .. w7: the row weights as register operands, for the variant that does not read them from LDS)
        const float w0 = wl[(tid >> 5) * 8], w1 = wl[(tid >> 5) * 8 + 1], w2 = wl[(tid >> 5) * 8 + 2], w3 = wl[(tid >> 5) * 8 + 3];
        const float w4 = wl[(tid >> 5) * 8 + 4], w5 = wl[(tid >> 5) * 8 + 5], w6 = wl[(tid >> 5) * 8 + 6], w7 = wl[(tid >> 5) * 8 + 7];
#define PK_REPLAY_IN : : [seed] "v"(seed), [slot] "v"(slot), [base] "s"(fast), [lds] "v"(lds), [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), \
                         [w4] "v"(w4), [w5] "v"(w5), [w6] "v"(w6), [w7] "v"(w7) : PK_REPLAY_CLOBBERS
#define PK_REPLAY_CASE(n) else if (V == n) asm volatile(PK_REPLAY_FAST_##n PK_REPLAY_IN)
        if (V == 0) asm volatile(PK_REPLAY_FAST_0 PK_REPLAY_IN);
        PK_REPLAY_CASE(1); PK_REPLAY_CASE(2); PK_REPLAY_CASE(3); PK_REPLAY_CASE(4); PK_REPLAY_CASE(5); PK_REPLAY_CASE(6); PK_REPLAY_CASE(7);
        PK_REPLAY_CASE(8); PK_REPLAY_CASE(9); PK_REPLAY_CASE(10); PK_REPLAY_CASE(11); PK_REPLAY_CASE(12); PK_REPLAY_CASE(13); PK_REPLAY_CASE(14); PK_REPLAY_CASE(15); PK_REPLAY_CASE(16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ground truth: the block is eight sums of 16 terms input x row weight (PK_TRUTH_*: the generator's symbolic execution of the block), every
        // step one fused multiply-add like the block's own
        const v4f f0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(fast + (size_t)slot * 256)), f1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(fast + (size_t)slot * 256 + 4));
        const float wv[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
        float e[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_fmaf(PK_TRUTH_F[c][k] * seed, wv[PK_TRUTH_W[c][k]], acc);
            e[c] = acc;
        }
        const v4f s0 = {e[0], e[1], e[2], e[3]}, s1 = {e[4], e[5], e[6], e[7]};
        bool bad = false;
        for (int i = 0; i < 4; ++i) bad |= (f0[i] != s0[i]) | (f1[i] != s1[i]);
        if (bad) {
            for (int i = 0; i < 4; ++i) { if (f0[i] != s0[i]) atomicAdd(colhist + i, 1u); if (f1[i] != s1[i]) atomicAdd(colhist + 4 + i, 1u); }
            atomicAdd(colhist + 8 + ((tid & 63) >> 4), 1u);
            const unsigned n = atomicAdd(count + 5, 1u);
            if (n < 8) {
                Hit &h = hits[n];
                h.mode = 5; h.block = blockIdx.x; h.tid = tid; h.iter = it;
                for (int i = 0; i < 4; ++i) { h.got[i] = f1[i]; h.want[i] = s1[i]; }
                hits[8 + n] = h;
                for (int i = 0; i < 4; ++i) { hits[8 + n].got[i] = f0[i]; hits[8 + n].want[i] = s0[i]; }
            }
        }
    }
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'r') {           // pkfma_probe replay [iters] [bg]
        const int iters = argc > 2 ? atoi(argv[2]) : 20000, bg = argc > 3 ? atoi(argv[3]) : 4, blocks = 512;
        const int v0 = argc > 4 ? atoi(argv[4]) : 0, v1 = argc > 5 ? atoi(argv[5]) : PK_REPLAY_NVARIANTS - 1;        // variant range
        float *dfast, *dsafe, *dsink, *dgin; unsigned *dcount; Hit *dhits;
        const size_t bytes = (size_t)blocks * 256 * 1024;
        hipMalloc(&dfast, bytes); dsafe = dfast; hipMalloc(&dsink, 512 * 4); hipMalloc(&dcount, 8 * 4); hipMalloc(&dhits, 16 * sizeof(Hit));
        hipMalloc(&dgin, 4096 * 4); hipMemset(dgin, 0, 4096 * 4);
        unsigned *dhist; hipMalloc(&dhist, 16 * 4);
        const char *names[PK_REPLAY_NVARIANTS] = {PK_REPLAY_NAMES};
        for (int v = 0; v < PK_REPLAY_NVARIANTS; ++v) {
            hipMemset(dcount, 0, 8 * 4); hipMemset(dfast, 0, bytes); hipMemset(dhist, 0, 16 * 4);
#define PK_LAUNCH(n) case n: hipLaunchKernelGGL(k_replay<n>, dim3(blocks), dim3(512), 0, 0, dfast, dsafe, iters, bg, dcount, dhits, dsink, dgin, dhist); break
            if (v < v0 || v > v1) continue;
            switch (v) {
                PK_LAUNCH(0); PK_LAUNCH(1); PK_LAUNCH(2); PK_LAUNCH(3); PK_LAUNCH(4); PK_LAUNCH(5); PK_LAUNCH(6); PK_LAUNCH(7);
                PK_LAUNCH(8); PK_LAUNCH(9); PK_LAUNCH(10); PK_LAUNCH(11); PK_LAUNCH(12); PK_LAUNCH(13); PK_LAUNCH(14); PK_LAUNCH(15); PK_LAUNCH(16);
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            unsigned count[8]; std::vector<Hit> hits(16); std::vector<float> probe(8);
            hipMemcpy(count, dcount, 8 * 4, hipMemcpyDeviceToHost); hipMemcpy(hits.data(), dhits, 16 * sizeof(Hit), hipMemcpyDeviceToHost);
            hipMemcpy(probe.data(), dfast + (size_t)5 * 256, 8 * 4, hipMemcpyDeviceToHost);
            unsigned hist[16]; hipMemcpy(hist, dhist, 16 * 4, hipMemcpyDeviceToHost);
            printf("{\"probe\": \"pkfma_replay\", \"stream\": \"%s\", \"iters\": %d, \"bg\": %d, \"wave_executions\": %.3g, \"lanes_whose_results_differ_from_the_exact_result\": %u, "
                   "\"wrong_by_column\": [%u, %u, %u, %u, %u, %u, %u, %u], \"wrong_by_16_lane_group\": [%u, %u, %u, %u]}\n",
                   names[v], iters, bg, (double)blocks * 4 * iters, count[5], hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8], hist[9], hist[10], hist[11]);
            for (unsigned i = 0; i < (count[5] < 3 ? count[5] : 3); ++i) {
                const Hit &h = hits[i], &g = hits[8 + i];
                printf("  block %d tid %d (lane %d) iter %d: columns 4..7 got %.9g %.9g %.9g %.9g want %.9g %.9g %.9g %.9g | columns 0..3 got %.9g %.9g %.9g %.9g want %.9g %.9g %.9g %.9g\n",
                       h.block, h.tid, h.tid & 63, h.iter, h.got[0], h.got[1], h.got[2], h.got[3], h.want[0], h.want[1], h.want[2], h.want[3],
                       g.got[0], g.got[1], g.got[2], g.got[3], g.want[0], g.want[1], g.want[2], g.want[3]);
            }
        }
        return 0;
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 100000, bg = argc > 2 ? atoi(argv[2]) : 1;
    const int blocks = 512;
    std::vector<float> in(4096);
    unsigned s = 12345u;
    for (auto &v : in) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.f - 0.5f) * 4.f; }
    float *din, *dslots, *dsink; unsigned *dcount; Hit *dhits;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dslots, (size_t)blocks * 256 * 8 * 4); hipMalloc(&dsink, 512 * 4);
    hipMalloc(&dcount, 8 * 4); hipMalloc(&dhits, 8 * 8 * sizeof(Hit));
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dcount, 0, 8 * 4); hipMemset(dslots, 0, (size_t)blocks * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, din, dslots, iters, bg, dcount, dhits, dsink);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned count[8]; std::vector<Hit> hits(64);
    hipMemcpy(count, dcount, 8 * 4, hipMemcpyDeviceToHost); hipMemcpy(hits.data(), dhits, 64 * sizeof(Hit), hipMemcpyDeviceToHost);
    const double execs = (double)blocks * 4 * iters;
    printf("{\"probe\": \"pkfma\", \"iters\": %d, \"bg\": %d, \"wave_executions_per_mode\": %.3g, \"ms\": %.1f, \"mismatching_lane_results\": "
           "{\"interleaved_store_behind\": %u, \"interleaved_nop4_store\": %u, \"back2back_nop4_store\": %u, \"interleaved_vmov_store\": %u, \"interleaved_nop0_store\": %u, "
           "\"each_pk_fma_followed_by_s_nop_7\": %u, \"each_pk_fma_followed_by_s_waitcnt\": %u, \"each_pk_fma_followed_by_s_waitcnt_and_two_s_nop_7\": %u}}\n",
           iters, bg, execs, ms, count[0], count[1], count[2], count[3], count[4], count[5], count[6], count[7]);
    for (int m = 0; m < 8; ++m)
        for (unsigned i = 0; i < (count[m] < 3 ? count[m] : 3); ++i) {
            const Hit &h = hits[m * 8 + i];
            printf("  mode %d block %d tid %d (lane %d) iter %d: got %.9g %.9g %.9g %.9g want %.9g %.9g %.9g %.9g\n", h.mode, h.block, h.tid, h.tid & 63, h.iter,
                   h.got[0], h.got[1], h.got[2], h.got[3], h.want[0], h.want[1], h.want[2], h.want[3]);
        }
    return 0;
}
