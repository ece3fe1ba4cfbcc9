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
// bg = 1: waves 4 .. 7 of every workgroup run MFMA loops beside the probing waves (the tile kernels' neighbours on a SIMD do).
// usage: pkfma_probe [iters = 100000] [bg = 1]
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
#define PK_OPERANDS  : : [a0] "v"(xa[0]), [a1] "v"(xa[1]), [a2] "v"(xa[2]), [a3] "v"(xa[3]), [a4] "v"(xa[4]), [a5] "v"(xa[5]),                       \
                         [b0] "v"(xb[0]), [b1] "v"(xb[1]), [b2] "v"(xb[2]), [b3] "v"(xb[3]), [b4] "v"(xb[4]), [b5] "v"(xb[5]),                       \
                         [w] "v"(wv), [p] "v"(slot), [o] "v"(oth)                                                                                  \
                     : "memory", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137"

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
        else asm volatile(PK_INIT PK_INTERLEAVED "s_nop 0\n\t" PK_STORES PK_END PK_OPERANDS);
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
    if (tid >= 256) {           // background: MFMA loops on the same SIMDs (waves 4 .. 7)
        if (!bg) return;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        h8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)(tid + i)); b[i] = (_Float16)(0.002f * (float)(tid - i)); }
        for (int it = 0; it < iters * 5 * 4; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (acc[0] == 12345.678f) sink[tid] = acc[1];
        return;
    }
    float *slot = slots + ((size_t)blockIdx.x * 256 + tid) * 8;
    run_mode<0>(in, slot, iters, count, hits, tid);
    run_mode<1>(in, slot, iters, count, hits, tid);
    run_mode<2>(in, slot, iters, count, hits, tid);
    run_mode<3>(in, slot, iters, count, hits, tid);
    run_mode<4>(in, slot, iters, count, hits, tid);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000, bg = argc > 2 ? atoi(argv[2]) : 1;
    const int blocks = 512;
    std::vector<float> in(4096);
    unsigned s = 12345u;
    for (auto &v : in) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.f - 0.5f) * 4.f; }
    float *din, *dslots, *dsink; unsigned *dcount; Hit *dhits;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dslots, (size_t)blocks * 256 * 8 * 4); hipMalloc(&dsink, 512 * 4);
    hipMalloc(&dcount, 8 * 4); hipMalloc(&dhits, 5 * 8 * sizeof(Hit));
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dcount, 0, 8 * 4); hipMemset(dslots, 0, (size_t)blocks * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, din, dslots, iters, bg, dcount, dhits, dsink);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned count[8]; std::vector<Hit> hits(40);
    hipMemcpy(count, dcount, 8 * 4, hipMemcpyDeviceToHost); hipMemcpy(hits.data(), dhits, 40 * sizeof(Hit), hipMemcpyDeviceToHost);
    const double execs = (double)blocks * 4 * iters;
    printf("{\"probe\": \"pkfma\", \"iters\": %d, \"bg_mfma\": %d, \"wave_executions_per_mode\": %.3g, \"ms\": %.1f, \"mismatching_lane_results\": "
           "{\"interleaved_store_behind\": %u, \"interleaved_nop4_store\": %u, \"back2back_nop4_store\": %u, \"interleaved_vmov_store\": %u, \"interleaved_nop0_store\": %u}}\n",
           iters, bg, execs, ms, count[0], count[1], count[2], count[3], count[4]);
    for (int m = 0; m < 5; ++m)
        for (unsigned i = 0; i < (count[m] < 8 ? count[m] : 8); ++i) {
            const Hit &h = hits[m * 8 + i];
            printf("  mode %d block %d tid %d (lane %d) iter %d: got %.9g %.9g %.9g %.9g want %.9g %.9g %.9g %.9g\n", h.mode, h.block, h.tid, h.tid & 63, h.iter,
                   h.got[0], h.got[1], h.got[2], h.got[3], h.want[0], h.want[1], h.want[2], h.want[3]);
        }
    return 0;
}
