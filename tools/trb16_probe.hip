// dev probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS is filled with u16 value = element index; every lane supplies a byte
// address from one of several patterns; the 4 u16 each lane receives are printed as (source lane whose 8-byte slot held it, element).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int *addr, unsigned short *out, int npat) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    for (int p = 0; p < npat; ++p) {
        unsigned a = (unsigned)addr[p * 64 + threadIdx.x] + (unsigned)(size_t)lds;   // LDS byte address
        unsigned long long r;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
        out[(p * 64 + threadIdx.x) * 4 + 0] = (unsigned short)(r & 0xffff);
        out[(p * 64 + threadIdx.x) * 4 + 1] = (unsigned short)((r >> 16) & 0xffff);
        out[(p * 64 + threadIdx.x) * 4 + 2] = (unsigned short)((r >> 32) & 0xffff);
        out[(p * 64 + threadIdx.x) * 4 + 3] = (unsigned short)((r >> 48) & 0xffff);
    }
}
int main() {
    const int NP = 4;
    std::vector<int> addr(NP * 64);
    for (int l = 0; l < 64; ++l) {
        addr[0 * 64 + l] = 0;                                   // P0: uniform address
        addr[1 * 64 + l] = l * 8;                               // P1: lane-linear 8-byte slots
        addr[2 * 64 + l] = ((l >> 2) & 3) * 592 + (l & 3) * 8 + (l >> 4) * 32;      // P2: per 16-lane group a [4 rows][16 cols] block of a 592-byte-stride tile: lane i -> row (i>>2)&3, col quad i&3; group g -> cols 16g
        addr[3 * 64 + l] = (l & 3) * 592 + ((l >> 2) & 3) * 8 + (l >> 4) * 32;       // P3: the other assignment: lane i -> row i&3, col quad (i>>2)&3
    }
    int *da; unsigned short *dout;
    hipMalloc(&da, addr.size() * 4); hipMalloc(&dout, NP * 64 * 4 * 2);
    hipMemcpy(da, addr.data(), addr.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, NP);
    std::vector<unsigned short> out(NP * 64 * 4);
    hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost);
    for (int p = 0; p < NP; ++p) {
        printf("pattern %d (value = LDS element index; for P2/P3: row = idx / 296, col = idx %% 296)\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %5d :", l, addr[p * 64 + l]);
            for (int j = 0; j < 4; ++j) { int v = out[(p * 64 + l) * 4 + j]; printf("  %5d (r%d c%d)", v, v / 296, v % 296); }
            printf("\n");
        }
    }
    return 0;
}
