// ds_read_b128 bank-conflict check (gfx950) for the GEMM fragment-read patterns: lane (frow = l & 15, fq = l >> 4) reads 16 B of
// row frow.  Prints LDS read GB/s per CU for each layout; the conflict-free rate is the "linear" row.
//   build: scripts/build_labs.sh lab_ldsread    run: build/lab_ldsread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void __launch_bounds__(512) rd(int pattern, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    unsigned off;
    switch (pattern) {
        case 0: off = lane * 16; break;                                                   // linear
        case 1: off = frow * 128 + ((fq ^ (frow & 7)) * 16); break;                       // 128-B rows, chunk ^= row & 7 (gemm.hip)
        case 2: off = frow * 128 + ((fq ^ ((frow >> 1) & 7)) * 16); break;                // 128-B rows, chunk ^= (row >> 1) & 7
        case 3: off = frow * 64 + ((fq ^ ((frow >> 2) & 3)) * 16); break;                 // 64-B rows, chunk ^= (row >> 2) & 3
        case 4: off = frow * 64 + fq * 16; break;                                         // 64-B rows, no swizzle
        case 5: off = frow * 128 + fq * 16; break;                                        // 128-B rows, no swizzle
        default: off = frow * 80 + fq * 16; break;                                        // 80-B rows (attention40 K tile)
    }
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds + wave * 4096) + off;
    u32x4 a0, a1, a2, a3, acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:2048\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:2048\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(base) : "memory");
        acc += a0 ^ a1 ^ a2 ^ a3;
    }
    if (acc[0] == 0x12345u) sink[0] = acc[1];
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    const char* names[] = {"linear", "128B rows ^ (row&7)", "128B rows ^ ((row>>1)&7)", "64B rows ^ ((row>>2)&3)", "64B rows plain", "128B rows plain", "80B rows plain"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 512})
        for (int p = 0; p < 7; ++p) {
            const int iters = 20000;
            hipLaunchKernelGGL(rd, dim3(256), dim3(threads), 8 * 4096 + 4096, st, p, 100, sink);
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(rd, dim3(256), dim3(threads), 8 * 4096 + 4096, st, p, iters, sink);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d waves/CU  %-28s %7.1f GB/s per CU\n", threads / 64, names[p], (double)(threads / 64) * iters * 4096.0 / ms * 1e-6);
        }
    return 0;
}
