// Whole-chip MFMA rate (gfx950) with and without a concurrent LDS-DMA fill stream and LDS fragment reads: the ceilings the GEMM
// main loop is judged against (DESIGN.md "GEMM").   build: scripts/build_labs.sh lab_peak    run: build/lab_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// per iteration and wave: 64 MFMA 16x16x32 (or 16 MFMA 32x32x16), optionally `fills` LDS-DMA KB and `reads` ds_read_b128
template <int KIND, int fills, int reads>
__global__ void __launch_bounds__(512) peak(const char* __restrict__ src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(0.001f * (j + i + lane)); b[i][j] = (__bf16)(0.002f * (j - i)); }
    f32x4 c[16];
    f32x16 d[4];
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) d[i][j] = 0;
    const char* g = src + ((size_t)(blockIdx.x & 7) * 65536 + wave * 8192 + lane * 16);
    char* slot = lds + wave * 16384;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < fills; ++f)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + ((it * fills + f) & 7) * 1024),
                                             (__attribute__((address_space(3))) void*)(slot + (f & 7) * 1024), 16, 0, 0);
#pragma unroll
        for (int r = 0; r < reads; ++r) {
            bf16x8 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) char*)(slot + 8192 + ((r & 7) * 1024) + lane * 16)) : "memory");
            a[r & 3] = v;
        }
        if (reads) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[k], c[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[k], d[i], 0, 0, 0);
        }
        if (fills) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c[i][0];
    for (int i = 0; i < 4; ++i) s += d[i][0];
    if (s == 1234.5f) sink[0] = s;
}

template <int KIND, int fills, int reads>
static void run(const char* src, int threads, float* sink, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, waves = threads / 64;
    CK(hipFuncSetAttribute((const void*)&peak<KIND, fills, reads>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((peak<KIND, fills, reads>), dim3(256), dim3(threads), waves * 16384, st, src, 100, sink);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((peak<KIND, fills, reads>), dim3(256), dim3(threads), waves * 16384, st, src, iters, sink);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * waves * iters * 64 * 16384.0 * (KIND == 0 ? 1.0 : 0.5);   // 16 MFMA 32x32x16 = half the FLOPs of 64 MFMA 16x16x32
    const double fill = 256.0 * waves * iters * (double)fills * 1024.0, rd = 256.0 * waves * iters * (double)reads * 1024.0;
    printf("%s  %d waves/CU  fills %d reads %2d per 64 MFMA: %7.1f TFLOP/s  (%.2f ms)  fill %6.1f GB/s/CU  lds-read %6.1f GB/s/CU\n",
           KIND == 0 ? "16x16x32" : "32x32x16", waves, fills, reads, flops / ms * 1e-9, ms, fill / ms * 1e-6 / 256, rd / ms * 1e-6 / 256);
    fflush(stdout);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    char* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float* sink; CK(hipMalloc(&sink, 64));
    for (int threads : {256, 512}) {
        run<0, 0, 0>(d, threads, sink, st);
        run<1, 0, 0>(d, threads, sink, st);
        run<0, 0, 12>(d, threads, sink, st);      // the 256x256 tile: 12 b128 reads per 32 MFMAs = 24 per 64
        run<0, 0, 24>(d, threads, sink, st);
        run<0, 0, 32>(d, threads, sink, st);      // the 128x128 tile: 8 reads per 16 MFMAs
        run<0, 4, 0>(d, threads, sink, st);
        run<0, 8, 0>(d, threads, sink, st);       // 256x256: 64 KB per 8 waves x 64 MFMA -> 8 KB per wave
        run<0, 8, 24>(d, threads, sink, st);
        run<0, 16, 32>(d, threads, sink, st);     // 128x128: 32 KB per 4 waves x 32 MFMA -> 16 KB per wave per 64 MFMA
    }
    return 0;
}
