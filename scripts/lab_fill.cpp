// L2 / HBM -> LDS fill-rate calibration (gfx950): what one CU and the whole chip sustain through global_load_lds (16 B per lane)
// as a function of waves per CU, instructions in flight per wave and where the data lives.  The GEMM main loops are sized from
// these numbers (DESIGN.md "GEMM").   build: scripts/build_labs.sh lab_fill    run: build/lab_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int DEPTH>
__device__ __forceinline__ void wait_depth() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory"); }

// each wave streams `per_wave` KB-sized rows; row r of wave w of block b is at base + ((start(b) + r * waves + w) % region_kb) KB
// mode 0: every block streams the SAME region (all hits after the first toucher); mode 1: block-private regions
template <int DEPTH, bool TO_LDS>
__global__ void __launch_bounds__(512) fill(const char* __restrict__ base, long long region_kb, int per_wave, int mode, int stride_kb,
                                            float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    long long pos = mode == 0 ? (long long)(blockIdx.x & 7) * 64 : (long long)blockIdx.x * stride_kb;
    char* slot = lds + wave * (DEPTH * 1024);
    float acc = 0.f;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 regs[DEPTH];
    for (int r = 0; r < per_wave; ++r) {
        const long long kb = (pos + (long long)r * waves + wave) % region_kb;
        const char* src = base + kb * 1024 + lane * 16;
        if (TO_LDS) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(slot + (r % DEPTH) * 1024), 16, 0, 0);
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(regs[r % DEPTH]) : "v"(src) : "memory");
        }
        if (r >= DEPTH - 1) wait_depth<DEPTH>();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!TO_LDS) { for (int i = 0; i < DEPTH; ++i) acc += regs[i][0]; }
    else acc = *(float*)(slot + lane * 4);
    if (acc == 12345.678f) sink[0] = acc;
}

// GEMM-shaped gathers: one instruction = ROWB-byte pieces of 1024/ROWB different rows of a row-major matrix with `pitch` bytes
// per row (ROWB = 128: the BK = 64 tile rows of gemm.hip; ROWB = 64: BK = 32 half tiles); consecutive instructions walk along K.
template <int DEPTH, int ROWB>
__global__ void __launch_bounds__(512) fill_rows(const char* __restrict__ base, int pitch, int rows_total, int per_wave, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR;         // lanes per row, rows per instruction
    char* slot = lds + wave * (DEPTH * 1024);
    const int ksteps = pitch / ROWB;
    // block b streams the row panel [256 (b % panels), +256): wave w takes rows 32 w .. 32 w + 31 of it in 32 / RPI instructions
    const int panels = rows_total / 256;
    const long long row0 = (long long)(blockIdx.x % panels) * 256 + wave * (256 / waves);
    int r = 0;
    for (int it = 0; it < per_wave; ++it) {
        const int k = (it / ((256 / waves) / RPI)) % ksteps, sub = it % ((256 / waves) / RPI);
        const char* src = base + (row0 + sub * RPI + lane / LPR) * pitch + k * ROWB + (lane % LPR) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(slot + (r % DEPTH) * 1024), 16, 0, 0);
        if (r >= DEPTH - 1) wait_depth<DEPTH>();
        ++r;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*(float*)(slot + lane * 4) == 12345.678f) sink[0] = 1.f;
}

template <int DEPTH, int ROWB>
static void run_rows(const char* d, int pitch, int rows_total, int threads, float* sink, hipStream_t st) {
    const int waves = threads / 64, per_wave = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)&fill_rows<DEPTH, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((fill_rows<DEPTH, ROWB>), dim3(256), dim3(threads), waves * DEPTH * 1024, st, d, pitch, rows_total, per_wave, sink);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill_rows<DEPTH, ROWB>), dim3(256), dim3(threads), waves * DEPTH * 1024, st, d, pitch, rows_total, per_wave, sink);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 256.0 * waves * per_wave * 1024.0 * 5;
    printf("row gather: %3d-B pieces, pitch %5d B, %5d rows (%.1f MB), depth %2d, %d waves: %8.1f GB/s total, %6.1f GB/s per CU\n", ROWB, pitch, rows_total,
           (double)pitch * rows_total / 1048576.0, DEPTH, waves, bytes / ms * 1e-6, bytes / ms * 1e-6 / 256);
    fflush(stdout);
}

template <int DEPTH, bool TO_LDS>
static void run(const char* d, long long region_kb, int blocks, int threads, int mode, float* sink, hipStream_t st, const char* what) {
    const int waves = threads / 64;
    const int per_wave = 4096 / waves * 2;        // 8 MB per block
    const int stride_kb = 8192;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t ldsb = TO_LDS ? (size_t)waves * DEPTH * 1024 : 0;
    CK(hipFuncSetAttribute((const void*)&fill<DEPTH, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((fill<DEPTH, TO_LDS>), dim3(blocks), dim3(threads), ldsb, st, d, region_kb, per_wave, mode, stride_kb, sink);
    CK(hipEventRecord(e0, st));
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fill<DEPTH, TO_LDS>), dim3(blocks), dim3(threads), ldsb, st, d, region_kb, per_wave, mode, stride_kb, sink);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * waves * per_wave * 1024.0 * iters;
    printf("%-34s %s depth %2d  blocks %4d x %d waves: %8.1f GB/s total, %6.1f GB/s per block (%.0f KB in flight per block)\n", what, TO_LDS ? "lds" : "reg",
           DEPTH, blocks, waves, bytes / ms * 1e-6, bytes / ms * 1e-6 / (blocks < 256 ? blocks : 256), waves * DEPTH * 1.0);
    fflush(stdout);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const long long total = (long long)4 << 30;
    char* d; CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total));
    float* sink; CK(hipMalloc(&sink, 64));
    for (int threads : {256, 512})
        for (int pitch : {640, 2560}) {
            run_rows<8, 128>(d, pitch, 24576, threads, sink, st);
            run_rows<8, 64>(d, pitch, 24576, threads, sink, st);
            run_rows<8, 128>(d, pitch, 1536, threads, sink, st);
            run_rows<8, 64>(d, pitch, 1536, threads, sink, st);
        }
    struct { const char* name; long long region_kb; int mode; } where[] = {
        {"shared 1 MB (L2 hits)", 1024, 0}, {"shared 16 MB (MALL hits)", 16384, 0}, {"private 8 MB/block (HBM stream)", total / 1024, 1}};
    for (auto& w : where) {
        for (int threads : {256, 512}) {
            run<2, true>(d, w.region_kb, 256, threads, w.mode, sink, st, w.name);
            run<4, true>(d, w.region_kb, 256, threads, w.mode, sink, st, w.name);
            run<8, true>(d, w.region_kb, 256, threads, w.mode, sink, st, w.name);
            run<16, true>(d, w.region_kb, 256, threads, w.mode, sink, st, w.name);
        }
        run<8, false>(d, w.region_kb, 256, 512, w.mode, sink, st, w.name);
        run<8, true>(d, w.region_kb, 512, 256, w.mode, sink, st, w.name);
        run<8, true>(d, w.region_kb, 1, 512, w.mode, sink, st, w.name);
    }
    return 0;
}
