// What the fused feed-forward's H loop costs in isolation: one wave per SIMD, per sub step 6 MFMAs 16x16x32 bf16 (2 weight fragments x
// 3 activation fragments), the activation fragments (a) held in registers, (b) read from LDS by ds_read_b128 three sub steps ahead
// with counted waits; the weight fragments and / or the accumulators in accumulation registers (inline-asm MFMAs pin the classes).
//     hipcc --offload-arch=gfx950 -O3 scripts/lab_mfma_lds.cpp -o build/lab_mfma_lds && build/lab_mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <bool ACC_A, bool W_A> __device__ __forceinline__ void mfma(f32x4& acc, const u32x4& w, const u32x4& x) {
    if (ACC_A && W_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(x));
    if (ACC_A && !W_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
    if (!ACC_A && W_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
    if (!ACC_A && !W_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
struct F3 { u32x4 r[3]; };
__device__ __forceinline__ void req(F3& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:2048\n\tds_read_b128 %2, %3 offset:4096"
                 : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]) : "v"(addr) : "memory");
}
template <int N> __device__ __forceinline__ void got(F3& f) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]) : "n"(N) : "memory");
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// the same FLOPs per sub step as 3 MFMAs 32x32x16 (one weight fragment x 3 activation fragments), V registers
template <int LDS> __global__ void __launch_bounds__(256, 1) kbig(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 61440 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const int frow = lane & 31, fq = lane >> 5;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const unsigned a0 = base + frow * 128 + ((fq ^ (frow & 7)) * 16), a1 = base + frow * 128 + (((4 + fq) ^ (frow & 7)) * 16);
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 w = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(w));
    F3 p[4];
    for (int b = 0; b < 4; ++b) { req(p[b], (b & 1) ? a1 : a0); got<0>(p[b]); }
    if (LDS == 1) { req(p[0], a0); req(p[1], a1); req(p[2], a0 + 12288); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned adr = ((j & 1) ? a1 : a0) + ((it + j) % 5) * 12288;
            if (LDS == 1) { got<6>(p[j]); req(p[(j + 3) & 3], adr); }
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(p[j].r[i]));
        }
    }
    if (LDS == 1) { got<0>(p[0]); got<0>(p[1]); got<0>(p[2]); got<0>(p[3]); }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int LDS> void runbig(float* out, const char* what) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kbig<LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    hipLaunchKernelGGL((kbig<LDS>), dim3(256), dim3(256), 61440, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kbig<LDS>), dim3(256), dim3(256), 61440, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 1024.0 * iters * 12 * 32768.0;
    printf("  %-82s %.3f ms = %6.0f TFLOP/s (%.1f ns per sub step of 3 MFMAs)\n", what, ms, fl / ms * 1e-9, ms * 1e6 / (iters * 4.0));
}

// LDS: 0 = fragments in registers, 1 = from LDS three sub steps ahead, 2 = from LDS, waited for right behind the request
template <bool ACC_A, bool W_A, int LDS> __global__ void __launch_bounds__(256, 1) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 61440 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)smem;
    const unsigned a0 = base + frow * 128 + ((fq ^ (frow & 7)) * 16), a1 = base + frow * 128 + (((4 + fq) ^ (frow & 7)) * 16);
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 w[2];
    for (int i = 0; i < 2; ++i) w[i] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i};
    if (W_A) { asm volatile("" : "+a"(w[0]), "+a"(w[1])); } else { asm volatile("" : "+v"(w[0]), "+v"(w[1])); }
    F3 p[4];
    for (int b = 0; b < 4; ++b) { req(p[b], (b & 1) ? a1 : a0); got<0>(p[b]); }
    auto six = [&](const F3& f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { mfma<ACC_A, W_A>(acc[2 * i], w[0], f.r[i]); mfma<ACC_A, W_A>(acc[2 * i + 1], w[1], f.r[i]); }
    };
    if (LDS == 1) { req(p[0], a0); req(p[1], a1); req(p[2], a0 + 12288); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {           // four sub steps, buffer j; panel (it * 4 + j) mod 5, halves alternate
            const unsigned adr = ((j & 1) ? a1 : a0) + ((it + j) % 5) * 12288;
            if (LDS == 1) { got<6>(p[j]); req(p[(j + 3) & 3], adr); }
            if (LDS == 2) { req(p[j], adr); got<0>(p[j]); }
            six(p[j]);
        }
    }
    if (LDS == 1) { got<0>(p[0]); got<0>(p[1]); got<0>(p[2]); got<0>(p[3]); }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) { if (ACC_A) asm volatile("" : "+a"(acc[i])); s += acc[i][0] + acc[i][3]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool ACC_A, bool W_A, int LDS> void run(float* out, const char* what) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<ACC_A, W_A, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
    hipLaunchKernelGGL((k<ACC_A, W_A, LDS>), dim3(256), dim3(256), 61440, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC_A, W_A, LDS>), dim3(256), dim3(256), 61440, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 1024.0 * iters * 24 * 16384.0;
    printf("  %-82s %.3f ms = %6.0f TFLOP/s (%.1f ns per sub step of 6 MFMAs)\n", what, ms, fl / ms * 1e-9, ms * 1e6 / (iters * 4.0));
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 256 * 4) != hipSuccess) return 1;
    printf("one wave per SIMD, 256 workgroups; per sub step 6 MFMAs 16x16x32 bf16 = 2 weight x 3 activation fragments\n");
    run<false, false, 0>(out, "all operands in V registers, nothing from LDS");
    run<true, false, 0>(out, "accumulators in accumulation registers");
    run<true, true, 0>(out, "accumulators and weight fragments in accumulation registers");
    run<false, false, 1>(out, "V registers; activation fragments from LDS (3 ds_read_b128 per sub step), 3 ahead, counted waits");
    run<true, true, 1>(out, "accumulation registers; activation fragments from LDS, 3 ahead, counted waits");
    run<true, true, 2>(out, "accumulation registers; activation fragments from LDS, waited for behind the request");
    printf("the same with 3 MFMAs 32x32x16 per sub step (the same FLOPs)\n");
    runbig<0>(out, "V registers, nothing from LDS");
    runbig<1>(out, "V registers; 3 ds_read_b128 per sub step from LDS, 3 ahead, counted waits");
    return 0;
}
