// Issue-rate calibration on one CU (gfx950): cycles per instruction for the instructions of the d=40 attention loop at
// 1..4 waves per SIMD, and how MFMA and VALU / transcendental streams of DIFFERENT waves of a SIMD overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// mode: which instruction stream; role split: waves with (wave >= split) run stream B, others stream A
__global__ void issue(int modeA, int modeB, int split, int iters, long long* out) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave >= split ? modeB : modeA;
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    unsigned p0 = threadIdx.x, p1 = threadIdx.x + 1;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * j); b[j] = (__bf16)(0.002f * j); }
    f32x16 c16 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    f32x4 c4a = {0,0,0,0}, c4b = {0,0,0,0}, c4c = {0,0,0,0}, c4d = {0,0,0,0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {        // 64 independent v_exp_f32
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (mode == 1) { // 64 v_cvt_pk_bf16_f32
            asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %1\n v_cvt_pk_bf16_f32 %0, %2, %4\n v_cvt_pk_bf16_f32 %3, %5, %7\n v_cvt_pk_bf16_f32 %6, %1, %4\n v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (mode == 2) { // 64 v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %6, %6, %7, %1\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %6, %6, %7, %1\n v_fma_f32 %2, %2, %4, %5\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (mode == 3) { // 64 v_permlane16_swap
            asm volatile(REP16("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (mode == 4) { // 16 dependent-free (same accumulator) 32x32x16 MFMAs
            for (int k = 0; k < 16; ++k) c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
        } else if (mode == 5) { // 32 16x16x32 MFMAs on 4 accumulators
            for (int k = 0; k < 8; ++k) {
                c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4a, 0, 0, 0);
                c4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4b, 0, 0, 0);
                c4c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4c, 0, 0, 0);
                c4d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4d, 0, 0, 0);
            }
        } else if (mode == 6) { // the attention mix of one (32q x 32k) block: 3 MFMA32 + 16 exp + 8 cvt + 4 swaps + 6 MFMA16, in program order
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n") REP8("v_cvt_pk_bf16_f32 %2, %3, %4\n")
                         "v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %2, %5\n"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4a, 0, 0, 0);
            c4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4b, 0, 0, 0);
            c4c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4c, 0, 0, 0);
            c4d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4d, 0, 0, 0);
            c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4a, 0, 0, 0);
            c4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4b, 0, 0, 0);
        } else if (mode == 7) { // same mix, MFMAs and VALU independent and interleaved 1 MFMA32 : 5 VALU, 1 MFMA16 : 2 VALU
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4a, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(v0), "+v"(v1));
            c4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4b, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(v0), "+v"(v1));
            c4c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4c, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(v0), "+v"(v1));
            c4d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4d, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_cvt_pk_bf16_f32 %3, %4, %5\n v_cvt_pk_bf16_f32 %6, %7, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c4a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4a, 0, 0, 0);
            asm volatile("v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            c4b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4b, 0, 0, 0);
            asm volatile("v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (mode == 8) { // idle
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + c16[0] + c4a[0] + c4b[0] + c4c[0] + c4d[0] + (float)(p0 + p1);
    if (acc == 12345.678f) out[1000] = 1;      // keep everything alive
    if ((threadIdx.x & 63) == 0) { out[wave] = t1 - t0; out[32 + wave] = t0; out[64 + wave] = t1; }
}

int main() {
    long long* d;
    hipMalloc(&d, 8192);
    const char* names[] = {"64 v_exp_f32", "64 v_cvt_pk_bf16_f32", "64 v_fma_f32", "64 v_permlane16_swap", "16 mfma 32x32x16", "32 mfma 16x16x32",
                           "attention block, program order (3 M32, 16 exp, 8 cvt, 4 swap, 6 M16)", "attention block, interleaved"};
    const int iters = 200;
    auto run = [&](int mA, int mB, int split, int nw, const char* label) {
        hipMemset(d, 0, 8192);
        issue<<<1, 64 * nw>>>(mA, mB, split, iters, d);
        long long h[96];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        // the SIMD serves its oldest wave first: the span until the LAST wave of a role is done is the throughput figure
        auto span = [&](int w0, int w1) { long long a = h[32 + w0], b = h[64 + w0]; for (int w = w0; w < w1; ++w) { a = std::min(a, h[32 + w]); b = std::max(b, h[64 + w]); } return (double)(b - a) / iters; };
        printf("%-70s waves/SIMD %d:", label, nw / 4);
        if (split >= nw) printf(" span %.1f  (wave0 alone %.1f)", span(0, nw), (double)h[0] / iters);
        else printf(" span A %.1f  span B %.1f", span(0, split), span(split, nw));
        printf("  ticks per loop body\n");
    };
    for (int m = 0; m < 8; ++m)
        for (int nw = 4; nw <= 16; nw += 4) run(m, m, 99, nw, names[m]);
    // two roles on every SIMD: waves 0-3 stream A, waves 4-7 stream B
    run(4, 0, 4, 8, "A = 16 mfma32 | B = 64 exp");
    run(4, 2, 4, 8, "A = 16 mfma32 | B = 64 fma");
    run(4, 1, 4, 8, "A = 16 mfma32 | B = 64 cvt_pk");
    run(5, 0, 4, 8, "A = 32 mfma16 | B = 64 exp");
    run(0, 2, 4, 8, "A = 64 exp    | B = 64 fma");
    run(0, 1, 4, 8, "A = 64 exp    | B = 64 cvt_pk");
    run(4, 8, 4, 8, "A = 16 mfma32 | B = idle");
    return 0;
}
