// Which VALU instructions run in the shadow of MFMAs on gfx950?  One wave per SIMD (and two), a loop of 4 independent MFMAs 32x32x16
// bf16 with 32 VALU instructions of ONE kind between them in program order (8 per MFMA), against the same two loops apart.
// hidden = (t_mfma + t_valu - t_both) / min(t_mfma, t_valu): 1 = fully in the shadow, 0 = serialised.
//     hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/lab_mfma_valu.cpp -o build/lab_mfma_valu && build/lab_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { PK_FMA, FMA, EXP, PK_MUL, PK_ADD, CVT_PK_BF16, MAX, MUL, ADD_U32, PERM, MOV_DPP, MAX_E64, MUL_E64, SUB_E64, MAX3, SUB, NOPS };
static const char* NAMES[] = {"v_pk_fma_f32", "v_fma_f32", "v_exp_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_max_f32",
                              "v_mul_f32", "v_add_u32", "v_perm_b32", "v_mov_b32 dpp row_shr:1", "v_max_f32_e64", "v_mul_f32_e64", "v_sub_f32_e64", "v_max3_f32",
                              "v_sub_f32"};

template <int OP> __device__ __forceinline__ void valu(f32x2& v, float m, float c) {
    if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(f32x2{m, m}), "v"(f32x2{c, c}));
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(c));
    if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
    if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(f32x2{m, m}));
    if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(f32x2{c, c}));
    if (OP == CVT_PK_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[0]) : "v"(m));
    if (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[0]) : "v"(m));
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(c));
    if (OP == MAX_E64) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    if (OP == MUL_E64) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[0]) : "v"(m));
    if (OP == SUB_E64) asm volatile("v_sub_f32_e64 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c), "v"(m));
    if (OP == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[0]) : "v"(v[1]));
}

// MODE bit 0: MFMA, bit 1: VALU
template <int MODE, int OP, int THREADS> __global__ void __launch_bounds__(THREADS) k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, 1.0f};
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE & 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) valu<OP>(v[r], 0.999f, 1e-3f);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int OP, int THREADS> double run(float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, OP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, OP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int OP, int THREADS> void row(float* out, double tm) {
    const int iters = 20000;
    const double tv = run<2, OP, THREADS>(out, iters), tb = run<3, OP, THREADS>(out, iters);
    const double ns_per = tv * 1e6 / (iters * 32.0);
    printf("  %-26s alone %.3f ms (%.2f ns each) | with the MFMAs %.3f ms | hidden %.2f\n", NAMES[OP], tv, ns_per, tb,
           (tm + tv - tb) / (tm < tv ? tm : tv));
}

template <int THREADS> void table(float* out) {
    const int iters = 20000;
    const double tm = run<1, FMA, THREADS>(out, iters);
    const double fl = 256.0 * THREADS / 64 * iters * 4 * 32768.0;
    printf("%d wave(s) per SIMD: 4 MFMA 32x32x16 bf16 per iteration alone %.3f ms = %.0f TFLOP/s (%.2f ns each)\n", THREADS / 256, tm,
           fl / tm * 1e-9, tm * 1e6 / (iters * 4.0));
    row<PK_FMA, THREADS>(out, tm); row<FMA, THREADS>(out, tm); row<EXP, THREADS>(out, tm); row<PK_MUL, THREADS>(out, tm);
    row<PK_ADD, THREADS>(out, tm); row<CVT_PK_BF16, THREADS>(out, tm); row<MAX, THREADS>(out, tm); row<MUL, THREADS>(out, tm);
    row<ADD_U32, THREADS>(out, tm); row<PERM, THREADS>(out, tm); row<MOV_DPP, THREADS>(out, tm);
    row<SUB, THREADS>(out, tm); row<MAX_E64, THREADS>(out, tm); row<MUL_E64, THREADS>(out, tm); row<SUB_E64, THREADS>(out, tm); row<MAX3, THREADS>(out, tm);
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 1024 * 4) != hipSuccess) return 1;
    table<256>(out);
    table<512>(out);
    return 0;
}
