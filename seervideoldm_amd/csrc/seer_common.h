// Shared device helpers for the gfx950 kernels of libseer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "seer_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SEER_WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __builtin_bit_cast(float, (unsigned int)b << 16);
}
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }

// unpack 8 bf16 held in a u32x4 to floats
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    bf16x2 p;
    p[0] = (bf16)lo;
    p[1] = (bf16)hi;
    return __builtin_bit_cast(unsigned int, p);
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
    return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (F.gelu default, attention.py:785-793)
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off of 1+erf): 2 transcendentals + 8 VALU instead
// of the ~30-instruction branchy libm erff -- the GEGLU epilogue evaluates it on M*N/2 elements per GEMM.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);                 // erf(|x|/sqrt2)
    const float erf_v = __builtin_copysignf(erf_abs, x);
    return 0.5f * x * (1.0f + erf_v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#define SEER_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return SEER_ELAUNCH;           \
    } while (0)
