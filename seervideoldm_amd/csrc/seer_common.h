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
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
