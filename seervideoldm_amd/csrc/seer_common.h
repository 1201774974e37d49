// Shared device helpers for the gfx950 kernels of libseer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "seer_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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

// the same for IEEE half storage (the VAE path, SEER_EPI_F16 / the *_dt entry points with dtype = SEER_DT_F16).  Scalar
// conversions on the two 16-bit halves of a dword: the ext_vector_type(2) _Float16 form of these loops compiled to code that
// read element 0 twice (ROCm 7.2; scripts/dbg_gn.py: y[.., 2] == y[.., 0])
__device__ __forceinline__ float half_bits_to_f32(unsigned int b) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(b & 0xffffu));
}
__device__ __forceinline__ unsigned int pack2h(float lo, float hi) {
    const unsigned int l = __builtin_bit_cast(unsigned short, (_Float16)lo), h = __builtin_bit_cast(unsigned short, (_Float16)hi);
    return l | (h << 16);
}
template <bool F16>
__device__ __forceinline__ void unpack8t(const u32x4& v, float (&f)[8]) {
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int u = v[i];
            f[2 * i] = half_bits_to_f32(u);
            f[2 * i + 1] = half_bits_to_f32(u >> 16);
        }
    } else {
        unpack8(v, f);
    }
}
template <bool F16>
__device__ __forceinline__ u32x4 pack8t(const float (&f)[8]) {
    if constexpr (F16) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pack2h(f[2 * i], f[2 * i + 1]);
        return v;
    } else {
        return pack8(f);
    }
}

// F16 instantiations of the MFMA kernels: A, A2, W, residual and C hold IEEE half instead of bf16 (SEER_EPI_F16: the VAE, which the
// reference runs in fp32, and the UNet engine under fp16 autocast --
// 11 significand bits instead of 8 at the same MFMA rate).  Same 16-bit loads and LDS image; only the MFMA opcode and the
// pack / unpack of the epilogue differ.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16>
__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned int pack2t(float lo, float hi) {
    if constexpr (F16) return pack2h(lo, hi);
    else return pack2(lo, hi);
}
template <bool F16>
__device__ __forceinline__ f32x2 unpack2t(unsigned int v) {
    if constexpr (F16) {
        return f32x2{half_bits_to_f32(v), half_bits_to_f32(v >> 16)};
    } else {
        return f32x2{__builtin_bit_cast(float, v << 16), __builtin_bit_cast(float, v & 0xffff0000u)};
    }
}

// 32x32x16 MFMA and the 16-bit storage value of a float, by storage type (the attention kernels: bf16 vectors are containers of bits)
template <bool F16>
__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ bf16 to16(float v) {
    if constexpr (F16) return __builtin_bit_cast(bf16, (_Float16)v);
    else return (bf16)v;
}

// ---- write-through output stores.  A plain store leaves its line dirty in the XCD's L2 and the dependent-kernel boundary behind
// the launch waits for the write-back (microarch guide, "boundary": + dirty bytes / 6 TB/s -- 2.6 us behind a 15.7 MB activation,
// a quarter of a 10 us normalisation kernel).  `sc1` stores write through while the kernel still runs; 16-byte stores only (the
// guide prices 8-byte sc1 stores at 2.7x per byte).  The store is inline asm, so the compiler's hazard recogniser does not see a
// VMEM store of more than 64 bits: the two wait states gfx940+ needs before a VALU instruction may overwrite the data VGPRs are
// the `s_nop 1` behind it (without them the weight-stationary GEMM stored garbage: the next erf argument landed in the same
// registers one instruction later).  -DSEER_WT_STORES=0 builds the plain-store library for A/B runs (profiles/r02_wt_stores.log).
#ifndef SEER_WT_STORES
#define SEER_WT_STORES 1
#endif
__device__ __forceinline__ void store16_out(void* dst, const u32x4 v) {
#if SEER_WT_STORES
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#else
    *reinterpret_cast<u32x4*>(dst) = v;
#endif
}
// the same with a wave-uniform base pointer and a 32-bit per-lane byte offset (one address VGPR)
__device__ __forceinline__ void store16_out(void* base, unsigned voff, const u32x4 v) {
#if SEER_WT_STORES
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
#else
    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + voff) = v;
#endif
}

// One interleaved rotary pair (x0, x1) -> (x0 c - x1 s, x1 c + x0 s) with the roundings of fma(x0, c, -(x1 s)), fma(x1, c, x0 s):
// the crossed products t = (-x1 s, x0 s) as TWO SCALAR multiplies, then ONE packed fma x * (c, c) + t.  Do not write the rotation
// as scalar fp32 lines, and do not leave the crossing to a packed instruction: hipcc (SLP vectoriser, shuffle folding) then emits
// packed instructions that read the halves of a register pair crossed, and a crossed read is only safe on SOME operand positions.
// Measured on MI355X (scripts/lab_pkswap.cpp, profiles/r03_flake_root_cause.md): a v_pk_{fma,mul,add}_f32 whose LOW result reads
// the HIGH half of its SECOND source (op_sel[1] = 1) computes that result in lanes 48..63 as if the operand were zero whenever a
// second process runs the denoising network on the GPU, and never otherwise.  Round 2's rotary epilogue contained
//     v_pk_fma_f32 v[36:37], v[76:77], v[36:37], v[84:85] op_sel:[0,1,0] op_sel_hi:[1,0,0]
// which returned the addend alone: 2 700 wrong launches of the q|k|v projection in 3.5 M, 0 in 3.5 M without the form.
// seervideoldm_amd/asm_check.py fails the build if any kernel of the library contains it.
__device__ __forceinline__ f32x2 rot_pair(f32x2 x, float c, float s) {
    float t0, t1;                                       // the crossed products as two scalar multiplies: no packed half-swap
    asm("v_mul_f32_e64 %0, -%1, %2" : "=v"(t0) : "v"(s), "v"(x[1]));
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t1) : "v"(s), "v"(x[0]));
    return __builtin_elementwise_fma(x, f32x2{c, c}, f32x2{t0, t1});
}

// folded LayerNorm term of a GEMM epilogue: acc * rstd - wsum * (mean * rstd), as two scalar instructions (written in C++ the
// row scalars sit in the halves of an LDS read pair and hipcc broadcasts the high half with op_sel[1] = 1: the form above)
__device__ __forceinline__ float ln_term(float acc, float rstd, float wsum, float mean_rstd) {
    float t, o;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(t) : "v"(wsum), "v"(mean_rstd));
    asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(o) : "v"(acc), "v"(rstd), "v"(t));
    return o;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (F.gelu default, attention.py:785-793) without libm and with ONE transcendental:
//     gelu(x) = relu(x) - |x| * Phi(-|x|),      Phi(-a) = 0.5 * erfc(a / sqrt 2) = 2^(a * R(a) - 1)
// R = degree-5 fit of (log2(erfc(a/sqrt2)/2) + 1) / a on [0, 4*sqrt2], weighted for the absolute error of gelu; beyond the
// interval a is clamped (Phi(-5.66) = 7.7e-9).  Max |error| vs fp64 gelu 3.1e-7 over [-9, 9], relative error < 7.1e-4 wherever
// |gelu| > 1e-4 (tests/test_gpu_kernels.py::test_gelu_erf_accuracy).  The GEGLU epilogue evaluates it on M*N/2 elements per
// GEMM and was VALU-bound with the rcp + exp2 form (A&S 7.1.26): 6 packed FMAs + 1 v_exp per element now.
#define SEER_GELU_C0 (-1.1511471271514893f)
#define SEER_GELU_C1 (-0.45891568064689636f)
#define SEER_GELU_C2 (-0.05323818698525429f)
#define SEER_GELU_C3 (0.00797746330499649f)
#define SEER_GELU_C4 (-0.0007398744928650558f)
#define SEER_GELU_C5 (2.9924221962573938e-05f)
#define SEER_GELU_AMAX (5.656854249492381f)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float a = fminf(fabsf(x), SEER_GELU_AMAX);
    float p = fmaf(SEER_GELU_C5, a, SEER_GELU_C4);
    p = fmaf(p, a, SEER_GELU_C3);
    p = fmaf(p, a, SEER_GELU_C2);
    p = fmaf(p, a, SEER_GELU_C1);
    p = fmaf(p, a, SEER_GELU_C0);
    const float h = __builtin_amdgcn_exp2f(fmaf(p, a, -1.0f));
    return fmaf(-a, h, fmaxf(x, 0.f));
}
// two elements at a time: the Horner chain maps to v_pk_fma_f32
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), SEER_GELU_AMAX), fminf(fabsf(x[1]), SEER_GELU_AMAX)};
    f32x2 p = __builtin_elementwise_fma(f32x2{SEER_GELU_C5, SEER_GELU_C5}, a, f32x2{SEER_GELU_C4, SEER_GELU_C4});
    p = __builtin_elementwise_fma(p, a, f32x2{SEER_GELU_C3, SEER_GELU_C3});
    p = __builtin_elementwise_fma(p, a, f32x2{SEER_GELU_C2, SEER_GELU_C2});
    p = __builtin_elementwise_fma(p, a, f32x2{SEER_GELU_C1, SEER_GELU_C1});
    p = __builtin_elementwise_fma(p, a, f32x2{SEER_GELU_C0, SEER_GELU_C0});
    const f32x2 q = __builtin_elementwise_fma(p, a, f32x2{-1.0f, -1.0f});
    const f32x2 h = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-a, h, r);
}

// fixed-point accumulation of a column-sum partial (seer_gemm_desc::colsum_fx): integer adds commute, so the total does not
// depend on arrival order.  No-return agent-scope atomic (executed at the memory side).
__device__ __forceinline__ void fx_add(int64_t* p, float v) {
    const long long q = __float2ll_rn(v * (float)(1 << SEER_GN_FX_SHIFT));
    __hip_atomic_fetch_add(reinterpret_cast<long long*>(p), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fx_add_ln(int64_t* p, float v) {
    const long long q = __float2ll_rn(v * (float)(1 << SEER_LN_FX_SHIFT));
#ifdef SEER_RS_PROBE_STORE      // measurement build: a plain store in place of the atomic (wrong sums) -- what the atomic itself costs
    *reinterpret_cast<long long*>(p) = q;
#else
    __hip_atomic_fetch_add(reinterpret_cast<long long*>(p), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// planes (sum, sum of squares) of the replica / batch element a partial adds to: colsum_fx[rep][b][2][N]
__device__ __forceinline__ int64_t* fx_slot(const seer_gemm_desc& p, int partial_index, int first_row) {
    const int nb = p.M / p.colsum_fx_rows;
    return p.colsum_fx + (int64_t)(((partial_index % p.colsum_fx_reps) * nb + first_row / p.colsum_fx_rows) * 2) * p.N;
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
