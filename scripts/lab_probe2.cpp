// replicate one sub tile of the d=40 kernel on the data of a failing row (same RNG as lab_attn "sharp Sk=32 (amp 8)")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void one_tile(const uint16_t* Q, const uint16_t* K, int ldq, int ldk, float cscale, float* S_out, float* S2_out, float* P_out) {
    const int lane = threadIdx.x, lq = lane & 31, lh = lane >> 5;
    bf16x8 qf[3], kf[3];
    for (int s = 0; s < 3; ++s) {
        u32x4 rq = {0, 0, 0, 0}, rk = {0, 0, 0, 0};
        if (s < 2 || lh == 0) {
            rq = *reinterpret_cast<const u32x4*>(Q + (long)lq * ldq + 16 * s + 8 * lh);
            rk = *reinterpret_cast<const u32x4*>(K + (long)lq * ldk + 16 * s + 8 * lh);
        } else {
            rk[0] = 0x3f80u;
        }
        float f[8];
        for (int i = 0; i < 4; ++i) { f[2*i] = __builtin_bit_cast(float, rq[i] << 16); f[2*i+1] = __builtin_bit_cast(float, rq[i] & 0xffff0000u); }
        bf16x8 q8;
        for (int j = 0; j < 8; ++j) q8[j] = (bf16)(f[j] * cscale);
        qf[s] = q8;
        kf[s] = __builtin_bit_cast(bf16x8, rk);
    }
    const f32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], s, 0, 0, 0);
    for (int r = 0; r < 16; ++r) S_out[lane * 16 + r] = s[r];
    float mx = s[0];
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    {
        const unsigned u = __builtin_bit_cast(unsigned, mx);
        const auto r2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__builtin_bit_cast(float, r2[0]), __builtin_bit_cast(float, r2[1]));
    }
    unsigned um = __builtin_bit_cast(unsigned, mx);
    if (!(um >> 31)) um += 0xffffu;
    const float m = __builtin_bit_cast(float, um & 0xffff0000u);
    if (lh) qf[2][0] = (bf16)(-m);
    f32x16 s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], z, 0, 0, 0);
    s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], s2, 0, 0, 0);
    s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], s2, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { S2_out[lane * 16 + r] = s2[r]; P_out[lane * 16 + r] = __builtin_amdgcn_exp2f(s2[r]); }
    if (lane < 64) S_out[64 * 16 + lane] = m;
}

int main() {
    const int heads = 8, d = 40, C = 320, ld = 960, batch = 4, Sq = 1024, Sk = 32;
    const float amp = 8.0f;
    std::vector<uint16_t> hq((long)batch * Sq * ld), hkv((long)batch * Sk * ld);
    uint32_t rng = 12345u;
    auto rnd = [&]() { float a = 0.f; for (int i = 0; i < 4; ++i) { rng = rng * 1664525u + 1013904223u; a += (float)(rng >> 8) * (1.0f / 16777216.0f) - 0.5f; } return a * 1.7320508f; };
    for (auto& x : hq) x = f2bf(rnd() * amp);
    for (auto& x : hkv) x = f2bf(rnd() * amp);
    uint16_t *dq, *dk; float *dS, *dS2, *dP;
    hipMalloc(&dq, hq.size() * 2); hipMalloc(&dk, hkv.size() * 2);
    hipMalloc(&dS, (64 * 16 + 64) * 4); hipMalloc(&dS2, 64 * 16 * 4); hipMalloc(&dP, 64 * 16 * 4);
    hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dk, hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice);
    const int h = 1;
    const float cs = (1.0f / sqrtf(40.f)) * 1.4426950408889634f;
    one_tile<<<1, 64>>>(dq + h * d, dk + C + h * d, ld, ld, cs, dS, dS2, dP);
    std::vector<float> S(64 * 16 + 64), S2(64 * 16), P(64 * 16);
    hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(S2.data(), dS2, S2.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost);
    // query 2 (lanes 2 and 34)
    for (int q : {2, 0, 5}) {
        printf("query %d: m = %g / %g\n", q, S[64 * 16 + q], S[64 * 16 + 32 + q]);
        for (int lh = 0; lh < 2; ++lh) for (int r = 0; r < 16; ++r) {
            const int lane = q + 32 * lh, key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            double ref = 0; for (int e = 0; e < d; ++e) ref += (double)bf2f(hq[(long)q * ld + h * d + e]) * bf2f(hkv[(long)key * ld + C + h * d + e]);
            ref *= cs;
            printf("  key %2d: host %9.3f  S %9.3f  S' %9.3f  P %g\n", key, ref, S[lane * 16 + r], S2[lane * 16 + r], P[lane * 16 + r]);
        }
    }
    return 0;
}
