// hardware probes: v_exp_f32 on very negative inputs, bf16 conversion of the results, MFMA on denormal operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(const float* x, int n, float* y, unsigned short* yb, float* mm) {
    const int i = threadIdx.x;
    if (i < n) {
        const float e = __builtin_amdgcn_exp2f(x[i]);
        y[i] = e;
        yb[i] = __builtin_bit_cast(unsigned short, (__bf16)e);
    }
    // MFMA: A row = lane&15, all elements 3.0; B col = lane&15: element 0 = bf16(exp2(x[lane&15])), others 0
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)3.0f; b[j] = (__bf16)0.0f; }
    if ((threadIdx.x >> 4) == 0) b[0] = (__bf16)__builtin_amdgcn_exp2f(x[threadIdx.x & 15]);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (threadIdx.x < 16) mm[threadIdx.x] = c[0];
}
int main() {
    const float hx[16] = {0.f, -0.7f, -48.f, -100.f, -126.f, -127.f, -130.f, -140.f, -149.f, -150.f, -160.f, -200.f, -273.f, -1000.f, -1e9f, -__builtin_inff()};
    float *dx, *dy, *dm; unsigned short* db;
    hipMalloc(&dx, 64); hipMalloc(&dy, 64); hipMalloc(&db, 32); hipMalloc(&dm, 64);
    hipMemcpy(dx, hx, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, 16, dy, db, dm);
    float hy[16], hm[16]; unsigned short hb[16];
    hipMemcpy(hy, dy, 64, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 32, hipMemcpyDeviceToHost); hipMemcpy(hm, dm, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) { uint32_t u; memcpy(&u, &hy[i], 4); uint32_t um; memcpy(&um, &hm[i], 4);
        printf("exp2(%g) = %g (0x%08x)  bf16 0x%04x   mfma 3*p = %g (0x%08x)\n", hx[i], hy[i], u, hb[i], hm[i], um); }
    return 0;
}
