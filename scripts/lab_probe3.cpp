// what do the two results of the permlane{16,32}_swap builtins hold? (hipcc ROCm 7.2, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k32(float* x, float* y) {
    const unsigned u = __builtin_bit_cast(unsigned, x[threadIdx.x]);
    const unsigned w = __builtin_bit_cast(unsigned, x[threadIdx.x + 64]);
    const auto r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
    y[threadIdx.x] = __builtin_bit_cast(float, r[0]);
    y[threadIdx.x + 64] = __builtin_bit_cast(float, r[1]);
}
__global__ void k16(float* x, float* y) {
    const unsigned u = __builtin_bit_cast(unsigned, x[threadIdx.x]);
    const unsigned w = __builtin_bit_cast(unsigned, x[threadIdx.x + 64]);
    const auto r = __builtin_amdgcn_permlane16_swap(u, w, false, false);
    y[threadIdx.x] = __builtin_bit_cast(float, r[0]);
    y[threadIdx.x + 64] = __builtin_bit_cast(float, r[1]);
}
__global__ void kasm(float* x, float* y) {
    float a = x[threadIdx.x], b = x[threadIdx.x + 64];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    y[threadIdx.x] = a;
    y[threadIdx.x + 64] = b;
}
int main() {
    float hx[128], hy[128], *dx, *dy;
    for (int i = 0; i < 128; ++i) hx[i] = (float)i;
    hipMalloc(&dx, 512); hipMalloc(&dy, 512);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    for (int t = 0; t < 3; ++t) {
        hipMemset(dy, 0, 512);
        if (t == 0) k32<<<1, 64>>>(dx, dy); else if (t == 1) k16<<<1, 64>>>(dx, dy); else kasm<<<1, 64>>>(dx, dy);
        hipMemcpy(hy, dy, 512, hipMemcpyDeviceToHost);
        printf("%s\n r[0]:", t == 0 ? "permlane32_swap builtin (a = 0..63, b = 64..127)" : t == 1 ? "permlane16_swap builtin" : "permlane32_swap inline asm");
        for (int i = 0; i < 64; ++i) printf(" %g", hy[i]);
        printf("\n r[1]:");
        for (int i = 0; i < 64; ++i) printf(" %g", hy[64 + i]);
        printf("\n");
    }
    return 0;
}
