/* A C99 host calling libseer_hip.so through include/seer_hip.h alone (no Python, no C++): SURVEY 8(b) names a non-Python
 * caller of the ABI.  Built by __graft_entry__.build() / scripts/build_labs.sh with gcc; tests/test_abi.py runs it on the GPU
 * box.  Checks seer_gemm_bf16 (A = I against an asymmetric W: C = W^T exactly) and seer_attn_fwd (head_dim 40, 256 keys)
 * against host arithmetic.  Exit code 0 and "ABI_CALLER_OK" on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "seer_hip.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(void) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    /* ---- GEMM: C[m][n] = sum_k A[m][k] W[n][k], A = identity, W asymmetric */
    enum { M = 128, N = 192, K = 128 };
    static uint16_t hA[M * K], hW[N * K], hC[M * N];
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hA[m * K + k] = f2bf(m == k ? 1.0f : 0.0f);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) hW[n * K + k] = f2bf((float)((n * 7 + k * 3) % 64) / 32.0f - 1.0f);
    void *dA, *dW, *dC;
    CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dW, sizeof hW)); CK(hipMalloc(&dC, sizeof hC));
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW, sizeof hW, hipMemcpyHostToDevice));
    seer_gemm_desc g; memset(&g, 0, sizeof g);
    g.A = dA; g.W = dW; g.C = dC; g.M = M; g.N = N; g.K = K; g.K1 = K; g.lda = K; g.ldc = N; g.mode = SEER_GEMM_PLAIN; g.batch = 1;
    int rc = seer_gemm_bf16(&g, st);
    if (rc != SEER_OK) { fprintf(stderr, "seer_gemm_bf16: %s\n", seer_strerror(rc)); return 1; }
    CK(hipStreamSynchronize(st)); CK(hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost));
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n)
        if (hC[m * N + n] != hW[n * K + m]) { fprintf(stderr, "gemm mismatch at (%d,%d)\n", m, n); return 1; }
    /* ---- attention: one batch element, 8 heads of 40 channels, 256 queries x 256 keys */
    enum { S = 256, H = 8, D = 40, C = H * D };
    static uint16_t hq[S * C], hk[S * C], hv[S * C], ho[S * C];
    uint32_t r = 1u;
    for (int i = 0; i < S * C; ++i) {
        r = r * 1664525u + 1013904223u; hq[i] = f2bf((float)(r >> 8) / 8388608.0f - 1.0f);
        r = r * 1664525u + 1013904223u; hk[i] = f2bf((float)(r >> 8) / 8388608.0f - 1.0f);
        r = r * 1664525u + 1013904223u; hv[i] = f2bf((float)(r >> 8) / 8388608.0f - 1.0f);
    }
    void *dq, *dk, *dv, *dout;
    CK(hipMalloc(&dq, sizeof hq)); CK(hipMalloc(&dk, sizeof hk)); CK(hipMalloc(&dv, sizeof hv)); CK(hipMalloc(&dout, sizeof ho));
    CK(hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk, sizeof hk, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice));
    seer_attn_desc a; memset(&a, 0, sizeof a);
    a.Q = dq; a.K = dk; a.V = dv; a.O = dout; a.q_ss = a.k_ss = a.v_ss = a.o_ss = C; a.q_bs = a.k_bs = a.v_bs = a.o_bs = (int64_t)S * C;
    a.batch = 1; a.heads = H; a.head_dim = D; a.Sq = S; a.Sk = S; a.scale = 1.0f / sqrtf((float)D);
    rc = seer_attn_fwd(&a, st);
    if (rc != SEER_OK) { fprintf(stderr, "seer_attn_fwd: %s\n", seer_strerror(rc)); return 1; }
    CK(hipStreamSynchronize(st)); CK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int h = 0; h < H; ++h) for (int q = 0; q < S; q += 17) {
        double s[S], m = -1e30, l = 0.0, o[D];
        for (int j = 0; j < S; ++j) { double d = 0; for (int e = 0; e < D; ++e) d += bf2f(hq[q * C + h * D + e]) * bf2f(hk[j * C + h * D + e]); s[j] = d * a.scale; if (s[j] > m) m = s[j]; }
        for (int e = 0; e < D; ++e) o[e] = 0.0;
        for (int j = 0; j < S; ++j) { double p = exp(s[j] - m); l += p; for (int e = 0; e < D; ++e) o[e] += p * bf2f(hv[j * C + h * D + e]); }
        for (int e = 0; e < D; ++e) { double err = fabs(o[e] / l - bf2f(ho[q * C + h * D + e])); if (err > worst) worst = err; }
    }
    if (!(worst < 2e-2)) { fprintf(stderr, "attention max error %g\n", worst); return 1; }
    printf("ABI_CALLER_OK abi=%d arch=%s attention_max_err=%.3g\n", seer_abi_version(), seer_build_arch(), worst);
    return 0;
}
