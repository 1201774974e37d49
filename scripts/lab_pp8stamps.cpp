// In-kernel timeline of the 256x256 8-phase tile (tile code 21) on the level-1 GEGLU projection (M 6144, N 5120, K 640):
// wall_clock64 stamps (10 ns) from a -DSEER_GEMM_STAMPS build of gemm.hip linked as build/libprobe/libseer_hip.so
// (profiles/r02_pp8_stamps.log).  Usage: lab_pp8stamps M N K geglu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "seer_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
extern "C" long long* seer_lab_pp8_stamps();
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 6144, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 640;
    const int geglu = argc > 4 ? atoi(argv[4]) : 1;
    uint16_t *A, *W, *C; float* B;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&B, N * 4));
    CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(W, 0x3c, (size_t)N * K * 2)); CK(hipMemset(B, 0, N * 4));
    seer_gemm_desc d; memset(&d, 0, sizeof d);
    d.A = A; d.W = W; d.C = C; d.bias = B; d.M = M; d.N = N; d.K = K; d.K1 = K; d.lda = K; d.ldc = geglu ? N / 2 : N; d.batch = 1; d.tile = 21;
    d.mode = SEER_GEMM_PLAIN; if (geglu) d.epilogue = SEER_EPI_GEGLU;
    for (int i = 0; i < 3; ++i) { int rc = seer_gemm_bf16(&d, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
    CK(hipDeviceSynchronize());
    std::vector<long long> st(64 * 8 * 16);
    CK(hipMemcpy(st.data(), seer_lab_pp8_stamps(), st.size() * 8, hipMemcpyDeviceToHost));
    printf("stamps: start | descriptors done | prologue landed | K loop done | rows re-aligned | epilogue math done | C staged | stored\n");
    for (int b : {0, 9, 40}) for (int w : {0, 5}) {
        const long long* t = st.data() + ((size_t)b * 8 + w) * 16;
        printf("block %2d wave %d:", b, w);
        for (int i = 1; i < 8; ++i) printf(" %lld", t[i] - t[0]);
        printf("  (10 ns ticks)\n");
    }
    return 0;
}
