// In-kernel timeline of one GEMM / conv tile: wall_clock64 stamps (10 ns) from a -DSEER_GEMM_STAMPS build of gemm.hip linked as
// build/libprobe/libseer_hip.so (profiles/r02_pp8_stamps.log, r02_tile_stamps.log).
//   lab_pp8stamps M N K geglu [tile [residual]]           plain GEMM (tile 21 = the 256x256 8-phase tile, 0 = auto)
//   lab_pp8stamps conv n_img H Cin Cout [tile [splits]]   3x3 conv, stride 1
// Stamp columns, relative to the wave's entry.  256x256 tile: prologue issued | prologue landed | K loop done | rows re-aligned |
// epilogue math done | C staged | stored.  Ring tiles: set-up done | first K tile landed | K loop done | epilogue math done |
// C staged | stored  (split-K: ... K loop done | partial stored).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "seer_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
extern "C" long long* seer_lab_pp8_stamps();
extern "C" long long* seer_lab_block_span();
int main(int argc, char** argv) {
    const bool conv = argc > 1 && !strcmp(argv[1], "conv");
    seer_gemm_desc d; memset(&d, 0, sizeof d);
    size_t a_elems, w_elems, c_elems;
    if (conv) {
        const int n_img = argc > 2 ? atoi(argv[2]) : 24, H = argc > 3 ? atoi(argv[3]) : 32, Cin = argc > 4 ? atoi(argv[4]) : 320;
        const int Cout = argc > 5 ? atoi(argv[5]) : 320;
        d.mode = SEER_GEMM_CONV3X3; d.M = n_img * H * H; d.N = Cout; d.K = 9 * Cin; d.K1 = d.K; d.Hin = d.Win = d.Hout = d.Wout = H;
        d.Cin = Cin; d.stride = 1; d.ldc = Cout; d.batch = 1;
        d.tile = argc > 6 ? atoi(argv[6]) : 0; d.splits = argc > 7 ? atoi(argv[7]) : 0;
        a_elems = (size_t)d.M * Cin; w_elems = (size_t)Cout * d.K; c_elems = (size_t)d.M * Cout;
    } else {
        const int M = argc > 1 ? atoi(argv[1]) : 6144, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 640;
        const int geglu = argc > 4 ? atoi(argv[4]) : 1;
        d.mode = SEER_GEMM_PLAIN; d.M = M; d.N = N; d.K = K; d.K1 = K; d.lda = K; d.ldc = geglu ? N / 2 : N; d.batch = 1;
        d.tile = argc > 5 ? atoi(argv[5]) : 21;
        if (geglu) d.epilogue = SEER_EPI_GEGLU;
        a_elems = (size_t)M * K; w_elems = (size_t)N * K; c_elems = (size_t)M * N;
    }
    uint16_t *A, *W, *C, *R; float* B; void* ws = nullptr;
    CK(hipMalloc(&A, a_elems * 2)); CK(hipMalloc(&W, w_elems * 2)); CK(hipMalloc(&C, c_elems * 2)); CK(hipMalloc(&R, c_elems * 2));
    CK(hipMalloc(&B, d.N * 4));
    CK(hipMemset(A, 0x3c, a_elems * 2)); CK(hipMemset(W, 0x3c, w_elems * 2)); CK(hipMemset(R, 0x3c, c_elems * 2)); CK(hipMemset(B, 0, d.N * 4));
    d.A = A; d.W = W; d.C = C; d.bias = B;
    if (!conv && argc > 6 && atoi(argv[6])) { d.residual = R; d.ldr = d.ldc; }
    const int64_t wsb = seer_gemm_workspace_bytes(&d);
    if (wsb < 0) { printf("workspace query: %s\n", seer_strerror((int)wsb)); return 1; }
    if (wsb > 0) { CK(hipMalloc(&ws, wsb)); d.workspace = ws; d.workspace_bytes = wsb; }
    for (int i = 0; i < 3; ++i) { int rc = seer_gemm_bf16(&d, nullptr); if (rc) { printf("rc %d (%s)\n", rc, seer_strerror(rc)); return 1; } }
    CK(hipDeviceSynchronize());
    std::vector<long long> st(64 * 8 * 16);
    CK(hipMemcpy(st.data(), seer_lab_pp8_stamps(), st.size() * 8, hipMemcpyDeviceToHost));
    printf("M %d N %d K %d tile %d%s: stamps relative to entry, 10 ns ticks (columns: see the header of scripts/lab_pp8stamps.cpp)\n",
           d.M, d.N, d.K, d.tile, wsb > 0 ? " split-K" : "");
    for (int b : {0, 9, 40}) for (int w : {0, 3}) {
        const long long* t = st.data() + ((size_t)b * 8 + w) * 16;
        printf("block %2d wave %d:", b, w);
        for (int i = 1; i < 8 && t[i] >= t[0] && t[i] - t[0] < 100000000; ++i) printf(" %lld", t[i] - t[0]);
        printf("\n");
    }
    // dispatch ramp and tail over ALL blocks of the launch (wave 0 entry / exit)
    {
        const int tiles = 8192;
        std::vector<long long> sp(tiles * 2);
        CK(hipMemcpy(sp.data(), seer_lab_block_span(), sp.size() * 8, hipMemcpyDeviceToHost));
        long long e_lo = 0, e_hi = 0, x_lo = 0, x_hi = 0; int nb = 0;
        std::vector<long long> entries;
        for (int b = 0; b < tiles; ++b) {
            const long long e = sp[b * 2], x = sp[b * 2 + 1];
            if (!e || !x) continue;
            if (!nb) { e_lo = e_hi = e; x_lo = x_hi = x; }
            if (e < e_lo) e_lo = e; if (e > e_hi) e_hi = e; if (x < x_lo) x_lo = x; if (x > x_hi) x_hi = x;
            entries.push_back(e); ++nb;
        }
        int first_round = 0;                       // blocks that entered before the first block left
        for (long long e : entries) if (e < x_lo) ++first_round;
        if (getenv("LAB_SPAN_DUMP")) {            // one line per block: id, entry and exit relative to the first entry (10 ns ticks)
            for (int b = 0; b < tiles; ++b)
                if (sp[b * 2] && sp[b * 2 + 1]) printf("span %d %lld %lld\n", b, sp[b * 2] - e_lo, sp[b * 2 + 1] - e_lo);
        }
        printf("blocks %d: first entry 0, last entry %lld, first exit %lld, last exit %lld; %d blocks entered before the first exit\n",
               nb, e_hi - e_lo, x_lo - e_lo, x_hi - e_lo, first_round);
    }
    // spread of the start stamps over the first 64 blocks: how staggered the launch is
    long long lo = st[0], hi = st[0];
    for (int b = 0; b < 64; ++b) { const long long v = st[(size_t)b * 8 * 16]; if (v < lo) lo = v; if (v > hi) hi = v; }
    printf("start spread over blocks 0..63: %lld ticks\n", hi - lo);
    return 0;
}
