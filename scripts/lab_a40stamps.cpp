// In-kernel timeline of the head-dim-40 attention kernel on the spatial block [192, 1024, 40]: s_memrealtime stamps (10 ns) of
// the first blocks, from a measurement build of the library (attention40.hip compiled with -DSEER_ATTN40_STAMPS into
// build/libprobe; see scripts/probe_a40stamps.sh).  Prints, per wave: Q load, then per 128-key tile
// [wait, barrier, issue, sub-block x4] durations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "seer_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
extern "C" long long* seer_lab_a40_stamps();

int main() {
    const int batch = 24, heads = 8, d = 40, S = 1024, C = heads * d, ld = 3 * C;
    const size_t n = (size_t)batch * S * ld;
    std::vector<uint16_t> h(n);
    uint32_t r = 5u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; float f = ((float)(r >> 8) / 8388608.0f - 1.0f) * 0.7f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    uint16_t *qkv, *out;
    CK(hipMalloc(&qkv, n * 2)); CK(hipMalloc(&out, (size_t)batch * S * C * 2));
    CK(hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice));
    seer_attn_desc ad; memset(&ad, 0, sizeof ad);
    ad.Q = qkv; ad.K = qkv + C; ad.V = qkv + 2 * C; ad.O = out;
    ad.q_ss = ad.k_ss = ad.v_ss = ld; ad.o_ss = C;
    ad.q_bs = ad.k_bs = ad.v_bs = (int64_t)S * ld; ad.o_bs = (int64_t)S * C;
    ad.batch = batch; ad.heads = heads; ad.head_dim = d; ad.Sq = S; ad.Sk = S; ad.scale = 0.158f;
    if (getenv("LAB_VARIANT")) ad.variant = atoi(getenv("LAB_VARIANT"));
    for (int i = 0; i < 3; ++i) { int rc = seer_attn_fwd(&ad, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
    CK(hipDeviceSynchronize());
    long long* dst = seer_lab_a40_stamps();
    std::vector<long long> st(16 * 4 * 128);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    for (int b : {0, 5, 11}) for (int w = 0; w < 4; ++w) {
        const long long* t = st.data() + ((size_t)b * 4 + w) * 128;
        printf("block %2d wave %d: start->Q %lld |", b, w, t[1] - t[0]);
        int i = 1;
        for (int tile = 0; tile < 8; ++tile) {
            printf(" T%d[w%lld b%lld i%lld s", tile, t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2]);
            for (int s = 0; s < 4; ++s) printf(" %lld", t[i + 4 + s] - t[i + 3 + s]);
            printf("]");
            i += 8;
        }
        printf(" | loop end %lld, epilogue %lld, total %lld ticks of 10 ns\n", t[i] - t[0], t[i + 1] - t[i], t[i + 1] - t[0]);
    }
    return 0;
}
