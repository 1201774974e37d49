// Stand-alone A/B harness for seer_attn_fwd at the head_dim-40 shapes of the Seer UNet (no Python, no torch: starts in
// seconds on a fresh gpurun box).  Links libseer_hip.so through the C ABI only.
//   build:  scripts/build_labs.sh      run:  build/lab_attn [iters]
// For every case x variant: max |err| and relative L2 against a naive fp32 kernel (one thread per query, two-pass
// softmax) on the same bf16 inputs, and the median launch time over `iters` launches (HIP events on the launch stream).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "seer_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ static inline float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __builtin_bit_cast(float, u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}

// naive reference: thread per (sequence, head, query); qtok/ktok give the token row of every sequence position
__global__ void ref_attn(const uint16_t* Q, const uint16_t* K, const uint16_t* V, float* O, const int* qtok,
                         const int* ktok, int nseq, int heads, int d, int Sq, int Sk, int ld, float scale, int causal,
                         int q_off) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)nseq * heads * Sq) return;
    const int q = idx % Sq;
    const int h = (idx / Sq) % heads;
    const int s = idx / ((long)Sq * heads);
    const uint16_t* qr = Q + (long)qtok[(long)s * Sq + q] * ld + h * d;
    float qv[160];
    for (int i = 0; i < d; ++i) qv[i] = bf2f(qr[i]);
    const int kmax = causal ? min(Sk, q + q_off + 1) : Sk;
    float m = -INFINITY;
    for (int j = 0; j < kmax; ++j) {
        const uint16_t* kr = K + (long)ktok[(long)s * Sk + j] * ld + h * d;
        float a = 0.f;
        for (int i = 0; i < d; ++i) a += qv[i] * bf2f(kr[i]);
        m = fmaxf(m, a * scale);
    }
    float l = 0.f, acc[160];
    for (int i = 0; i < d; ++i) acc[i] = 0.f;
    for (int j = 0; j < kmax; ++j) {
        const long tk = ktok[(long)s * Sk + j];
        const uint16_t* kr = K + tk * ld + h * d;
        float a = 0.f;
        for (int i = 0; i < d; ++i) a += qv[i] * bf2f(kr[i]);
        const float p = expf(a * scale - m);
        l += p;
        const uint16_t* vr = V + tk * ld + h * d;
        for (int i = 0; i < d; ++i) acc[i] += p * bf2f(vr[i]);
    }
    float* orow = O + ((long)s * Sq + q) * heads * d + h * d;
    for (int i = 0; i < d; ++i) orow[i] = acc[i] / l;
}

// stress: how many 16-bit words of `a` differ from `b`
__global__ void count_diff(const uint16_t* a, const uint16_t* b, long n, unsigned* cnt) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(cnt, 1u);
}

struct Case {
    const char* name;
    int batch, Sq, Sk, causal;
    int ws, F, H, W;      // window form when ws > 0
    float amp;            // input scale (larger -> sharper softmax)
};

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 30;
    const int heads = 8, d = 40, C = heads * d, ld = 3 * C;
    const Case cases[] = {
        {"spatial L0 [192,1024,40]", 24, 1024, 1024, 0, 0, 0, 0, 0, 1.0f},
        {"cross L0 Sk=77", 24, 1024, 77, 0, 0, 0, 0, 0, 1.0f},
        {"overhead Sk=128", 24, 1024, 128, 0, 0, 0, 0, 0, 1.0f},
        {"overhead Sk=256", 24, 1024, 256, 0, 0, 0, 0, 0, 1.0f},
        {"overhead Sk=512", 24, 1024, 512, 0, 0, 0, 0, 0, 1.0f},
        {"temporal L0 ws8 causal 768", 2, 768, 768, 1, 8, 12, 32, 32, 1.0f},
        {"ragged Sq=1000 Sk=930", 3, 1000, 930, 0, 0, 0, 0, 0, 1.0f},
        {"ragged causal Sq=Sk=333", 3, 333, 333, 1, 0, 0, 0, 0, 1.0f},
        {"sharp spatial (amp 6)", 4, 1024, 1024, 0, 0, 0, 0, 0, 6.0f},
        {"sharp Sk=32 (amp 8)", 4, 1024, 32, 0, 0, 0, 0, 0, 8.0f},
        {"sharp Sk=128 (amp 8)", 4, 1024, 128, 0, 0, 0, 0, 0, 8.0f},
        {"sharp spatial (amp 2)", 4, 1024, 1024, 0, 0, 0, 0, 0, 2.0f},
        {"sharp spatial (amp 3)", 4, 1024, 1024, 0, 0, 0, 0, 0, 3.0f},
        {"sharp spatial (amp 4)", 4, 1024, 1024, 0, 0, 0, 0, 0, 4.0f},
        {"sharp spatial (amp 5)", 4, 1024, 1024, 0, 0, 0, 0, 0, 5.0f},
        {"spatial 64^2 [192,4096,40]", 24, 4096, 4096, 0, 0, 0, 0, 0, 1.0f},
    };
    const int variants[] = {1, 2, 3, 4, 5, 7};
    const char* only = getenv("LAB_CASE");
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("seer ABI %d arch %s, iters %d\n", seer_abi_version(), seer_build_arch(), iters);

    for (const Case& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        const int tq = c.ws ? c.F * c.H * c.W : c.Sq;      // tokens per batch element in memory
        const int tk = c.ws ? c.F * c.H * c.W : c.Sk;
        const int nwin = c.ws ? (c.H / c.ws) * (c.W / c.ws) : 1;
        const int nseq = c.batch * nwin;
        const long rows_q = (long)c.batch * tq, rows_k = (long)c.batch * tk;
        // fused q|k|v rows for self attention (q at col 0, k at C, v at 2C); separate K/V rows for cross attention
        std::vector<uint16_t> hq(rows_q * ld), hkv(rows_k * ld);
        uint32_t rng = 12345u;
        auto rnd = [&]() {      // sum of 4 uniforms: roughly gaussian
            float a = 0.f;
            for (int i = 0; i < 4; ++i) {
                rng = rng * 1664525u + 1013904223u;
                a += (float)(rng >> 8) * (1.0f / 16777216.0f) - 0.5f;
            }
            return a * 1.7320508f;
        };
        for (auto& x : hq) x = f2bf(rnd() * c.amp);
        for (auto& x : hkv) x = f2bf(rnd() * c.amp);
        uint16_t *dq, *dkv, *dout;
        float* dref;
        CK(hipMalloc(&dq, hq.size() * 2));
        CK(hipMalloc(&dkv, hkv.size() * 2));
        CK(hipMalloc(&dout, rows_q * C * 2));
        CK(hipMalloc(&dref, (long)nseq * c.Sq * C * 4));
        CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dkv, hkv.data(), hkv.size() * 2, hipMemcpyHostToDevice));
        const bool self = (tq == tk);
        const uint16_t* Qp = dq;
        const uint16_t* Kp = self ? dq + C : dkv + C;
        const uint16_t* Vp = self ? dq + 2 * C : dkv + 2 * C;

        // token maps for the reference (sequence index = win * batch + b, as the kernel decodes it)
        std::vector<int> qtok((long)nseq * c.Sq), ktok((long)nseq * c.Sk);
        for (int s = 0; s < nseq; ++s) {
            const int win = s / c.batch, b = s % c.batch;
            for (int pos = 0; pos < std::max(c.Sq, c.Sk); ++pos) {
                int t = pos;
                if (c.ws) {
                    const int nwx = c.W / c.ws, ws2 = c.ws * c.ws;
                    const int f = pos / ws2, rem = pos % ws2, wy = rem / c.ws, wx = rem % c.ws;
                    t = f * c.H * c.W + ((win / nwx) * c.ws + wy) * c.W + (win % nwx) * c.ws + wx;
                }
                if (pos < c.Sq) qtok[(long)s * c.Sq + pos] = b * tq + t;
                if (pos < c.Sk) ktok[(long)s * c.Sk + pos] = b * tk + t;
            }
        }
        int *dqt, *dkt;
        CK(hipMalloc(&dqt, qtok.size() * 4));
        CK(hipMalloc(&dkt, ktok.size() * 4));
        CK(hipMemcpy(dqt, qtok.data(), qtok.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dkt, ktok.data(), ktok.size() * 4, hipMemcpyHostToDevice));
        const float scale = 1.0f / sqrtf((float)d);
        {
            const long n = (long)nseq * heads * c.Sq;
            ref_attn<<<(unsigned)((n + 127) / 128), 128, 0, st>>>(Qp, Kp, Vp, dref, dqt, dkt, nseq, heads, d, c.Sq, c.Sk, ld,
                                                                   scale, c.causal, 0);
            CK(hipStreamSynchronize(st));
        }
        std::vector<float> href((long)nseq * c.Sq * C);
        CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));

        seer_attn_desc ad;
        memset(&ad, 0, sizeof(ad));
        ad.Q = Qp; ad.K = Kp; ad.V = Vp; ad.O = dout;
        ad.q_ss = ld; ad.k_ss = ld; ad.v_ss = ld; ad.o_ss = C;
        ad.q_bs = (int64_t)tq * ld; ad.k_bs = (int64_t)tk * ld; ad.v_bs = (int64_t)tk * ld; ad.o_bs = (int64_t)tq * C;
        uint16_t* dhm = nullptr;
        if (getenv("LAB_HEADMAJOR")) {
            // the same q | k | v values as HEAD-MAJOR operands [batch][head][token][d] (seer_attn_desc::q_hs / k_hs / v_hs): which
            // LAB_HEADMAJOR digit selects: 1 = K and V, 2 = Q too
            const int mode = atoi(getenv("LAB_HEADMAJOR"));
            const uint16_t* src_q = hq.data();
            const uint16_t* src_kv = self ? hq.data() : hkv.data();
            std::vector<uint16_t> hm((rows_q + 2 * rows_k) * C);
            uint16_t* hmq = hm.data();
            uint16_t* hmk = hmq + rows_q * C;
            uint16_t* hmv = hmk + rows_k * C;
            for (int b = 0; b < c.batch; ++b)
                for (int h = 0; h < heads; ++h) {
                    for (int t = 0; t < tq; ++t)
                        memcpy(hmq + (((long)b * heads + h) * tq + t) * d, src_q + ((long)b * tq + t) * ld + h * d, d * 2);
                    for (int t = 0; t < tk; ++t) {
                        memcpy(hmk + (((long)b * heads + h) * tk + t) * d, src_kv + ((long)b * tk + t) * ld + C + h * d, d * 2);
                        memcpy(hmv + (((long)b * heads + h) * tk + t) * d, src_kv + ((long)b * tk + t) * ld + 2 * C + h * d, d * 2);
                    }
                }
            CK(hipMalloc(&dhm, hm.size() * 2));
            CK(hipMemcpy(dhm, hm.data(), hm.size() * 2, hipMemcpyHostToDevice));
            ad.K = dhm + rows_q * C; ad.V = dhm + (rows_q + rows_k) * C;
            ad.k_ss = ad.v_ss = d; ad.k_hs = ad.v_hs = (int64_t)tk * d; ad.k_bs = ad.v_bs = (int64_t)heads * tk * d;
            if (mode >= 2) { ad.Q = dhm; ad.q_ss = d; ad.q_hs = (int64_t)tq * d; ad.q_bs = (int64_t)heads * tq * d; }
            printf("   [head-major %s operands]\n", mode >= 2 ? "Q, K, V" : "K, V");
        }
        ad.batch = c.batch; ad.heads = heads; ad.head_dim = d; ad.Sq = c.Sq; ad.Sk = c.Sk; ad.causal = c.causal;
        ad.scale = scale;
        if (c.ws) { ad.window_ws = c.ws; ad.F = c.F; ad.H = c.H; ad.W = c.W; ad.Fq = c.F; }
        const double flops = 4.0 * nseq * heads * (double)c.Sq * c.Sk * d;
        printf("== %s  (%.2f GFLOP dense-equivalent)\n", c.name, flops * 1e-9);
        std::vector<uint16_t> hout(rows_q * C);
        const char* vsel = getenv("LAB_VARIANTS");      // e.g. "23": only variants 2 and 3
        for (int v : variants) {
            if (vsel && !strchr(vsel, '0' + v)) continue;
            ad.variant = v;
            CK(hipMemsetAsync(dout, 0xff, rows_q * C * 2, st));
            int rc = seer_attn_fwd(&ad, st);
            if (rc != 0) { printf("   variant %d: rc %d (%s)\n", v, rc, seer_strerror(rc)); continue; }
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
            double num = 0, den = 0, maxe = 0;
            long nanc = 0;
            for (int s = 0; s < nseq; ++s)
                for (int q = 0; q < c.Sq; ++q) {
                    const long orow = (long)qtok[(long)s * c.Sq + q] * C;
                    const long rrow = ((long)s * c.Sq + q) * C;
                    for (int i = 0; i < C; ++i) {
                        const float o = bf2f(hout[orow + i]), r = href[rrow + i];
                        if (!(o == o) || fabsf(o) > 1e30f) {
                            if (nanc < 2 * d && i % d == 0) {
                                // diagnose: the row's scores on the host
                                const int h = i / d;
                                const uint16_t* hqv = hq.data();
                                const uint16_t* hkp = self ? hq.data() + C : hkv.data() + C;
                                const uint16_t* qr = hqv + (long)qtok[(long)s * c.Sq + q] * ld + h * d;
                                double m1 = -1e30, m2 = -1e30, mn = 1e30; int a1 = -1;
                                for (int j = 0; j < c.Sk; ++j) {
                                    const uint16_t* kr = hkp + (long)ktok[(long)s * c.Sk + j] * ld + h * d;
                                    double a = 0; for (int e2 = 0; e2 < d; ++e2) a += (double)bf2f(qr[e2]) * bf2f(kr[e2]);
                                    a *= scale * 1.4426950408889634;
                                    if (a > m1) { m2 = m1; m1 = a; a1 = j; } else if (a > m2) m2 = a;
                                    mn = std::min(mn, a);
                                }
                                printf("      NaN row s=%d q=%d h=%d: ref[0]=%g  log2-scores: max %.1f at key %d, 2nd %.1f, min %.1f | out bits:",
                                       s, q, h, r, m1, a1, m2, mn);
                                for (int e2 = 0; e2 < d; e2 += 5) printf(" %04x", hout[orow + i + e2]);
                                printf("\n");
                            }
                            ++nanc; continue;
                        }
                        const double e = (double)o - r;
                        num += e * e; den += (double)r * r;
                        maxe = std::max(maxe, fabs(e));
                    }
                }
            if (getenv("LAB_STRESS")) {
                // run-to-run determinism: every launch must reproduce the first one bit for bit
                const int n_st = atoi(getenv("LAB_STRESS"));
                uint16_t* dsave; unsigned* dcnt;
                CK(hipMalloc(&dsave, rows_q * C * 2)); CK(hipMalloc(&dcnt, 4));
                CK(hipMemcpyAsync(dsave, dout, rows_q * C * 2, hipMemcpyDeviceToDevice, st));
                int bad_launches = 0; unsigned worst = 0;
                for (int it = 0; it < n_st; ++it) {
                    CK(hipMemsetAsync(dcnt, 0, 4, st));
                    CK(hipMemsetAsync(dout, 0xff, rows_q * C * 2, st));
                    seer_attn_fwd(&ad, st);
                    const long n = rows_q * C;
                    count_diff<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dout, dsave, n, dcnt);
                    unsigned h = 0;
                    CK(hipMemcpyAsync(&h, dcnt, 4, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    if (h) { ++bad_launches; worst = std::max(worst, h); }
                }
                printf("   variant %d: stress %d launches: %d differ from the first (worst %u words)\n", v, n_st, bad_launches, worst);
                CK(hipFree(dsave)); CK(hipFree(dcnt));
            }
            std::vector<float> ts;
            for (int it = 0; it < 3; ++it) seer_attn_fwd(&ad, st);
            for (int it = 0; it < iters; ++it) {
                CK(hipEventRecord(e0, st));
                seer_attn_fwd(&ad, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            // back-to-back launches (the kernel's own duration without the event pair's overhead)
            CK(hipEventRecord(e0, st));
            for (int it = 0; it < iters; ++it) seer_attn_fwd(&ad, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float msb;
            CK(hipEventElapsedTime(&msb, e0, e1));
            const double us = msb * 1e3 / iters;
            printf("   variant %d: rel_l2 %.3e  max_abs %.3e  nan %ld | median %.1f us  min %.1f us | back-to-back %.1f us = %.0f TF (%.3f of 2.5 PF)\n",
                   v, sqrt(num / std::max(den, 1e-30)), maxe, nanc, ts[ts.size() / 2], ts[0], us, flops / us * 1e-6,
                   flops / us * 1e-6 / 2500.0);
            fflush(stdout);
        }
        CK(hipFree(dq)); CK(hipFree(dkv)); CK(hipFree(dout)); CK(hipFree(dref)); CK(hipFree(dqt)); CK(hipFree(dkt));
    }
    return 0;
}
