// Which packed-fp32 instruction forms lose a term on MI355X next to a co-tenant process?  (profiles/r03_flake_root_cause.md)
//
// Every form is one kernel: each lane holds a = (a0, a1), b = (b0, b1), c = (c0, c1), and ITER times over
//     d <- b (two v_mov_b32);  <the instruction under test, in inline asm, on d / a / c>;  compare d with the host's expectation
// and counts the iterations whose result differs.  The process forks BEFORE touching HIP: the child loops over the same kernels
// as the co-tenant (`--cotenant 0` runs alone).  Prints per form: launches, wrong launches, wrong lanes histogram by lane/16.
//
//     hipcc --offload-arch=gfx950 -O2 -o build/lab_pkswap scripts/lab_pkswap.cpp && build/lab_pkswap --seconds 12
#include <hip/hip_runtime.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

struct Form {
    const char* name;
    const char* text;
};

// %0 = d (in/out pair), %1 = o (separate output pair, pre-set to b), %2 = a, %3 = c
#define FORMS(X)                                                                                                              \
    X(0, "scalar v_fma_f32 x2, aliased (control)", "(in C++)")              \
    X(1, "A  pk_fma d,a,d,c  src1 swapped, dst=src1", "v_pk_fma_f32 %0, %2, %0, %3 op_sel:[0,1,0] op_sel_hi:[1,0,0]")          \
    X(2, "A' pk_fma o,a,d,c  src1 swapped, no alias", "v_pk_fma_f32 %1, %2, %0, %3 op_sel:[0,1,0] op_sel_hi:[1,0,0]")          \
    X(3, "B  pk_mul d,a,d    src0 swapped, dst=src1", "v_pk_mul_f32 %0, %2, %0 op_sel:[1,0] op_sel_hi:[0,1]")                 \
    X(4, "C  pk_add d,d,a    src1 swapped, dst=src0", "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]")                 \
    X(5, "C' pk_add o,d,a    src1 swapped, no alias", "v_pk_add_f32 %1, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]")                 \
    X(6, "D  pk_fma d,d,a,c  src1 broadcast lo, dst=src0", "v_pk_fma_f32 %0, %0, %2, %3 op_sel_hi:[1,0,1]")                    \
    X(7, "E  pk_fma d,a,c,d  plain accumulate", "v_pk_fma_f32 %0, %2, %3, %0")                                                \
    X(8, "F  pk_fma d,d,a,c  src0 swapped, dst=src0", "v_pk_fma_f32 %0, %0, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]")          \
    X(9, "G  pk_fma d,a,c,d  src2 swapped, dst=src2", "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,1] op_sel_hi:[1,1,0]")          \
    X(10, "H  pk_mul d,d,a    src1 swapped, dst=src0", "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]")                \
    X(11, "I  pk_mul o,a,d    src0 swapped, no alias", "v_pk_mul_f32 %1, %2, %0 op_sel:[1,0] op_sel_hi:[0,1]")                \
    X(12, "J  pk_mul d,d,a    src1 hi broadcast, dst=src0", "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1]")                          \
    X(13, "K  pk_fma d,a,c,d  src1 hi broadcast, dst=src2", "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0]")                     \
    X(14, "L  pk_mul d,a,d    src0 hi broadcast, dst=src1", "v_pk_mul_f32 %0, %2, %0 op_sel:[1,0]")                          \
    X(15, "M  pk_mov d,a,c    lo<-src0.hi, hi<-src1.lo", "v_pk_mov_b32 %0, %2, %3 op_sel:[1,0]")                             \
    X(16, "N  pk_add d,d,a    src1 swapped + negated, dst=src0", "v_pk_add_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]") \
    X(17, "O  pk_fma d,a,c,d  src2 hi broadcast, dst=src2", "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,1]")

constexpr int NFORMS = 18;

// the expectation, on the host, with the same roundings (fmaf is fused; plain products and sums round once)
static void expect(int form, const float a[2], const float b[2], const float c[2], float out[2]) {
    const float d0 = b[0], d1 = b[1];
    switch (form) {
        case 0: { const float lo = fmaf(a[0], d1, c[0]); out[0] = lo; out[1] = fmaf(a[1], lo, c[1]); break; }   // sequential!
        case 1: case 2: out[0] = fmaf(a[0], d1, c[0]); out[1] = fmaf(a[1], d0, c[0]); break;    // op_sel_hi[2] = 0: c.lo twice
        case 3: case 11: out[0] = a[1] * d0; out[1] = a[0] * d1; break;
        case 4: case 5: out[0] = d0 + a[1]; out[1] = d1 + a[0]; break;
        case 6: out[0] = fmaf(d0, a[0], c[0]); out[1] = fmaf(d1, a[0], c[1]); break;
        case 7: out[0] = fmaf(a[0], c[0], d0); out[1] = fmaf(a[1], c[1], d1); break;
        case 8: out[0] = fmaf(d1, a[0], c[0]); out[1] = fmaf(d0, a[1], c[1]); break;
        case 9: out[0] = fmaf(a[0], c[0], d1); out[1] = fmaf(a[1], c[1], d0); break;
        case 10: out[0] = d0 * a[1]; out[1] = d1 * a[0]; break;
        case 12: out[0] = d0 * a[1]; out[1] = d1 * a[1]; break;
        case 13: out[0] = fmaf(a[0], c[1], d0); out[1] = fmaf(a[1], c[1], d1); break;
        case 14: out[0] = a[1] * d0; out[1] = a[1] * d1; break;
        case 15: out[0] = a[1]; out[1] = c[0]; break;
        case 16: out[0] = d0 - a[1]; out[1] = d1 - a[0]; break;
        case 17: out[0] = fmaf(a[0], c[0], d1); out[1] = fmaf(a[1], c[1], d1); break;
    }
}

template <int FORM>
__global__ void __launch_bounds__(256) form_kernel(const f32x2* __restrict__ A, const f32x2* __restrict__ B,
                                                   const f32x2* __restrict__ C, const f32x2* __restrict__ E, int iters,
                                                   unsigned* __restrict__ wrong, f32x2* __restrict__ sample) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const f32x2 a = A[t], b = B[t], c = C[t], e = E[t];
    unsigned bad = 0;
    f32x2 got = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x2 d = b, o = b;
        asm volatile("; d <- b" : "+v"(d), "+v"(o));         // opaque: both pairs are re-materialised from b every iteration
        if constexpr (FORM == 0) {
            float lo = d[0], hi = d[1];
            asm volatile("v_fma_f32 %0, %2, %1, %4\n v_fma_f32 %1, %3, %0, %5" : "+v"(lo), "+v"(hi) : "v"(a[0]), "v"(a[1]), "v"(c[0]), "v"(c[1]));
            d = f32x2{lo, hi};
        }
#define BODY(ID, NAME, TEXT)                                                        \
    if constexpr (FORM == ID && ID != 0) {                                          \
        asm volatile("s_nop 1\n " TEXT : "+v"(d), "+v"(o) : "v"(a), "v"(c));        \
    }
        FORMS(BODY)
#undef BODY
        const f32x2 r = (FORM == 2 || FORM == 5 || FORM == 11) ? o : d;
        if (__float_as_uint(r[0]) != __float_as_uint(e[0]) || __float_as_uint(r[1]) != __float_as_uint(e[1])) {
            ++bad;
            got = r;
        }
    }
    if (bad) {
        atomicAdd(&wrong[0], 1u);                         // lanes with at least one wrong iteration
        atomicAdd(&wrong[1 + (threadIdx.x & 63) / 16], bad);
        sample[0] = got;
        sample[1] = e;
        sample[2] = a;
        sample[3] = b;
        sample[4] = c;
        sample[5] = f32x2{(float)(threadIdx.x & 63), (float)blockIdx.x};
    }
}

typedef void (*kern_t)(const f32x2*, const f32x2*, const f32x2*, const f32x2*, int, unsigned*, f32x2*);
#define KPTR(ID, NAME, TEXT) form_kernel<ID>,
static kern_t KERNELS[NFORMS] = {FORMS(KPTR)};
#define KNAME(ID, NAME, TEXT) NAME,
static const char* NAMES[NFORMS] = {FORMS(KNAME)};

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Bufs {
    f32x2 *A, *B, *C, *E[NFORMS], *sample;
    unsigned* wrong;
    int n;
};

static Bufs setup(int blocks, bool verify) {
    Bufs u{};
    u.n = blocks * 256;
    std::vector<f32x2> a(u.n), b(u.n), c(u.n), e(u.n);
    unsigned s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return 0.25f + (float)((s >> 8) & 0xffff) / 65536.0f * 3.5f;      // [0.25, 3.75): no denormals, no overflow
    };
    for (int i = 0; i < u.n; ++i) {
        a[i] = f32x2{rnd(), -rnd()};
        b[i] = f32x2{rnd(), rnd()};
        c[i] = f32x2{-rnd(), rnd()};
    }
    const size_t bytes = sizeof(f32x2) * u.n;
    CK(hipMalloc(&u.A, bytes));
    CK(hipMalloc(&u.B, bytes));
    CK(hipMalloc(&u.C, bytes));
    CK(hipMemcpy(u.A, a.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(u.B, b.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(u.C, c.data(), bytes, hipMemcpyHostToDevice));
    for (int f = 0; f < NFORMS; ++f) {
        for (int i = 0; i < u.n; ++i) {
            const float aa[2] = {a[i][0], a[i][1]}, bb[2] = {b[i][0], b[i][1]}, cc[2] = {c[i][0], c[i][1]};
            float o[2];
            expect(f, aa, bb, cc, o);
            e[i] = f32x2{o[0], o[1]};
        }
        CK(hipMalloc(&u.E[f], bytes));
        CK(hipMemcpy(u.E[f], e.data(), bytes, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&u.wrong, 8 * sizeof(unsigned)));
    CK(hipMalloc(&u.sample, 8 * sizeof(f32x2)));
    (void)verify;
    return u;
}

int main(int argc, char** argv) {
    double seconds = 10.0;
    int cotenant = 1, blocks = 2048, iters = 4096, only = -1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--seconds")) seconds = atof(argv[++i]);
        else if (!strcmp(argv[i], "--cotenant")) cotenant = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--blocks")) blocks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--form")) only = atoi(argv[++i]);
    }
    pid_t child = 0;
    if (cotenant) {
        child = fork();                                   // before any HIP call
        if (child == 0) {
            Bufs u = setup(blocks, false);
            for (;;)
                for (int f = 0; f < NFORMS; ++f) {
                    hipLaunchKernelGGL(KERNELS[f], dim3(blocks), dim3(256), 0, 0, u.A, u.B, u.C, u.E[f], iters, u.wrong, u.sample);
                    CK(hipDeviceSynchronize());
                }
        }
        sleep(8);                                         // let the co-tenant reach its loop
    }
    Bufs u = setup(blocks, true);
    printf("lab_pkswap: %d lanes, %d iterations per launch, %.0f s per form, cotenant=%d\n", u.n, iters, seconds, cotenant);
    for (int f = 0; f < NFORMS; ++f) {
        if (only >= 0 && f != only) continue;
        long launches = 0, wrong_launches = 0;
        unsigned long q[4] = {0, 0, 0, 0};
        f32x2 smp[8];
        bool have = false;
        const double t0 = now();
        while (now() - t0 < seconds) {
            CK(hipMemset(u.wrong, 0, 8 * sizeof(unsigned)));
            hipLaunchKernelGGL(KERNELS[f], dim3(blocks), dim3(256), 0, 0, u.A, u.B, u.C, u.E[f], iters, u.wrong, u.sample);
            unsigned w[8];
            CK(hipMemcpy(w, u.wrong, sizeof(w), hipMemcpyDeviceToHost));
            ++launches;
            if (w[0]) {
                ++wrong_launches;
                for (int k = 0; k < 4; ++k) q[k] += w[1 + k];
                if (!have) {
                    CK(hipMemcpy(smp, u.sample, sizeof(smp), hipMemcpyDeviceToHost));
                    have = true;
                }
            }
        }
        printf("form %2d %-52s launches %6ld wrong %5ld  wrong iterations by lane quarter [%lu %lu %lu %lu]\n", f, NAMES[f],
               launches, wrong_launches, q[0], q[1], q[2], q[3]);
        if (have)
            printf("         sample: lane %.0f block %.0f got (%.9g, %.9g) want (%.9g, %.9g)  a (%.9g, %.9g) b (%.9g, %.9g) c (%.9g, %.9g)\n",
                   smp[5][0], smp[5][1], smp[0][0], smp[0][1], smp[1][0], smp[1][1], smp[2][0], smp[2][1], smp[3][0], smp[3][1],
                   smp[4][0], smp[4][1]);
        fflush(stdout);
    }
    if (child > 0) {
        kill(child, SIGKILL);
        waitpid(child, nullptr, 0);
    }
    return 0;
}
