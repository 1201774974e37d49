// Flash attention forward for head_dim 40 on gfx950 (MI355X): the 32x32-latent level of the Seer UNet
// (spatial self-attention [B*F*8, 1024, 40], text cross-attention Sk = 77, temporal window attention 768 causal keys).
// Second-generation kernel next to attention.hip (which keeps head dims 80 / 96 / 160); same descriptor, same semantics
// (include/seer_hip.h, seer_attn_fwd; reference: seer/models/attention.py:622-630, 632-703).
//
// Why a kernel of its own.  At d = 40 a score costs 80 MFMA FLOPs and one exp: measured on one SIMD (scripts/lab_issue.cpp,
// profiles/r02_lab_issue.log) v_exp_f32 issues every 9.3 cycles, v_cvt_pk_bf16_f32 every 5.2, v_permlane16_swap every 9.1,
// and every MFMA takes 8 cycles of the same vector issue port.  A 32-query x 32-key block therefore needs
//     16 exp = 149 cycles of the issue port,  7 MFMAs (3 QK^T + 4 PV, 32x32x16) = 56 more and 240 of the matrix pipe,
// so the kernel is bound by the transcendental rate, not by the matrix pipe, and everything else on the vector port has to go:
//   * Q is pre-multiplied by scale*log2(e) (by the producing GEMM, SEER_ATTN_Q_PRESCALED, or here once per block), so the
//     MFMA result is already in the exp2 domain;
//   * the softmax reference m rides in the contraction's padding: 40-wide rows are contracted in three 16-wide steps,
//     column 40 of K' is 1.0 and column 40 of Q' is -m (a bf16-exact value), so S' = s - m comes out of the MFMA;
//   * the denominator is row 40 of O^T (column 40 of V' is 1.0): no adds;
//   * fast path: m is fixed after the first 32 keys.  With fp32 accumulation and bf16 P (8 exponent bits) any reference
//     within ~2^100 of the true maximum is exact to rounding; a block whose scores run further than that overflows to
//     inf, is detected at the end (an accumulator is not finite) and is redone in the same launch by the TRACK form,
//     which re-bases m whenever a score exceeds it by more than 2^16;
//   * fast path: P is packed to bf16 by truncation (one v_perm_b32 per pair instead of v_cvt_pk_bf16_f32).  The same
//     truncated P feeds numerator and denominator, so the bias cancels and the random error has the variance of
//     round-to-nearest.  The TRACK form (also used whenever lse is requested: the backward rebuilds P from it) rounds.
//   * no lane movement between the two products: S^T = K' Q'^T by 32x32x16 MFMAs puts one query on a lane; the packed
//     accumulators ARE the B operand of O^T += V'^T P^T, again 32x32x16 (the 16x16x32 form would save a quarter of the
//     matrix cycles but needs four v_permlane16_swap per block: 36 cycles of the port that is the bottleneck).
//   * K / V tiles (128 keys) arrive by LDS-DMA (global_load_lds, 16 B per lane) into dense 80-byte rows, one barrier per
//     tile, the next tile in flight under the current one.  Padding columns are not stored: the lanes that would read
//     them read a small constant region instead.  V rows sit in a permuted order that makes the transposed reads
//     (ds_read_b64_tr_b16) bank-conflict free on 80-byte rows.
//   * blocks are handed out in XCD-contiguous chunks so that the heads of one frame (adjacent columns of the fused q|k|v
//     rows) meet in one L2.
#include "seer_common.h"

namespace {

constexpr int A40_D = 40;
constexpr int A40_KT = 128;                         // keys per LDS stage
constexpr int A40_ROWB = 80;                        // bytes per K / V row in LDS
constexpr int A40_HALF = A40_KT * A40_ROWB;         // 10240: K image, then V image
constexpr int A40_STAGE = 2 * A40_HALF;             // 20480
constexpr int A40_ZBYTES = 1472;                    // constant region: zeros with bf16 1.0 at bytes 0, 160, 1280, 1440
constexpr float A40_NEG_INF = -__builtin_inff();
constexpr float A40_THR = 16.0f;                    // TRACK: re-base when a score exceeds the reference by 2^16
constexpr float A40_THR_F16 = 14.0f;                // ... by 2^14 on IEEE-half operands (P <= 2^14: the half range ends at 2^16)

struct TokMap40 {
    int ws_log2;   // -1: identity
    int HW, W_, wy0, wx0;
    __device__ __forceinline__ int operator()(int pos) const {
        if (ws_log2 < 0) return pos;
        const int ws2 = 2 * ws_log2;
        const int f = pos >> ws2;
        const int rem = pos & ((1 << ws2) - 1);
        const int wy = rem >> ws_log2, wx = rem & ((1 << ws_log2) - 1);
        return f * HW + (wy0 + wy) * W_ + wx0 + wx;
    }
};

// a half-representable value >= x, at most one half ulp above it (x finite, |x| < 60000): the softmax reference of the half form
__device__ __forceinline__ float f16_ceil(float x) {
    x = fminf(fmaxf(x, -60000.f), 60000.f);
    if (x > 0.f && x < 6.2e-5f) return 6.103515625e-5f;
    if (x <= 0.f && x > -6.2e-5f) return 0.f;
    _Float16 h = (_Float16)x;
    float f = (float)h;
    if (f < x) {
        unsigned short b = __builtin_bit_cast(unsigned short, h);
        b = (b & 0x8000u) ? (unsigned short)(b - 1) : (unsigned short)(b + 1);
        f = (float)__builtin_bit_cast(_Float16, b);
    }
    return f;
}
// smallest bf16-representable value >= x (x finite)
__device__ __forceinline__ float bf16_ceil(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    if (!(u >> 31)) u += 0xffffu;        // positive: round the magnitude up; negative: truncation already rounds up
    return __builtin_bit_cast(float, u & 0xffff0000u);
}
// v_permlane32_swap a, b:  a' = [a.lo | b.lo], b' = [a.hi | b.hi].  Inline asm: the second result of
// __builtin_amdgcn_permlane{16,32}_swap is not reliable with hipcc ROCm 7.2 (see attention.hip, scripts/lab_probe3.cpp);
// the s_nops are the VALU -> permlane hazard, which nobody pads inside an asm statement.
__device__ __forceinline__ float xhalf_max40(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float lower_half_value40(float v) {      // value of lane (l & 31) in every lane l
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a;
}

// LDS fragment reads in inline asm.  hipcc (ROCm 7.2) orders every LDS load it can see behind ALL pending LDS-DMA
// (s_waitcnt vmcnt(0) in front of the first ds_read after each global_load_lds: the next tile's whole L2 / HBM latency,
// exposed once per tile -- with one LDS array or with one per stage); the ordering that matters is ours: vmcnt, then the
// workgroup barrier, then the reads.  Loads and their lgkmcnt wait share ONE statement with early-clobber outputs, so no
// compiler-placed instruction can touch a destination register before the data has landed.
__device__ __forceinline__ unsigned lds_addr(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}
__device__ __forceinline__ void lds_read_k3(unsigned a01, unsigned a2, bf16x8& k0, bf16x8& k1, bf16x8& k2) {
    u32x4 r0, r1, r2;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:32\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(a01), "v"(a2) : "memory");
    k0 = __builtin_bit_cast(bf16x8, r0);
    k1 = __builtin_bit_cast(bf16x8, r1);
    k2 = __builtin_bit_cast(bf16x8, r2);
}
// the V'^T fragments of one 32-key sub tile: 2 key steps x 2 d tiles, each two transposed 8-byte reads.
// a0: d tile 0 (value columns 0..31), a1: d tile 1 (columns 32..39 | 1.0 | zeros; constant region in the padding lanes).
// Key step 1 is 16 rows = 1280 B further on, the second read of a fragment 2 rows = 160 B.
__device__ __forceinline__ void lds_read_v4(unsigned a0, unsigned a1, bf16x8& v00, bf16x8& v01, bf16x8& v10, bf16x8& v11) {
    u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:160\n\t"
                 "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:160\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:1280\n\tds_read_b64_tr_b16 %5, %8 offset:1440\n\t"
                 "ds_read_b64_tr_b16 %6, %9 offset:1280\n\tds_read_b64_tr_b16 %7, %9 offset:1440\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(a0), "v"(a1) : "memory");
    v00 = __builtin_bit_cast(bf16x8, u32x4{r0[0], r0[1], r1[0], r1[1]});      // key step 0, d tile 0
    v01 = __builtin_bit_cast(bf16x8, u32x4{r2[0], r2[1], r3[0], r3[1]});      // key step 0, d tile 1
    v10 = __builtin_bit_cast(bf16x8, u32x4{r4[0], r4[1], r5[0], r5[1]});      // key step 1, d tile 0
    v11 = __builtin_bit_cast(bf16x8, u32x4{r6[0], r6[1], r7[0], r7[1]});      // key step 1, d tile 1
}

// K' and V'^T fragments of one sub tile in one issue: the three K' reads are waited for here (LDS returns in order, so
// lgkmcnt(8) = "all but the 8 V reads"), the V'^T reads stay in flight under the QK^T MFMAs and the exponentials and are
// waited for by lds_wait_v right before the PV MFMAs.  The V registers are written by the hardware AFTER this statement
// ends: between the two statements they must not be read, copied or spilled -- they are only ever named again as the
// read-write operands of lds_wait_v, and the build is checked for moves of those registers (scripts/check_attn40_asm.py).
struct VRegs { u32x2 r[8]; };
__device__ __forceinline__ void lds_issue_kv(unsigned a01, unsigned a2, unsigned v0, unsigned v1, bf16x8& k0, bf16x8& k1,
                                             bf16x8& k2, VRegs& v) {
    u32x4 r0, r1, r2;
    asm volatile("ds_read_b128 %0, %11\n\tds_read_b128 %1, %11 offset:32\n\tds_read_b128 %2, %12\n\t"
                 "ds_read_b64_tr_b16 %3, %13\n\tds_read_b64_tr_b16 %4, %13 offset:160\n\t"
                 "ds_read_b64_tr_b16 %5, %14\n\tds_read_b64_tr_b16 %6, %14 offset:160\n\t"
                 "ds_read_b64_tr_b16 %7, %13 offset:1280\n\tds_read_b64_tr_b16 %8, %13 offset:1440\n\t"
                 "ds_read_b64_tr_b16 %9, %14 offset:1280\n\tds_read_b64_tr_b16 %10, %14 offset:1440\n\ts_waitcnt lgkmcnt(8)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(v.r[0]), "=&v"(v.r[1]), "=&v"(v.r[2]), "=&v"(v.r[3]), "=&v"(v.r[4]),
                   "=&v"(v.r[5]), "=&v"(v.r[6]), "=&v"(v.r[7])
                 : "v"(a01), "v"(a2), "v"(v0), "v"(v1) : "memory");
    k0 = __builtin_bit_cast(bf16x8, r0);
    k1 = __builtin_bit_cast(bf16x8, r1);
    k2 = __builtin_bit_cast(bf16x8, r2);
}
__device__ __forceinline__ void lds_wait_v(VRegs& v, bf16x8& v00, bf16x8& v01, bf16x8& v10, bf16x8& v11) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v.r[0]), "+v"(v.r[1]), "+v"(v.r[2]), "+v"(v.r[3]), "+v"(v.r[4]), "+v"(v.r[5]), "+v"(v.r[6]), "+v"(v.r[7])
                 :: "memory");
    v00 = __builtin_bit_cast(bf16x8, u32x4{v.r[0][0], v.r[0][1], v.r[1][0], v.r[1][1]});      // key step 0, d tile 0
    v01 = __builtin_bit_cast(bf16x8, u32x4{v.r[2][0], v.r[2][1], v.r[3][0], v.r[3][1]});      // key step 0, d tile 1
    v10 = __builtin_bit_cast(bf16x8, u32x4{v.r[4][0], v.r[4][1], v.r[5][0], v.r[5][1]});      // key step 1, d tile 0
    v11 = __builtin_bit_cast(bf16x8, u32x4{v.r[6][0], v.r[6][1], v.r[7][0], v.r[7][1]});      // key step 1, d tile 1
}

// Ring form (NST = 3): the K' fragments of sub tile i + 1 are requested right behind the Q K^T MFMAs of sub tile i (no wait: they land
// under the exponentials), sub tile i + 1 then issues only its V'^T reads.  LDS returns in order, so lgkmcnt(0) in lds_wait_vk covers
// both; the three K registers obey the same rule as the V registers: between the issuing statement and the waiting one nothing may
// name them (seervideoldm_amd/asm_check.py, rule attn40_vregs).
struct KRegs { u32x4 r[3]; };
__device__ __forceinline__ void lds_prefetch_k(unsigned a01, unsigned a2, KRegs& k) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:32\n\tds_read_b128 %2, %4"
                 : "=&v"(k.r[0]), "=&v"(k.r[1]), "=&v"(k.r[2]) : "v"(a01), "v"(a2) : "memory");
}
__device__ __forceinline__ void lds_issue_v(unsigned v0, unsigned v1, VRegs& v) {
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:160\n\t"
                 "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:160\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:1280\n\tds_read_b64_tr_b16 %5, %8 offset:1440\n\t"
                 "ds_read_b64_tr_b16 %6, %9 offset:1280\n\tds_read_b64_tr_b16 %7, %9 offset:1440"
                 : "=&v"(v.r[0]), "=&v"(v.r[1]), "=&v"(v.r[2]), "=&v"(v.r[3]), "=&v"(v.r[4]), "=&v"(v.r[5]), "=&v"(v.r[6]), "=&v"(v.r[7])
                 : "v"(v0), "v"(v1) : "memory");
}
__device__ __forceinline__ void lds_wait_vk(VRegs& v, KRegs& k, bf16x8& v00, bf16x8& v01, bf16x8& v10, bf16x8& v11) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v.r[0]), "+v"(v.r[1]), "+v"(v.r[2]), "+v"(v.r[3]), "+v"(v.r[4]), "+v"(v.r[5]), "+v"(v.r[6]), "+v"(v.r[7]),
                   "+v"(k.r[0]), "+v"(k.r[1]), "+v"(k.r[2])
                 :: "memory");
    v00 = __builtin_bit_cast(bf16x8, u32x4{v.r[0][0], v.r[0][1], v.r[1][0], v.r[1][1]});
    v01 = __builtin_bit_cast(bf16x8, u32x4{v.r[2][0], v.r[2][1], v.r[3][0], v.r[3][1]});
    v10 = __builtin_bit_cast(bf16x8, u32x4{v.r[4][0], v.r[4][1], v.r[5][0], v.r[5][1]});
    v11 = __builtin_bit_cast(bf16x8, u32x4{v.r[6][0], v.r[6][1], v.r[7][0], v.r[7][1]});
}

template <bool B> struct BoolTag { static constexpr bool value = B; };

// TRACK_ONLY: run the tracked-reference form directly (variant 4 / 5, and whenever lse is requested)
#ifdef SEER_ATTN40_STAMPS      // measurement-only build (scripts/lab_a40stamps.cpp): s_memrealtime stamps of the first 16 blocks
__device__ long long seer_a40_stamps[16 * 4 * 128];
extern "C" long long* seer_lab_a40_stamps() {
    long long* ptr = nullptr;
    (void)hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(seer_a40_stamps));
    return ptr;
}
#define A40_STAMP()                                                                                           \
    do {                                                                                                      \
        if (blockIdx.x < 16 && lane == 0 && nst < 128) seer_a40_stamps[(blockIdx.x * 4 + wave) * 128 + nst] = wall_clock64(); \
        ++nst;                                                                                                \
    } while (0)
#else
#define A40_STAMP() do {} while (0)
#endif

// PLAIN: not causal, not windowed (the spatial self-attention and text cross-attention blocks): the window / diagonal
// arithmetic folds away -- and the two uses of the kernel carry different names in a profile (the spatial [192,1024,40] block is
// the north star's kernel target; the causal window form serves the temporal blocks at 36 us, and one name for both reads 43)
// NST: K|V stages in LDS.  2 = the next tile is requested behind this tile's barrier (five LDS-DMA pieces per wave in one burst) and
// waited for with vmcnt(0) at the top of the next tile.  3 (ring): tile t + 2 is requested DURING tile t, one piece per sub tile behind
// the sub tile's Q K^T MFMAs, and the top of a tile waits only for ITS pieces (vmcnt(5): the next tile's five stay in flight) -- the
// per-tile wait + barrier + issue burst was a third of a tile period (profiles/r02_attn40_stamps.log).
// F16 (SEER_ATTN_F16): IEEE-half operands.  Only the TRACKED form exists there: the fast path fixes its reference after 32 keys and lives
// off bf16's 8 exponent bits (P up to 2^127); with the reference tracked P <= 2^14 fits the half range.
template <int QB, bool TRACK_ONLY, bool PLAIN, int NST = 2, bool F16 = false>
__global__ void __launch_bounds__(256, (QB == 1 && NST == 2) ? 3 : 2) seer_attn40_kernel(const seer_attn_desc p, const int ws_log2_arg, const int nqb) {
    const int ws_log2 = PLAIN ? -1 : ws_log2_arg;
    const bool causal = PLAIN ? false : (p.causal != 0);
    static_assert(!F16 || TRACK_ONLY, "IEEE-half operands: the tracked form only");
    constexpr int D = A40_D;
    constexpr int QW = 32 * QB;                      // queries per wave
    // NST K|V stages + the constant region
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * A40_STAGE + A40_ZBYTES];
    unsigned char* const stage_a = lds;
    unsigned char* const stage_b = lds + A40_STAGE;
    [[maybe_unused]] unsigned char* const stage_c = lds + (NST - 1) * A40_STAGE;
    unsigned char* const zreg = lds + NST * A40_STAGE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;        // S^T / O^T layout: query column, key (or d row) half
    [[maybe_unused]] int nst = 0;
    A40_STAMP();

    // ---- block -> (batch, head, query block): XCD-contiguous chunks of the (b, head, qblk) order
    int wg;
    {
        const int nwg = gridDim.x, L = blockIdx.x;
        const int xcd = L & 7, q_ = nwg >> 3, r_ = nwg & 7;
        wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + (L >> 3);
    }
    const int qblk = wg % nqb;
    const int y = wg / nqb;
    const int head = y % p.heads;
    int b = y / p.heads;
    TokMap40 tok;
    tok.ws_log2 = ws_log2;
    tok.HW = p.H * p.W;
    tok.W_ = p.W;
    tok.wy0 = tok.wx0 = 0;
    if (ws_log2 >= 0) {
        const int win = b / p.batch;
        b = b - win * p.batch;
        const int nwx = p.W >> ws_log2;
        tok.wy0 = (win / nwx) << ws_log2;
        tok.wx0 = (win % nwx) << ws_log2;
    }
    const bf16* __restrict__ Qg = reinterpret_cast<const bf16*>(p.Q) + (int64_t)b * p.q_bs + head * (p.q_hs ? p.q_hs : D);
    const bf16* __restrict__ Kg = reinterpret_cast<const bf16*>(p.K) + (int64_t)b * p.k_bs + head * (p.k_hs ? p.k_hs : D);
    const bf16* __restrict__ Vg = reinterpret_cast<const bf16*>(p.V) + (int64_t)b * p.v_bs + head * (p.v_hs ? p.v_hs : D);
    bf16* __restrict__ Og = reinterpret_cast<bf16*>(p.O) + (int64_t)b * p.o_bs + head * D;

    const int qblk0 = qblk * (4 * QW);
    const int q0w = qblk0 + wave * QW;               // first query of this wave
    const int q_off = PLAIN ? 0 : p.causal_offset;

    // ---- constant region: zeros, with {1.0, 0, 0, 0} at the four offsets a padding lane's reads land on.
    //      K' column 40.. = {1, 0 x7} = its first 16 bytes; V' columns 40..43 = {1, 0, 0, 0} = zreg + 0 (+160, +1280, +1440),
    //      V' columns 44.. = zeros = zreg + 8 (+160, ...).
    for (int i = tid; i < A40_ZBYTES / 4; i += 256) {
        const int off = i * 4;
        reinterpret_cast<unsigned*>(zreg)[i] = (off == 0 || off == 160 || off == 1280 || off == 1440) ? (F16 ? 0x3c00u : 0x3f80u) : 0u;
    }
    const unsigned ZONE = lds_addr(zreg);

    // ---- keys this block needs
    int k_end = p.Sk;
    if (causal) k_end = min(p.Sk, min(qblk0 + 4 * QW, p.Sq) + q_off);
    const int ntiles = (k_end + A40_KT - 1) / A40_KT;

    // ---- LDS-DMA plan: 20 wave instructions per tile (10 K + 10 V, 1 KiB each), wave w issues e = w + 4 i.
    // Source pointers are kept per slot and advanced by a constant per tile (128 keys = a whole number of frames of a
    // window, so the step is constant in the window form too); only a partial last tile recomputes them with the clamp.
    // Kept per slot: ONE 32-bit byte offset from the wave-uniform K / V base of the (batch, head) -- the row / chunk of a lane are
    // recomputed where they are needed (set-up and partial tiles), so the plan costs 5 registers, not 20.
    unsigned dma_o[5];
    const int tok_step = ws_log2 < 0 ? A40_KT : (A40_KT >> (2 * ws_log2)) * tok.HW;   // tokens per tile
    auto dma_off = [&](int i, int kt0) {
        const int e = wave + 4 * i;
        const bool is_v = e >= 10;
        const int j = is_v ? e - 10 : e;
        const int idx = 64 * j + lane;               // 16-byte chunk index inside the 128 x 5 image
        const int pos = idx / 5;
        const int c8 = (idx - pos * 5) * 8;          // element offset of the lane's 16-byte chunk
        int key = pos;                               // key (inside the tile) whose row this lane fetches
        if (is_v)     // V rows are stored transposed inside every group of 16: key 4 y + x sits at row 4 x + y, so that the four
            key = (pos & ~15) + 4 * (pos & 3) + ((pos >> 2) & 3);     // keys of a transposed read are 4 rows (320 B) apart
        int kg = kt0 + key;
        kg = kg < p.Sk ? kg : p.Sk - 1;
        const int64_t tk = tok(kg);
        return (unsigned)((tk * (is_v ? p.v_ss : p.k_ss) + c8) * 2);      // < 2^32: checked by the launcher
    };
    auto issue_piece = [&](int t, unsigned char* stage, int i) {   // piece i (0..4) of this wave's share of tile t
        const bool partial = (t + 1) * A40_KT > p.Sk;                 // wave-uniform
        const int e = wave + 4 * i;
        const bool is_v = e >= 10;
        const int j = is_v ? e - 10 : e;
        const unsigned off = partial ? dma_off(i, t * A40_KT) : dma_o[i];
        dma_o[i] += (unsigned)(tok_step * (is_v ? p.v_ss : p.k_ss) * 2);
        const unsigned char* base = reinterpret_cast<const unsigned char*>(is_v ? Vg : Kg);      // wave-uniform
        unsigned char* dst = stage + (is_v ? A40_HALF : 0) + j * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto issue_tile = [&](int t, unsigned char* stage) {
#pragma unroll
        for (int i = 0; i < 5; ++i) issue_piece(t, stage, i);
    };

    // ---- Q' fragments (B operand of S^T = K' Q'^T: col = query, k = 16 s + 8 h + j), scaled to the exp2 domain
    const float cscale = (p.flags & SEER_ATTN_Q_PRESCALED) ? 1.0f : p.scale * 1.4426950408889634f;
    // start of the K|V stream: tile 0 (ring: and tile 1) requested behind the Q loads, in front of everything that waits for Q
    // (ring: tile 1 follows in start_stream_2, BEHIND the conversion of Q -- the compiler's wait for the Q loads is vmcnt(0), and
    //  the cold start is bandwidth-bound: every workgroup of the chip asks for Q and its first tile at once, ~11 B per cycle and CU)
    auto start_stream = [&]() {
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_o[i] = dma_off(i, 0);
        issue_tile(0, stage_a);
    };
    auto start_stream_2 = [&]() {
        if constexpr (NST == 3) {
            if (ntiles > 1) issue_tile(1, stage_b);
        }
    };
    bf16x8 qf[QB][3];
    {
        // all 3 QB loads in flight together (a wait per load serialised six L2 / HBM round trips in front of the first tile)
        u32x4 qraw[QB][3];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            int qi = q0w + 32 * qb + lq;
            qi = qi < p.Sq ? qi : p.Sq - 1;
            const bf16* qrow = Qg + (int64_t)tok(qi) * p.q_ss;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                // (s, lh) = (2, 1) is the padding: those lanes re-read chunk (2, 0) and drop it -- no divergent load
                const u32x4 v = *reinterpret_cast<const u32x4*>(qrow + 16 * s + ((s == 2) ? 0 : 8 * lh));
                qraw[qb][s] = v;
            }
        }
        start_stream();
        const bool pre = (p.flags & SEER_ATTN_Q_PRESCALED) != 0;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                u32x4 raw = qraw[qb][s];
                if (s == 2 && lh) raw = u32x4{0u, 0u, 0u, 0u};
                float f[8];
                unpack8t<F16>(raw, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= cscale;
                const u32x4 scaled = pack8t<F16>(f);
                qf[qb][s] = __builtin_bit_cast(bf16x8, pre ? raw : scaled);
            }
        start_stream_2();
    }

    // ---- per-lane LDS offsets (bytes, relative to the sub tile's first row)
    // K' A fragments: row lq, chunk 2 s + lh; (s, lh) = (2, 1) is the padding -> constant region
    const unsigned k_off = lq * A40_ROWB + 16 * lh;
    // V'^T A fragments (32 d rows x 16 keys) by transposed reads.  Lane l: 16-lane group g = (l >> 4) & 1 takes value columns
    // 16 g .. 16 g + 15 of the d tile; inside the group lane i supplies the address of block row q' = i >> 2 (key 4 h + q' of
    // the key step, stored at row 4 q' + h), column piece c = i & 3.  The second read of a fragment is key + 8 = row + 2.
    const int vg = (lane >> 4) & 1, vq = (lane & 15) >> 2, vc = lane & 3;
    const unsigned v_off = A40_HALF + (4 * vq + lh) * A40_ROWB + 32 * vg + 8 * vc;
    const bool v1_pad = vg == 1 || vc >= 2;          // d tile 1: columns 40.. do not exist in the 80-byte rows
    const unsigned v1_const = ZONE + ((vg == 0 && vc == 2) ? 0 : 8);

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 oacc[QB][2];
    float m_run[QB];

    auto load_kfrags = [&](unsigned kst, bf16x8 (&kf)[3]) {          // kst: LDS address of the sub tile's first K row
        const unsigned kp = kst + k_off;
        lds_read_k3(kp, lh ? ZONE : kp + 64, kf[0], kf[1], kf[2]);
    };
    auto qk = [&](const bf16x8 (&kf)[3], int qb) {
        f32x16 s = mma32<F16>(kf[0], qf[qb][0], zero16);
        s = mma32<F16>(kf[1], qf[qb][1], s);
        s = mma32<F16>(kf[2], qf[qb][2], s);
        return s;
    };
    auto mask_scores = [&](f32x16& s, int kb, int qb) {
        const int qi = q0w + 32 * qb + lq;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool ok = (key < p.Sk) && (!causal || key <= qi + q_off);
            s[r] = ok ? s[r] : A40_NEG_INF;
        }
    };
    auto set_ref = [&](int qb, float m) {            // the reference rides in Q' column 40 (lanes of the upper half)
        m_run[qb] = m;
        if (lh) qf[qb][2][0] = to16<F16>(-m);
    };

    // one pass over the keys; `track` selects the reference handling and the rounding of P.
    // (Measured and dropped, profiles/r02_attn40_variants.log: issuing the QK^T MFMAs of block i+1 before the exponentials of
    //  block i -- +16 registers for the second score tile, 0...-8 %: with three waves per SIMD the other waves already fill
    //  the matrix pipe, and the extra registers cost the 64-query form its third wave.)
    auto run = [&](auto track, const bool started) {
        constexpr bool TRACK = decltype(track)::value;
        if (!started) {
            start_stream();
            start_stream_2();
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            oacc[qb][0] = zero16;
            oacc[qb][1] = zero16;
            set_ref(qb, 0.f);
        }

        // st_a: LDS address of tile t's stage; st_next: the stage of the tile requested during tile t (t + 1, ring: t + 2)
        auto tile = [&](const int t, const unsigned st_a, unsigned char* st_next) {
            A40_STAMP();
            // this wave's part of tile t (and, at t = 0, Q) has landed; ring: the five pieces of tile t + 1 stay in flight
            if (NST == 3 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            A40_STAMP();
            __builtin_amdgcn_s_barrier();                         // all of tile t has landed; everyone left tile t-1
            A40_STAMP();
            if (NST == 2 && t + 1 < ntiles) issue_tile(t + 1, st_next);
            A40_STAMP();
            const int kt0 = t * A40_KT;
            const bool ring_issue = NST == 3 && t + 2 < ntiles;   // wave-uniform
            // ring: this wave's pieces of tile t + 2, spread over the sub tiles (piece 4 rides with piece 1: sub tile 0 already
            // pays the K' read behind the barrier)
            auto ring_pieces = [&](int sub) {
                if (!ring_issue) return;
                issue_piece(t + 2, st_next, sub);
                if (sub == 1) issue_piece(t + 2, st_next, 4);
            };

            if (t == 0) {
                // reference of the softmax: the maximum over the first 32 keys (every query sees key 0)
                bf16x8 kf[3];
                load_kfrags(st_a, kf);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    f32x16 s = qk(kf, qb);
                    if ((31 >= p.Sk) || (causal && 31 > q0w + 32 * qb + q_off)) mask_scores(s, 0, qb);
                    float mx = s[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                    mx = xhalf_max40(mx);
                    set_ref(qb, mx == A40_NEG_INF ? 0.f : (F16 ? f16_ceil(mx) : bf16_ceil(mx)));
                }
            }

            [[maybe_unused]] KRegs knext;                         // ring: K' fragments requested one sub tile ahead
            // ring fast path: whole tiles only (the launcher admits Sq % 256 == 0 and Sk % 128 == 0), so which sub tiles carry
            // prefetched K' is known at compile time -- a run-time choice between two issue statements would make the register
            // allocator merge (copy) registers the LDS is still writing
            constexpr bool RING_FAST = NST == 3 && QB == 2 && PLAIN && !TRACK;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int kb = kt0 + 32 * sub;
                // wave-uniform: past the last key, or above the causal diagonal of every query of the wave
                if (!RING_FAST && (kb >= k_end || q0w >= p.Sq || (causal && kb > q0w + QW - 1 + q_off))) {
                    if constexpr (NST == 3) ring_pieces(sub);     // the wave still owes its share of the K|V stream
                    continue;
                }
                const unsigned kst = st_a + sub * 32 * A40_ROWB;
                bf16x8 kf[3];
                VRegs vraw;
                if constexpr (RING_FAST) {
                    const unsigned kp = kst + k_off, va = kst + v_off;
                    if (sub > 0) {                                // K' landed under the previous sub tile (lds_wait_vk waited)
                        lds_issue_v(va, v1_pad ? v1_const : va + 64, vraw);
                        kf[0] = __builtin_bit_cast(bf16x8, knext.r[0]);
                        kf[1] = __builtin_bit_cast(bf16x8, knext.r[1]);
                        kf[2] = __builtin_bit_cast(bf16x8, knext.r[2]);
                    } else {
                        lds_issue_kv(kp, lh ? ZONE : kp + 64, va, v1_pad ? v1_const : va + 64, kf[0], kf[1], kf[2], vraw);
                    }
                } else {
                    const unsigned kp = kst + k_off, va = kst + v_off;
                    lds_issue_kv(kp, lh ? ZONE : kp + 64, va, v1_pad ? v1_const : va + 64, kf[0], kf[1], kf[2], vraw);
                }
                bf16x8 vf[2][2];                                  // [key step][d tile], shared by the wave's query blocks
                // generic path: the V'^T reads (behind the K' reads in the same LDS queue: a few cycles later) are waited for here, in
                // straight-line code -- a wait under a run-time flag inside the query-block loop leaves the assembly with a path from the
                // issue statement to a use that skips it (never taken, but no check of the assembly can know)
                if constexpr (!(QB == 2 && PLAIN && !TRACK)) lds_wait_v(vraw, vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                if constexpr (NST == 3 && !RING_FAST) ring_pieces(sub);
                if constexpr (QB == 2 && PLAIN && !TRACK) {
                    // two query blocks, software-pipelined: both score tiles are issued before the first exponentials, so the
                    // matrix pipe works on block 1's Q K^T under block 0's exponentials and on block 0's P V under block 1's
                    f32x16 s0 = qk(kf, 0);
                    if constexpr (RING_FAST) {
                        // behind block 0's score MFMAs: the next sub tile's K' reads and this wave's LDS-DMA piece, so that neither
                        // stands between a barrier and the first MFMA of a sub tile
                        if (sub < 3) {
                            const unsigned kpn = kst + 32 * A40_ROWB + k_off;
                            lds_prefetch_k(kpn, lh ? ZONE : kpn + 64, knext);
                        }
                        ring_pieces(sub);
                    }
                    f32x16 s1 = qk(kf, 1);
                    if constexpr (!RING_FAST) {       // (whole tiles only on the ring: no key is ever masked there -- hipcc turns this
                        if (kb + 31 >= p.Sk) {        //  branch into 32 v_cndmask per sub tile, executed every time)
                            mask_scores(s0, kb, 0);
                            mask_scores(s1, kb, 1);
                        }
                    }
                    auto exp_pack = [&](const f32x16& sc, u32x4 (&pk)[2]) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float e0 = __builtin_amdgcn_exp2f(sc[2 * i]), e1 = __builtin_amdgcn_exp2f(sc[2 * i + 1]);
                            pk[i >> 2][i & 3] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, e1), __builtin_bit_cast(unsigned, e0),
                                                                      0x07060302u);
                        }
                    };
                    auto pv = [&](int qb, const u32x4 (&pk)[2]) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const bf16x8 pb = __builtin_bit_cast(bf16x8, pk[s2]);
                            oacc[qb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][0], pb, oacc[qb][0], 0, 0, 0);
                            oacc[qb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s2][1], pb, oacc[qb][1], 0, 0, 0);
                        }
                    };
                    u32x4 pk0[2], pk1[2];
                    exp_pack(s0, pk0);
                    // (measured and dropped, profiles/r05_attn40_ring.log: block 0's exponentials scheduled into the gaps of block 1's three
                    //  score MFMAs by sched_group_barrier -- [192,1024,40] 50.9 -> 52.3 us, [192,4096,40] 556 -> 570 us)
                    if constexpr (NST == 3) {
                        if (sub < 3) lds_wait_vk(vraw, knext, vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                        else lds_wait_v(vraw, vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                    } else {
                        lds_wait_v(vraw, vf[0][0], vf[0][1], vf[1][0], vf[1][1]);
                    }
                    pv(0, pk0);
                    exp_pack(s1, pk1);
                    pv(1, pk1);
                    A40_STAMP();
                    continue;
                }
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (causal && kb > q0w + 32 * qb + 31 + q_off) continue;      // this query block is above the diagonal
                    f32x16 s = qk(kf, qb);                                          // S' = s - m, exp2 domain
                    if ((kb + 31 >= p.Sk) || (causal && kb + 31 > q0w + 32 * qb + q_off)) mask_scores(s, kb, qb);
                    if constexpr (TRACK) {
                        float mx = s[0];
#pragma unroll
                        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                        mx = xhalf_max40(mx);
                        constexpr float THR = F16 ? A40_THR_F16 : A40_THR;
                        if (!__all(mx <= THR)) {
                            const float m_new = mx > THR ? (F16 ? f16_ceil(m_run[qb] + mx) : bf16_ceil(m_run[qb] + mx)) : m_run[qb];
                            const float delta = m_new - m_run[qb];
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                s[r] -= delta;
                                oacc[qb][0][r] *= alpha;
                                oacc[qb][1][r] *= alpha;
                            }
                            set_ref(qb, m_new);
                        }
                    }
                    // P'^T = exp2(S') packed to bf16: registers 8 s2 .. 8 s2 + 7 are the B fragment of key step s2
                    u32x4 pk[2];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float e0 = __builtin_amdgcn_exp2f(s[2 * i]), e1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
                        if constexpr (TRACK)
                            pk[i >> 2][i & 3] = pack2t<F16>(e0, e1);
                        else        // truncation: {e1[31:16], e0[31:16]}
                            pk[i >> 2][i & 3] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, e1),
                                                                      __builtin_bit_cast(unsigned, e0), 0x07060302u);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 pb = __builtin_bit_cast(bf16x8, pk[s2]);
                        oacc[qb][0] = mma32<F16>(vf[s2][0], pb, oacc[qb][0]);
                        oacc[qb][1] = mma32<F16>(vf[s2][1], pb, oacc[qb][1]);
                    }
                }
                A40_STAMP();
            }
        };
        const unsigned st0 = lds_addr(stage_a);
        if constexpr (NST == 3) {
            for (int t = 0; t < ntiles; t += 3) {
                tile(t, st0, stage_c);
                if (t + 1 < ntiles) tile(t + 1, st0 + A40_STAGE, stage_a);
                if (t + 2 < ntiles) tile(t + 2, st0 + 2 * A40_STAGE, stage_b);
            }
        } else {
            for (int t = 0; t < ntiles; t += 2) {
                tile(t, st0, stage_b);
                if (t + 1 < ntiles) tile(t + 1, st0 + A40_STAGE, stage_a);
            }
        }
        __syncthreads();                              // every wave has left the stages (re-run, or the O staging below)
    };

    // the fast path overflowed if the denominator or any numerator is not finite (inf * 0 = NaN: one FMA per register)
    auto accumulators_finite = [&]() {
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float l = lower_half_value40(oacc[qb][1][4]);
            float z = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) z = fmaf(oacc[qb][0][r], 0.f, z);
#pragma unroll
            for (int r = 0; r < 5; ++r) z = fmaf(oacc[qb][1][r], 0.f, z);      // rows 32..39 (both halves) and the denominator
            ok &= (l > 0.f && l < 0x1p100f && z == 0.f) || (q0w + 32 * qb + lq >= p.Sq);      // 1 / l must not underflow
        }
        return ok;
    };

    if constexpr (TRACK_ONLY) {
        run(BoolTag<true>{}, true);
    } else {
        run(BoolTag<false>{}, true);
        // a score more than 2^127 above the reference taken from the first 32 keys: exp2 overflowed.  The workgroup shares
        // the K / V stream, so it re-runs as a whole, with the tracked reference.
        if (__syncthreads_or(!accumulators_finite())) run(BoolTag<true>{}, false);
    }

    A40_STAMP();
    // ---- finalize: O[q][d] = O^T[d][q] / l[q], staged through LDS (the stages are free) and stored as whole 80-byte rows
    unsigned char* ost = stage_a + wave * (QW * A40_ROWB);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l = lower_half_value40(oacc[qb][1][4]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qi = q0w + 32 * qb + lq;
        // training: log2-domain log-sum-exp of the scaled scores, read back by seer_attn_bwd
        if (p.lse && lh == 0 && qi < p.Sq) p.lse[(int64_t)y * p.Sq + qi] = m_run[qb] + __builtin_amdgcn_logf(l);
        unsigned char* orow = ost + (32 * qb + lq) * A40_ROWB + 8 * lh;      // lane holds d = 32 dt + 8 g + 4 h + (0..3)
#pragma unroll
        for (int g = 0; g < 5; ++g) {                // g = 4: d tile 1, rows 32..39
            const f32x16& o = oacc[qb][g >> 2];
            const int r0 = 4 * (g & 3);
            u32x2 w;
            w[0] = pack2t<F16>(o[r0] * inv, o[r0 + 1] * inv);
            w[1] = pack2t<F16>(o[r0 + 2] * inv, o[r0 + 3] * inv);
            *reinterpret_cast<u32x2*>(orow + 16 * g) = w;
        }
    }
    // 16-byte chunks of whole output rows: chunk i = row * 5 + c (the wave reads back only what it wrote)
#pragma unroll
    for (int it = 0; it < (QW * 5 + 63) / 64; ++it) {
        const int i = lane + 64 * it;
        if ((QW * 5) % 64 != 0 && i >= QW * 5) break;
        const int row = i / 5, c = i - row * 5;
        const int qi = q0w + row;
        if (qi < p.Sq) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ost + i * 16);
            // (plain store: the write-through form of seer_common.h measured +2-3 us on this kernel, profiles/r02_wt_stores.log)
            *reinterpret_cast<u32x4*>(Og + (int64_t)tok(qi) * p.o_ss + c * 8) = v;
        }
    }
    A40_STAMP();
}

}  // namespace

// called by seer_attn_fwd (attention.hip) for head_dim 40.  Two shapes: 32 queries per wave, 128 per workgroup, three workgroups
// per CU; and 64 queries per wave, 256 per workgroup, two workgroups per CU (under the three-workgroup register bound that form
// spilled at 168 registers and measured 10-20 % slower, profiles/r02_attn40_variants.log; at its own 240 it does not).
// variant 5, or lse != NULL, runs the tracked form directly.
int seer_attn40_launch(const seer_attn_desc& d, int ws_log2, hipStream_t st) {
    int nbatch = d.batch;
    if (ws_log2 >= 0) nbatch *= (d.H >> ws_log2) * (d.W >> ws_log2);
    // the LDS-DMA plan addresses K / V rows by 32-bit byte offsets from the (batch, head) base
    {
        const int64_t ntok = ws_log2 >= 0 ? (int64_t)d.F * d.H * d.W : (int64_t)d.Sk;
        if (ntok * (d.k_ss > d.v_ss ? d.k_ss : d.v_ss) * 2 >= (int64_t)1 << 32) return SEER_ENOSYS;
    }
    const bool track = d.variant == 5 || d.lse != nullptr;
    const bool plain = ws_log2 < 0 && !d.causal && d.causal_offset == 0;
    if (d.flags & SEER_ATTN_F16) {          // IEEE-half operands: the tracked form (32 queries per wave), nothing else
        const int nqbh = (d.Sq + 127) / 128;
        dim3 gridh((unsigned)(nqbh * nbatch * d.heads));
        if (plain) hipLaunchKernelGGL((seer_attn40_kernel<1, true, true, 2, true>), gridh, dim3(256), 0, st, d, ws_log2, nqbh);
        else hipLaunchKernelGGL((seer_attn40_kernel<1, true, false, 2, true>), gridh, dim3(256), 0, st, d, ws_log2, nqbh);
        SEER_LAUNCH_CHECK();
        return SEER_OK;
    }
    // 64 queries per wave: K / V fragments, LDS-DMA issue and barriers are shared by two query blocks, and the two blocks are
    // software-pipelined (both Q K^T chains first, then block 0's exponentials under block 1's chain, block 0's P V under block
    // 1's exponentials); 240 registers, two waves per SIMD.  [192, 1024, 40]: 53.2 vs 56.3 us, [192, 4096, 40]: 582 vs 643 us
    // (profiles/r03_lab_attn_qb2.log).  Taken by non-causal launches that fill at least one round of the 512 resident workgroups;
    // variant 2 forces it, variant 3 the 32-query form (A/B runs, tests)
    const int nqb2 = (d.Sq + 255) / 256;
    // the 64-query form on a three-stage ring: whole tiles only (RING_FAST).  [192, 1024, 40] 54.1 -> 50.6 us, [192, 4096, 40]
    // 583 -> 566 us back to back (profiles/r05_attn40_ring.log); variant 7 forces it, variant 2 keeps the two-stage form (A/B runs)
    const bool ring_ok = plain && d.Sq % 256 == 0 && d.Sk % A40_KT == 0;
    if (d.variant == 7 && (track || !ring_ok)) return SEER_EINVAL;
    if (!track && ring_ok && (d.variant == 7 || (d.variant == 0 && (long)nqb2 * nbatch * d.heads >= 512))) {
        dim3 grid7((unsigned)(nqb2 * nbatch * d.heads));
        hipLaunchKernelGGL((seer_attn40_kernel<2, false, true, 3>), grid7, dim3(256), 0, st, d, ws_log2, nqb2);
        SEER_LAUNCH_CHECK();
        return SEER_OK;
    }
    // (plain launches only: under a causal mask the 64-query wave does the work of its later query block for the earlier
    //  one too -- temporal window block 47 vs 36 us)
    if (!track && (d.variant == 2 || (d.variant == 0 && plain && (long)nqb2 * nbatch * d.heads >= 512))) {
        dim3 grid2((unsigned)(nqb2 * nbatch * d.heads));
        if (plain) hipLaunchKernelGGL((seer_attn40_kernel<2, false, true>), grid2, dim3(256), 0, st, d, ws_log2, nqb2);
        else hipLaunchKernelGGL((seer_attn40_kernel<2, false, false>), grid2, dim3(256), 0, st, d, ws_log2, nqb2);
        SEER_LAUNCH_CHECK();
        return SEER_OK;
    }
    const int nqb = (d.Sq + 127) / 128;
    dim3 grid((unsigned)(nqb * nbatch * d.heads));
    if (track && plain) hipLaunchKernelGGL((seer_attn40_kernel<1, true, true>), grid, dim3(256), 0, st, d, ws_log2, nqb);
    else if (track) hipLaunchKernelGGL((seer_attn40_kernel<1, true, false>), grid, dim3(256), 0, st, d, ws_log2, nqb);
    else if (plain) hipLaunchKernelGGL((seer_attn40_kernel<1, false, true>), grid, dim3(256), 0, st, d, ws_log2, nqb);
    else hipLaunchKernelGGL((seer_attn40_kernel<1, false, false>), grid, dim3(256), 0, st, d, ws_log2, nqb);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
