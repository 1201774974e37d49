// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )        (see include/seer_hip.h, seer_gemm_bf16)
//
// Structure (v1): 256 threads = 4 waves (2 x 2), block tile BM x BN x 64, v_mfma_f32_16x16x32_bf16,
// LDS tiles of 128-byte rows with a 16-byte-chunk XOR swizzle (chunk ^= row & 7: conflict-free
// ds_read_b128 fragment reads), register-staged global->LDS double buffering with ONE barrier per K tile
// (loads for tile t+1 are issued before the MFMAs of tile t and written to the other buffer after them).
// The MFMA is issued "swapped" (W fragment as the A operand, activation fragment as the B operand) so that
// every lane ends up holding 4 CONSECUTIVE output columns of one output row: the epilogue (bias, time-embedding
// row vector, residual, GEGLU, SiLU) works on 4-wide vectors and stores 8 bytes per lane.
//
// Implicit GEMM conv: k = (ky, kx, ci); Cin % 64 == 0 so a 64-wide K tile never straddles a filter tap; the
// tap -> pixel offset is wave-uniform scalar work, the per-row pixel decode is hoisted out of the K loop.
#include "seer_common.h"
#include <mutex>
#include <type_traits>

// Measurement build only (-DSEER_GEMM_STAMPS, scripts/lab_pp8stamps.cpp): every wave of the first 64 blocks records
// wall_clock64() at the phase boundaries of its tile.  256x256 tile: entry, prologue issued, prologue landed, K loop done, rows
// re-aligned, epilogue math done, tile staged, tile stored.  LDS-direct ring tiles: entry, set-up done, first K tile landed,
// K loop done, then epilogue math done, tile staged, tile stored (split-K: partial tile stored).
#ifdef SEER_GEMM_STAMPS
__device__ long long seer_pp8_stamps[64 * 8 * 16];
extern "C" long long* seer_lab_pp8_stamps() {
    long long* p = nullptr;
    (void)hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(seer_pp8_stamps));
    return p;
}
// entry / exit time of wave 0 of every block (first 8192 blocks of z = 0): the dispatch ramp and the tail of a launch
__device__ long long seer_block_span[8192 * 2];
extern "C" long long* seer_lab_block_span() {
    long long* p = nullptr;
    (void)hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(seer_block_span));
    return p;
}
#define PSPAN(which)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.z == 0 && blockIdx.x < 8192) seer_block_span[blockIdx.x * 2 + (which)] = wall_clock64(); \
    } while (0)
#define PSTAMP()                                                                                                       \
    do {                                                                                                               \
        if (blockIdx.x < 64 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 && nst_ < 16)                                \
            seer_pp8_stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + nst_] = wall_clock64();                       \
        ++nst_;                                                                                                        \
    } while (0)
#else
#define PSTAMP() do { } while (0)
#define PSPAN(which) do { } while (0)
#endif

bool seer_gemm_ws_eligible(const seer_gemm_desc& d);             // gemm_ws.hip: weight-stationary persistent kernel (short K)
bool seer_gemm_ws_profitable(const seer_gemm_desc& d);
int seer_gemm_ws_launch(const seer_gemm_desc& d, hipStream_t st);
bool seer_gemm_t320_eligible(const seer_gemm_desc& d);           // gemm_t320.hip: 256 x 320 tile, in-launch split-K
int64_t seer_gemm_t320_workspace_bytes(const seer_gemm_desc& d, int splits);
int64_t seer_gemm_t320_sync_bytes(const seer_gemm_desc& d, int splits);
int seer_gemm_t320_launch(const seer_gemm_desc& d, int splits, hipStream_t st);

namespace {

constexpr int BK = 64;


#ifndef SEER_GEMM_EARLY_REFILL
#define SEER_GEMM_EARLY_REFILL 1
#endif
// measurement-only builds (scripts/probe_gemm.sh; results are WRONG): bit 0 = no global->LDS refills inside the K loop,
// bit 1 = no LDS fragment reads, bit 2 = no MFMAs.  The shipped library is built with 0.
#ifndef SEER_GEMM_PROBE
#define SEER_GEMM_PROBE 0
#endif

// 16 zero bytes: the source of out-of-image taps for the LDS-direct (global_load_lds) conv gather
__device__ __attribute__((aligned(16))) unsigned int seer_zero_page[4] = {0u, 0u, 0u, 0u};

// NS == 0: register-staged double buffer (global_load -> VGPR -> ds_write), one barrier per K tile.
// NS >= 2: NS-stage ring filled by global_load_lds (16 B per lane straight into LDS, no VGPR / ds_write), NS-1 tiles in
//          flight across raw s_barriers behind counted s_waitcnt vmcnt(N).  Same LDS image either way.
// WM x WN: the wave grid.  WN = 2: WM = 2 -> 256 threads, 4 -> 512 threads (256-row tiles, same 64x64 wave tile).
// WM = 2, WN = 4 with a 256 x 256 tile is the 8-phase ping-pong kernel (PP8 below): 128 x 64 wave tiles, its own main loop.
// whether a tile instantiation can produce column sums (seer_gemm_desc::colsum): the staged C tile plus the row-segment
// partials have to fit in the K-loop LDS, one thread per output column adds them
template <int BM, int BN, int NS, int WM, int WN>
constexpr bool tile_colsum_ok() {
    constexpr int NT = 64 * WM * WN, STAGE = (BM + BN) * BK;
    constexpr bool PP8 = WM == 2 && WN == 4 && BM == 256 && BN == 256;
    constexpr bool CSWZ = BM * (BN * 2 + 16) > 2 * STAGE * (int)sizeof(bf16);
    constexpr int CPITCH = BN * 2 + (CSWZ ? 0 : 16);
    return !PP8 && BN <= NT && BM * CPITCH + (NT / (BN / 2)) * BN * 8 <= (NS == 0 ? 2 : NS) * STAGE * (int)sizeof(bf16);
}

// whether a tile instantiation can take part in a folded LayerNorm (seer_gemm_desc::rowstat / ln_rowstat): the consumer parks
// (mean * rstd, rstd) of its BM rows behind the staged C tile and the column-sum scratch
template <int BM, int BN, int NS, int WM, int WN>
constexpr bool tile_ln_ok() {
    constexpr int NT = 64 * WM * WN, STAGE = (BM + BN) * BK;
    constexpr bool PP8 = WM == 2 && WN == 4 && BM == 256 && BN == 256;
    constexpr bool CSWZ = BM * (BN * 2 + 16) > 2 * STAGE * (int)sizeof(bf16);
    constexpr int CPITCH = BN * 2 + (CSWZ ? 0 : 16);
    return !PP8 && BM <= NT && BM * CPITCH + 4096 + BM * 8 <= (NS == 0 ? 2 : NS) * STAGE * (int)sizeof(bf16);
}

template <int BM, int BN, bool CONV, bool GEGLU, bool SPLIT, int NS, int WM = 2, int WN = 2, bool F16 = false>
__global__ void __launch_bounds__(64 * WM * WN) seer_gemm_kernel(const seer_gemm_desc p) {
    constexpr int NT = 64 * WM * WN;               // threads per block
    constexpr int RPP = NT / 8;                    // tile rows staged per pass (8 lanes x 16 B cover one 128-B row)
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr bool PP8 = WM == 2 && WN == 4 && BM == 256 && BN == 256;
    static_assert(WN == 2 || PP8, "wave grids other than WM x 2 exist only as the 256 x 256 ping-pong kernel");
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_CH = BM / RPP, B_CH = BN / RPP;  // 16-byte chunks per thread per K tile
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    static_assert(!GEGLU || (TN % 2 == 0), "GEGLU needs value/gate n-tile pairs inside one wave");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE = (BM + BN) * BK;        // elements per stage: [A tile BM x 64][B tile BN x 64]
    bf16* const smem_b = reinterpret_cast<bf16*>(smem);

    [[maybe_unused]] int nst_ = 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations stay in SGPRs
    const int wm = wave / WN, wn = wave % WN;
    PSTAMP();
    PSPAN(0);

    // ---- block -> tile mapping: XCD-contiguous chunks (blocks b and b+8 share an XCD), grouped along M
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int wg;
    {
        const int L = blockIdx.x;
        const int xcd = L & 7;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
#ifndef SEER_GEMM_GM
#define SEER_GEMM_GM 4          // M tiles per block group (round 1 chose 8 from a back-to-back sweep, profiles/r01_gemm_group_sweep.log; inside the
                               // step -- cold operands -- 4 is the best of 1, 2, 3, 4, 6, 8, 16, 32: 9.59 against 9.64 ms and 9.77 against 9.81 on two
                               // boxes, profiles/r06_gemm_group_in_step.log)
#endif
    constexpr int GM = SEER_GEMM_GM;
    const int group = wg / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (wg % (GM * tiles_n)) % gsz;
    const int tn = (wg % (GM * tiles_n)) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = SPLIT ? 0 : blockIdx.z;          // SPLIT: blockIdx.z is the K slice, not a batch index
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A) + (int64_t)z * p.strideA;
    const bf16* __restrict__ A2 = reinterpret_cast<const bf16*>(p.A2);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W) + (int64_t)z * p.strideW;

    // ---- per-thread staging descriptors
    const int c8 = (tid & 7) * 8;     // element offset of this thread's chunk inside the K tile
    const int srow = tid >> 3;        // 0..RPP-1
    int64_t a_off[A_CH];              // plain: row offset into A ; conv: image base offset
    int64_t a_off2[A_CH];
    int a_oy[A_CH], a_ox[A_CH];
    // PP8 stages half tiles: chunk c = 2 * half + i is one 8-row group of "half" (the rows of quadrant-row `half` of both wave
    // rows: tile rows wm * 128 + half * 64 + [0, 64); 16 of them per wave), see the main loop
    auto a_tile_row = [&](int c) { return PP8 ? (wave >> 2) * 128 + (c >> 1) * 64 + 16 * (wave & 3) + 8 * (c & 1) + (lane >> 3) : srow + RPP * c; };
    auto b_tile_row = [&](int c) {
        const int hn = 16 * wave + 8 * (c & 1) + (lane >> 3);
        return PP8 ? (hn >> 5) * 64 + (c >> 1) * 32 + (hn & 31) : srow + RPP * c;
    };
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int gm = m0 + a_tile_row(i);
        gm = gm < p.M ? gm : p.M - 1;
        if constexpr (CONV) {
            // upsample == 2: one of the four 2x2 phase convs of "nearest 2x + conv3x3" (seer_hip.h): rows index the SOURCE
            // grid, output pixel (2 y + a, 2 x + b) of phase (a, b) = blockIdx.z reads source rows y + a - 1 + {0, 1}
            const bool phase_mode = p.upsample == 2;
            const int gw = phase_mode ? p.Win : p.Wout;
            const int hw = phase_mode ? p.Hin * p.Win : p.Hout * p.Wout;
            const int img = gm / hw;
            const int rem = gm - img * hw;
            const int oy = rem / gw;
            const int pad0 = p.pad_after_only ? 0 : 1;     // rows / columns of padding before the image
            a_oy[i] = phase_mode ? oy + ((int)blockIdx.z >> 1) - 1 : oy * p.stride - pad0;
            a_ox[i] = phase_mode ? (rem - oy * gw) + ((int)blockIdx.z & 1) - 1 : (rem - oy * gw) * p.stride - pad0;
            a_off[i] = (int64_t)img * p.Hin * p.Win * p.Cin;
            a_off2[i] = 0;
        } else {
            a_off[i] = (int64_t)gm * p.lda;
            a_off2[i] = (int64_t)gm * p.lda2;
            // the same row relative to the tile's first row, in ELEMENTS (32 bits): the LDS-direct fetches address
            // "wave-uniform 64-bit base + one VGPR" (global_load_lds v, s[..]) instead of forming a 64-bit address per piece
            a_oy[i] = (gm - m0) * p.lda;
            a_ox[i] = (gm - m0) * p.lda2;
        }
    }
    // LDS-direct conv gather without per-piece address arithmetic (not the nearest-2x read-through form, whose source pixel is not
    // affine in the tap): a_pix = byte offset of tap (0, 0) of the piece row's pixel, this lane's chunk included; bit 9 i + tap of
    // a_okb[i / 3]: tap inside the image.  A K tile then costs one scalar tap offset and, per piece, a select between
    // a_pix + tap offset and an offset past the end of the tensor, which the buffer bounds check turns into zeros -- no 64-bit
    // address, no zero page, no branch, no division (round 4: the loop is bound by instruction issue, profiles/r04_t320_stamps.log)
    [[maybe_unused]] int a_pix[A_CH];
    [[maybe_unused]] unsigned a_okb[(A_CH + 2) / 3] = {};
    [[maybe_unused]] unsigned a_bytes = 0;
    bool conv_fast = false;
    if constexpr (CONV && NS >= 2) {
        const int64_t hw_out = p.upsample == 2 ? (int64_t)p.Hin * p.Win : (int64_t)p.Hout * p.Wout;
        const int64_t in_bytes = (p.M / hw_out) * p.Hin * p.Win * p.Cin * 2;
        conv_fast = p.upsample != 1 && in_bytes < ((int64_t)1 << 31) && (int64_t)p.K * p.Cin < ((int64_t)1 << 32);
        a_bytes = (unsigned)in_bytes;
        if (conv_fast) {
            const int ksz = p.upsample == 2 ? 2 : 3;
            const int sch = ((tid & 7) ^ (srow & 7)) * 8;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int iy0 = a_oy[i], ix0 = a_ox[i];
                a_pix[i] = (int)(a_off[i] + ((int64_t)iy0 * p.Win + ix0) * p.Cin + sch) * 2;
                unsigned bits = 0u;
                for (int ky = 0; ky < ksz; ++ky)
                    for (int kx = 0; kx < ksz; ++kx)
                        if (iy0 + ky >= 0 && iy0 + ky < p.Hin && ix0 + kx >= 0 && ix0 + kx < p.Win) bits |= 1u << (ky * ksz + kx);
                a_okb[i / 3] |= bits << (9 * (i % 3));
            }
        }
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (int)a_bytes, 0x00020000);
    const unsigned cin_magic = CONV ? 0xFFFFFFFFu / (unsigned)(p.Cin > 0 ? p.Cin : 1) + 1u : 0u;     // tap = umulhi(kbase, magic)
    // plain operands of the LDS-direct ring as buffer resources over the tile's rows (see issue_tile)
    const __amdgpu_buffer_rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (CONV ? 0 : (int64_t)m0 * p.lda)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A2 && !CONV ? A2 + (int64_t)m0 * p.lda2 : A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(W + (int64_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);

    int64_t b_off[B_CH];
    int b_rel[B_CH];                  // (gn - n0) * K: the W row relative to the tile's first, in elements
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int gn = n0 + b_tile_row(i);
        gn = gn < p.N ? gn : p.N - 1;
        b_off[i] = (int64_t)gn * p.K;
        b_rel[i] = (gn - n0) * p.K;
    }

    u32x4 areg[A_CH], breg[B_CH];

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        if constexpr (CONV) {
            const int tap = kbase / p.Cin;
            const int ci0 = kbase - tap * p.Cin;
            const int ksz = p.upsample == 2 ? 2 : 3;
            const int ky = tap / ksz, kx = tap - ky * ksz;
            const int Hs = p.upsample == 1 ? p.Hin * 2 : p.Hin;
            const int Ws = p.upsample == 1 ? p.Win * 2 : p.Win;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int iy = a_oy[i] + ky, ix = a_ox[i] + kx;
                const bool ok = (iy >= 0) & (iy < Hs) & (ix >= 0) & (ix < Ws);
                const int sy = p.upsample == 1 ? (iy >> 1) : iy;
                const int sx = p.upsample == 1 ? (ix >> 1) : ix;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) {
                    const bf16* src = A + a_off[i] + ((int64_t)sy * p.Win + sx) * p.Cin + ci0 + c8;
                    v = *reinterpret_cast<const u32x4*>(src);
                }
                areg[i] = v;
            }
        } else {
            const bool second = kbase >= p.K1;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const bf16* src = second ? (A2 + a_off2[i] + (kbase - p.K1) + c8) : (A + a_off[i] + kbase + c8);
                areg[i] = *reinterpret_cast<const u32x4*>(src);
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            breg[i] = *reinterpret_cast<const u32x4*>(W + b_off[i] + kbase + c8);
        }
    };
    auto store_tile = [&](int buf) {
        bf16* as = smem_b + buf * STAGE;
        bf16* bs = as + BM * BK;
        const int sw = ((tid & 7) ^ (srow & 7)) * 8;   // (row & 7) == (srow & 7) because rows step by RPP (a multiple of 8)
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *reinterpret_cast<u32x4*>(as + (srow + RPP * i) * BK + sw) = areg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4*>(bs + (srow + RPP * i) * BK + sw) = breg[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- residual tile prefetch: the epilogue's bf16 residual (8 B per accumulator quad) is requested before the K loop so
    // its HBM/L2 latency hides under the main loop instead of adding a dependent round trip to every block's tail
    const bf16* R = reinterpret_cast<const bf16*>(p.residual);
    constexpr bool RES_PRE = !GEGLU && !SPLIT && TM * TN <= 16;     // the 256x256 tile has no registers to spare for it
    u32x2 rpre[RES_PRE ? TM : 1][RES_PRE ? TN : 1];
    if constexpr (RES_PRE) {
        if (R) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + (lane >> 4) * 4;
                    rpre[i][j] = (m < p.M && n < p.N) ? *reinterpret_cast<const u32x2*>(R + (int64_t)m * p.ldr + n)
                                                        : u32x2{0u, 0u};
                }
            }
        }
    }

    // folded LayerNorm (seer_gemm_desc::ln_rowstat): thread t < BM REQUESTS the accumulated (sum, sum of squares) of tile row t here
    // and keeps the raw pair through the K loop; the epilogue converts it and hands (mean * rstd, rstd) to the lanes that own the
    // row through LDS.  (Converting here put a wait for the load -- an L2 round trip -- in front of the first LDS-DMA of every
    // tile: +3 us on the GEGLU projections with their 7.5 tiles per CU, profiles/r04_ln_fold_overheads.md.)
    constexpr bool LN_OK = !SPLIT && tile_ln_ok<BM, BN, NS, WM, WN>();
    typedef __attribute__((ext_vector_type(2))) long long i64x2;
    i64x2 ln_raw = {0, 0};
    if constexpr (LN_OK) {
        if (p.ln_rowstat && tid < BM) {
            const int m = min(m0 + tid, p.M - 1);
            ln_raw = *reinterpret_cast<const i64x2*>(reinterpret_cast<const long long*>(p.ln_rowstat) + (int64_t)m * 2);
        }
    }
    // ... and the lane's wsum quads, requested before the K loop like the bias (a dependent L2 round trip at the head of the
    // epilogue otherwise)
    f32x4 spre[LN_OK ? TN : 1];
    if constexpr (LN_OK) {
        if (p.ln_rowstat) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + (lane >> 4) * 4;
                spre[j] = n < p.N ? *reinterpret_cast<const f32x4*>(p.ln_wsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // bias prefetch, same reason (a dependent L2 round trip at the head of every block's epilogue otherwise); the 160-wide
    // tiles have no registers to spare for it
    constexpr bool BIAS_PRE = !SPLIT && TN <= 4 && !PP8;
    f32x4 bpre[TN];
    if constexpr (BIAS_PRE) {
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + (lane >> 4) * 4;
                bpre[j] = n < p.N ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // per-batch row vector (time embedding): one row serves the whole tile whenever the tile does not straddle two batch
    // elements -> prefetch it as well
    constexpr bool RV_PRE = !SPLIT && !GEGLU && TN <= 4 && !PP8;
    f32x4 rvpre[TN];
    bool rv_pre_ok = false;
    if constexpr (RV_PRE) {
        if (p.rowvec) {
            const int rb0 = m0 / p.rows_per_batch;
            rv_pre_ok = ((min(m0 + BM, p.M) - 1) / p.rows_per_batch) == rb0;
            if (rv_pre_ok) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + (lane >> 4) * 4;
                    rvpre[j] = n < p.N ? *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)rb0 * p.rowvec_ld + n)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }

    const int nk_all = p.K / BK;
    const int kt0 = SPLIT ? (int)((int64_t)nk_all * blockIdx.z / p.splits) : 0;
    const int nk = SPLIT ? (int)((int64_t)nk_all * (blockIdx.z + 1) / p.splits) : nk_all;
    const int frow = lane & 15;        // operand row inside a 16-row fragment
    const int fq = lane >> 4;          // 16-byte chunk inside a 32-wide k-step

    // all fragment reads of the K tile (both 32-wide k-steps) are issued before the first MFMA: one exposed LDS latency
    // per tile instead of one per k-step (with 1-2 waves per SIMD nothing else hides it); the compiler's counted
    // lgkmcnt waits let the first MFMAs start as soon as the k-step-0 fragments are back.
    auto compute_tile = [&](int buf) {
        const bf16* as = smem_b + buf * STAGE + (wm * WTM) * BK;
        const bf16* bs = smem_b + buf * STAGE + BM * BK + (wn * WTN) * BK;
        bf16x8 af[2][TM], wf[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bs + (j * 16 + frow) * BK + sw);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(as + (i * 16 + frow) * BK + sw);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mma16<F16>(wf[ks][j], af[ks][i], acc[i][j]);
    };

    if constexpr (PP8) {
        // ---- 256 x 256 x 64 tile, 8 waves as 2 (M) x 4 (N), 128 x 64 wave tiles: the 8-phase ping-pong loop of the CDNA guide's
        // "256^2 8-phase template" (cdna_hip_programming.md), built from its description.  Why: a 128 x 128 tile needs 1 byte of
        // LDS fill per 64 FLOP and the fill path gives a CU ~50-70 GB/s (profiles/r02_lab_fill.log) -> at most ~1.1 PF; 256 x 256
        // halves the bytes per FLOP, but a block-wide "all read, all multiply" loop leaves LDS and matrix pipe idle in turn
        // (0.5-0.8 PF, profiles/r02_lab_gemm_t256.log).  Here:
        //   * a K tile is four 16 KB HALF tiles  h0 = A rows of quadrant-row 0, h1 = W columns of quadrant-column 1,
        //     h2 = A rows of quadrant-row 1, h3 = W columns of quadrant-column 0  (quadrant-row r of the block = rows
        //     wm * 128 + r * 64 + [0, 64) of both wave rows; quadrant-column c = columns wn * 64 + c * 32 + [0, 32));
        //   * a K tile is four PHASES, one 64 x 32 quadrant of the wave tile each (16 MFMAs over K = 64):
        //       q0 = (r0, c0) reads h0, h3 | q1 = (r0, c1) reads h1 | q2 = (r1, c1) reads h2 | q3 = (r1, c0) reads h3
        //     so every half tile has its LAST read one phase before the phase that restages it (tile u + 2 into the same buffer):
        //       q1 stages h0, q2 stages h1, q3 stages h2, q0 of the next tile stages h3  -> 3-4 half tiles always in flight;
        //   * every phase is  [fragment reads + 2 LDS-DMA + lgkmcnt(0)]  s_barrier  [16 MFMAs]  s_barrier,  and wave row 1 runs
        //     ONE BARRIER behind wave row 0: while one wave of a SIMD multiplies, the other reads and stages;
        //   * one counted wait per K tile (q3): vmcnt(6) leaves the three youngest half tiles in flight, never 0 in the loop.
        // Ordering (guide, "Read a staged buffer one phase AFTER the wait that retires it"): the q3 wait sits before q3's first
        // barrier and the first read of the tile it retires is in q0; fragment reads are retired (lgkmcnt(0)) BEFORE the phase's
        // first barrier, which is what makes restaging one phase later safe with the two wave rows a barrier apart.
        static_assert(NS == 2 && A_CH == 4 && B_CH == 4, "two 64 KB buffers of four half tiles");
        constexpr int HALF = 128 * BK;                               // elements per half tile
        const int T = nk - kt0;
        const int schunk = ((tid & 7) ^ ((tid >> 3) & 7)) * 8;
        // LDS: [buffer][h0 = A r0 | h2 = A r1 | h3 = W c0 | h1 = W c1], 16 KB each
        auto half_base = [&](int u, int h) { return smem_b + ((u & 1) * 4 + (h == 0 ? 0 : h == 2 ? 1 : h == 3 ? 2 : 3)) * HALF; };
        auto stage_half = [&](int u, int h) {                         // this wave's 2 x 1 KB of half tile h of K tile u
            if (u >= T) return;
            const int kbase = (kt0 + u) * BK;
            bf16* dst = half_base(u, h) + (16 * wave) * BK;
            if (h == 0 || h == 2) {
                const int c0 = h == 0 ? 0 : 2;
                if constexpr (CONV) {
                    const int tap = kbase / p.Cin;
                    const int ci0 = kbase - tap * p.Cin;
                    const int ksz = p.upsample == 2 ? 2 : 3;
                    const int ky = tap / ksz, kx = tap - ky * ksz;
                    const int Hs = p.upsample == 1 ? p.Hin * 2 : p.Hin;
                    const int Ws = p.upsample == 1 ? p.Win * 2 : p.Win;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int iy = a_oy[c0 + i] + ky, ix = a_ox[c0 + i] + kx;
                        const bool ok = (iy >= 0) & (iy < Hs) & (ix >= 0) & (ix < Ws);
                        const int sy = p.upsample == 1 ? (iy >> 1) : iy;
                        const int sx = p.upsample == 1 ? (ix >> 1) : ix;
                        const bf16* src = ok ? (A + a_off[c0 + i] + ((int64_t)sy * p.Win + sx) * p.Cin + ci0 + schunk)
                                             : reinterpret_cast<const bf16*>(seer_zero_page);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(dst + 8 * i * BK), 16, 0, 0);
                    }
                } else {
                    const bool second = kbase >= p.K1;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const bf16* src = second ? (A2 + a_off2[c0 + i] + (kbase - p.K1) + schunk) : (A + a_off[c0 + i] + kbase + schunk);
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)(dst + 8 * i * BK), 16, 0, 0);
                    }
                }
            } else {
                const int c0 = h == 3 ? 0 : 2;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16* src = W + b_off[c0 + i] + kbase + schunk;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + 8 * i * BK), 16, 0, 0);
                }
            }
        };
        bf16x8 fa[2][4], fb[2][2];
        auto read_a = [&](int u, int r) {                             // 64 rows of this wave x K 64: 8 ds_read_b128
            const bf16* as = half_base(u, r == 0 ? 0 : 2) + (wm * 64) * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(as + (i * 16 + frow) * BK + sw);
            }
        };
        auto read_b = [&](int u, int c) {                             // 32 columns of this wave x K 64: 4 ds_read_b128
            const bf16* bs = half_base(u, c == 0 ? 3 : 1) + (wn * 32) * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(bs + (j * 16 + frow) * BK + sw);
            }
        };
        auto barrier = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // the MFMA half of a phase: quadrant (r, c) of the wave tile, between the phase's two barriers
#define SEER_PP8_MMA(R, C)                                                                                                   \
        do {                                                                                                                 \
            barrier();                                                                                                       \
            __builtin_amdgcn_s_setprio(1);                                                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                 \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
                        acc[4 * (R) + i][2 * (C) + j] =                                                                      \
                            mma16<F16>(fb[ks][j], fa[ks][i], acc[4 * (R) + i][2 * (C) + j]);                                                \
            __builtin_amdgcn_s_setprio(0);                                                                                   \
            barrier();                                                                                                       \
        } while (0)

        PSTAMP();
        // prologue: K tile 0 whole, K tile 1 up to h2 (its h3 goes out in q0 of tile 0)
        stage_half(0, 0); stage_half(0, 1); stage_half(0, 2); stage_half(0, 3);
        stage_half(1, 0); stage_half(1, 1); stage_half(1, 2);
        if (T > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        PSTAMP();
        if (wm == 1) barrier();                                       // wave row 1 runs one barrier behind from here on
        for (int u = 0; u < T; ++u) {
            // q0
            read_b(u, 0);
            read_a(u, 0);
            stage_half(u + 1, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEER_PP8_MMA(0, 0);
            // q1
            read_b(u, 1);
            stage_half(u + 2, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEER_PP8_MMA(0, 1);
            // q2
            read_a(u, 1);
            stage_half(u + 2, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEER_PP8_MMA(1, 1);
            // q3: the wait that retires K tile u + 1 (all but the halves h0..h2 of tile u + 2 staged in q1..q3)
            read_b(u, 0);
            stage_half(u + 2, 2);
            if (u + 2 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEER_PP8_MMA(1, 0);
        }
#undef SEER_PP8_MMA
        PSTAMP();
        if (wm == 0) barrier();                                       // re-align the two wave rows
        PSTAMP();
    } else if constexpr (NS == 0) {
        load_tile(kt0);
        store_tile(0);
        __syncthreads();
        for (int kt = kt0; kt < nk; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < nk) load_tile(kt + 1);
            compute_tile(buf);
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
    } else {
        // ---- LDS-direct ring.  One wave instruction writes 1 KiB = 8 rows x 128 B, lane l -> (row l>>3, slot l&7); the
        // XOR swizzle goes on the SOURCE chunk (slot ^ row&7), the LDS image stays lane-linear (cdna guide, rule 21).
        constexpr int LPT = A_CH + B_CH;           // global_load_lds per wave per K tile
        const int schunk = ((tid & 7) ^ (srow & 7)) * 8;
        // plain operands as buffer loads: wave-uniform resource at the tile's first row, per-lane byte offset that is constant
        // over K (hoisted here), scalar offset that advances with K -- per piece one M0 write and one buffer_load ... lds.
        // (The offset arguments are cast to int explicitly: without the casts hipcc (ROCm 7.2) silently drops the HOST stub of every
        //  instantiation that reaches this call -- no diagnostic, undefined __device_stub__ symbols at load time.)
        unsigned av1[A_CH], av2[A_CH], bv[B_CH];
        if constexpr (!CONV) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                av1[i] = (unsigned)(a_oy[i] + schunk) * 2u;
                av2[i] = (unsigned)(a_ox[i] + schunk) * 2u;
            }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) bv[i] = (unsigned)(b_rel[i] + schunk) * 2u;

        auto issue_tile = [&](int kt, int stage) {
            const int kbase = kt * BK;
            bf16* as = smem_b + stage * STAGE + (8 * wave) * BK;      // this wave's 8-row group (wave-uniform)
            bf16* bs = smem_b + stage * STAGE + BM * BK + (8 * wave) * BK;
            if constexpr (CONV) {
                if (conv_fast) {
                    const int tap = (int)__umulhi((unsigned)kbase, cin_magic);
                    const int ci0 = kbase - tap * p.Cin;
                    const int ksz = p.upsample == 2 ? 2 : 3;
                    const int ky = tap / ksz, kx = tap - ky * ksz;
                    const int tap_b = (ky * p.Win + kx) * p.Cin * 2;
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) {
                        const bool ok = ((a_okb[i / 3] >> (9 * (i % 3) + tap)) & 1u) != 0u;
                        const unsigned voff = ok ? (unsigned)(a_pix[i] + tap_b) : a_bytes;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(as + RPP * i * BK), 16, voff,
                                                                 ci0 * 2, 0, 0);
                    }
                } else {
                const int tap = kbase / p.Cin;
                const int ci0 = kbase - tap * p.Cin;
                const int ksz = p.upsample == 2 ? 2 : 3;
                const int ky = tap / ksz, kx = tap - ky * ksz;
                const int Hs = p.upsample == 1 ? p.Hin * 2 : p.Hin;
                const int Ws = p.upsample == 1 ? p.Win * 2 : p.Win;
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    const int iy = a_oy[i] + ky, ix = a_ox[i] + kx;
                    const bool ok = (iy >= 0) & (iy < Hs) & (ix >= 0) & (ix < Ws);
                    const int sy = p.upsample == 1 ? (iy >> 1) : iy;
                    const int sx = p.upsample == 1 ? (ix >> 1) : ix;
                    const bf16* src = ok ? (A + a_off[i] + ((int64_t)sy * p.Win + sx) * p.Cin + ci0 + schunk)
                                         : reinterpret_cast<const bf16*>(seer_zero_page);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(as + RPP * i * BK), 16, 0, 0);
                }
                }
            } else {
                if (kbase >= p.K1) {               // (wave-uniform: the second source of a skip concat)
#pragma unroll
                    for (int i = 0; i < A_CH; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a2, (__attribute__((address_space(3))) void*)(as + RPP * i * BK), 16, (int)av2[i],
                                                                 (int)((kbase - p.K1) * 2), 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < A_CH; ++i)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a1, (__attribute__((address_space(3))) void*)(as + RPP * i * BK), 16, (int)av1[i],
                                                                 (int)(kbase * 2), 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(bs + RPP * i * BK), 16, (int)bv[i], (int)(kbase * 2), 0, 0);
        };
#if SEER_GEMM_EARLY_REFILL
        // Early refill: a stage is only occupied from "landed" to "fragments in registers".  Per K tile: wait for the tile,
        // barrier, read ALL its fragments (both k-steps) into VGPRs, barrier, refill the SAME stage with tile kt+NS, then run
        // the MFMAs from registers -> NS tiles (not NS-1) are in flight while the MFMAs of a tile run, at the same LDS bytes.
        bf16x8 af[2][TM], wf[2][TN];
        auto read_frags = [&](int buf) {
            const bf16* as = smem_b + buf * STAGE + (wm * WTM) * BK;
            const bf16* bs = smem_b + buf * STAGE + BM * BK + (wn * WTN) * BK;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bs + (j * 16 + frow) * BK + sw);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(as + (i * 16 + frow) * BK + sw);
            }
        };
#if SEER_GEMM_PROBE & 8
        // probe build: the same operand registers through 32x32x16 MFMAs (half the MFMA issues for the same pipe time; the numbers
        // are meaningless) -- prices what a 32x32 fragment layout could buy before anyone writes it
        constexpr bool P32 = (TM % 2 == 0) && (TN % 2 == 0) && !F16;
        f32x16 acc32[P32 ? TM / 2 : 1][P32 ? TN / 2 : 1] = {};
#endif
        auto mma_tile = [&]() {
#if SEER_GEMM_PROBE & 8
            if constexpr (P32) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < TM / 2; ++i)
#pragma unroll
                            for (int j = 0; j < TN / 2; ++j)
                                acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][2 * j + h], af[ks][2 * i + h], acc32[i][j], 0, 0, 0);
                return;
            }
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mma16<F16>(wf[ks][j], af[ks][i], acc[i][j]);
        };
        {
        PSTAMP();
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
            if (kt0 + s_ < nk) issue_tile(kt0 + s_, s_);
#if SEER_GEMM_PROBE & 2
        read_frags(0);
#endif
        int stage = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            const int pending = min(NS - 1, nk - 1 - kt);   // tiles issued after tile kt
            if (NS >= 4 && pending >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
            else if (NS >= 3 && pending == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
            else if (pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");            // every wave's part of tile kt has landed
#ifdef SEER_GEMM_STAMPS
            if (kt == kt0) PSTAMP();
#endif
#if !(SEER_GEMM_PROBE & 2)
            read_frags(stage);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");            // every wave holds its fragments: the stage is free
#if !(SEER_GEMM_PROBE & 1)
            if (kt + NS < nk) issue_tile(kt + NS, stage);
#endif
#if !(SEER_GEMM_PROBE & 4)
            mma_tile();
#else
            // probe build (scripts/probe_gemm.sh): keep the fragment reads alive without issuing MFMAs
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(wf[ks][j]));
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[ks][i]));
            }
#endif
            stage = (stage + 1 == NS) ? 0 : stage + 1;
        }
        PSTAMP();
#if SEER_GEMM_PROBE & 8
        if constexpr (P32) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = acc32[i / 2][j / 2][((i & 1) * 2 + (j & 1)) * 4 + e];
        }
#endif
        }
#else
#pragma unroll
        for (int s_ = 0; s_ < NS - 1; ++s_)
            if (kt0 + s_ < nk) issue_tile(kt0 + s_, s_);
        int stage = 0;
        for (int kt = kt0; kt < nk; ++kt) {
            // tiles issued after tile kt and still allowed in flight while we wait for tile kt
            const int pending = min(NS - 2, nk - 1 - kt);
            if (NS >= 5 && pending >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
            else if (NS >= 4 && pending == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
            else if (NS >= 3 && pending == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // every wave's part of tile kt has landed AND every wave is done reading stage (kt-1): safe to refill it
            asm volatile("s_barrier" ::: "memory");
            const int nxt = kt + NS - 1;
            if (nxt < nk) issue_tile(nxt, (stage + NS - 1) % NS);
            compute_tile(stage);
            stage = (stage + 1 == NS) ? 0 : stage + 1;
        }
#endif
        // (measured and rejected on MI355X, profiles/r01_gemm_microbench_v5.log: issuing the LDS-DMA pieces between the MFMAs
        //  with sched_group_barrier instead of in one burst is 5-20 % SLOWER at 2 blocks/CU: the other block already covers
        //  the burst, and spreading the pieces delays the next tile's arrival at the barrier.)
    }

#ifdef SEER_GEMM_EPI_NOP
    // bug-hunt build (scripts/exp_flake.py): a long pause between the last MFMA of the K loop and the first read of an accumulator
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
    // ---- split-K: raw fp32 partial tile to the workspace slice of this K range; the reduce kernel does the epilogue
    if constexpr (SPLIT) {
        float* ws = reinterpret_cast<float*>(p.workspace) + (int64_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + frow;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + fq * 4;
                if (n >= p.N) continue;
                // write-through: up to 31 MB of partials per launch would otherwise sit dirty in L2 until the boundary in front
                // of the reduce pass
                store16_out(ws + (int64_t)m * p.N + n, __builtin_bit_cast(u32x4, acc[i][j]));
            }
        }
        PSTAMP();
        PSPAN(1);
        return;
    }

    // ---- epilogue: lane holds rows n = 4*fq + r (r = 0..3) of the 16x16 D tile, column m = frow
    const bool out_f32 = (p.epilogue & SEER_EPI_OUT_F32) != 0;
    const bool do_silu = (p.epilogue & SEER_EPI_SILU) != 0;
    const bool trans = (p.epilogue & SEER_EPI_TRANS_OUT) != 0;
    bf16* Cb = reinterpret_cast<bf16*>(p.C) + (int64_t)z * p.strideC;
    // output row of GEMM row m: the identity, except for the phase convs, whose row (img, y, x) is pixel (2 y + a, 2 x + b)
    auto crow = [&](int m) -> int64_t {
        if constexpr (CONV) {
            if (p.upsample == 2) {
                const int hw = p.Hin * p.Win;
                const int img = m / hw, rem = m - img * hw;
                const int y = rem / p.Win, x = rem - y * p.Win;
                return ((int64_t)img * p.Hout + 2 * y + ((int)blockIdx.z >> 1)) * p.Wout + 2 * x + ((int)blockIdx.z & 1);
            }
        }
        return m;
    };
    float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)z * p.strideC;
    const bool do_rot = (p.epilogue & SEER_EPI_ROTARY) != 0;

    // bf16 row-major outputs leave through LDS: the MFMA layout gives a lane 8 B in each of 16 different rows (32-B row
    // segments per wave instruction, measured ~1.5 TB/s); staged through the (now idle) K-loop LDS the block stores 16 B per
    // lane along whole output rows instead.
    constexpr int BNO = GEGLU ? BN / 2 : BN;            // output columns of the block tile
    // staged row pitch in bytes: 16 B of padding spreads the 16 rows a wave instruction writes over the banks; when the padded
    // tile does not fit (256 x 256 bf16 = the whole 128 KB ring) the rows are dense and the 16-byte chunk index is XORed with
    // the row instead
    constexpr bool CSWZ = BM * (BNO * 2 + 16) > 2 * STAGE * (int)sizeof(bf16);
    constexpr int CPITCH = BNO * 2 + (CSWZ ? 0 : 16);
    static_assert(BM * CPITCH <= 2 * STAGE * (int)sizeof(bf16), "staged C tile must fit in the K-loop LDS");
    const bool staged = !out_f32 && !trans && (p.ldc % 8 == 0) && (p.N % (GEGLU ? 16 : 8) == 0) &&
                        ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
    if (staged) __syncthreads();                        // every wave is done with the K-loop stages
    // folded LayerNorm: acc <- rstd * acc - (mean * rstd) * wsum[n], first term of the epilogue (the host admits it only for
    // staged launches: the row statistics travel through the idle K-loop LDS)
    constexpr int LNROW_OFF = BM * CPITCH + 4096;
    bool do_ln = false;
    float lmr[TM], lr[TM];
    if constexpr (LN_OK) {
        do_ln = p.ln_rowstat != nullptr && staged;
        if (do_ln) {
            float* lnrow = reinterpret_cast<float*>(smem + LNROW_OFF);
            if (tid < BM) {
                // E[x^2] - mean^2 in DOUBLE on the exact integer totals: a row whose mean is 100 standard deviations away keeps
                // its variance (in fp32 the difference of the two ~1e4-times-larger terms loses it); BM threads, once per tile
                const double k = 1.0 / ((double)(1 << SEER_LN_FX_SHIFT) * (double)p.K);
                const double mean = (double)ln_raw[0] * k;
                double var = (double)ln_raw[1] * k - mean * mean;
                var = var > 0.0 ? var : 0.0;
                const float r = (float)(1.0 / __builtin_sqrt(var + (double)p.ln_eps));
                *reinterpret_cast<f32x2*>(lnrow + tid * 2) = f32x2{(float)mean * r, r};
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(lnrow + (wm * WTM + i * 16 + frow) * 2);
                lmr[i] = v[0];
                lr[i] = v[1];
            }
        }
    }

    // ---- fast epilogue: a full tile whose epilogue is bias / GEGLU / preloaded row vector / rotary / column scale / preloaded
    // residual into the staged C tile (every FF, projection, q|k|v and conv GEMM of the U-Net).  The general body below tests each descriptor flag for each of the
    // TM x TN accumulator quads and recomputes the staging address per quad: ~30 VALU instructions per quad, 3.2 us of VALU
    // issue per 256x256 tile and 1.6 us per 128x128 tile (profiles/r02_pp8_stamps.log).  Here every term is its own pass over
    // the accumulators behind ONE wave-uniform branch, in the general body's order of additions (bit-identical results), and
    // the staging address is one XOR per fragment column plus an immediate offset per fragment row.
    // row statistics of the output (seer_gemm_desc::rowstat): each lane adds the fp32 values of its quads per fragment row, the
    // four lanes of a row meet by two lane exchanges, one atomic pair per (row, wave column)
    constexpr bool RS_OK = !GEGLU && !SPLIT && tile_ln_ok<BM, BN, NS, WM, WN>();
    bool do_rs = false;
    float rsum[TM], rsq[TM];
    if constexpr (RS_OK) {
        do_rs = p.rowstat != nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) { rsum[i] = 0.f; rsq[i] = 0.f; }
    }
    const bool fast_epi = staged && m0 + BM <= p.M && n0 + BN <= p.N && !do_silu &&
                          (!GEGLU || !(do_rot || (p.epilogue & SEER_EPI_COLSCALE))) && (!p.rowvec || (RV_PRE && rv_pre_ok)) &&
                          (!R || RES_PRE);          // a residual that was not prefetched takes the general body
    if (fast_epi) {
        if constexpr (LN_OK) {
            if (do_ln) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 sv = spre[j];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = ln_term(acc[i][j][r], lr[i], sv[r], lmr[i]);
                }
            }
        }
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 bv;
                if constexpr (BIAS_PRE) bv = bpre[j];
                else bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WTN + j * 16 + fq * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] += bv;
            }
        }
        if constexpr (GEGLU) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const f32x4 g = acc[i][j + 1];
                    const f32x2 ge0 = gelu_erf_f2(f32x2{g[0], g[1]}), ge1 = gelu_erf_f2(f32x2{g[2], g[3]});
                    acc[i][j][0] *= ge0[0]; acc[i][j][1] *= ge0[1]; acc[i][j][2] *= ge1[0]; acc[i][j][3] *= ge1[1];
                }
                if constexpr (PP8) __builtin_amdgcn_sched_barrier(0);    // one fragment row of erf temporaries at a time
            }
        }
        if constexpr (RV_PRE) {
            if (p.rowvec) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] += rvpre[j];
                }
            }
        }
        if constexpr (!GEGLU) {
            if (do_rot && n0 + wn * WTN < p.rot_cols) {      // (wave-uniform: a wave whose columns are all v columns has nothing to rotate)
                // rotary on the q|k columns (attention.py:649-651), as in the general body, with the channel of each fragment
                // column and the position of each fragment row computed once per lane instead of once per quad.
                // (Measured and dropped, profiles/r03_rotary_loads.log: unconditional (cos, sin) loads issued together per fragment
                //  row, the unrotated quads kept by a select -- 33.5 vs 32.2 us on the 24 576-row projection: the cost of the
                //  epilogue is the 63 MB of table reads per launch, not their latency.)
                int tj[TN];                                 // float offset of the lane's (cos, sin) pairs in a table row, or -1
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + fq * 4;
                    const int ch = n % p.rot_head_dim;
                    tj[j] = (n < p.rot_cols && ch < p.rot_dim) ? (ch / 2) * 2 : -1;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wm * WTM + i * 16 + frow;
                    const int pos = m % p.rot_tokens_per_batch + p.rot_pos_offset;
                    const float* trow = p.rot_table + (int64_t)pos * (p.rot_dim / 2) * 2;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (tj[j] >= 0) {
                            const f32x4 cs = *reinterpret_cast<const f32x4*>(trow + tj[j]);
                            const f32x2 r01 = rot_pair(f32x2{acc[i][j][0], acc[i][j][1]}, cs[0], cs[1]);
                            const f32x2 r23 = rot_pair(f32x2{acc[i][j][2], acc[i][j][3]}, cs[2], cs[3]);
                            acc[i][j] = f32x4{r01[0], r01[1], r23[0], r23[1]};
                        }
                    }
                }
            }
            if (p.epilogue & SEER_EPI_COLSCALE) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // x * 1.0f is exact: the columns past col_scale_cols keep their bits
                    const float sc = (n0 + wn * WTN + j * 16 + fq * 4) < p.col_scale_cols ? p.col_scale : 1.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] *= sc;
                }
            }
        }
        if constexpr (RES_PRE) {
            if (R) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const u32x2 rv = rpre[i][j];
                        const f32x2 r01 = unpack2t<F16>(rv[0]), r23 = unpack2t<F16>(rv[1]);
                        acc[i][j][0] += r01[0];
                        acc[i][j][1] += r01[1];
                        acc[i][j][2] += r23[0];
                        acc[i][j][3] += r23[1];
                    }
                }
            }
        }
        if constexpr (RS_OK) {
            if (do_rs) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { sm += acc[i][j][r]; sq = fmaf(acc[i][j][r], acc[i][j][r], sq); }
                    rsum[i] = sm;
                    rsq[i] = sq;
                }
            }
        }
        // byte column of the lane's quad inside the staged row: GEGLU halves the column pitch (value columns only)
        const int cb0 = GEGLU ? (wn * WTN + fq * 8) : (wn * WTN * 2 + fq * 8);
        const int rowb = (wm * WTM + frow) * CPITCH;
#pragma unroll
        for (int j = 0; j < TN; j += (GEGLU ? 2 : 1)) {
            const int cb = cb0 + (GEGLU ? j * 16 : j * 32);
            const int at = rowb + (CSWZ ? (cb ^ (frow << 4)) : cb);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                u32x2 o;
                o[0] = pack2t<F16>(acc[i][j][0], acc[i][j][1]);
                o[1] = pack2t<F16>(acc[i][j][2], acc[i][j][3]);
                *reinterpret_cast<u32x2*>(smem + at + i * 16 * CPITCH) = o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + frow;
            if (m >= p.M) continue;
            const int rb = p.rowvec ? (m / p.rows_per_batch) : 0;
#pragma unroll
            for (int j = 0; j < TN; j += (GEGLU ? 2 : 1)) {
                const int n = n0 + wn * WTN + j * 16 + fq * 4;   // first of 4 consecutive GEMM columns
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
                if constexpr (LN_OK) {
                    if (do_ln) {
                        const f32x4 sv = spre[j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ln_term(v[r], lr[i], sv[r], lmr[i]);
                    }
                }
                if (p.bias) {
                    f32x4 bv;
                    if constexpr (BIAS_PRE) bv = bpre[j];
                    else bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bv[r];
                }
                int nc = n;   // output column
                if constexpr (GEGLU) {
                    float g[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[r] = acc[i][j + 1][r];
                    if constexpr (LN_OK) {
                        if (do_ln) {
                            const f32x4 sv = spre[j + 1];
#pragma unroll
                            for (int r = 0; r < 4; ++r) g[r] = ln_term(g[r], lr[i], sv[r], lmr[i]);
                        }
                    }
                    if (p.bias) {
                        f32x4 bg;
                        if constexpr (BIAS_PRE) bg = bpre[j + 1];
                        else bg = *reinterpret_cast<const f32x4*>(p.bias + n + 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] += bg[r];
                    }
                    const f32x2 ge0 = gelu_erf_f2(f32x2{g[0], g[1]}), ge1 = gelu_erf_f2(f32x2{g[2], g[3]});
                    v[0] *= ge0[0]; v[1] *= ge0[1]; v[2] *= ge1[0]; v[3] *= ge1[1];
                    nc = ((n - fq * 4) >> 1) + fq * 4;
                }
                if (p.rowvec) {
                    f32x4 tv;
                    if (RV_PRE && rv_pre_ok) tv = rvpre[j];
                    else tv = *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)rb * p.rowvec_ld + nc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += tv[r];
                }
                if (do_silu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
                }
                if (do_rot && n < p.rot_cols) {
                    // rotary on q|k columns (attention.py:649-651): channel inside its head = n % head_dim; the lane's 4
                    // consecutive columns are two interleaved pairs (x0,x1) -> (x0 c - x1 s, x1 c + x0 s)
                    const int ch = n % p.rot_head_dim;
                    if (ch < p.rot_dim) {
                        const int pos = m % p.rot_tokens_per_batch + p.rot_pos_offset;
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rot_table + ((int64_t)pos * (p.rot_dim / 2) + ch / 2) * 2);
                        const f32x2 r01 = rot_pair(f32x2{v[0], v[1]}, cs[0], cs[1]);
                        const f32x2 r23 = rot_pair(f32x2{v[2], v[3]}, cs[2], cs[3]);
                        v[0] = r01[0]; v[1] = r01[1]; v[2] = r23[0]; v[3] = r23[1];
                    }
                }
                if ((p.epilogue & SEER_EPI_COLSCALE) && n < p.col_scale_cols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= p.col_scale;
                }
                if (R) {
                    u32x2 rv;
                    if constexpr (RES_PRE) rv = rpre[i][j];
                    else rv = *reinterpret_cast<const u32x2*>(R + (int64_t)m * p.ldr + nc);
                    const f32x2 r01 = unpack2t<F16>(rv[0]), r23 = unpack2t<F16>(rv[1]);
                    v[0] += r01[0];
                    v[1] += r01[1];
                    v[2] += r23[0];
                    v[3] += r23[1];
                }
                if constexpr (RS_OK) {
                    if (do_rs) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { rsum[i] += v[r]; rsq[i] = fmaf(v[r], v[r], rsq[i]); }
                    }
                }
                if (trans) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (out_f32) Cf[(int64_t)(nc + r) * p.ldc + m] = v[r];
                        else if constexpr (F16) reinterpret_cast<_Float16*>(Cb)[(int64_t)(nc + r) * p.ldc + m] = (_Float16)v[r];
                        else Cb[(int64_t)(nc + r) * p.ldc + m] = (bf16)v[r];
                    }
                } else if (out_f32) {
                    *reinterpret_cast<f32x4*>(Cf + (int64_t)m * p.ldc + nc) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    u32x2 o;
                    o[0] = pack2t<F16>(v[0], v[1]);
                    o[1] = pack2t<F16>(v[2], v[3]);
                    if (staged) {
                        const int row_l = wm * WTM + i * 16 + frow;
                        const int col_l = nc - (GEGLU ? (n0 >> 1) : n0);
                        const int cb = col_l * 2;
                        *reinterpret_cast<u32x2*>(smem + row_l * CPITCH + (CSWZ ? (cb ^ ((row_l & 15) << 4)) : cb)) = o;
                    } else {
                        *reinterpret_cast<u32x2*>(Cb + crow(m) * p.ldc + nc) = o;
                    }
                }
            }
        }
    }   // general epilogue
    if constexpr (RS_OK) {
        if (do_rs) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float sm = rsum[i], sq = rsq[i];
                sm += __shfl_xor(sm, 16); sq += __shfl_xor(sq, 16);
                sm += __shfl_xor(sm, 32); sq += __shfl_xor(sq, 32);
                // ONE instruction for both: lanes fq = 0 add the sums, lanes fq = 1 the squares of the same 16 rows -- 256 contiguous
                // bytes = four fully used 64-byte atomic requests (the requests are the cost: ~12 ns each per CU)
                const int m = m0 + wm * WTM + i * 16 + frow;
                if (fq < 2 && m < p.M) fx_add_ln(p.rowstat + (int64_t)m * 2 + fq, fq ? sq : sm);
            }
        }
    }
    PSTAMP();
    if (staged) {
        __syncthreads();
        PSTAMP();
        constexpr int CPR = BNO / 8;                    // 16-byte chunks per staged row
        const bool wt_store = p.K <= 3072;
        const int n0o = GEGLU ? (n0 >> 1) : n0;
        const int n_out = GEGLU ? (p.N >> 1) : p.N;
        // ---- column sums of the tile as stored (bf16-rounded): sum and sum of squares per output column over the tile's rows,
        // written to colsum[z][tile_m][N][2].  The GroupNorm that consumes C adds them per (batch element, group) in a fixed
        // order (seer_groupnorm_stats_from_colsums) instead of re-reading C.  Thread -> (column pair, row segment): a wave reads
        // 256 contiguous bytes of one staged row per instruction (conflict-free), row segments are added in order by the
        // column's thread: deterministic, no atomics.
        constexpr int CS_CP = BNO / 2;                  // column pairs
        constexpr int CS_RS = NT / CS_CP;               // row segments
        constexpr bool COLSUM_OK = !GEGLU && !SPLIT && tile_colsum_ok<BM, BN, NS, WM, WN>();
        float* cs_scratch = reinterpret_cast<float*>(smem + BM * CPITCH);
        if constexpr (COLSUM_OK) {
            if (p.colsum || p.colsum_fx) {
                const int cp = tid % CS_CP, rs = tid / CS_CP;
                if (rs < CS_RS) {
                    const int rows = min(BM, p.M - m0);
                    const int chunk = cp >> 2, within = (cp & 3) * 4;
                    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
                    for (int r = rs; r < rows; r += CS_RS) {
                        const uint32_t v = *reinterpret_cast<const uint32_t*>(
                            smem + r * CPITCH + (CSWZ ? (chunk ^ (r & 15)) : chunk) * 16 + within);
                        const f32x2 f01 = unpack2t<F16>(v);
                        const float f0 = f01[0], f1 = f01[1];
                        s0 += f0; q0 += f0 * f0;
                        s1 += f1; q1 += f1 * f1;
                    }
                    *reinterpret_cast<f32x4*>(cs_scratch + (rs * CS_CP + cp) * 4) = f32x4{s0, q0, s1, q1};
                }
            }
        }
#pragma unroll
        for (int it = 0; it < (BM * CPR + NT - 1) / NT; ++it) {
            const int c = tid + it * NT;
            const int row = c / CPR, ch = c - row * CPR;
            const int m = m0 + row, n = n0o + ch * 8;
            if (((BM * CPR) % NT == 0 || c < BM * CPR) && m < p.M && n < n_out)
            {
                // write-through below ~3000 of K, where the boundary write-back is a visible share of the launch (projections
                // 11.3 -> 9.6 us, 1x1 shortcuts 15.8 -> 13.8); the long-K convs measured 1-3 % slower with it and keep plain stores
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CPITCH + (CSWZ ? (ch ^ (row & 15)) : ch) * 16);
                if (wt_store) store16_out(Cb + crow(m) * p.ldc + n, v);
                else *reinterpret_cast<u32x4*>(Cb + crow(m) * p.ldc + n) = v;
            }
        }
        if constexpr (COLSUM_OK) {
            if (p.colsum || p.colsum_fx) {
                __syncthreads();                        // the row-segment partials are parked
                if (tid < BNO && n0o + tid < n_out) {
                    float sm = 0.f, sq = 0.f;
#pragma unroll
                    for (int rs = 0; rs < CS_RS; ++rs) {
                        const float* e = cs_scratch + (rs * CS_CP + (tid >> 1)) * 4 + (tid & 1) * 2;
                        sm += e[0];
                        sq += e[1];
                    }
                    if (p.colsum_fx) {                   // accumulate per batch element (every phase of an upsampling conv adds here)
                        int64_t* o = fx_slot(p, m0 / BM, m0) + n0o + tid;
                        fx_add(o, sm);
                        fx_add(o + p.N, sq);
                    } else {
                        const int tiles_m = (p.M + BM - 1) / BM;
                        float* o = p.colsum + (((int64_t)blockIdx.z * tiles_m + m0 / BM) * p.N + n0o + tid) * 2;
                        *reinterpret_cast<f32x2*>(o) = f32x2{sm, sq};
                    }
                }
            }
        }
    }
    PSTAMP();
    PSPAN(1);
}

// split-K second pass: C = epilogue( sum over slices, in slice order ).  One quad of 4 output columns of row m; returns the
// four values as stored (bf16-rounded unless the output is fp32).
__device__ __forceinline__ f32x4 splitk_reduce_quad(const seer_gemm_desc& p, int m, int n) {
    const float* ws = reinterpret_cast<const float*>(p.workspace) + (int64_t)m * p.N + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(ws);
#pragma unroll 4
    for (int z = 1; z < p.splits; ++z) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (int64_t)z * p.M * p.N);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] += b[r];
    }
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[r];
    }
    if (p.rowvec) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(m / p.rows_per_batch) * p.rowvec_ld + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += tv[r];
    }
    if (p.epilogue & SEER_EPI_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    }
    if ((p.epilogue & SEER_EPI_COLSCALE) && n < p.col_scale_cols) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.col_scale;
    }
    const bool h16 = (p.epilogue & SEER_EPI_F16) != 0;      // IEEE-half storage (the fp16 engine): same bytes, other conversions
    if (p.residual) {
        const u32x2 rv = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(p.residual) + (int64_t)m * p.ldr + n);
        if (h16) {
            const f32x2 r01 = unpack2t<true>(rv[0]), r23 = unpack2t<true>(rv[1]);
            v[0] += r01[0]; v[1] += r01[1]; v[2] += r23[0]; v[3] += r23[1];
        } else {
            v[0] += __builtin_bit_cast(float, rv[0] << 16);
            v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
            v[2] += __builtin_bit_cast(float, rv[1] << 16);
            v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
        }
    }
    if (p.epilogue & SEER_EPI_OUT_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
        return f32x4{v[0], v[1], v[2], v[3]};
    }
    u32x2 o;
    if (h16) {
        o[0] = pack2h(v[0], v[1]);
        o[1] = pack2h(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n) = o;
        const f32x2 s01 = unpack2t<true>(o[0]), s23 = unpack2t<true>(o[1]);
        return f32x4{s01[0], s01[1], s23[0], s23[1]};
    }
    o[0] = pack2(v[0], v[1]);
    o[1] = pack2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n) = o;
    return f32x4{__builtin_bit_cast(float, o[0] << 16), __builtin_bit_cast(float, o[0] & 0xffff0000u),
                 __builtin_bit_cast(float, o[1] << 16), __builtin_bit_cast(float, o[1] & 0xffff0000u)};
}

__global__ void __launch_bounds__(256) seer_splitk_reduce_kernel(const seer_gemm_desc p) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N / 4;
    if (idx >= (int64_t)p.M * n4) return;
    const int m = (int)(idx / n4);
    const int n = (int)(idx - (int64_t)m * n4) * 4;
    (void)splitk_reduce_quad(p, m, n);
}

// the same pass with column sums (seer_gemm_desc::colsum): a block owns splitk_cs_rows(M) rows x 256 columns, thread -> (column
// quad, row lane) with the rows of a lane added in order, the four row lanes of a column added in order by its thread;
// colsum[ceil(M / rows)][N][2].  Few rows (the 4x4 / 8x8 levels): one row per thread, as many blocks as the plain reduce pass --
// at 16 rows per block the 384-row convs ran 120 blocks of 64-deep load chains, +12 us (profiles/r02_gn_colsums.log).
// (Round 2 took this kernel out when a two-process test differed in the last bits with column sums on; the cause was elsewhere --
// a packed-fp32 instruction form in the rotary epilogue, profiles/r03_flake_root_cause.md -- and it is back.)
__host__ __device__ inline int splitk_cs_rows(int M) { return M >= 2048 ? 16 : 4; }
__global__ void __launch_bounds__(256) seer_splitk_reduce_colsum_kernel(const seer_gemm_desc p) {
    __shared__ float part[4][64][8];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + cq) * 4;
    const int cs_rows = splitk_cs_rows(p.M);
    const int mb = blockIdx.y * cs_rows;
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        for (int m = mb + rl; m < min(mb + cs_rows, p.M); m += 4) {
            const f32x4 v = splitk_reduce_quad(p, m, n);
#pragma unroll
            for (int r = 0; r < 4; ++r) { sm[r] += v[r]; sq[r] += v[r] * v[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[rl][cq][r * 2] = sm[r]; part[rl][cq][r * 2 + 1] = sq[r]; }
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < p.N) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a += part[k][threadIdx.x >> 2][(threadIdx.x & 3) * 2];
            b += part[k][threadIdx.x >> 2][(threadIdx.x & 3) * 2 + 1];
        }
        *reinterpret_cast<f32x2*>(p.colsum + ((int64_t)blockIdx.y * p.N + col) * 2) = f32x2{a, b};
    }
}

// the same pass for ACCUMULATED column sums (seer_gemm_desc::colsum_fx).  The atomics are the cost here (one per address every
// ~12 ns, ~1.3 TB/s of them chip-wide): a block owns RB = RL * RPL rows x CB columns and adds ONE pair per column -- 16 x fewer
// than one per 4-row strip (measured with 4 / 16-row strips: +10..15 us per conv, profiles/r04_gn_fx_producers.md).  Thread -> (column
// quad, row lane); the row lanes of a column are added in order by the column's thread.
template <int CB, int RL, int RPL>
__global__ void __launch_bounds__(256) seer_splitk_reduce_fx_kernel(const seer_gemm_desc p) {
    constexpr int NQ = CB / 4, RB = RL * RPL;
    static_assert(NQ * RL == 256, "one thread per (column quad, row lane)");
    __shared__ float part[RL][NQ][8];
    const int cq = threadIdx.x % NQ, rl = threadIdx.x / NQ;
    const int n = blockIdx.x * CB + cq * 4;
    const int mb = blockIdx.y * RB;
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int m = mb + rl + i * RL;
            if (m < p.M) {
                const f32x4 v = splitk_reduce_quad(p, m, n);
#pragma unroll
                for (int r = 0; r < 4; ++r) { sm[r] += v[r]; sq[r] += v[r] * v[r]; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[rl][cq][r * 2] = sm[r]; part[rl][cq][r * 2 + 1] = sq[r]; }
    __syncthreads();
    const int col = blockIdx.x * CB + threadIdx.x;
    if (threadIdx.x < CB && col < p.N) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            a += part[k][threadIdx.x >> 2][(threadIdx.x & 3) * 2];
            b += part[k][threadIdx.x >> 2][(threadIdx.x & 3) * 2 + 1];
        }
        int64_t* o = fx_slot(p, blockIdx.y, mb) + col;
        fx_add(o, a);
        fx_add(o + p.N, b);
    }
}
__host__ inline int splitk_fx_rows(int M) { return M >= 1024 ? 64 : 32; }

// the kernel's `staged` condition, host side: column sums are taken from the staged bf16 tile
bool colsum_store_ok(const seer_gemm_desc& d) {
    return !(d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_GEGLU)) && d.ldc % 8 == 0 && d.N % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(d.C) & 15) == 0 && (d.batch <= 1 || (d.strideC % 8) == 0);
}

// one flag per tile instantiation, at namespace scope (no function-local statics in the library): the dynamic-LDS opt-in of its
// kernels has run
template <int BM, int BN, int NS, int WM, int WN, bool F16>
std::once_flag g_tile_lds_once;

// F16: the IEEE-half instantiation of the same tile (SEER_EPI_F16: the VAE, and the UNet engine under fp16 autocast)
template <int BM, int BN, int NS, int WM = 2, int WN = 2, bool F16 = false>
int launch_tile(const seer_gemm_desc& d, hipStream_t st) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, d.batch > 1 ? d.batch : 1);
    const size_t lds = (size_t)(NS == 0 ? 2 : NS) * (BM + BN) * BK * sizeof(bf16);
    constexpr bool GEGLU_OK = ((BN / 32) % 2) == 0;      // value / gate n-tile pairs must sit in one wave
    if (lds > 64 * 1024) {
        // above the default dynamic-LDS limit: opt in once per instantiation (160 KiB per CU on gfx950); std::call_once keeps
        // concurrent first calls from different host threads safe (the header promises thread safety)
        std::call_once(g_tile_lds_once<BM, BN, NS, WM, WN, F16>, [lds] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_kernel<BM, BN, true, false, false, NS, WM, WN, F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if constexpr (GEGLU_OK)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_kernel<BM, BN, false, true, false, NS, WM, WN, F16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_kernel<BM, BN, false, false, false, NS, WM, WN, F16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
    }
    const bool conv = d.mode == SEER_GEMM_CONV3X3;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (conv && geglu) return SEER_EINVAL;
    if ((d.colsum || d.colsum_fx) && (geglu || !tile_colsum_ok<BM, BN, NS, WM, WN>() || !colsum_store_ok(d))) return SEER_EINVAL;
    if ((d.rowstat || d.ln_rowstat) && !tile_ln_ok<BM, BN, NS, WM, WN>()) return SEER_EINVAL;
    if (conv) {
        hipLaunchKernelGGL((seer_gemm_kernel<BM, BN, true, false, false, NS, WM, WN, F16>), grid, dim3(64 * WM * WN), lds, st, d);
    } else if (geglu) {
        if constexpr (GEGLU_OK)
            hipLaunchKernelGGL((seer_gemm_kernel<BM, BN, false, true, false, NS, WM, WN, F16>), grid, dim3(64 * WM * WN), lds, st, d);
        else
            return SEER_EINVAL;
    } else {
        hipLaunchKernelGGL((seer_gemm_kernel<BM, BN, false, false, false, NS, WM, WN, F16>), grid, dim3(64 * WM * WN), lds, st, d);
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

template <int BM, int BN, int NS, bool F16 = false>
int launch_split_tile(const seer_gemm_desc& d, hipStream_t st) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, d.splits);
    const size_t lds = (size_t)(NS == 0 ? 2 : NS) * (BM + BN) * BK * sizeof(bf16);
    if (d.mode == SEER_GEMM_CONV3X3)
        hipLaunchKernelGGL((seer_gemm_kernel<BM, BN, true, false, true, NS, 2, 2, F16>), grid, dim3(256), lds, st, d);
    else
        hipLaunchKernelGGL((seer_gemm_kernel<BM, BN, false, false, true, NS, 2, 2, F16>), grid, dim3(256), lds, st, d);
    SEER_LAUNCH_CHECK();
    if (d.colsum_fx) {
        if (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_GEGLU)) return SEER_EINVAL;
        if (splitk_fx_rows(d.M) == 64)
            hipLaunchKernelGGL((seer_splitk_reduce_fx_kernel<64, 16, 4>), dim3((unsigned)((d.N + 63) / 64), (unsigned)((d.M + 63) / 64)),
                               dim3(256), 0, st, d);
        else
            hipLaunchKernelGGL((seer_splitk_reduce_fx_kernel<32, 32, 1>), dim3((unsigned)((d.N + 31) / 32), (unsigned)((d.M + 31) / 32)),
                               dim3(256), 0, st, d);
    } else if (d.colsum) {
        if (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_GEGLU)) return SEER_EINVAL;
        const int cs_rows = splitk_cs_rows(d.M);
        hipLaunchKernelGGL(seer_splitk_reduce_colsum_kernel, dim3((unsigned)((d.N + 255) / 256),
                           (unsigned)((d.M + cs_rows - 1) / cs_rows)), dim3(256), 0, st, d);
    } else {
        const int64_t n = (int64_t)d.M * (d.N / 4);
        hipLaunchKernelGGL(seer_splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d);
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

template <bool F16>
int launch_split_t(const seer_gemm_desc& d, hipStream_t st) {
    if (d.tile == SEER_TILE_64x64) return launch_split_tile<64, 64, 0, F16>(d, st);        // register-staged (A/B testing)
    if (d.tile == SEER_TILE_G64x64_3) return launch_split_tile<64, 64, 3, F16>(d, st);
    if (d.tile == SEER_TILE_G128x128_2) return launch_split_tile<128, 128, 2, F16>(d, st);
    if (d.tile == SEER_TILE_G96x160_2) return launch_split_tile<96, 160, 2, F16>(d, st);
    // auto: prepare() already wrote its tile choice into d.tile; anything else keeps the 64x64 ring
    return launch_split_tile<64, 64, 3, F16>(d, st);
}
int launch_split(const seer_gemm_desc& d, hipStream_t st) {
    return (d.epilogue & SEER_EPI_F16) ? launch_split_t<true>(d, st) : launch_split_t<false>(d, st);
}

// validate + normalise a descriptor; returns SEER_OK and the number of K slices the call will use in *splits
int prepare(seer_gemm_desc& d, int* splits) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return SEER_EINVAL;
    if (d.K % BK) return SEER_EINVAL;
    if (!d.A || !d.W || !d.C) return SEER_EINVAL;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (d.N % (geglu ? 32 : 4)) return SEER_EINVAL;
    if (d.mode == SEER_GEMM_CONV3X3) {
        if (d.Cin <= 0 || d.Cin % BK) return SEER_EINVAL;
        if (d.Hin <= 0 || d.Win <= 0 || d.Hout <= 0 || d.Wout <= 0) return SEER_EINVAL;
        if (d.upsample == 2) {
            // four 2x2 phase convs as the four batch elements of one launch: same input, W[phase][N][4 Cin], rows scattered
            if (d.K != 4 * d.Cin || d.stride != 1 || d.pad_after_only || d.Hout != 2 * d.Hin || d.Wout != 2 * d.Win) return SEER_EINVAL;
            if (d.M % (d.Hin * d.Win) || d.residual || d.rowvec) return SEER_EINVAL;
            if (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_ROTARY | SEER_EPI_GEGLU)) return SEER_EINVAL;
            d.batch = 4;
            d.strideA = 0;
            d.strideW = (int64_t)d.N * d.K;
            d.strideC = 0;
        } else {
            if (d.K != 9 * d.Cin) return SEER_EINVAL;
            if (d.stride != 1 && d.stride != 2) return SEER_EINVAL;
            if (d.M % (d.Hout * d.Wout)) return SEER_EINVAL;
        }
        d.K1 = d.K;
    } else if (d.mode == SEER_GEMM_PLAIN) {
        if (!d.A2) { d.K1 = d.K; d.lda2 = 0; }
        if (d.K1 % BK || d.K1 > d.K || d.lda % 8 || (d.A2 && d.lda2 % 8)) return SEER_EINVAL;
    } else {
        return SEER_EINVAL;
    }
    if (!(d.epilogue & SEER_EPI_TRANS_OUT) && (d.ldc % 4)) return SEER_EINVAL;
    if (d.residual && (d.ldr % 4)) return SEER_EINVAL;
    if (d.rowvec && d.rows_per_batch <= 0) return SEER_EINVAL;
    if (d.epilogue & SEER_EPI_ROTARY) {
        if (!d.rot_table || d.rot_head_dim <= 0 || d.rot_head_dim % 4 || d.rot_dim <= 0 || d.rot_dim % 4 ||
            d.rot_dim > d.rot_head_dim || d.rot_tokens_per_batch <= 0 || d.rot_cols % d.rot_head_dim || geglu)
            return SEER_EINVAL;
    }
    if (d.epilogue & SEER_EPI_COLSCALE) {
        if (geglu || d.col_scale_cols <= 0 || d.col_scale_cols % 4 || d.col_scale_cols > d.N) return SEER_EINVAL;
    }
    if (d.batch <= 1) d.batch = 1;

    // split-K decision: few output tiles and a long K loop (deep-level convs / linears, M = 384 .. 1536)
    int s = 1;
    const bool can_split = d.batch == 1 && !geglu && !(d.epilogue & (SEER_EPI_TRANS_OUT | SEER_EPI_ROTARY)) &&
                           (d.tile == SEER_TILE_AUTO || d.tile == SEER_TILE_64x64 || d.tile == SEER_TILE_G64x64_3 ||
                            d.tile == SEER_TILE_G128x128_2 || d.tile == SEER_TILE_G96x160_2) &&
                           d.splits != 1;
    if (can_split) {
        const int nk = d.K / BK;
        if (d.splits > 1) {
            s = d.splits < nk ? d.splits : nk;
        } else if (d.splits == 0) {
            // measured on MI355X (profiles/r01_splitk_sweep.log, r01_sweep_shapes.log): the reduce pass + second launch
            // cost ~4-5 us, so splitting pays only when a slice keeps >= ~11 K tiles and the unsplit grid leaves CUs idle.
            // Wide outputs (N a multiple of 128, >= 640): 128x128 tiles (half the L2->LDS bytes per FLOP of 64x64), K
            // sliced until the grid holds ~2 blocks per CU -- the 16x16-level convs at B*F = 24 (x2), the 8x8 level (x4),
            // the 4x4 level (x16) and, with fewer frames or one CFG half per rank, the same levels sliced deeper.
            const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
            const long blocks64 = (long)((d.M + 63) / 64) * ((d.N + 63) / 64);
            // (round 2 kept a plain GEMM whose 64x64 grid already holds ~2 blocks per CU and whose K is only moderately long -- the
            // 8x8-level feed-forward output projection, 1536 x 1280 x 5120 -- on the 5-stage 64x64 ring unsplit, 40.8 against 47.4 us
            // for 128x128 x 4 slices + the reduce pass; since the staging path lost its address arithmetic the slices win, 35.2
            // against 40.0: profiles/r04_ff2_tile_sweep.log)
            const bool unsplit_ring = false;
            int s128 = 1;
            // (cold weights, r06_lab_cold_weights.log / r06_lab_cold_train.log: a PLAIN product is sliced to ONE round of the chip, not two
            //  -- 1536 x 1280 x 6400: 2 slices 41.3 against 43.3 us for 4; 768 x 1280 x 10240: 4 slices 37.0 against 45.3 for 8; 924 x 768 x
            //  6144: 4 slices 23.8 against 30.7 -- while the convs, whose slices are ten times longer, keep two)
            const long split_target = d.mode == SEER_GEMM_PLAIN ? 180 : 400, split_accept = d.mode == SEER_GEMM_PLAIN ? 180 : 200;
            if (d.N % 128 == 0 && d.N >= 640 && d.M >= 256 && t128 < 256 && nk >= 64 && !unsplit_ring)
                while (t128 * s128 < split_target && s128 < 16 && nk / (2 * s128) >= 11) s128 *= 2;
            // N = 320 on half the rows (one CFG half per rank: 12 288 rows = 256 tiles of 96x160, one lone workgroup per CU, which
            // runs a K tile no faster than two co-resident ones do): two K slices bring the second workgroup back.  3x3 convs only
            // (K >= 2880): 47.1 -> 44.9, 84.4 -> 76.8, 128.0 -> 103.5 us; the K = 1280 GEMM loses (profiles/r02_half_rows.log)
            const long t96160 = (long)((d.M + 95) / 96) * ((d.N + 159) / 160);
            // Round 6, COLD weights (profiles/r06_lab_cold_weights.log: each launch of a shape reads another weight tensor, >= 400 MB in
            // rotation, as inside the step, where a launch's weights were last touched 1.7 GB of other weights ago and come from HBM,
            // not from the 256 MB memory-side cache a back-to-back sweep over ONE tensor reads them from): cold, every launch costs
            // 3-15 us more, and the ranking moves towards FEWER workgroups re-reading a weight tile -- fewer K slices, wider tiles.
            //  * the 4x4-level convs (384 rows): 96x160 tiles x 8 slices = 256 workgroups, ONE round (29.5 / 44.6 us cold against 36.1 /
            //    48.2 for 128x128 x 16 slices; hot the two are level)
            //    (fewer rows -- a rank's share of the frames -- keep the one round: 256 / tiles slices, each >= 11 K tiles)
            if (d.tile == SEER_TILE_AUTO && d.mode == SEER_GEMM_CONV3X3 && d.M <= 384 && d.N % 160 == 0 && nk >= 160 && t96160 <= 32) {
                s = (int)(256 / t96160);
                while (s > 1 && nk / s < 11) s >>= 1;
                d.tile = SEER_TILE_G96x160_2;
            } else
            if (s128 > 1 && t128 * s128 >= split_accept && d.tile == SEER_TILE_AUTO) {
                s = s128;
                d.tile = SEER_TILE_G128x128_2;
            // (with COLD weights the 320 -> 320 conv, K = 2880, is faster unsplit -- 34.8 against 40.6 us -- and the longer ones keep their
            //  two slices: 60.9 against 63.7, 81.3 against 91.7; profiles/r06_lab_cold_train_convs.log)
            } else if (d.N == 320 && d.mode == SEER_GEMM_CONV3X3 && t96160 >= 128 && t96160 <= 256 && nk >= 80 &&
                       d.tile == SEER_TILE_AUTO) {
                s = 2;
                d.tile = SEER_TILE_G96x160_2;
            } else {
                const long blocks = blocks64;
                if (blocks <= 160 && nk >= 160) s = blocks <= 40 ? 16 : 8;      // 4x4 / 8x8 level convs of a frame shard
                else if (blocks <= 160 && nk >= 40) s = nk / 20 < 8 ? nk / 20 : 8;
                else if (blocks <= 512 && nk >= 80 && !unsplit_ring) s = 4;
                if (s > 1 && d.tile == SEER_TILE_AUTO) d.tile = SEER_TILE_G64x64_3;
            }
        }
    }
    *splits = s;
    return SEER_OK;
}

// the tile an unsplit, non-weight-stationary launch of `d` runs (d.tile, or the AUTO choice)
int resolve_tile(const seer_gemm_desc& d) {
    int tile = d.tile;
    if (tile == SEER_TILE_AUTO) {
        // from the MI355X sweep (profiles/r01_gemm_tile_sweep.log): LDS-direct 2-stage 128x128 wherever it fills the chip,
        // LDS-direct 3-stage 128x64 / 64x64 for narrow N or few rows once the K loop is long enough to amortise the ring
        // prologue, register-staged 64x64 for short K.
        const int nk = d.K / BK;
        const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch;
        const long t12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64) * d.batch;
        const int n128 = (d.N + 127) / 128 * 128;
        const bool n_fits_128 = (n128 - d.N) * 8 <= d.N;           // <= 12.5 % padded columns
        const long t128160 = (long)((d.M + 127) / 128) * ((d.N + 159) / 160) * d.batch;
        const long t96160_ = (long)((d.M + 95) / 96) * ((d.N + 159) / 160) * d.batch;
        // (12 288 rows x 320 -- one CFG half, or the b = 1 fine-tuning step -- are 256 tiles of 96x160: one round; 16.2 / 28.6 / 12.9 us
        //  against 19.6 / 33.9 / 15.1 on 128x64 with cold weights, r06_lab_cold_train.log)
        if ((d.N == 320 || d.N == 960) && nk >= 5 && (t128160 >= 256 || (d.N == 320 && t96160_ >= 224 && t96160_ <= 256)) &&
            !(d.epilogue & SEER_EPI_GEGLU)) {
            // (with the fast epilogue the 160-wide tiles also win at K = 320 .. 960, where the register-staged 64x64 tile used to:
            //  projections of the 320-wide level 17.5 -> 15.6 / 14.4 -> 12.0 us, its 1x1 shortcuts 20.0 -> 16.2 / 26.5 -> 20.6, and
            //  its q|k|v projection, N = 960 = 6 x 160, 32.7 (weight-stationary) -> 26.4: profiles/r02_tile_sweep_fastepi.log)
            // N = 320 in two 160-wide tiles: no padded columns, 0.45x the L2->LDS traffic of 64x64.  Rows per tile: whichever
            // of 128 / 96 leaves the last round of tiles fuller -- 24 576 rows (the 32x32 level at CFG batch 2) are 384 tiles
            // of 128 rows (1.5 per CU) but 512 of 96 rows (2 per CU): +13 % on the 320->320 conv (profiles/r01_tile96.log)
            const long t96160 = (long)((d.M + 95) / 96) * ((d.N + 159) / 160) * d.batch;
            auto fill = [](long t) { return (double)t / (256.0 * (double)((t + 255) / 256)); };
            tile = fill(t96160) > fill(t128160) + 0.05 ? SEER_TILE_G96x160_2 : SEER_TILE_G128x160_2;
            // the rotary epilogue (temporal q|k|v) reads a (cos, sin) row per output row: fewer, taller tiles re-read less of the
            // table -- 24 576 x 960 x 320: 30.0 us on 128x160 against 34.8 on 96x160 (profiles/r03_tile_ab_rotary.log)
            if ((d.epilogue & SEER_EPI_ROTARY) && t128160 >= 256) tile = SEER_TILE_G128x160_2;
        }
        else if ((d.epilogue & SEER_EPI_ROTARY) && d.N == 1920 && nk >= 5 && (long)((d.M + 95) / 96) * 12 * d.batch >= 256)
            tile = SEER_TILE_G96x160_2;      // 6 144 x 1 920 x 640 rotary: 24.0 us against 26.8 on 128x128 (same log)
        // (cold weights, r06_lab_cold_weights.log: with a short K loop the fuller last round of 96-row tiles does not pay for their extra
        //  weight reads -- the 16x16-level q|k|v projection, 6144 x 1920 x 640: 21.5 us on 128x128 against 24.4 cold, 20.8 / 19.2 hot)
        else if (t128 >= 256 && n_fits_128 && d.N >= 640 && nk <= 10 && d.mode == SEER_GEMM_PLAIN && !(d.epilogue & SEER_EPI_GEGLU))
            tile = SEER_TILE_G128x128_2;
        else if (t128 >= 256 && n_fits_128 && d.N >= 640) {
            // 128x128, unless 96-row tiles leave the last round of resident workgroups (two per CU) clearly fuller: the q|k|v
            // projections of the 8x8 / 16x16 level are 360 / 720 tiles of 128 rows (0.70 of one / two rounds) but 480 / 960 of 96
            // rows (0.94): 22.0 -> 19.0, 23.1 -> 20.2 (rotary), 22.3 -> 20.9 us (profiles/r04_plain_tile_sweep.log)
            const long t96128 = (long)((d.M + 95) / 96) * ((d.N + 127) / 128) * d.batch;
            auto fill2 = [](long t) { return (double)t / (512.0 * (double)((t + 511) / 512)); };
            tile = (d.mode == SEER_GEMM_PLAIN && fill2(t96128) > fill2(t128) + 0.15) ? SEER_TILE_G96x128_2 : SEER_TILE_G128x128_2;
        }
        // a conv whose 128x128 grid just misses one round of the chip but whose 96x128 grid fills it (the 16x16-level 320 -> 640
        // conv: 240 / 320 tiles): 34.5 against 45.0 us on 128x64 (profiles/r04_conv_tile_sweep.log); plain GEMMs of those sizes stay
        // on 128x64 (ff.net.2 at that level: 34.0 against 31.5, r04_ff2_tile_sweep.log)
        // ... unless 96x160 tiles make exactly one round of the chip (that conv: 256 tiles; 35.0 against 41.7 us with cold weights,
        // r06_lab_cold_weights.log)
        else if (d.mode == SEER_GEMM_CONV3X3 && d.N % 160 == 0 && d.N >= 640 && nk >= 40 &&
                 (long)((d.M + 95) / 96) * (d.N / 160) * d.batch >= 224 && (long)((d.M + 95) / 96) * (d.N / 160) * d.batch <= 256)
            tile = SEER_TILE_G96x160_2;
        else if (d.mode == SEER_GEMM_CONV3X3 && n_fits_128 && d.N >= 640 && t128 >= 192 && nk >= 40 &&
                 (long)((d.M + 95) / 96) * ((d.N + 127) / 128) * d.batch >= 256)
            tile = SEER_TILE_G96x128_2;
        // cold weights (r06_lab_cold_weights.log): a 128x128 grid that almost fills one round (200-255 tiles: the 16x16-level projections,
        // shortcuts and ff.net.2 | proj_out, 6144 rows x 640) beats twice as many 128x64 tiles -- 10.2 / 16.5 / 22.0 / 34.1 us against 11.6 /
        // 19.2 / 26.5 / 38.7 (hot: level); and where the 128x64 grid itself is 200-255 tiles (the 8x8-level projections and shortcuts,
        // 1536 rows x 1280) it beats the 64x64 ring: 12.5 / 20.6 against 13.7 / 24.9
        else if (d.mode == SEER_GEMM_PLAIN && n_fits_128 && d.N >= 640 && t128 >= 200 && nk >= 10) tile = SEER_TILE_G128x128_2;
        else if (d.mode == SEER_GEMM_PLAIN && n_fits_128 && d.N >= 640 && t128 < 200 && nk >= 10 &&
                 (long)((d.M + 95) / 96) * ((d.N + 127) / 128) * d.batch >= 200 && (long)((d.M + 95) / 96) * ((d.N + 127) / 128) * d.batch <= 256)
            tile = SEER_TILE_G96x128_2;         // 768 x 3840 x 1280 (b = 1 step, 8x8 level): 240 tiles, 14.9 against 20.6 us cold on 128x64
        else if (d.mode == SEER_GEMM_PLAIN && t12864 >= 200 && t12864 < 256 && nk >= 10) tile = SEER_TILE_G128x64_3;   // (3072 x 640 x 640: 7.7 against 10.5)
        else if (t12864 >= 256 && nk >= 5) tile = SEER_TILE_G128x64_3;   // (K = 320 too: 8.9 vs 9.7 us on 12 288 x 320, r02_half_rows.log)
        else if (nk >= 64) tile = SEER_TILE_G64x64_5;        // long K on few tiles: deeper ring (see prepare(), unsplit_ring)
        else if (nk >= 12) tile = SEER_TILE_G64x64_3;
        else tile = SEER_TILE_64x64;
    }
    return tile;
}

// The 256 x 320 tile kernel (gemm_t320.hip): 0 = this launch does not go there, else the number of K slices it runs with
// (reduced inside the launch; needs desc.workspace and desc.sync).
//
// AUTO takes it where a cost model calibrated on MI355X (profiles/r04_t320_*.log) puts it ahead of the smaller tiles:
//   t = rounds * (K tiles per slice * 1.9 us [2.0 conv] + 7 us of prologue / epilogue [12 GEGLU]) + slab bytes / 4.3 TB/s + 3 us
// rounds = ceil(tiles * S / 256 CUs); slab bytes = 2 * tiles * S * 320 KB when S > 1 -- the partial tiles of a split launch go
// through the fabric (write-through stores, sc1 loads) at HBM-like rates whatever the split, which is what keeps the big tile
// away from the few-tile shapes of a 32x32-latent step: 256 workgroups x 320 KB x 2 = 164 MB = 38 us per split launch.
// The smaller tiles are priced at the best rate they reach on these shape classes (0.78-1.1 PFLOP/s): the big tile is only
// chosen where it beats that optimistic figure.
double t320_model_us(const seer_gemm_desc& d, int S) {
    const int nb = (d.mode == SEER_GEMM_CONV3X3 && d.upsample == 2) ? 4 : (d.batch > 1 ? d.batch : 1);
    const int tiles = ((d.M + 255) / 256) * (d.N / 320) * nb, nk = d.K / BK;
    const int rounds = (tiles * S + 255) / 256;
    const double t_iter = d.mode == SEER_GEMM_CONV3X3 ? 2.0 : 1.9;
    const double slab = S > 1 ? 2.0 * tiles * S * 327680.0 / 4.3e6 : 0.0;
    const double fixed = (d.epilogue & SEER_EPI_GEGLU) ? 12.0 : 7.0;      // prologue + epilogue of a tile (the erf epilogue: +5 us)
    return rounds * (((nk + S - 1) / S) * t_iter + fixed) + slab + 3.0;
}
int t320_best_split(const seer_gemm_desc& d, bool can_split, double* t_best) {
    static const int cand[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    const int nk = d.K / BK;
    int best = 1;
    double tb = t320_model_us(d, 1);
    if (can_split)
        for (int S : cand) {
            if (S > nk / 4) break;                   // a slice keeps at least four K tiles
            const double t = t320_model_us(d, S);
            if (t < tb) { tb = t; best = S; }
        }
    if (t_best) *t_best = tb;
    return best;
}
bool t320_buffers_ok(const seer_gemm_desc& d, int s) {
    return d.workspace && d.workspace_bytes >= seer_gemm_t320_workspace_bytes(d, s) && d.sync &&
           d.sync_bytes >= seer_gemm_t320_sync_bytes(d, s);
}
// assume_buffers: a size query -- plan as if workspace and sync will be provided; the launch plans with what it was given
int t320_plan(const seer_gemm_desc& d, bool assume_buffers = false) {
    if (d.rowstat || d.ln_rowstat) return 0;          // row statistics / folded LayerNorm live in the tile kernel
    if (d.tile != SEER_TILE_T256x320 && d.tile != SEER_TILE_AUTO) return 0;
    if (d.mode != SEER_GEMM_PLAIN && d.mode != SEER_GEMM_CONV3X3) return 0;
    if (!seer_gemm_t320_eligible(d)) return 0;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    const bool phases = d.mode == SEER_GEMM_CONV3X3 && d.upsample == 2;
    const bool can_split = !geglu && d.batch <= 1 && !phases && d.splits != 1;
    if (d.tile == SEER_TILE_T256x320) {
        int s = 1;
        if (can_split) {
            const int nk = d.K / BK;
            s = d.splits > 1 ? d.splits : t320_best_split(d, true, nullptr);
            if (s > nk / 4) s = nk / 4;
            if (s > 16) s = 16;
            if (s < 1) s = 1;
            if (!assume_buffers && s > 1 && !t320_buffers_ok(d, s)) s = 1;      // asked for by name: run it unsplit
        }
        return s;
    }
    // AUTO
    if (d.M % 256 || d.splits > 1) return 0;          // (ragged row tiles and explicit split requests stay with the smaller tiles)
    double t = 0.0;
    // AUTO never splits K on this kernel: the in-launch reduction waits for peer workgroups, which is only safe when this process
    // has the GPU to itself (gemm_t320.hip); ask for SEER_TILE_T256x320 + splits by name where that holds
    (void)can_split;
    const int s = t320_best_split(d, false, &t);
    const double flops = 2.0 * d.M * d.N * (double)d.K * (phases ? 4 : d.batch > 1 ? d.batch : 1);
    // FLOP per us.  (Convs: since the small tiles' gather lost its address arithmetic they reach 1.0-1.2 PFLOP/s on long-K convs
    // with many rows -- profiles/r04_t320_recalibration.log -- and the big tile is rarely ahead there.)
    const double rate = d.mode == SEER_GEMM_CONV3X3 ? 1.10e9 : geglu ? 0.78e9 : 0.85e9;
    if (!(t < flops / rate)) return 0;
    if (!assume_buffers && s > 1 && !t320_buffers_ok(d, s)) return 0;         // no room to split: the smaller tiles take it
    return s;
}

template <int BM_, int BN_, int NS_, int WM_ = 2, int WN_ = 2>
struct TileTag { static constexpr int BM = BM_, BN = BN_, NS = NS_, WM = WM_, WN = WN_; };

template <class F>
int dispatch_tile(int tile, F&& f) {
    switch (tile) {
        case SEER_TILE_128x128: return f(TileTag<128, 128, 0>{});
        case SEER_TILE_128x64: return f(TileTag<128, 64, 0>{});
        case SEER_TILE_64x64: return f(TileTag<64, 64, 0>{});
        case SEER_TILE_G128x128_2: return f(TileTag<128, 128, 2>{});
        case SEER_TILE_G128x128_3: return f(TileTag<128, 128, 3>{});
        case SEER_TILE_G128x64_3: return f(TileTag<128, 64, 3>{});
        case SEER_TILE_G64x64_3: return f(TileTag<64, 64, 3>{});
        case SEER_TILE_G64x64_4: return f(TileTag<64, 64, 4>{});
        case SEER_TILE_G64x64_5: return f(TileTag<64, 64, 5>{});
        case SEER_TILE_G128x64_4: return f(TileTag<128, 64, 4>{});
        case SEER_TILE_G128x160_2: return f(TileTag<128, 160, 2>{});
        case SEER_TILE_G64x160_3: return f(TileTag<64, 160, 3>{});
        case SEER_TILE_G256x128_2: return f(TileTag<256, 128, 2, 4>{});
        case SEER_TILE_G256x64_3: return f(TileTag<256, 64, 3, 4>{});
        case SEER_TILE_G256x256_2: return f(TileTag<256, 256, 2, 2, 4>{});
        case SEER_TILE_G96x160_2: return f(TileTag<96, 160, 2>{});
        case SEER_TILE_G96x160_3: return f(TileTag<96, 160, 3>{});
        case SEER_TILE_G96x128_2: return f(TileTag<96, 128, 2>{});
        default: return SEER_EINVAL;
    }
}

}  // namespace

extern "C" int64_t seer_gemm_workspace_bytes(const seer_gemm_desc* desc) {
    if (!desc) return SEER_EINVAL;
    seer_gemm_desc d = *desc;
    const int s320 = t320_plan(d, true);
    const int64_t w320 = s320 ? seer_gemm_t320_workspace_bytes(d, s320) : 0;
    if (d.tile == SEER_TILE_T256x320) d.tile = SEER_TILE_AUTO;
    if (d.tile == SEER_TILE_WS || d.tile == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;
    int s = 1;
    const int rc = prepare(d, &s);
    if (rc != SEER_OK) return rc;
    // (a launch that is planned for the 256 x 320 tile but arrives without `sync` falls back to the smaller tiles: room for both)
    const int64_t wold = s > 1 ? (int64_t)s * d.M * d.N * (int64_t)sizeof(float) : 0;
    return w320 > wold ? w320 : wold;
}

extern "C" int64_t seer_gemm_sync_bytes(const seer_gemm_desc* desc) {
    if (!desc) return SEER_EINVAL;
    if (const int s320 = t320_plan(*desc, true)) return seer_gemm_t320_sync_bytes(*desc, s320);
    return 0;
}

extern "C" int32_t seer_gemm_colsum_rows(const seer_gemm_desc* desc) {
    if (!desc) return 0;
    seer_gemm_desc d = *desc;
    if (const int s320 = t320_plan(d))            // one partial per wave row (64 rows) unsplit, per 16-row fragment when K is split
        return ((d.epilogue & SEER_EPI_GEGLU) || !colsum_store_ok(d) || d.M % 256) ? 0 : (s320 > 1 ? 16 : 64);
    if (d.tile == SEER_TILE_T256x320) d.tile = SEER_TILE_AUTO;
    const int requested = d.tile;
    if (requested == SEER_TILE_WS || requested == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;
    int s = 1;
    if (prepare(d, &s) != SEER_OK) return 0;
    if (!colsum_store_ok(d)) return 0;
    if (s > 1 && d.workspace && d.workspace_bytes >= (int64_t)s * d.M * d.N * (int64_t)sizeof(float))
        return d.batch <= 1 ? splitk_cs_rows(d.M) : 0;          // split-K: the reduce pass leaves them
    d.splits = 1;
    d.tile = desc->tile == SEER_TILE_T256x320 ? SEER_TILE_AUTO : desc->tile;
    if (requested == SEER_TILE_WS || requested == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;
    const int rows = dispatch_tile(resolve_tile(d), [&](auto t) {
        using T = decltype(t);
        return tile_colsum_ok<T::BM, T::BN, T::NS, T::WM, T::WN>() ? (int)T::BM : 0;
    });
    return rows > 0 ? rows : 0;           // an unknown tile code comes back as a negative status
}

// ---- folded LayerNorm (seer_gemm_desc::rowstat / ln_rowstat): such launches stay on the tile kernel, unsplit, with a staged
// bf16 output.  Returns the tile code the launch takes, 0 when it cannot carry row statistics.
int ln_resolve(const seer_gemm_desc& in) {
    seer_gemm_desc d = in;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (d.mode != SEER_GEMM_PLAIN || d.batch > 1 || (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_SILU)))
        return 0;
    if (d.ldc % 8 || d.N % (geglu ? 16 : 8) || (reinterpret_cast<uintptr_t>(d.C) & 15)) return 0;      // the kernel's `staged`
    if (d.rowstat && (geglu || d.colsum || d.colsum_fx)) return 0;
    if (d.ln_rowstat && (d.A2 || !d.ln_wsum || (reinterpret_cast<uintptr_t>(d.ln_wsum) & 15))) return 0;
    const bool special = d.tile == SEER_TILE_T256x320 || d.tile == SEER_TILE_WS || d.tile == SEER_TILE_AUTO_TILED;
    if (special) d.tile = SEER_TILE_AUTO;
    int sp = 1;
    if (prepare(d, &sp) != SEER_OK) return 0;
    d.splits = 1;
    d.tile = special ? SEER_TILE_AUTO : in.tile;
    const int tile = resolve_tile(d);
    const int ok = dispatch_tile(tile, [&](auto t) {
        using T = decltype(t);
        return tile_ln_ok<T::BM, T::BN, T::NS, T::WM, T::WN>() ? 1 : 0;
    });
    return ok == 1 ? tile : 0;
}

extern "C" int32_t seer_gemm_rowstat_ok(const seer_gemm_desc* desc) {
    if (!desc) return 0;
    seer_gemm_desc d = *desc;
    if (!d.rowstat) d.rowstat = reinterpret_cast<int64_t*>(16);      // the question is about the launch WITH row statistics
    return ln_resolve(d) ? 1 : 0;
}

extern "C" int32_t seer_gemm_lnfold_ok(const seer_gemm_desc* desc) {
    if (!desc || !desc->ln_rowstat) return 0;
    // a launch AUTO gives to the weight-stationary kernel (the level-0 GEGLU projection) keeps LayerNorm + that kernel: faster
    // than the tile kernel with the fold (58.7 vs 65.5 us, profiles/r02_tile_sweep_fastepi.log, against 7.6 us of LayerNorm)
    seer_gemm_desc d = *desc;
    d.ln_rowstat = nullptr;
    if (d.tile == SEER_TILE_AUTO && !d.colsum && !d.colsum_fx && !d.rowstat && seer_gemm_ws_eligible(d) && seer_gemm_ws_profitable(d)) return 0;
    return ln_resolve(*desc) ? 1 : 0;
}

extern "C" int32_t seer_gemm_colsum_fx_layout(const seer_gemm_desc* desc, int32_t rows_per_batch, int32_t* reps) {
    if (reps) *reps = 1;
    if (!desc || rows_per_batch <= 0 || desc->M % rows_per_batch) return 0;
    int rows = seer_gemm_colsum_rows(desc);       // tile rows; 64 / 16 on the 256 x 320 tile; the strip of the split-K reduce pass
    if (rows <= 0) return 0;
    seer_gemm_desc d = *desc;
    if (!t320_plan(d)) {
        if (d.tile == SEER_TILE_T256x320 || d.tile == SEER_TILE_WS || d.tile == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;
        int s = 1;
        if (prepare(d, &s) == SEER_OK && s > 1 && d.workspace && d.workspace_bytes >= (int64_t)s * d.M * d.N * (int64_t)sizeof(float))
            rows = splitk_fx_rows(d.M);           // the accumulating reduce pass owns bigger row blocks
    }
    if (rows_per_batch % rows) return 0;
    const int adds = rows_per_batch / rows;       // adds per address and launch without replicas
    if (reps) *reps = adds <= 24 ? 1 : adds <= 48 ? 2 : adds <= 96 ? 4 : 8;
    return rows;
}

extern "C" int seer_gemm_bf16(const seer_gemm_desc* desc, void* stream) {
    if (!desc) return SEER_EINVAL;
    seer_gemm_desc d = *desc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.colsum_fx) {                  // accumulated column sums: no partial of this launch may straddle two batch elements
        if (d.colsum) return SEER_EINVAL;
        int reps = 1;
        if (seer_gemm_colsum_fx_layout(desc, d.colsum_fx_rows, &reps) <= 0 || d.colsum_fx_reps < 1) return SEER_EINVAL;
    }
    const bool rows_ln = d.rowstat || d.ln_rowstat;       // row statistics / folded LayerNorm: tile kernel, unsplit
    if (rows_ln && !ln_resolve(d)) return SEER_EINVAL;
    if (const int s320 = rows_ln ? 0 : t320_plan(d)) {
        int sp = 1;
        seer_gemm_desc chk = d;
        chk.tile = SEER_TILE_AUTO;
        const int rc320 = prepare(chk, &sp);         // the same argument checks as every other launch
        if (rc320 != SEER_OK) return rc320;
        d.K1 = chk.K1; d.lda2 = chk.lda2; d.batch = chk.batch; d.strideA = chk.strideA; d.strideW = chk.strideW; d.strideC = chk.strideC;
        if ((d.colsum || d.colsum_fx) && ((d.epilogue & SEER_EPI_GEGLU) || !colsum_store_ok(d) || d.M % 256)) return SEER_EINVAL;
        return seer_gemm_t320_launch(d, s320, st);
    }
    if (d.tile == SEER_TILE_T256x320) d.tile = SEER_TILE_AUTO;      // not eligible: the tile kernels take it
    const int requested = d.tile;       // WS / AUTO_TILED are AUTO as far as tile and split-K selection go
    if (requested == SEER_TILE_WS || requested == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;
    int s = 1;
    const int rc = prepare(d, &s);
    if (rc != SEER_OK) return rc;
    if (s > 1 && !rows_ln && d.workspace && d.workspace_bytes >= (int64_t)s * d.M * d.N * (int64_t)sizeof(float)) {
        d.splits = s;
        return launch_split(d, st);
    }
    d.splits = 1;
    d.tile = desc->tile == SEER_TILE_T256x320 ? SEER_TILE_AUTO : desc->tile;            // prepare() may have picked a split tile; unsplit launches choose their own below
    if (!d.colsum && !d.colsum_fx && !rows_ln && seer_gemm_ws_eligible(d) &&
        (requested == SEER_TILE_WS || (requested == SEER_TILE_AUTO && seer_gemm_ws_profitable(d)))) {
        const int rc_ws = seer_gemm_ws_launch(d, st);
        if (rc_ws != SEER_ENOSYS) return rc_ws;
    }
    if (requested == SEER_TILE_WS || requested == SEER_TILE_AUTO_TILED) d.tile = SEER_TILE_AUTO;   // the tile kernel picks its own

    const int tile = resolve_tile(d);
    if (d.epilogue & SEER_EPI_F16)
        return dispatch_tile(tile, [&](auto t) {
            using T = decltype(t);
            return launch_tile<T::BM, T::BN, T::NS, T::WM, T::WN, true>(d, st);
        });
    return dispatch_tile(tile, [&](auto t) {
        using T = decltype(t);
        return launch_tile<T::BM, T::BN, T::NS, T::WM, T::WN>(d, st);
    });
}
