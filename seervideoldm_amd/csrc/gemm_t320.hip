// 256 x 320 x 64 tile GEMM / implicit-GEMM conv for gfx950: one 512-thread workgroup per CU, in-launch split-K reduction.
//
// Why this tile.  A CU fills LDS from L2 / MALL at ~70 GB/s whatever issues the loads (profiles/r02_lab_fill.log: one block
// alone 74 GB/s, 256 blocks 68, 3x3 gathers ~51), so a tile's FLOPs per filled byte bound its rate: 128 x 128 -> 64 FLOP/B ->
// ~1.1 PF, measured 1.07.  Every N of the denoising network is a multiple of 320 (320 * {1, 2, 3, 4, 6, 8, 12, 16, 32}) and
// every M of its three upper levels a multiple of 256, so ONE tile shape, 256 x 320 (142 FLOP/B), covers them without a padded
// column.  8 waves as 4 (M) x 2 (N), 64 x 160 wave tiles = 160 accumulator registers, v_mfma_f32_16x16x32_bf16 issued swapped
// (a lane holds 4 consecutive output columns of one row, as in gemm.hip).
//
// Main loop = the 8-phase ping-pong schedule of gemm.hip's 256 x 256 tile (cdna_hip_programming.md, "256^2 8-phase template")
// re-derived for this wave grid: a K tile is four LDS regions (A rows of quadrant-row 0 / 1 of every wave row: 16 KB each;
// W columns of quadrant-column 0 / 1 of both wave columns: 20 KB each), four phases each multiply one 32 x 80 quadrant of the
// wave tile (20 MFMAs) in the order (r0,c0) (r1,c0) (r1,c1) (r0,c1) so that the 40-register W fragments are read twice and the
// 16-register A fragments three times per K tile; every region is restaged for K tile u + 2 one phase after its last read;
// waves 4..7 run one barrier behind waves 0..3 (each SIMD hosts one wave of either half: one multiplies while the other reads
// LDS and issues LDS-DMA); one counted s_waitcnt vmcnt(8) per K tile, never 0 inside the loop.
//
// Split-K without a second launch (seer_gemm_desc::sync).  Few-tile, long-K problems (every conv below the 32x32 level, the
// feed-forward output projections) need K slices to occupy 256 CUs; the two-launch form pays a reduce kernel that re-reads
// splits x M x N floats.  Here slice s of a tile writes its raw accumulators to workspace in REGISTER order (each store
// instruction = 1 KB contiguous, write-through), arrives on the tile's counter, waits for its S - 1 peers (all resident: the
// grid is at most one round of CUs, peers have adjacent block ids), then reduces ITS share of the accumulator quads over all
// S slabs in slice order (deterministic, bit-identical from launch to launch), runs the epilogue on that share and stores it.
// The hand-off follows MI355X_MICROARCH.md "Valid forms", table row 3: sc1 stores of whole 128-B lines, every storing wave's
// s_waitcnt vmcnt(0), workgroup barrier, one lane's agent-scope atomic add; consumer: one lane polls with sc1 loads, workgroup
// barrier, then sc1 loads only.  The last slice to finish reading resets both counters: `sync` stays zero between launches.
#include "seer_common.h"
#include <mutex>

#ifndef SEER_T320_PROBE
#define SEER_T320_PROBE 0      // measurement builds (results wrong): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 4 = no MFMA
#endif

namespace {

constexpr int BM = 256, BN = 320, BK = 64, NT = 512;
constexpr int A_HALF = 128 * BK;                       // elements (16 KB)
constexpr int B_HALF = 160 * BK;                       // elements (20 KB)
constexpr int BUF = 2 * A_HALF + 2 * B_HALF;           // one K tile: [A r0 | A r1 | W c0 | W c1] = 72 KB
constexpr int LDS_BYTES = 2 * BUF * 2 + 1024;          // two K tiles + 1 KB that absorbs the padding LDS-DMA pieces
#ifdef SEER_T320_STAMPS
constexpr int LDS_ALLOC = LDS_BYTES + 8 * 64 * 8;      // + the stamp area of the measurement build
#else
constexpr int LDS_ALLOC = LDS_BYTES;
#endif
constexpr int NQ = 40;                                 // accumulator quads (f32x4) per lane: 4 row fragments x 10 column fragments
constexpr int SMAX = 16;                               // K slices per tile, at most
constexpr int QMAX = 20;                               // quads of one slice's share (S >= 2)

__device__ __attribute__((aligned(16))) unsigned int seer_t320_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void lds_dma16(const void* src, void* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <bool CONV, bool GEGLU, bool SPLIT>
__global__ void __launch_bounds__(NT) seer_gemm_t320_kernel(const seer_gemm_desc p) {
    static_assert(!(GEGLU && (CONV || SPLIT)), "GEGLU: plain unsplit launches only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* const smem_b = reinterpret_cast<bf16*>(smem);
    bf16* const dummy = smem_b + 2 * BUF;
    // LDS-DMA destinations as LDS-address-space byte offsets from the start (a generic -> LDS cast per piece costs a null check)
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte* const lds3 = (lds_byte*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations and region offsets live in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2;                         // 1: runs one barrier behind
    const int frow = lane & 15, fq = lane >> 4;

    // ---- block -> (tile, slice): slices of a tile have adjacent block ids (dispatched together); tiles in XCD-contiguous
    // chunks grouped along M like gemm.hip (blocks b and b + 8 share an XCD)
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = p.N / BN;
    const int S = SPLIT ? p.splits : 1;
    int tile, slice;
    if constexpr (SPLIT) {
        tile = blockIdx.x / S;
        slice = blockIdx.x - tile * S;
    } else {
        const int nwg = tiles_m * tiles_n;
        const int L = blockIdx.x, xcd = L & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        slice = 0;
    }
    constexpr int GM = 8;
    const int group = tile / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + (tile % (GM * tiles_n)) % gsz;
    const int tn = (tile % (GM * tiles_n)) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = SPLIT ? 0 : blockIdx.z;
    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A) + (int64_t)z * p.strideA;
    const bf16* __restrict__ A2 = reinterpret_cast<const bf16*>(p.A2);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W) + (int64_t)z * p.strideW;

    // ---- staging descriptors.  One LDS-DMA piece = 8 rows x 128 B; lane -> (row lane >> 3, 16-B slot lane & 7), the XOR swizzle
    // goes on the SOURCE chunk.  A region rr: local row lr = 32 wm' + i <-> tile row 64 wm' + 32 rr + i; this wave stages local
    // rows 16 wave + 8 c + (lane >> 3), c = 0, 1.  W region cc: local row lc = 80 wn' + j <-> tile column 160 wn' + 80 cc + j;
    // this wave stages pieces wave, wave + 8, wave + 16 (the last only for wave < 4: 20 pieces per region).
    const int schunk = ((lane & 7) ^ (lane >> 3)) * 8;
    int a_row[4];                                      // [rr * 2 + c]: tile row of this lane's piece row (clamped to M)
    // conv: byte offset of tap (0, 0) of the piece row's pixel, this lane's chunk included (negative above / left of the image),
    // and one validity bit per tap: bit 9 c + tap of a_ok[rr]
    int a_pix[4];
    unsigned a_ok[2] = {0u, 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int lr = 16 * wave + 8 * (c & 1) + (lane >> 3);
        int gm = m0 + (lr >> 5) * 64 + (c >> 1) * 32 + (lr & 31);
        gm = gm < p.M ? gm : p.M - 1;
        a_row[c] = gm - m0;
        if constexpr (CONV) {
            const bool phase_mode = p.upsample == 2;
            const int gw = phase_mode ? p.Win : p.Wout;
            const int hw = phase_mode ? p.Hin * p.Win : p.Hout * p.Wout;
            const int img = gm / hw;
            const int rem = gm - img * hw;
            const int oy = rem / gw, ox = rem - oy * gw;
            const int pad0 = p.pad_after_only ? 0 : 1;
            const int iy0 = phase_mode ? oy + ((int)blockIdx.z >> 1) - 1 : oy * p.stride - pad0;
            const int ix0 = phase_mode ? ox + ((int)blockIdx.z & 1) - 1 : ox * p.stride - pad0;
            a_pix[c] = ((img * p.Hin * p.Win + iy0 * p.Win + ix0) * p.Cin + schunk) * 2;
            const int ksz = phase_mode ? 2 : 3;
            unsigned bits = 0u;
            for (int ky = 0; ky < ksz; ++ky)
                for (int kx = 0; kx < ksz; ++kx)
                    if (iy0 + ky >= 0 && iy0 + ky < p.Hin && ix0 + kx >= 0 && ix0 + kx < p.Win) bits |= 1u << (ky * ksz + kx);
            a_ok[c >> 1] |= bits << (9 * (c & 1));
        }
    }
    unsigned b_rel[3];                                 // [t]: byte offset of this lane's chunk in W region 0 (region 1: + 80 rows of W)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int pc = wave + 8 * t;
        const int lc = 8 * pc + (lane >> 3);
        const int col = (lc / 80) * 160 + (lc % 80);
        b_rel[t] = pc < 20 ? (unsigned)(col * p.K + schunk) * 2u : (unsigned)schunk * 2u;   // (a padding piece re-reads row 0)
    }
    // plain: LDS-DMA as buffer loads -- wave-uniform resource (the tile's first row), per-lane byte offset that is constant over K,
    // scalar offset that advances with K: one s_mov m0 + one buffer_load per piece, no per-piece address arithmetic.  (The loop is
    // bound by instruction ISSUE: in-kernel stamps, profiles/r04_t320_stamps.log -- a phase's 20 MFMAs take 160 issue cycles of
    // their 320, the partner wave's reads + LDS-DMA + address arithmetic have to fit in the other 160.)
    unsigned a_voff[4];                                // (row * lda + chunk) * 2 for the source the K loop is in
#pragma unroll
    for (int c = 0; c < 4; ++c) a_voff[c] = (unsigned)(a_row[c] * p.lda + schunk) * 2u;
    const __amdgpu_buffer_rsrc_t a1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (int64_t)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A2 ? A2 + (int64_t)m0 * p.lda2 : A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(W + (int64_t)n0 * p.K), 0, 0x7fffffff, 0x00020000);
    int a_second = 0;                                  // a_voff holds the offsets of A2 (wave-uniform)
    // conv: the input tensor as a raw buffer (bytes; eligibility keeps it under 2 GB)
    const unsigned a_bytes = CONV ? (unsigned)((int64_t)(p.M / ((p.upsample == 2 ? p.Hin * p.Win : p.Hout * p.Wout))) * p.Hin * p.Win * p.Cin * 2) : 0u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (int)a_bytes, 0x00020000);
    const unsigned cin_magic = CONV ? 0xFFFFFFFFu / (unsigned)p.Cin + 1u : 0u;   // tap = umulhi(kbase, magic): exact for kbase * Cin < 2^32

    const int nk_all = p.K / BK;
    const int kt0 = SPLIT ? (int)((int64_t)nk_all * slice / S) : 0;
    const int T = (SPLIT ? (int)((int64_t)nk_all * (slice + 1) / S) : nk_all) - kt0;

    auto stage_a = [&](int u, int rr) {
        if (u >= T) return;
#if !(SEER_T320_PROBE & 1)
        const int kbase = (kt0 + u) * BK;
        lds_byte* const dst = lds3 + (((u & 1) * BUF + rr * A_HALF + (16 * wave) * BK) * 2);
        if constexpr (CONV) {
            // LDS-DMA through a buffer resource over the input tensor: a lane whose tap falls outside the image asks for the byte
            // just past the tensor and the bounds check returns zeros -- no zero page, no 64-bit address select, no branch
            const int tap = (int)__umulhi((unsigned)kbase, cin_magic);
            const int ci0 = kbase - tap * p.Cin;
            const int ksz = p.upsample == 2 ? 2 : 3;
            const int ky = tap / ksz, kx = tap - ky * ksz;
            const int tap_b = (ky * p.Win + kx) * p.Cin * 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool ok = ((a_ok[rr] >> (9 * c + tap)) & 1u) != 0u;
                const unsigned voff = ok ? (unsigned)(a_pix[rr * 2 + c] + tap_b) : a_bytes;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, dst + 8 * c * BK * 2, 16, voff, ci0 * 2, 0, 0);
            }
        } else {
            const bool second = kbase >= p.K1;
            if ((int)second != __builtin_amdgcn_readfirstlane(a_second)) {                 // (once per tile, at the seam of a two-source K: the skip concat)
                a_second = (int)second;
#pragma unroll
                for (int c = 0; c < 4; ++c) a_voff[c] = (unsigned)(a_row[c] * (second ? p.lda2 : p.lda) + schunk) * 2u;
            }
            const int soff = (second ? kbase - p.K1 : kbase) * 2;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? a2_rsrc : a1_rsrc, dst + 8 * c * BK * 2, 16, a_voff[rr * 2 + c], soff, 0, 0);
        }
#endif
    };
    auto stage_b = [&](int u, int cc) {
        if (u >= T) return;
#if !(SEER_T320_PROBE & 1)
        const int kbase = (kt0 + u) * BK;
        lds_byte* const dst = lds3 + (((u & 1) * BUF + 2 * A_HALF + cc * B_HALF + (8 * wave) * BK) * 2);
        const int soff = (80 * cc * p.K + kbase) * 2;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst + 64 * t * BK * 2, 16, b_rel[t], soff, 0, 0);
        // third piece: waves 4..7 have none -- theirs lands in the spare 1 KB, so that every wave counts 3 per W region
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, wave < 4 ? dst + 128 * BK * 2 : lds3 + 2 * BUF * 2, 16, b_rel[2], soff, 0, 0);
#endif
    };

    bf16x8 fa[2][2], fb[2][5];
    auto read_a = [&](int u, int rr) {
#if !(SEER_T320_PROBE & 2)
        const bf16* as = smem_b + (u & 1) * BUF + rr * A_HALF + (wm * 32) * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(as + (i * 16 + frow) * BK + sw);
        }
#endif
    };
    auto read_b = [&](int u, int cc) {
#if !(SEER_T320_PROBE & 2)
        const bf16* bs = smem_b + (u & 1) * BUF + 2 * A_HALF + cc * B_HALF + (wn * 80) * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = (((ks * 4 + fq) ^ (frow & 7)) * 8);
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(bs + (j * 16 + frow) * BK + sw);
        }
#endif
    };
    auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

#ifdef SEER_T320_STAMPS
    // measurement build (scripts/run_r04_t320_stamps.sh): s_memtime (shader cycles) when the wave ARRIVES at each of the two
    // barriers of a phase (reads landed + LDS-DMA issued | MFMAs issued), K tiles 4..10, lane 0 of every wave of blocks 0..15, parked in the spare LDS behind the K-loop buffers and copied to desc.workspace
    // [block][wave][64] at the end (slot 63: hardware id).  Unsplit launches only.
    long long* const st_lds = reinterpret_cast<long long*>(smem + LDS_BYTES) + wave * 64;
    int st_n = 0;
    const bool st_on = blockIdx.x < 16 && lane == 0 && p.workspace != nullptr;
#define TSTAMP(u_)                                                                                     \
    do {                                                                                               \
        if (st_on && (u_) >= 4 && (u_) < 11 && st_n < 62) st_lds[st_n++] = (long long)__builtin_readcyclecounter();              \
    } while (0)
#else
#define TSTAMP(u_) do { } while (0)
#endif
    f32x4 acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#if SEER_T320_PROBE & 4
#define SEER_T320_MFMA(R, C)                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(fa[ks][i]));                                 \
            _Pragma("unroll") for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(fb[ks][j]));                                 \
        }
#else
#define SEER_T320_MFMA(R, C)                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                    \
                _Pragma("unroll") for (int j = 0; j < 5; ++j)                                                                \
                    acc[2 * (R) + i][5 * (C) + j] =                                                                          \
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[2 * (R) + i][5 * (C) + j], 0, 0, 0);
#endif
#ifndef SEER_T320_PRIO
#define SEER_T320_PRIO 1       // wave priority while it multiplies: 1 = raised (the template's form), 0 = left alone, -1 = raised while it LOADS instead
#endif
#define SEER_T320_MMA(R, C)                                                                                                  \
        do {                                                                                                                 \
            TSTAMP(u);                                                                                                       \
            if (SEER_T320_PRIO < 0) __builtin_amdgcn_s_setprio(0);                                                           \
            barrier();                                                                                                       \
            if (SEER_T320_PRIO > 0) __builtin_amdgcn_s_setprio(1);                                                           \
            SEER_T320_MFMA(R, C)                                                                                             \
            if (SEER_T320_PRIO > 0) __builtin_amdgcn_s_setprio(0);                                                           \
            TSTAMP(u);                                                                                                       \
            barrier();                                                                                                       \
            if (SEER_T320_PRIO < 0) __builtin_amdgcn_s_setprio(1);                                                           \
        } while (0)

    // prologue: K tile 0 whole (10 pieces per wave), K tile 1 except its A r0 (8 pieces; A r0 goes out in phase 0 of tile 0)
    stage_b(0, 0); stage_a(0, 0); stage_a(0, 1); stage_b(0, 1);
    stage_b(1, 0); stage_a(1, 1); stage_b(1, 1);
    if (T > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
#ifndef SEER_T320_NOSTAGGER
    if (grp == 1) barrier();                           // waves 4..7 run one barrier behind from here on
#endif
    for (int u = 0; u < T; ++u) {
        // phase 0: quadrant (r0, c0)
        read_b(u, 0);
        read_a(u, 0);
        stage_a(u + 1, 0);                             // its buffer's A r0 was last read in phase 3 of tile u - 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEER_T320_MMA(0, 0);
        // phase 1: (r1, c0); W c0 stays in registers
        read_a(u, 1);
        stage_b(u + 2, 0);                             // W c0 of this buffer: last read in phase 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEER_T320_MMA(1, 0);
        // phase 2: (r1, c1); A r1 stays
        read_b(u, 1);
        stage_a(u + 2, 1);                             // A r1: last read in phase 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEER_T320_MMA(1, 1);
        // phase 3: (r0, c1); W c1 stays.  The wait retires K tile u + 1 (everything but the 8 pieces of tile u + 2 issued in
        // phases 1..3); its first read is in phase 0 of the next iteration, two barriers later.
        read_a(u, 0);
        stage_b(u + 2, 1);                             // W c1: last read in phase 2
        if (u + 2 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEER_T320_MMA(0, 1);
    }
#undef SEER_T320_MMA
#undef SEER_T320_MFMA
#ifndef SEER_T320_NOSTAGGER
    if (grp == 0) barrier();                           // re-align the two halves: every wave is done with the K-loop LDS
#else
    barrier();
#endif

    // =========================================================================================================================
    // epilogue.  acc[i][j][r] = C[m0 + 64 wm + 16 i + frow][n0 + 160 wn + 16 j + 4 fq + r]
    // =========================================================================================================================
    bf16* Cb = reinterpret_cast<bf16*>(p.C) + (int64_t)z * p.strideC;
    const bf16* R = reinterpret_cast<const bf16*>(p.residual);
    const bool do_rot = (p.epilogue & SEER_EPI_ROTARY) != 0;
    const bool do_cs = (p.epilogue & SEER_EPI_COLSCALE) != 0;
    auto crow = [&](int m) -> int64_t {               // output row of GEMM row m (the phase convs scatter their rows)
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
    constexpr int BNO = GEGLU ? BN / 2 : BN;            // output columns of the tile
    constexpr int CPITCH = BNO * 2 + 16;                // staged row pitch (bytes)
    constexpr int PASS_ROWS = GEGLU ? 256 : 128;        // rows staged per pass (128 x 656 B = 82 KB; GEGLU: 256 x 336 B = 84 KB)
    constexpr int NPASS = BM / PASS_ROWS;
    constexpr int CPR = BNO / 8;                        // 16-byte chunks per staged row

    const int n0o = GEGLU ? (n0 >> 1) : n0;             // first output column of the tile
    // the quads this block finishes: all of them, or (split) its share [q0, q1) of q = 10 i + j, reduced over the S slabs
    int q0 = 0, q1 = NQ;
    f32x4 mine[SPLIT ? QMAX : 1];
    int& timed_out_s = *reinterpret_cast<int*>(dummy);     // (inside the ONE LDS array: a second __shared__ object can cost the K loop its counted waits)
    if constexpr (SPLIT) {
        q0 = NQ * slice / S;
        q1 = NQ * (slice + 1) / S;
        // ---- publish: raw accumulators in register order, [tile][slice][wave][q][lane] x 16 B, write-through
        float* slabs = reinterpret_cast<float*>(p.workspace) + (int64_t)tile * S * (8 * NQ * 64 * 4);
        {
            float* my = slabs + (int64_t)slice * (8 * NQ * 64 * 4) + (wave * NQ * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 10; ++j) store16_out(my + (i * 10 + j) * 256, __builtin_bit_cast(u32x4, acc[i][j]));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int* cnt = reinterpret_cast<unsigned int*>(p.sync) + 2 * tile;
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S && ++spins < (1 << 24))
                __builtin_amdgcn_s_sleep(4);
            // The wait assumes the S slices of a tile become resident together: true for one process per GPU (block ids of a tile are
            // adjacent, dispatch is in order).  Two processes that BOTH run spin-waiting kernels on one GPU can hold each other's
            // CUs; the bound turns that into a loud failure (a NaN share) instead of a hang -- AUTO therefore never splits K here.
            timed_out_s = spins >= (1 << 24);
        }
        __syncthreads();
        const bool timed_out = timed_out_s != 0;
        // ---- reduce my share in slice order: sc1 loads only (the lines were written through by other CUs)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, S * (8 * NQ * 64 * 16), 0x00020000);
        const int nq = q1 - q0;
#pragma unroll
        for (int k = 0; k < QMAX; ++k) {
            if (k < nq) {
                const int off = ((wave * NQ + q0 + k) * 64 + lane) * 16;
                u32x4 v[SMAX];
#pragma unroll
                for (int sp = 0; sp < SMAX; ++sp)
                    if (sp < S) v[sp] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, sp * (8 * NQ * 64 * 16), 16);
                f32x4 sum = __builtin_bit_cast(f32x4, v[0]);
#pragma unroll
                for (int sp = 1; sp < SMAX; ++sp)
                    if (sp < S) sum += __builtin_bit_cast(f32x4, v[sp]);
                mine[k] = timed_out ? f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")} : sum;
            }
        }
        __syncthreads();                                // every wave holds its sums: this block no longer reads the slabs
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)S - 1u) {              // last reader of the tile: leave the counters as the next launch expects them
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- epilogue terms on one quad: v = the four values of row m (tile-relative row mt), GEMM columns n .. n + 3
    auto finish_quad = [&](f32x4 v, int m, int n) -> f32x4 {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.rowvec) {
            const int mc = m < p.M ? m : p.M - 1;
            v += *reinterpret_cast<const f32x4*>(p.rowvec + (int64_t)(mc / p.rows_per_batch) * p.rowvec_ld + n);
        }
        if (do_rot && n < p.rot_cols) {
            const int ch = n % p.rot_head_dim;
            if (ch < p.rot_dim) {
                const int pos = m % p.rot_tokens_per_batch + p.rot_pos_offset;
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rot_table + ((int64_t)pos * (p.rot_dim / 2) + ch / 2) * 2);
                const f32x2 r01 = rot_pair(f32x2{v[0], v[1]}, cs[0], cs[1]);
                const f32x2 r23 = rot_pair(f32x2{v[2], v[3]}, cs[2], cs[3]);
                v = f32x4{r01[0], r01[1], r23[0], r23[1]};
            }
        }
        if (do_cs && n < p.col_scale_cols) v *= p.col_scale;
        if (R) {
            const int mc = m < p.M ? m : p.M - 1;
            const u32x2 rv = *reinterpret_cast<const u32x2*>(R + (int64_t)mc * p.ldr + n);
            v[0] += __builtin_bit_cast(float, rv[0] << 16);
            v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
            v[2] += __builtin_bit_cast(float, rv[1] << 16);
            v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
        }
        return v;
    };

    // unsplit launches add the residual from LDS: the rows of a pass are fetched 16 B per lane along whole rows (the accumulator
    // layout would fetch 8 B per lane in 16 different rows per instruction) into the slots the output will be staged in; a lane
    // reads its quad's 8 B back, adds in fp32, rounds once and overwrites the slot.  The fetch of pass 0 is issued behind the other
    // epilogue terms (ahead of them its 40 registers spill next to the 160 accumulators), the fetch of pass 1 behind the barrier
    // that publishes pass 0.
    u32x4 rres[SPLIT ? 1 : (PASS_ROWS * CPR) / NT];
    auto fetch_residual = [&](int ps) {
        if constexpr (!SPLIT) {
#pragma unroll
            for (int it = 0; it < (PASS_ROWS * CPR) / NT; ++it) {
                const int c = tid + it * NT;
                const int row = c / CPR, ch = c - row * CPR;
                const int m = m0 + ps * PASS_ROWS + row;
                rres[it] = m < p.M ? *reinterpret_cast<const u32x4*>(R + (int64_t)m * p.ldr + n0o + ch * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    const bool res_lds = !SPLIT && R != nullptr && (p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(R) & 15) == 0);

    if constexpr (!SPLIT) {
        // unsplit: one pass per term over the register tile (the order of additions of gemm.hip: bias, GEGLU, row vector, rotary,
        // column scale, residual)
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 160 + j * 16 + fq * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] += bv;
            }
        }
        if constexpr (GEGLU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 10; j += 2) {
                    const f32x4 g = acc[i][j + 1];
                    const f32x2 ge0 = gelu_erf_f2(f32x2{g[0], g[1]}), ge1 = gelu_erf_f2(f32x2{g[2], g[3]});
                    acc[i][j][0] *= ge0[0]; acc[i][j][1] *= ge0[1]; acc[i][j][2] *= ge1[0]; acc[i][j][3] *= ge1[1];
                }
                __builtin_amdgcn_sched_barrier(0);      // one fragment row of erf temporaries at a time
            }
        }
        if (p.rowvec) {
            // GEGLU: the row vector is indexed by OUTPUT column
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = m0 + wm * 64 + i * 16 + frow;
                m = m < p.M ? m : p.M - 1;
                const float* rv = p.rowvec + (int64_t)(m / p.rows_per_batch) * p.rowvec_ld;
#pragma unroll
                for (int j = 0; j < 10; j += (GEGLU ? 2 : 1)) {
                    const int n = n0 + wn * 160 + j * 16 + fq * 4;
                    const int nc = GEGLU ? ((n - fq * 4) >> 1) + fq * 4 : n;
                    acc[i][j] += *reinterpret_cast<const f32x4*>(rv + nc);
                }
            }
        }
        if constexpr (!GEGLU) {
            if (do_rot && n0 + wn * 160 < p.rot_cols) {
                int tj[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const int n = n0 + wn * 160 + j * 16 + fq * 4;
                    const int ch = n % p.rot_head_dim;
                    tj[j] = (n < p.rot_cols && ch < p.rot_dim) ? (ch / 2) * 2 : -1;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + wm * 64 + i * 16 + frow;
                    const int pos = m % p.rot_tokens_per_batch + p.rot_pos_offset;
                    const float* trow = p.rot_table + (int64_t)pos * (p.rot_dim / 2) * 2;
#pragma unroll
                    for (int j = 0; j < 10; ++j) {
                        if (tj[j] >= 0) {
                            const f32x4 cs = *reinterpret_cast<const f32x4*>(trow + tj[j]);
                            const f32x2 r01 = rot_pair(f32x2{acc[i][j][0], acc[i][j][1]}, cs[0], cs[1]);
                            const f32x2 r23 = rot_pair(f32x2{acc[i][j][2], acc[i][j][3]}, cs[2], cs[3]);
                            acc[i][j] = f32x4{r01[0], r01[1], r23[0], r23[1]};
                        }
                    }
                }
            }
            if (do_cs) {
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const float sc = (n0 + wn * 160 + j * 16 + fq * 4) < p.col_scale_cols ? p.col_scale : 1.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] *= sc;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < QMAX; ++k) {
            if (k < q1 - q0) {
                const int q = q0 + k, i = q / 10, j = q - i * 10;
                mine[k] = finish_quad(mine[k], m0 + wm * 64 + i * 16 + frow, n0 + wn * 160 + j * 16 + fq * 4);
            }
        }
    }

    // ---- bf16 rows leave through the (now idle) K-loop LDS: 16 B per lane along whole output rows.  Column sums of the tile
    // as stored, per 64-row partial (row fragment i of the four wave rows): colsum[z][4 tile_m + i][N][2].
    const bool wt_store = p.K <= 3072;
    if (res_lds) fetch_residual(0);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (ps > 0) __syncthreads();                    // the previous pass has been copied out
        if constexpr (!SPLIT) {
            if (res_lds) {
#pragma unroll
                for (int it = 0; it < (PASS_ROWS * CPR) / NT; ++it) {
                    const int c = tid + it * NT;
                    const int row = c / CPR, ch = c - row * CPR;
                    *reinterpret_cast<u32x4*>(smem + row * CPITCH + ch * 16) = rres[it];
                }
                __syncthreads();
                if (ps + 1 < NPASS) fetch_residual(ps + 1);
            }
            if (GEGLU || (wm >> 1) == ps) {
                const int rowb = ((GEGLU ? wm * 64 : (wm & 1) * 64) + frow) * CPITCH;
#pragma unroll
                for (int j = 0; j < 10; j += (GEGLU ? 2 : 1)) {
                    const int cb = GEGLU ? (wn * 80 + (j >> 1) * 16 + fq * 4) * 2 : (wn * 160 + j * 16 + fq * 4) * 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        u32x2* slot = reinterpret_cast<u32x2*>(smem + rowb + i * 16 * CPITCH + cb);
                        f32x4 v = acc[i][j];
                        if (res_lds) {
                            const u32x2 rv = *slot;
                            v[0] += __builtin_bit_cast(float, rv[0] << 16);
                            v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, rv[1] << 16);
                            v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
                        } else if (R) {
                            int m = m0 + wm * 64 + i * 16 + frow;
                            m = m < p.M ? m : p.M - 1;
                            const int n = n0 + wn * 160 + j * 16 + fq * 4;
                            const u32x2 rv = *reinterpret_cast<const u32x2*>(R + (int64_t)m * p.ldr + (GEGLU ? ((n - fq * 4) >> 1) + fq * 4 : n));
                            v[0] += __builtin_bit_cast(float, rv[0] << 16);
                            v[1] += __builtin_bit_cast(float, rv[0] & 0xffff0000u);
                            v[2] += __builtin_bit_cast(float, rv[1] << 16);
                            v[3] += __builtin_bit_cast(float, rv[1] & 0xffff0000u);
                        }
                        u32x2 o;
                        o[0] = pack2(v[0], v[1]);
                        o[1] = pack2(v[2], v[3]);
                        *slot = o;
                    }
                }
            }
        } else {
            if ((wm >> 1) == ps) {
#pragma unroll
                for (int k = 0; k < QMAX; ++k) {
                    if (k < q1 - q0) {
                        const int q = q0 + k, i = q / 10, j = q - i * 10;
                        u32x2 o;
                        o[0] = pack2(mine[k][0], mine[k][1]);
                        o[1] = pack2(mine[k][2], mine[k][3]);
                        *reinterpret_cast<u32x2*>(smem + ((wm & 1) * 64 + i * 16 + frow) * CPITCH + (wn * 160 + j * 16 + fq * 4) * 2) = o;
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (!GEGLU) {
            // column sums of the pass as stored (bf16-rounded), per CONTIGUOUS row block: unsplit, one partial per wave row (64
            // rows: colsum[z][4 tile_m + wave row][N][2]); split, one per 16-row fragment of a wave row (a slice finishes whole
            // quads: colsum[16 tile_m + 4 wave row + fragment][N][2], every entry written by exactly one slice).  A thread owns a
            // column pair of one block and adds its rows in order: deterministic, no atomics.
            if (p.colsum || p.colsum_fx) {
                constexpr int NB = SPLIT ? 8 : 2;             // row blocks in a pass
                constexpr int RB = SPLIT ? 16 : 64;           // rows per block
                for (int e = tid; e < 160 * NB; e += NT) {
                    const int cp = e % 160, blk = e / 160;
                    if constexpr (SPLIT) {
                        const int qq = (blk & 3) * 10 + (cp % 80) / 8;
                        if (qq < q0 || qq >= q1) continue;
                    }
                    const int mrow = m0 + ps * 128 + blk * RB;
                    float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll 4
                    for (int r = 0; r < RB; ++r) {
                        if (mrow + r < p.M) {
                            const uint32_t v = *reinterpret_cast<const uint32_t*>(smem + (blk * RB + r) * CPITCH + cp * 4);
                            const float f0 = __builtin_bit_cast(float, v << 16), f1 = __builtin_bit_cast(float, v & 0xffff0000u);
                            s0 += f0; t0 += f0 * f0;
                            s1 += f1; t1 += f1 * f1;
                        }
                    }
                    if (p.colsum_fx) {               // accumulated per batch element in fixed point (seer_gemm_desc::colsum_fx)
                        int64_t* o = fx_slot(p, mrow / RB, mrow) + n0 + cp * 2;
                        fx_add(o, s0); fx_add(o + 1, s1); fx_add(o + p.N, t0); fx_add(o + p.N + 1, t1);
                    } else {
                        const int64_t part = ((int64_t)(SPLIT ? 0 : blockIdx.z) * tiles_m + tm) * (2 * NB) + ps * NB + blk;
                        *reinterpret_cast<f32x4*>(p.colsum + (part * p.N + n0 + cp * 2) * 2) = f32x4{s0, t0, s1, t1};
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < (PASS_ROWS * CPR + NT - 1) / NT; ++it) {
            const int c = tid + it * NT;
            const int row = c / CPR, ch = c - row * CPR;
            const int trow = ps * PASS_ROWS + row;
            const int m = m0 + trow;
            bool on = ((PASS_ROWS * CPR) % NT == 0 || c < PASS_ROWS * CPR) && m < p.M;
            if constexpr (SPLIT) {
                const int qq = ((trow >> 4) & 3) * 10 + (ch % 20) / 2;      // quad of this chunk: fragment row, fragment column
                on = on && qq >= q0 && qq < q1;
            }
            if (on) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CPITCH + ch * 16);
                bf16* dst = Cb + crow(m) * p.ldc + n0o + ch * 8;
                if (wt_store) store16_out(dst, v);
                else *reinterpret_cast<u32x4*>(dst) = v;
            }
        }
    }
#ifdef SEER_T320_STAMPS
    if (st_on) {
        long long* o = reinterpret_cast<long long*>(p.workspace) + ((int64_t)blockIdx.x * 8 + wave) * 64;
        for (int i = 0; i < 62; ++i) o[i] = i < st_n ? st_lds[i] : 0;
        o[62] = 0;
        o[63] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    }
#endif
}

std::once_flag g_t320_once;

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
// what the kernel handles (gemm.hip decides WHEN to use it)
bool seer_gemm_t320_eligible(const seer_gemm_desc& d) {
    if (d.N % 320 || d.K % 64 || d.M < 1) return false;
    if (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_TRANS_OUT | SEER_EPI_SILU | SEER_EPI_F16)) return false;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (geglu && (d.mode == SEER_GEMM_CONV3X3 || (d.epilogue & (SEER_EPI_ROTARY | SEER_EPI_COLSCALE)))) return false;
    if (d.ldc % 8 || (reinterpret_cast<uintptr_t>(d.C) & 15)) return false;
    if (d.batch > 1 && (d.strideC % 8)) return false;
    if (d.residual && (d.ldr % 4)) return false;
    // 32-bit element offsets in the staging path
    if (d.mode == SEER_GEMM_CONV3X3) {
        if (d.upsample == 1) return false;             // the nearest-2x read-through form stays with the smaller tiles
        if ((int64_t)d.M * d.Cin >= (1ll << 30) || (int64_t)d.K * d.Cin >= (1ll << 32)) return false;
        const int64_t hw_out = d.upsample == 2 ? (int64_t)d.Hin * d.Win : (int64_t)d.Hout * d.Wout;
        if (hw_out <= 0 || d.M % hw_out) return false;
        if ((d.M / hw_out) * d.Hin * d.Win * d.Cin * 2 >= (1ll << 31)) return false;     // the input tensor as one raw buffer
        if (d.Hin > 16000 || d.Win > 16000) return false;
    } else {
        if ((int64_t)256 * (d.lda > d.lda2 ? d.lda : d.lda2) >= (1ll << 30)) return false;
    }
    if ((int64_t)320 * d.K >= (1ll << 30)) return false;
    return true;
}

int64_t seer_gemm_t320_workspace_bytes(const seer_gemm_desc& d, int splits) {
    if (splits <= 1) return 0;
    const int64_t tiles = (int64_t)((d.M + 255) / 256) * (d.N / 320);
    return tiles * splits * (int64_t)(8 * NQ * 64 * 16);
}
int64_t seer_gemm_t320_sync_bytes(const seer_gemm_desc& d, int splits) {
    if (splits <= 1) return 0;
    const int64_t tiles = (int64_t)((d.M + 255) / 256) * (d.N / 320);
    return tiles * 2 * (int64_t)sizeof(unsigned int);
}

int seer_gemm_t320_launch(const seer_gemm_desc& d0, int splits, hipStream_t st) {
    seer_gemm_desc d = d0;
    std::call_once(g_t320_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_t320_kernel<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_t320_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_t320_kernel<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_t320_kernel<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_t320_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC);
    });
    const bool conv = d.mode == SEER_GEMM_CONV3X3;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    const int tiles = ((d.M + 255) / 256) * (d.N / 320);
    const int batch = d.batch > 1 ? d.batch : 1;
    if (splits > 1) {
        // the slices of a tile WAIT for each other inside the launch: all of them must be resident at once (one workgroup per CU at
        // this LDS size).  A request that does not fit runs with the largest split that does (same result up to the order of the
        // fp32 additions) instead of relying on dispatch order and the bounded wait.
        static const int n_cu = [] {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
                return v;
            return 256;
        }();
        if ((int64_t)tiles * splits > n_cu) splits = tiles >= n_cu ? 1 : n_cu / tiles;
    }
    if (splits > 1) {
        if (geglu || batch > 1 || splits > SMAX || splits > d.K / 64) return SEER_EINVAL;
        if (!d.workspace || d.workspace_bytes < seer_gemm_t320_workspace_bytes(d, splits)) return SEER_EINVAL;
        if (!d.sync || d.sync_bytes < seer_gemm_t320_sync_bytes(d, splits)) return SEER_EINVAL;
        d.splits = splits;
        dim3 grid(tiles * splits, 1, 1);
        if (conv) hipLaunchKernelGGL((seer_gemm_t320_kernel<true, false, true>), grid, dim3(NT), LDS_ALLOC, st, d);
        else hipLaunchKernelGGL((seer_gemm_t320_kernel<false, false, true>), grid, dim3(NT), LDS_ALLOC, st, d);
    } else {
        d.splits = 1;
        dim3 grid(tiles, 1, batch);
        if (conv) hipLaunchKernelGGL((seer_gemm_t320_kernel<true, false, false>), grid, dim3(NT), LDS_ALLOC, st, d);
        else if (geglu) hipLaunchKernelGGL((seer_gemm_t320_kernel<false, true, false>), grid, dim3(NT), LDS_ALLOC, st, d);
        else hipLaunchKernelGGL((seer_gemm_t320_kernel<false, false, false>), grid, dim3(NT), LDS_ALLOC, st, d);
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
