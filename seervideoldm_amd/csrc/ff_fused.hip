// The feed-forward of a transformer block at the 320-channel level of the Seer UNet as ONE launch on gfx950 (MI355X):
//     y = x + [Wp | Wp W2] [h | g] + (Wp b2 + bp),      g = GEGLU(LayerNorm(h) W1^T + b1)
// i.e. norm3 -> ff.net.0 (GEGLU) -> ff.net.2 + residual -> proj_out + residual of BasicTransformerBlock / SpatialTransformer3D
// (seer/models/attention.py:231-248, 308-327, 742-747, 783-793, 126, 141-145).  The launches it replaces: layernorm, the
// weight-stationary GEGLU projection, and the two-source [proj_out | proj_out ff.net.2] GEMM -- 98 us at 24 576 rows, the 4 x 1280
// intermediate (63 MB written and read back per block) and the normalised copy of the residual stream (31 MB) never reach memory.
//
// Shape of the kernel.  Rows of these operators are independent, so a workgroup OWNS 96 rows (24 576 rows = 256 workgroups = one
// per CU, one round) and keeps them in LDS for the whole launch:
//   * T: the 96 x 320 tile of h as five K panels [96 rows][128 B] (16-byte chunks XOR-swizzled by row & 7, written by LDS-DMA with
//     the swizzle on the source side); the proj_out part Y += h Wp^T runs on it first, then it is LayerNormed IN PLACE (a wave
//     owns 24 rows; two-pass statistics in registers, as seer_layernorm), and every later step reads LN(h) from it;
//   * the weights STREAM: wave w (one per SIMD, 4 per workgroup) owns 80 output columns of Y and, per 64-wide chunk of the inner
//     dimension, 32 rows of W1 (16 value + 16 gate columns: the packed GEGLU row order interleaves them in 16s) -- so every wave
//     streams only ITS rows, through a wave-private LDS ring (3 x 4 KB for the K steps of W1, 10 KB for the chunk's 80 x 64 slice of
//     [Wp | Wp W2]): no barrier orders weight traffic, each wave counts its own vmcnt;
//   * per chunk: H = LN(h) W1c^T (12 MFMAs per 32-wide k step and wave), GEGLU in registers, g (96 x 64 bf16) through a 12 KB LDS
//     panel, Y += g Wc^T (30 MFMAs per k step); two barriers per chunk, both about g;
//   * epilogue: x arrives in T by LDS-DMA under the last chunk, y leaves T as whole rows; the tile's column sums (sum, sum of
//     squares of the stored bf16 values) for the GroupNorm that consumes the block's output.
// LDS: 61 440 (T) + 12 288 (g) + 4 x 22 528 (rings) = 163 840 bytes = all of it.  Registers: 120 (Y) + 48 (H) accumulators per lane.
// MFMAs are issued "swapped" (the weight fragment is the A operand), so a lane holds 4 consecutive output columns of one row.
#include "seer_common.h"
#include <mutex>

// measurement builds (scripts/lab_ff_probe.py): bit mask of what to LEAVE OUT of the chunk loop -- wrong results, timing only
//   1 the W1 stream, 2 the [Wp | Wp W2] stream, 4 the MFMAs, 8 the GELU, 16 the fragment reads
#ifndef FF_PROBE
#define FF_PROBE 0
#endif

#ifdef FF_STAMPS    // measurement build: s_memrealtime (10 ns) stamps of the first and the last workgroup, per wave
__device__ long long seer_ff_stamps[2 * 4 * 256];
#define FF_STAMP()                                                                                                        \
    do {                                                                                                                  \
        if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0 && nst < 256)                                   \
            seer_ff_stamps[((blockIdx.x ? 1 : 0) * 4 + wave) * 256 + nst] = wall_clock64();                              \
        ++nst;                                                                                                            \
    } while (0)
extern "C" int seer_lab_ff_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(seer_ff_stamps), sizeof(long long) * 2 * 4 * 256);
}
#else
#define FF_STAMP() do {} while (0)
#endif

namespace {

constexpr int FF_C = 320;                  // channels of the level
constexpr int FF_INNER = 1280;             // GEGLU width (4 C)
constexpr int FF_BM = 96;                  // rows per workgroup
constexpr int FF_NCHUNK = FF_INNER / 64;   // 20
constexpr int FF_KS = FF_C / 64;           // 5 K steps of 64 over the channels
constexpr int FF_PANEL = FF_BM * 128;      // 12 288: one [96 rows][64 k] panel
constexpr int FF_T_BYTES = FF_KS * FF_PANEL;
constexpr int FF_W1_SLOT = 32 * 128;       // 4 096: 32 rows of W1 x one K step
constexpr int FF_W1_NSLOT = 3;
constexpr int FF_WC_BYTES = 80 * 128;      // 10 240: 80 rows of [Wp | Wp W2] x one K step
constexpr int FF_WAVE_BYTES = FF_W1_NSLOT * FF_W1_SLOT + FF_WC_BYTES;      // 22 528
constexpr int FF_LDS = FF_T_BYTES + FF_PANEL + 4 * FF_WAVE_BYTES;          // 163 840
static_assert(FF_LDS == 160 * 1024, "the kernel is laid out for the whole LDS of a CU");

struct FfArgs {
    const bf16* h; const bf16* x; bf16* y;
    int ldh, ldx, ldy, M;
    const float* gamma; const float* beta; float eps;
    const bf16* w1; const float* b1;        // [2 * FF_INNER][FF_C] in the packed GEGLU row order, its bias in the same order
    const bf16* wcat; const float* bcat;    // [FF_C][FF_C + FF_INNER] = [Wp | Wp W2], Wp b2 + bp
    int64_t* colsum_fx;                     // [reps][M / fx_rows][2][FF_C] fixed-point column sums (seer_gemm_desc::colsum_fx) or NULL
    int fx_rows, fx_reps;                   // rows per batch element (a multiple of 96), replicas
};

__device__ __forceinline__ unsigned lds_u32(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}
// Every wave issues its vector-memory operations in ONE fixed order, so every wait below is a compile-time count of the operations
// that were issued AFTER the one waited for (vmcnt retires in order).
template <int N> struct IntTag { static constexpr int value = N; };
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that leaves the vector-memory counter alone (__syncthreads() drains it: hipcc orders LDS-DMA before barriers)
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// fragment reads in inline asm (hipcc would order a C++ read of a ring slot behind ALL pending LDS-DMA); loads and their wait in
// ONE statement with early-clobber outputs: nothing can touch a destination before the data has landed
__device__ __forceinline__ void lds_read6(unsigned a, unsigned off_step, bf16x8 (&f)[6]) {
    u32x4 r0, r1, r2, r3, r4, r5;
    // rows frow, 16 + frow, ... 80 + frow of a panel: 16 rows = 2048 B apart
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:2048\n\tds_read_b128 %2, %6 offset:4096\n\t"
                 "ds_read_b128 %3, %6 offset:6144\n\tds_read_b128 %4, %6 offset:8192\n\tds_read_b128 %5, %6 offset:10240\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5) : "v"(a + off_step) : "memory");
    f[0] = __builtin_bit_cast(bf16x8, r0); f[1] = __builtin_bit_cast(bf16x8, r1); f[2] = __builtin_bit_cast(bf16x8, r2);
    f[3] = __builtin_bit_cast(bf16x8, r3); f[4] = __builtin_bit_cast(bf16x8, r4); f[5] = __builtin_bit_cast(bf16x8, r5);
}
__device__ __forceinline__ void lds_read2(unsigned a, bf16x8 (&f)[2]) {
    u32x4 r0, r1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1) : "v"(a) : "memory");
    f[0] = __builtin_bit_cast(bf16x8, r0); f[1] = __builtin_bit_cast(bf16x8, r1);
}
__device__ __forceinline__ void lds_read5(unsigned a, bf16x8 (&f)[5]) {
    u32x4 r0, r1, r2, r3, r4;
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:2048\n\tds_read_b128 %2, %5 offset:4096\n\t"
                 "ds_read_b128 %3, %5 offset:6144\n\tds_read_b128 %4, %5 offset:8192\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a) : "memory");
    f[0] = __builtin_bit_cast(bf16x8, r0); f[1] = __builtin_bit_cast(bf16x8, r1); f[2] = __builtin_bit_cast(bf16x8, r2);
    f[3] = __builtin_bit_cast(bf16x8, r3); f[4] = __builtin_bit_cast(bf16x8, r4);
}

__global__ void __launch_bounds__(256, 1) seer_ff_fused_c320_kernel(const FfArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const T = smem;
    unsigned char* const G = smem + FF_T_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const ring = smem + FF_T_BYTES + FF_PANEL + wave * FF_WAVE_BYTES;      // this wave's weight ring
    unsigned char* const wc = ring + FF_W1_NSLOT * FF_W1_SLOT;                             // ... and its [Wp | Wp W2] slice buffer
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * FF_BM;
    [[maybe_unused]] int nst = 0;
    FF_STAMP();                             // 0

    // ---- LDS-DMA helpers.  A piece = 8 rows x 128 B; lane l -> row l >> 3, LDS position l & 7 holds source chunk (l & 7) ^ (row & 7)
    const int drow = lane >> 3;
    const int dchunk = ((lane & 7) ^ (drow & 7)) * 8;       // element offset of the source chunk inside the 64-wide K step
    auto dma = [&](const bf16* src_row0, int ld, int k0, unsigned char* dst) {      // rows src_row0 + [0, 8), K step at k0
        const bf16* src = src_row0 + (int64_t)drow * ld + k0 + dchunk;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // the tile of `src` (h, later x): 60 pieces, 15 per wave
    auto load_tile = [&](const bf16* src, int ld) {
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
            dma(src + (int64_t)(m0 + rg * 8) * ld, ld, pnl * 64, T + pnl * FF_PANEL + rg * 1024);
        }
    };
    // 32 rows of W1 (this wave's 16 value + 16 gate columns of chunk c) x K step ks -> ring slot: 4 operations
    auto load_w1 = [&](int c, int ks, int slot) {
        const bf16* w = p.w1 + (int64_t)(128 * c + 32 * wave) * FF_C;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma(w + (int64_t)(8 * i) * FF_C, FF_C, ks * 64, ring + slot * FF_W1_SLOT + i * 1024);
    };
    // 80 rows of [Wp | Wp W2] (this wave's output columns) x K step s (0..4: the Wp part, 5 + c: chunk c) -> dst: 10 operations
    auto load_wc = [&](int s, unsigned char* dst) {
        const bf16* w = p.wcat + (int64_t)(80 * wave) * (FF_C + FF_INNER);
#pragma unroll
        for (int i = 0; i < 10; ++i) dma(w + (int64_t)(8 * i) * (FF_C + FF_INNER), FF_C + FF_INNER, s * 64, dst + i * 1024);
    };

    // ---- accumulators
    f32x4 Y[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS offsets: fragment of row frow (+ 16 i), logical 16-byte chunk ch of a 64-wide K step: position ch ^ (row & 7)
    const unsigned swz = (unsigned)(frow & 7);
    auto frag_off = [&](int k32) { return (unsigned)(frow * 128 + (((k32 * 4 + fq) ^ swz) * 16)); };      // rows 16 i + frow: same swizzle
    const unsigned T0 = lds_u32(T), G0 = lds_u32(G), R0 = lds_u32(ring), WC0 = lds_u32(wc);

    // Y += act(panel at LDS address `act`) x W(80 x 64 slice at `wbuf`)^T over one K step of 64
    auto y_step = [&](unsigned act, unsigned wbuf) {
#pragma unroll
        for (int k32 = 0; k32 < 2; ++k32) {
            bf16x8 wf[5], af[6];
            if (!(FF_PROBE & 16)) {
                lds_read5(wbuf + frag_off(k32), wf);
                lds_read6(act, frag_off(k32), af);
            } else {
                for (int j = 0; j < 5; ++j) asm volatile("" : "=v"(wf[j]));
                for (int i = 0; i < 6; ++i) asm volatile("" : "=v"(af[i]));
            }
            if (FF_PROBE & 4) {
                for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(wf[j]));
                for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(af[i]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) Y[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], Y[i][j], 0, 0, 0);
        }
    };

    // LayerNorm affine of this lane's channels (64 q + lane), ahead of everything the kernel counts
    float gm[5], bt[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { gm[q] = p.gamma[q * 64 + lane]; bt[q] = p.beta[q * 64 + lane]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FF_STAMP();                             // 1

    // ================= phase 0: the tile of h, and Y = h Wp^T (K steps 0..4 of [Wp | Wp W2]) =================
    // order of issue: tile (15), S0 -> wc (10), S1 -> ring (10), [read S0] S2 -> wc, [read S1] S3 -> ring, [read S2] S4 -> wc
    load_tile(p.h, p.ldh);
    load_wc(0, wc);
    load_wc(1, ring);                       // the W1 ring region doubles as the second slice buffer of this phase (12 KB >= 10 KB)
    wait_vm<20>();
    barrier_lds();                          // T complete
    FF_STAMP();                             // 2
    wait_vm<10>(); y_step(T0 + 0 * FF_PANEL, WC0); load_wc(2, wc);
    wait_vm<10>(); y_step(T0 + 1 * FF_PANEL, R0);  load_wc(3, ring);
    wait_vm<10>(); y_step(T0 + 2 * FF_PANEL, WC0); load_wc(4, wc);
    wait_vm<10>(); y_step(T0 + 3 * FF_PANEL, R0);
    wait_vm<0>();  y_step(T0 + 4 * FF_PANEL, WC0);
    // the weight streams of the main loop, in the order the loop keeps: W1 K steps s = 5 c + ks round-robin over the three slots,
    // the chunk's slice of [Wp | Wp W2] behind step 5 c + 2
    load_w1(0, 0, 0);
    load_w1(0, 1, 1);
    load_w1(0, 2, 2);
    load_wc(FF_KS + 0, wc);
    FF_STAMP();                             // 3

    // ================= phase 1: LayerNorm of the tile, in place (wave w: rows 24 w .. 24 w + 23) =================
    barrier_lds();                          // every wave has finished reading h
    FF_STAMP();                             // 4
    for (int r = 0; r < 24; ++r) {
        const int row = wave * 24 + r;
        unsigned short* e[5];
        float v[5];
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) {           // element k = 64 q + lane: panel q, chunk lane >> 3 at position (lane >> 3) ^ (row & 7)
            e[q] = reinterpret_cast<unsigned short*>(T + q * FF_PANEL + row * 128 + ((((lane >> 3) ^ (row & 7))) * 16) + (lane & 7) * 2);
            v[q] = bf16_bits_to_f32(*e[q]);
            sm += v[q];
        }
        const float mean = wave_sum(sm) * (1.0f / FF_C);
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) { const float d = v[q] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) * (1.0f / FF_C) + p.eps);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const bf16 o = (bf16)((v[q] - mean) * rstd * gm[q] + bt[q]);
            *e[q] = __builtin_bit_cast(unsigned short, o);
        }
    }
    __syncthreads();                        // T = LN(h)
    FF_STAMP();                             // 5

    // ================= phase 2: the chunks of the inner dimension =================
    // One K step of H: wait for W1 step s (PEND = operations issued after it), 12 MFMAs per 32 of k, then the slot takes step s + 3
    // (past the end of the stream the requests wrap to chunk 0: loads nobody reads, so that every count stays a constant).
#pragma unroll 1
    for (int c = 0; c < FF_NCHUNK; ++c) {
        FF_STAMP();                         // 6 + 6 c: chunk top
        f32x4 H[6][2];
#pragma unroll
        for (int i = 0; i < 6; ++i) { H[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; H[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // this wave's 32 biases of the chunk, through the scalar cache (not a vector-memory operation)
        f32x16 b1v, b1g;
        {
            const float* bp = p.b1 + 128 * c + 32 * wave;
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(b1v), "=&s"(b1g) : "s"(bp) : "memory");
        }
        FF_STAMP();                         // 6 + 6 c + 1: biases in
        const int slot0 = (c * FF_KS) % FF_W1_NSLOT;
        auto h_step = [&](int ks, auto pend) {
            int slot = slot0 + ks;
            slot -= slot >= FF_W1_NSLOT ? FF_W1_NSLOT : 0;
            slot -= slot >= FF_W1_NSLOT ? FF_W1_NSLOT : 0;
            wait_vm<decltype(pend)::value>();
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32) {
                bf16x8 wf[2], af[6];
                if (!(FF_PROBE & 16)) {
                    lds_read2(R0 + slot * FF_W1_SLOT + frag_off(k32), wf);
                    lds_read6(T0 + ks * FF_PANEL, frag_off(k32), af);
                } else {
                    for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(wf[j]));
                    for (int i = 0; i < 6; ++i) asm volatile("" : "=v"(af[i]));
                }
                if (FF_PROBE & 4) {
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(wf[j]));
                    for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(af[i]));
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    H[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], af[i], H[i][0], 0, 0, 0);
                    H[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], af[i], H[i][1], 0, 0, 0);
                }
            }
            int s3 = c * FF_KS + ks + 3;
            s3 -= s3 >= FF_NCHUNK * FF_KS ? FF_NCHUNK * FF_KS : 0;
            if (!(FF_PROBE & 1)) load_w1(s3 / FF_KS, s3 % FF_KS, slot);
        };
        // behind step s: s + 1, s + 2 (8) and, for ks <= 2, the chunk's [Wp | Wp W2] slice (10), issued behind step 5 c + 2
        h_step(0, IntTag<18>{}); h_step(1, IntTag<18>{}); h_step(2, IntTag<18>{}); h_step(3, IntTag<8>{}); h_step(4, IntTag<8>{});
        // ---- g = value * gelu(gate), bf16, into the g panel: row 16 i + frow, columns 16 wave + 4 fq .. + 3
        f32x4 bv, bg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bv[e] = fq == 0 ? b1v[e] : fq == 1 ? b1v[4 + e] : fq == 2 ? b1v[8 + e] : b1v[12 + e];
            bg[e] = fq == 0 ? b1g[e] : fq == 1 ? b1g[4 + e] : fq == 2 ? b1g[8 + e] : b1g[12 + e];
        }
        FF_STAMP();                         // + 2: H done
        if (c > 0) barrier_lds();           // every wave has finished reading the previous chunk's g
        FF_STAMP();                         // + 3
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 val = H[i][0] + bv, gat = H[i][1] + bg;
            const f32x2 ge0 = (FF_PROBE & 8) ? f32x2{gat[0], gat[1]} : gelu_erf_f2(f32x2{gat[0], gat[1]});
            const f32x2 ge1 = (FF_PROBE & 8) ? f32x2{gat[2], gat[3]} : gelu_erf_f2(f32x2{gat[2], gat[3]});
            u32x2 o;
            o[0] = pack2(val[0] * ge0[0], val[1] * ge0[1]);
            o[1] = pack2(val[2] * ge1[0], val[3] * ge1[1]);
            const int row = 16 * i + frow;
            lds_write8(G0 + row * 128 + (((2 * wave + (fq >> 1)) ^ (row & 7)) * 16) + (fq & 1) * 8, o);
        }
        barrier_lds();                      // g complete (and, in the last chunk, every wave is done with LN(h))
        FF_STAMP();                         // + 4
        // ---- Y += g Wc^T; behind the slice: W1 steps 5 c + 3 .. 5 c + 7 (20)
        wait_vm<20>();
        FF_STAMP();                         // + 5
        if (c == FF_NCHUNK - 1) load_tile(p.x, p.ldx);      // T is free: x rides in under the last Y step
        y_step(G0, WC0);
        if (!(FF_PROBE & 2) && c + 1 < FF_NCHUNK) load_wc(FF_KS + c + 1, wc);
    }

    // ================= phase 3: y = Y + bias + x, through T; column sums of the tile =================
    FF_STAMP();                             // 126
    wait_vm<0>();
    __syncthreads();                        // x complete in T (and every wave past its last read of g)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int n = 80 * wave + 16 * j + 4 * fq;          // this lane's 4 consecutive output columns
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bcat + n);
        const int pnl = n >> 6, ch = (n & 63) >> 3;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = 16 * i + frow;
            u32x2* cell = reinterpret_cast<u32x2*>(T + pnl * FF_PANEL + row * 128 + ((ch ^ (row & 7)) * 16) + (n & 7) * 2);
            const u32x2 xv = *cell;
            u32x2 o;
            o[0] = pack2(Y[i][j][0] + bb[0] + __builtin_bit_cast(float, xv[0] << 16),
                         Y[i][j][1] + bb[1] + __builtin_bit_cast(float, xv[0] & 0xffff0000u));
            o[1] = pack2(Y[i][j][2] + bb[2] + __builtin_bit_cast(float, xv[1] << 16),
                         Y[i][j][3] + bb[3] + __builtin_bit_cast(float, xv[1] & 0xffff0000u));
            *cell = o;
        }
    }
    __syncthreads();
    FF_STAMP();                             // 127
    // whole rows out: the inverse of load_tile (LDS position l & 7 of row r holds chunk (l & 7) ^ (r & 7))
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
        const u32x4 v = *reinterpret_cast<const u32x4*>(T + pnl * FF_PANEL + rg * 1024 + lane * 16);
        store16_out(p.y + (int64_t)(m0 + rg * 8 + drow) * p.ldy + pnl * 64 + dchunk, v);
    }
    FF_STAMP();                             // 128: rows stored
    if (p.colsum_fx) {
        // (sum, sum of squares) of the stored bf16 values per column: thread (16-row segment s6, 8-column chunk cc), then the six
        // segments through the g panel
        float* part = reinterpret_cast<float*>(G);          // [6][320][2] floats = 15 360 B > 12 288: two passes of three segments
        const int cc = tid % 40, s6 = tid / 40;             // 240 threads work
        float sm[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[e] = sq[e] = 0.f;
        if (s6 < 6) {
            const int pnl = cc >> 3, ch = cc & 7;
            for (int r = 0; r < 16; ++r) {
                const int row = 16 * s6 + r;
                const u32x4 v = *reinterpret_cast<const u32x4*>(T + pnl * FF_PANEL + row * 128 + ((ch ^ (row & 7)) * 16));
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sm[e] += f[e]; sq[e] += f[e] * f[e]; }
            }
        }
        float tot_s[8], tot_q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tot_s[e] = tot_q[e] = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (s6 < 6 && s6 / 3 == pass) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    part[((s6 % 3) * FF_C + cc * 8 + e) * 2] = sm[e];
                    part[((s6 % 3) * FF_C + cc * 8 + e) * 2 + 1] = sq[e];
                }
            }
            __syncthreads();
            if (tid < 40) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    for (int s = 0; s < 3; ++s) {           // fixed order: deterministic
                        tot_s[e] += part[(s * FF_C + tid * 8 + e) * 2];
                        tot_q[e] += part[(s * FF_C + tid * 8 + e) * 2 + 1];
                    }
            }
        }
        if (tid < 40) {
            const int nb = p.M / p.fx_rows;
            int64_t* dst = p.colsum_fx + (int64_t)((((int)blockIdx.x % p.fx_reps) * nb + m0 / p.fx_rows) * 2) * FF_C + tid * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { fx_add(dst + e, tot_s[e]); fx_add(dst + FF_C + e, tot_q[e]); }
        }
    }
}

std::once_flag g_ff_once;

}  // namespace

extern "C" int seer_ff_fused_c320(const void* h, int32_t ldh, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t M,
                                  const float* gamma, const float* beta, float eps, const void* w1, const float* b1, const void* wcat,
                                  const float* bcat, int64_t* colsum_fx, int64_t fx_rows, int32_t fx_reps, void* stream) {
    if (!h || !x || !y || !gamma || !beta || !w1 || !b1 || !wcat || !bcat) return SEER_EINVAL;
    if (M <= 0 || M % FF_BM || ldh % 8 || ldx % 8 || ldy % 8 || ldh < FF_C || ldx < FF_C || ldy < FF_C) return SEER_EINVAL;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w1) |
         reinterpret_cast<uintptr_t>(wcat) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(bcat)) & 15) return SEER_EINVAL;
    if (colsum_fx && (fx_rows <= 0 || fx_rows % FF_BM || M % fx_rows || fx_reps <= 0)) return SEER_EINVAL;
    std::call_once(g_ff_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_ff_fused_c320_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
    });
    FfArgs a;
    a.h = reinterpret_cast<const bf16*>(h); a.x = reinterpret_cast<const bf16*>(x); a.y = reinterpret_cast<bf16*>(y);
    a.ldh = ldh; a.ldx = ldx; a.ldy = ldy; a.M = (int)M;
    a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.w1 = reinterpret_cast<const bf16*>(w1); a.b1 = b1; a.wcat = reinterpret_cast<const bf16*>(wcat); a.bcat = bcat;
    a.colsum_fx = colsum_fx; a.fx_rows = (int)fx_rows; a.fx_reps = fx_reps;
    hipLaunchKernelGGL(seer_ff_fused_c320_kernel, dim3((unsigned)(M / FF_BM)), dim3(256), FF_LDS, reinterpret_cast<hipStream_t>(stream), a);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
