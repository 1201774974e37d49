// The feed-forward of a transformer block at the 320-channel level of the Seer UNet as ONE launch on gfx950 (MI355X):
//     y = x + [Wp | Wp W2] [h | g] + (Wp b2 + bp),      g = GEGLU(LayerNorm(h) W1^T + b1)
// i.e. norm3 -> ff.net.0 (GEGLU) -> ff.net.2 + residual -> proj_out + residual of BasicTransformerBlock / SpatialTransformer3D
// (seer/models/attention.py:231-248, 308-327, 742-747, 783-793, 126, 141-145).  The launches it replaces: layernorm, the
// weight-stationary GEGLU projection, and the two-source [proj_out | proj_out ff.net.2] GEMM -- 102 us at 24 576 rows; the 4 x 1280
// intermediate (63 MB written and read back per block) and the normalised copy of the residual stream (31 MB) never reach memory.
//
// Shape of the kernel.  Rows of these operators are independent, so a workgroup OWNS 96 rows (24 576 rows = 256 workgroups = one
// per CU, one round) and keeps them in LDS for the whole launch:
//   * T: the 96 x 320 tile of h as five K panels [96 rows][128 B] (16-byte chunks XOR-swizzled by row & 7, written by LDS-DMA with
//     the swizzle on the source side); the proj_out part Y += h Wp^T runs on it first, then it is LayerNormed IN PLACE (a wave
//     owns 24 rows, 8 lanes a row; two-pass statistics in registers, as seer_layernorm), and every later step reads LN(h) from it;
//   * the weights STREAM from L2 straight into registers: wave w (one per SIMD, 4 per workgroup) owns 80 output columns of Y and,
//     per 64-wide chunk of the inner dimension, 32 rows of W1 (16 value + 16 gate columns: the packed GEGLU row order interleaves
//     them in 16s) -- nobody else reads those rows, so staging them in LDS would buy nothing.  The host packs both matrices in
//     FRAGMENT order (seer_ff_fused_pack_*): every load is one contiguous KiB per wave.  A wave holds its 20 W1 fragments of a chunk
//     and its 10 [Wp | Wp W2] fragments in 120 registers; a fragment is requested again, for the NEXT chunk, as soon as its last
//     MFMA has issued (a whole chunk of lead, ~1.5 us), and every wait is a constant count: the order of issue never changes;
//   * per chunk: H = LN(h) W1c^T in two halves of 48 rows (6 MFMAs per 32 of k, half and wave), + b1, GEGLU in registers beside the
//     next half's MFMAs (plain fp32 instructions: packed ones stall MFMAs, profiles/r05_lab_mfma_valu.log), g (96 x 64 bf16) into one of
//     TWO 12 KB LDS panels, ONE barrier, Y += g Wc^T (30 MFMAs per 32 of k); the activation fragments come from LDS three sub steps
//     ahead of their MFMAs, every wait counted;
//   * epilogue: x arrives in T by LDS-DMA under the last chunk, y leaves T as whole rows; the tile's column sums (sum, sum of
//     squares of the stored bf16 values), added in 64-bit fixed point, for the GroupNorm that consumes the block's output.
// LDS: 61 440 (T) + 2 x 12 288 (g) + 12 800 (gamma, beta, b1) = 98 816 bytes.  MFMAs are issued "swapped" (the weight fragment is the
// A operand), so a lane holds 4 consecutive output columns of one row.
#include "seer_common.h"
#include <mutex>
#include <utility>

// measurement builds (scripts/lab_ff_probe.py): bit mask of what to LEAVE OUT of the chunk loop -- wrong results, timing only
//   1 the W1 stream, 2 the [Wp | Wp W2] stream, 4 the MFMAs, 8 the GELU, 16 the activation fragment reads
#ifndef FF_PROBE
#define FF_PROBE 0
#endif

#ifdef FF_STAMPS    // measurement build: s_memrealtime (10 ns) stamps of the first and the last workgroup, per wave
__device__ long long seer_ff_stamps[2 * 4 * 256];
#define FF_STAMP()                                                                                                        \
    do {                                                                                                                  \
        if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0 && nst < 256)                                   \
            seer_ff_stamps[((blockIdx.x ? 1 : 0) * 4 + wave) * 256 + nst] = wall_clock64();                              \
        if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0 && (nst == 0 || nst == 63))                      \
            seer_ff_stamps[((blockIdx.x ? 1 : 0) * 4 + wave) * 256 + 200 + (nst ? 1 : 0)] = clock64();                    \
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
constexpr int FF_CONST_BYTES = (2 * FF_C + 2 * FF_INNER) * 4;               // gamma | beta | b1, fp32
constexpr int FF_LDS = FF_T_BYTES + 2 * FF_PANEL + FF_CONST_BYTES;          // 98 816
constexpr int FF_W1_BLOCK = 20 * 1024;     // one wave's W1 fragments of a chunk: [ks 5][k32 2][value | gate][64 lanes][16 B]
constexpr int FF_WC_BLOCK = 10 * 1024;     // one wave's [Wp | Wp W2] fragments of a K step of 64: [k32 2][5 column fragments][64][16 B]

struct FfArgs {
    const bf16* h; const bf16* x; bf16* y;
    int ldh, ldx, ldy, M;
    const float* gamma; const float* beta; float eps;
    const unsigned char* w1f; const float* b1;      // W1 in fragment order (seer_ff_fused_pack_w1), its bias in the packed GEGLU order
    const unsigned char* wcf; const float* bcat;    // [Wp | Wp W2] in fragment order (seer_ff_fused_pack_wcat), Wp b2 + bp
    int64_t* colsum_fx;                     // [reps][M / fx_rows][2][FF_C] fixed-point column sums (seer_gemm_desc::colsum_fx) or NULL
    int fx_rows, fx_reps;                   // rows per batch element (a multiple of 16, at least 96), replicas
    float* colsum_tiles;                    // [M / 96][FF_C][2] fp32 per-tile column sums (seer_gemm_desc::colsum) or NULL
    // optional prologue (seer_ff_fused_c320_pre): the rows of `h` are h = h + a Wo^T + bo first -- the attention's to_out projection and
    // its residual (attention.py:316-322, 237-240), whose output nobody but this launch reads
    const bf16* a; int lda; const unsigned char* wof; const float* bo;
};

__device__ __forceinline__ unsigned lds_u32(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}
// Every wave issues its vector-memory operations in ONE fixed order, so every wait below is a compile-time count of the operations
// that were issued AFTER the one waited for (vmcnt retires in order).
template <int N> struct IntTag { static constexpr int value = N; };
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that leaves the vector-memory counter alone (__syncthreads() drains it)
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// ---- weight fragments: global -> registers, asynchronously.  The destination registers are outputs of the REQUEST and in-out operands
// of the WAIT; between the two nothing may name them (seervideoldm_amd/asm_check.py::check_ff_fused walks the assembly for that).
struct W4 { u32x4 r[4]; };                  // one K step of this wave's W1 rows: k32 0 value, gate; k32 1 value, gate (1 KiB apart)
struct W10 { u32x4 r[10]; };                // one K step of this wave's 80 rows of [Wp | Wp W2]: k32 0 x 5 column fragments, k32 1 x 5
__device__ __forceinline__ void req4(W4& w, unsigned voff, const unsigned char* base) {      // into ACCUMULATION registers
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&a"(w.r[0]), "=&a"(w.r[1]), "=&a"(w.r[2]), "=&a"(w.r[3]) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void req10(W10& w, unsigned voff, const unsigned char* base) {
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(w.r[0]), "=&v"(w.r[1]), "=&v"(w.r[2]), "=&v"(w.r[3]) : "v"(voff), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(w.r[4]), "=&v"(w.r[5]), "=&v"(w.r[6]), "=&v"(w.r[7]) : "v"(voff), "s"(base + 4096) : "memory");
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(w.r[8]), "=&v"(w.r[9]) : "v"(voff), "s"(base + 8192) : "memory");
}
template <int N> __device__ __forceinline__ void got4(W4& w) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+a"(w.r[0]), "+a"(w.r[1]), "+a"(w.r[2]), "+a"(w.r[3]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void got10(W10& w) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(w.r[0]), "+v"(w.r[1]), "+v"(w.r[2]), "+v"(w.r[3]), "+v"(w.r[4]), "+v"(w.r[5]), "+v"(w.r[6]), "+v"(w.r[7]),
                   "+v"(w.r[8]), "+v"(w.r[9])
                 : "n"(N) : "memory");
}
// ---- activation fragments: LDS -> registers one sub step ahead.  Rows frow, 16 + frow, ... 80 + frow of a panel (2 KiB apart).
struct AFrag { u32x4 r[6]; };
__device__ __forceinline__ void a_req(AFrag& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:2048\n\tds_read_b128 %2, %6 offset:4096\n\t"
                 "ds_read_b128 %3, %6 offset:6144\n\tds_read_b128 %4, %6 offset:8192\n\tds_read_b128 %5, %6 offset:10240"
                 : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]), "=&v"(f.r[3]), "=&v"(f.r[4]), "=&v"(f.r[5]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void a_got(AFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]), "+v"(f.r[4]), "+v"(f.r[5])::"memory");
}
struct AHalf { u32x4 r[3]; };               // three row fragments (48 rows) of one sub step
__device__ __forceinline__ void h_req(AHalf& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:2048\n\tds_read_b128 %2, %3 offset:4096"
                 : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]) : "v"(addr) : "memory");
}
// N: LDS operations issued behind this fragment's three reads that may stay in flight (the LDS returns in order)
template <int N = 0> __device__ __forceinline__ void h_got(AHalf& f) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]) : "n"(N) : "memory");
}
__device__ __forceinline__ void bias_req(unsigned addr, f32x4& a, f32x4& b) {          // a = [addr], b = [addr + 64]: two operations
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}
template <int N> __device__ __forceinline__ void bias_got(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// F16: h, x, y and both weight matrices hold IEEE half (the fp16 engine): same loads, LDS images and schedule; the MFMA opcode and
// the pack / unpack of LayerNorm, GEGLU, residual and column sums differ
template <bool F16, bool PRE = false>
__global__ void __launch_bounds__(256, 1) seer_ff_fused_c320_kernel(const FfArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const T = smem;
    unsigned char* const G = smem + FF_T_BYTES;                                 // two g panels
    float* const cst = reinterpret_cast<float*>(smem + FF_T_BYTES + 2 * FF_PANEL);     // gamma [320] | beta [320] | b1 [2560]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * FF_BM;
    [[maybe_unused]] int nst = 0;
    FF_STAMP();                             // 0

    // ---- LDS-DMA of an activation tile.  A piece = 8 rows x 128 B; lane l -> row l >> 3, LDS position l & 7 holds source chunk
    // (l & 7) ^ (row & 7); 60 pieces, 15 per wave
    const int drow = lane >> 3;
    const int dchunk = ((lane & 7) ^ (drow & 7)) * 8;       // element offset of the source chunk inside the 64-wide K step
    auto load_tile = [&](const bf16* src, int ld) {
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
            const int m = min(m0 + rg * 8 + drow, p.M - 1);            // (a ragged last tile reads its last row again)
            const bf16* s = src + (int64_t)m * ld + pnl * 64 + dchunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(T + pnl * FF_PANEL + rg * 1024), 16, 0, 0);
        }
    };

    // ---- the weight streams of this wave
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned char* const wc_wave = p.wcf + (int64_t)wave * FF_WC_BLOCK;           // + s * 4 * FF_WC_BLOCK: K step s of [Wp | Wp W2]
    const unsigned char* const w1_wave = p.w1f + (int64_t)wave * FF_W1_BLOCK;           // + c * 4 * FF_W1_BLOCK: chunk c of W1
    W4 w1r[5];                              // W1 fragments of one chunk, per K step
    W10 wcr;                                // [Wp | Wp W2] fragments of one K step

    // ---- accumulators
    f32x4 Y[6][5];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS offset of the fragment of row frow (+ 16 i), logical 16-byte chunk k32 * 4 + fq of a 64-wide K step
    const unsigned swz = (unsigned)(frow & 7);
    const unsigned fo0 = (unsigned)(frow * 128 + (((0 + fq) ^ swz) * 16)), fo1 = (unsigned)(frow * 128 + (((4 + fq) ^ swz) * 16));
    const unsigned T0 = lds_u32(T), G0 = lds_u32(G), CST0 = lds_u32(cst);

    auto mfma_y = [&](const u32x4* w5, const AFrag& a) {
        if (FF_PROBE & 4) {
            for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(w5[j]));
            for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(a.r[i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) Y[i][j] = mma16<F16>(as_bf(w5[j]), as_bf(a.r[i]), Y[i][j]);
    };

    // Y += T W^T for five K steps of a fragment-packed matrix whose steps 0 and 1 are requested (wcr, s1); every wait a count of the
    // requests issued behind the one waited for
    W10 s1;
    auto prod5 = [&](const unsigned char* wbase) {
        AFrag a0, a1;
        // sub step (s, k32) of K step s reads panel s; the next sub step's fragments are requested before this one's MFMAs
        a_req(a0, T0 + fo0);
        a_got(a0);
        got10<0>(wcr);
        got10<0>(s1);
        // S0 (wcr)
        a_req(a1, T0 + fo1);                             mfma_y(&wcr.r[0], a0); a_got(a1);
        a_req(a0, T0 + FF_PANEL + fo0);                  mfma_y(&wcr.r[5], a1); a_got(a0);
        req10(wcr, voff, wbase + 2 * 4 * FF_WC_BLOCK);                          // S2 -> wcr
        // S1
        a_req(a1, T0 + FF_PANEL + fo1);                  mfma_y(&s1.r[0], a0);  a_got(a1);
        a_req(a0, T0 + 2 * FF_PANEL + fo0);              mfma_y(&s1.r[5], a1);  a_got(a0);
        req10(s1, voff, wbase + 3 * 4 * FF_WC_BLOCK);                           // S3 -> s1; behind S2: S3
        // S2
        a_req(a1, T0 + 2 * FF_PANEL + fo1); got10<10>(wcr); mfma_y(&wcr.r[0], a0); a_got(a1);
        a_req(a0, T0 + 3 * FF_PANEL + fo0);              mfma_y(&wcr.r[5], a1); a_got(a0);
        req10(wcr, voff, wbase + 4 * 4 * FF_WC_BLOCK);                          // S4 -> wcr; behind S3: S4
        // S3
        a_req(a1, T0 + 3 * FF_PANEL + fo1); got10<10>(s1); mfma_y(&s1.r[0], a0); a_got(a1);
        a_req(a0, T0 + 4 * FF_PANEL + fo0);              mfma_y(&s1.r[5], a1);  a_got(a0);
        // S4
        a_req(a1, T0 + 4 * FF_PANEL + fo1); got10<0>(wcr); mfma_y(&wcr.r[0], a0); a_got(a1);
        mfma_y(&wcr.r[5], a1);
    };

    if constexpr (PRE) {
        // ================= prologue: h <- h + a Wo^T + bo, through T (nothing of it is stored: this launch is its only reader) =========
        // the residual quads of this lane (row 16 i + frow, columns 80 w + 16 j + 4 fq ..) ride in registers under the product
        load_tile(p.a, p.lda);
        const unsigned char* const wo_wave = p.wof + (int64_t)wave * FF_WC_BLOCK;
        req10(wcr, voff, wo_wave + 0 * 4 * FF_WC_BLOCK);
        req10(s1, voff, wo_wave + 1 * 4 * FF_WC_BLOCK);
        u32x2 rq[6][5];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int m = min(m0 + 16 * i + frow, p.M - 1);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                rq[i][j] = *reinterpret_cast<const u32x2*>(p.h + (int64_t)m * p.ldh + 80 * wave + 16 * j + 4 * fq);
        }
        wait_vm<0>();
        __syncthreads();                    // the tile of a complete
        prod5(wo_wave);
        barrier_lds();                      // every wave has finished reading a
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int n = 80 * wave + 16 * j + 4 * fq;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bo + n);
            const int pnl = n >> 6, ch = (n & 63) >> 3;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int row = 16 * i + frow;
                const f32x2 r01 = unpack2t<F16>(rq[i][j][0]), r23 = unpack2t<F16>(rq[i][j][1]);
                u32x2 o;
                o[0] = pack2t<F16>(Y[i][j][0] + bb[0] + r01[0], Y[i][j][1] + bb[1] + r01[1]);
                o[1] = pack2t<F16>(Y[i][j][2] + bb[2] + r23[0], Y[i][j][3] + bb[3] + r23[1]);
                *reinterpret_cast<u32x2*>(T + pnl * FF_PANEL + row * 128 + ((ch ^ (row & 7)) * 16) + (n & 7) * 2) = o;
                Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // ================= phase 0: the tile of h; gamma, beta, b1 into LDS; Y = h Wp^T (K steps 0..4 of [Wp | Wp W2]) =================
    // the first two K steps of the stream are requested before anything else
    if constexpr (!PRE) load_tile(p.h, p.ldh);
    req10(wcr, voff, wc_wave + 0 * 4 * FF_WC_BLOCK);
    req10(s1, voff, wc_wave + 1 * 4 * FF_WC_BLOCK);
    for (int i = tid; i < (2 * FF_C + 2 * FF_INNER) / 4; i += 256) {
        const f32x4* src = i < FF_C / 4 ? reinterpret_cast<const f32x4*>(p.gamma) + i
                           : i < FF_C / 2 ? reinterpret_cast<const f32x4*>(p.beta) + (i - FF_C / 4)
                                          : reinterpret_cast<const f32x4*>(p.b1) + (i - FF_C / 2);
        reinterpret_cast<f32x4*>(cst)[i] = *src;
    }
    wait_vm<0>();                           // this wave's pieces of the tile (and S0, S1) are in
    __syncthreads();                        // T (PRE: the rows of h written above) and the constants complete
    FF_STAMP();                             // 1
    prod5(wc_wave);
    // chunk 0's W1 fragments (its slice of [Wp | Wp W2] follows behind H(0), as in every iteration)
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) req4(w1r[ks], voff, w1_wave + ks * 4096);
    FF_STAMP();                             // 2

    // ================= phase 1: LayerNorm of the tile, in place (wave w: rows 24 w .. 24 w + 23; 8 lanes a row) =================
    barrier_lds();                          // every wave has finished reading h
    FF_STAMP();                             // 3
    {
        // lane l: row (l >> 3) of each pass of 8 rows, the 16-byte position l & 7 of every panel = logical chunk (l & 7) ^ (row & 7)
        const int lc = (lane & 7) ^ (lane >> 3);
        for (int pass = 0; pass < 3; ++pass) {
            const int row = wave * 24 + pass * 8 + (lane >> 3);
            float v[5][8];
            float sm = 0.f;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(T + q * FF_PANEL + row * 128 + (lane & 7) * 16);
                unpack8t<F16>(raw, v[q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += v[q][e];
            }
            sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
            const float mean = sm * (1.0f / FF_C);
            float sq = 0.f;
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[q][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            const float rstd = rsqrtf(sq * (1.0f / FF_C) + p.eps);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const f32x4* gp = reinterpret_cast<const f32x4*>(cst + q * 64 + lc * 8);
                const f32x4* bp = reinterpret_cast<const f32x4*>(cst + FF_C + q * 64 + lc * 8);
                const f32x4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (v[q][e] - mean) * rstd * g0[e] + b0[e];
                    o[4 + e] = (v[q][4 + e] - mean) * rstd * g1[e] + b1[e];
                }
                *reinterpret_cast<u32x4*>(T + q * FF_PANEL + row * 128 + (lane & 7) * 16) = pack8t<F16>(o);
            }
        }
    }
    barrier_lds();                          // T = LN(h)
    FF_STAMP();                             // 4

    // ================= phase 2: the chunks of the inner dimension =================
    // Iteration c: H(c) = LN(h) W1c^T in two halves of 48 rows -- the GEGLU of the first half runs in the shadow of the second half's
    // MFMAs; barrier (g(c - 1) is complete); Y += g(c - 1) Wc^T with the GEGLU of the second half in the shadow of its MFMAs; g(c) into
    // the other panel.  (Plain fp32 instructions only: packed ones do not run beside MFMAs, profiles/r05_lab_mfma_valu.log.)
    // Order of this wave's requests, per iteration c: W1(c + 1) steps 0..4 (4 each, during the second half: a step is requested again
    // as soon as its last MFMAs have issued), then slice(c) of [Wp | Wp W2] (10: its registers are free once Y(c - 1) has issued).
    // Behind W1(c) step ks, when the first half waits for it, are therefore: steps ks + 1..4 of chunk c and slice(c - 1) = 26 - 4 ks
    // requests (chunk 0: 16 - 4 ks, no slice yet); behind slice(c - 1): the five steps of W1(c + 1) = 20.  Past the end the W1
    // requests wrap to chunk 0: loads nobody uses, so that every count stays a constant and no register with a request in flight
    // meets a branch.
    f32x4 H[6][2];
    auto mfma_h3 = [&](const u32x4* w2, const AHalf& a, int i0) {
        if (FF_PROBE & 4) {
            for (int j = 0; j < 2; ++j) asm volatile("" ::"a"(w2[j]));
            for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(a.r[i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            H[i0 + i][0] = mma16<F16>(as_bf(w2[0]), as_bf(a.r[i]), H[i0 + i][0]);
            H[i0 + i][1] = mma16<F16>(as_bf(w2[1]), as_bf(a.r[i]), H[i0 + i][1]);
        }
    };
    // GEGLU of row fragment i of H: g = (value + b) * gelu(gate + b), bf16: columns 16 wave + 4 fq .. + 3 of row 16 i + frow
    f32x4 bv, bg, bvp, bgp;                 // biases of this wave's value / gate columns 4 fq .. 4 fq + 3: this chunk's, the previous chunk's
    auto geglu = [&](int i, const f32x4& b_val, const f32x4& b_gate) {
        const f32x4 val = H[i][0] + b_val, gat = H[i][1] + b_gate;
        float ge[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ge[e] = (FF_PROBE & 8) ? gat[e] : gelu_erf_f(gat[e]);
        u32x2 o;
        o[0] = pack2t<F16>(val[0] * ge[0], val[1] * ge[1]);
        o[1] = pack2t<F16>(val[2] * ge[2], val[3] * ge[3]);
        return o;
    };
    const unsigned gcell = (unsigned)(frow * 128 + (fq & 1) * 8) + (((unsigned)(2 * wave + (fq >> 1)) ^ swz) * 16);   // (16 i + frow) & 7 = frow & 7
    auto mfma_y_row = [&](const u32x4* w5, const u32x4& a, int i) {
        if (FF_PROBE & 4) {
            for (int j = 0; j < 5; ++j) asm volatile("" ::"v"(w5[j]));
            asm volatile("" ::"v"(a));
            return;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) Y[i][j] = mma16<F16>(as_bf(w5[j]), as_bf(a), Y[i][j]);
    };
    // The two halves of H(c) = 20 sub steps j = (half, ks, k32) of 6 MFMAs.  The activation fragments of sub step j + 3 are requested
    // before the MFMAs of sub step j (four buffers in rotation, three requests in flight, every wait counted: the LDS returns in order).
    // On entry both bias sets and the fragments of sub steps 0, 1 and 2 have been requested (pa, pb, pc), in this order.
    // ALL of the GEGLU runs beside these MFMAs: the first half (rows 0..47 -> H[0..2]) carries the GEGLU of the PREVIOUS chunk's rows
    // 48..95 (H[3..5] still hold them; its g values go straight to the previous chunk's panel gprev), the second half the GEGLU of this
    // chunk's rows 0..47 (ga: written behind the barrier, when the panel is free).  `first`: chunk 0 (no slice in flight, no previous chunk).
    u32x2 ga[3];
    AHalf pa, pb, pc, pd;
    auto sub_addr = [&](int j) {            // LDS address of sub step j's fragments
        const int half = j / 10, ks = (j % 10) >> 1, k32 = j & 1;
        return T0 + ks * FF_PANEL + half * 6144 + (k32 ? fo1 : fo0);
    };
    auto h_phase = [&](const unsigned char* w1_next, unsigned gprev, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        constexpr int SL = FIRST ? 0 : 10;
#pragma unroll
        for (int i = 0; i < 3; ++i) { H[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; H[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        auto step = [&](int j, AHalf& cur, AHalf& nxt3) {
            const int half = j / 10, ks = (j % 10) >> 1, k32 = j & 1;
            if (!(FF_PROBE & 16)) {
                if (j + 2 < 20) h_got<6>(cur); else if (j + 1 < 20) h_got<3>(cur); else h_got<0>(cur);
                if (j + 3 < 20) h_req(nxt3, sub_addr(j + 3));
            }
            if (j == 10) {
#pragma unroll
                for (int i = 3; i < 6; ++i) { H[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; H[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            if (half == 0 && k32 == 0) {
                if (ks == 0) got4<(FF_PROBE & 3) ? 0 : 16 + SL>(w1r[0]);
                if (ks == 1) got4<(FF_PROBE & 3) ? 0 : 12 + SL>(w1r[1]);
                if (ks == 2) got4<(FF_PROBE & 3) ? 0 : 8 + SL>(w1r[2]);
                if (ks == 3) got4<(FF_PROBE & 3) ? 0 : 4 + SL>(w1r[3]);
                if (ks == 4) got4<(FF_PROBE & 3) ? 0 : 0 + SL>(w1r[4]);
            }
            mfma_h3(&w1r[ks].r[2 * k32], cur, 3 * half);
            if (k32 == 1 && ks < 3) {
                if (half == 0) { if (!FIRST) lds_write8(gprev + (3 + ks) * 2048 + gcell, geglu(3 + ks, bvp, bgp)); }
                else ga[ks] = geglu(ks, bv, bg);
            }
            if (half == 1 && k32 == 1 && !(FF_PROBE & 1)) req4(w1r[ks], voff, w1_next + ks * 4096);
        };
        // four buffers in rotation: sub step j lives in buffer j mod 4 (pa, pb, pc, pd); on entry sub steps 0, 1, 2 are requested
        step(0, pa, pd);
        bias_got<9>(bv, bg);                // (older than sub step 0's fragments: in)
        bias_got<9>(bvp, bgp);
        step(1, pb, pa); step(2, pc, pb); step(3, pd, pc);
#pragma unroll
        for (int j = 4; j < 20; j += 4) {
            step(j, pa, pd);
            step(j + 1, pb, pa);
            step(j + 2, pc, pb);
            step(j + 3, pd, pc);
        }
    };
    // requests at the end of an iteration, for the next chunk cn: its biases and this chunk's again (the next iteration finishes this
    // chunk's GEGLU), then its first three sub steps
    auto next_requests = [&](int cn, int c) {
        bias_req(CST0 + (2 * FF_C + 128 * cn + 32 * wave + 4 * fq) * 4, bv, bg);
        bias_req(CST0 + (2 * FF_C + 128 * c + 32 * wave + 4 * fq) * 4, bvp, bgp);
        if (!(FF_PROBE & 16)) { h_req(pa, sub_addr(0)); h_req(pb, sub_addr(1)); h_req(pc, sub_addr(2)); }
    };
    // Y += g Wc^T over one chunk: the 12 fragments of the panel, 10 MFMAs per row fragment
    auto y_phase = [&](unsigned gpanel, auto pend) {
        AHalf q0, q1, q2, q3;               // rows 0..47 of k32 0, 1, then rows 48..95
        if (!(FF_PROBE & 16)) { h_req(q0, gpanel + fo0); h_req(q1, gpanel + fo1); h_req(q2, gpanel + 6144 + fo0); h_req(q3, gpanel + 6144 + fo1); }
        got10<(FF_PROBE & 3) ? 0 : decltype(pend)::value>(wcr);
        if (!(FF_PROBE & 16)) { h_got<6>(q0); h_got<6>(q1); }
#pragma unroll
        for (int i = 0; i < 3; ++i) { mfma_y_row(&wcr.r[0], q0.r[i], i); mfma_y_row(&wcr.r[5], q1.r[i], i); }
        if (!(FF_PROBE & 16)) { h_got<0>(q2); h_got<0>(q3); }
#pragma unroll
        for (int i = 0; i < 3; ++i) { mfma_y_row(&wcr.r[0], q2.r[i], 3 + i); mfma_y_row(&wcr.r[5], q3.r[i], 3 + i); }
    };

    next_requests(0, 0);
    FF_STAMP();                             // 5
    // ---- chunk 0: nothing to multiply yet
    {
        h_phase(w1_wave + (int64_t)1 * 4 * FF_W1_BLOCK, G0, IntTag<1>{});
#pragma unroll
        for (int i = 0; i < 3; ++i) lds_write8(G0 + i * 2048 + gcell, ga[i]);
        if (!(FF_PROBE & 2)) req10(wcr, voff, wc_wave + (int64_t)(FF_KS + 0) * 4 * FF_WC_BLOCK);
        next_requests(1, 0);
    }
#pragma unroll 1
    for (int c = 1; c < FF_NCHUNK; ++c) {
        FF_STAMP();                         // 6 + 3 (c - 1)
        const int cn = c + 1 < FF_NCHUNK ? c + 1 : 0;
        const unsigned gprev = G0 + ((c - 1) & 1) * FF_PANEL, gbuf = G0 + (c & 1) * FF_PANEL;
        h_phase(w1_wave + (int64_t)cn * 4 * FF_W1_BLOCK, gprev, IntTag<0>{});
        FF_STAMP();                         // + 1: H done
        barrier_lds();                      // g(c - 1) complete; every wave is past Y(c - 2), the last reader of the panel g(c) goes to
        FF_STAMP();                         // + 2
        y_phase(gprev, IntTag<20>{});       // behind slice(c - 1): the five steps of W1(c + 1)
        if (!(FF_PROBE & 2)) req10(wcr, voff, wc_wave + (int64_t)(FF_KS + c) * 4 * FF_WC_BLOCK);
#pragma unroll
        for (int i = 0; i < 3; ++i) lds_write8(gbuf + i * 2048 + gcell, ga[i]);
        next_requests(cn, c);
    }
    FF_STAMP();                             // 63
    // ---- the last chunk: the GEGLU of its rows 48..95, its product; x rides in under it (T is free: every wave is past its last H
    // at the barrier)
    {
        // (the requests for a chunk past the end are in flight: their registers stay theirs until this wait)
        h_got<0>(pa); h_got<0>(pb); h_got<0>(pc); bias_got<0>(bv, bg); bias_got<0>(bvp, bgp);
        const unsigned glast = G0 + ((FF_NCHUNK - 1) & 1) * FF_PANEL;
#pragma unroll
        for (int i = 3; i < 6; ++i) lds_write8(glast + i * 2048 + gcell, geglu(i, bvp, bgp));
        barrier_lds();
        load_tile(p.x, p.ldx);
        y_phase(glast, IntTag<15>{});       // behind the slice: the x tile's 15 pieces (the wrapped W1 requests are older)
    }

    // ================= phase 3: y = Y + bias + x, through T; column sums of the tile =================
    FF_STAMP();                             // 64
    // the W1 requests past the end may still be in flight: their registers stay the requests' until this wait
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) got4<0>(w1r[ks]);
    __syncthreads();                        // x complete in T
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
            const f32x2 x01 = unpack2t<F16>(xv[0]), x23 = unpack2t<F16>(xv[1]);
            o[0] = pack2t<F16>(Y[i][j][0] + bb[0] + x01[0], Y[i][j][1] + bb[1] + x01[1]);
            o[1] = pack2t<F16>(Y[i][j][2] + bb[2] + x23[0], Y[i][j][3] + bb[3] + x23[1]);
            *cell = o;
        }
    }
    __syncthreads();
    FF_STAMP();                             // 65
    // whole rows out: the inverse of load_tile (LDS position l & 7 of row r holds chunk (l & 7) ^ (r & 7))
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
        const u32x4 v = *reinterpret_cast<const u32x4*>(T + pnl * FF_PANEL + rg * 1024 + lane * 16);
        if (m0 + rg * 8 + drow < p.M) store16_out(p.y + (int64_t)(m0 + rg * 8 + drow) * p.ldy + pnl * 64 + dchunk, v);
    }
    FF_STAMP();                             // 66: rows stored
    if (p.colsum_fx || p.colsum_tiles) {
        // (sum, sum of squares) of the stored bf16 values per column: thread (16-row segment s6, 8-column chunk cc) sums its cell, the
        // six segments meet in LDS (the g panels: [6][320][2] floats), fixed order: deterministic
        float* part = reinterpret_cast<float*>(G);
        const int cc = tid % 40, s6 = tid / 40;             // 240 threads work
        if (s6 < 6) {
            float sm[8], sq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[e] = sq[e] = 0.f;
            const int pnl = cc >> 3, ch = cc & 7;
            const int valid = min(16, p.M - m0 - 16 * s6);   // rows of this segment inside the tensor (a ragged last tile)
            for (int r = 0; r < valid; ++r) {
                const int row = 16 * s6 + r;
                const u32x4 v = *reinterpret_cast<const u32x4*>(T + pnl * FF_PANEL + row * 128 + ((ch ^ (row & 7)) * 16));
                float f[8];
                unpack8t<F16>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { sm[e] += f[e]; sq[e] += f[e] * f[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                part[(s6 * FF_C + cc * 8 + e) * 2] = sm[e];
                part[(s6 * FF_C + cc * 8 + e) * 2 + 1] = sq[e];
            }
        }
        __syncthreads();
        // a tile may straddle two batch elements (their boundary is a multiple of 16 rows: between two segments): the segments of the
        // first go to its slot, the rest to the next one's
        const int nb = p.colsum_fx ? p.M / p.fx_rows : 1;
        const int b0 = p.colsum_fx ? m0 / p.fx_rows : 0;
        const int s_split = p.colsum_fx ? min(6, ((b0 + 1) * p.fx_rows - m0) / 16) : 6;      // segments [0, s_split) belong to b0
        int64_t* dst = p.colsum_fx ? p.colsum_fx + (int64_t)((((int)blockIdx.x % p.fx_reps) * nb + b0) * 2) * FF_C : nullptr;
        for (int i = tid; i < 2 * FF_C; i += 256) {         // i = plane * 320 + column
            const int col = i < FF_C ? i : i - FF_C, pl = i < FF_C ? 0 : 1;
            float t = 0.f, t1 = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float v = part[(s * FF_C + col) * 2 + pl];
                if (s < s_split) t += v; else t1 += v;
            }
            if (dst) {
                fx_add(dst + i, t);
                if (s_split < 6 && b0 + 1 < nb) fx_add(dst + 2 * FF_C + i, t1);
            }
            if (p.colsum_tiles) p.colsum_tiles[((int64_t)blockIdx.x * FF_C + col) * 2 + pl] = t;
        }
    }
}

std::once_flag g_ff_once;

// fragment-order packing, one thread per 16 bytes
__global__ void ff_pack_w1_kernel(const bf16* __restrict__ w1, u32x4* __restrict__ out) {
    // out[c][w][ks][k32][f][lane] = w1[128 c + 32 w + 16 f + (lane & 15)][64 ks + 32 k32 + 8 (lane >> 4) .. + 7]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * FF_INNER * FF_C / 8) return;
    const int lane = i & 63, f = (i >> 6) & 1, k32 = (i >> 7) & 1, r = i >> 8;      // r = (c * 4 + w) * 5 + ks
    const int ks = r % 5, cw = r / 5, w = cw & 3, c = cw >> 2;
    const int row = 128 * c + 32 * w + 16 * f + (lane & 15), k = 64 * ks + 32 * k32 + 8 * (lane >> 4);
    out[i] = *reinterpret_cast<const u32x4*>(w1 + (int64_t)row * FF_C + k);
}
__global__ void ff_pack_wcat_kernel(const bf16* __restrict__ wcat, u32x4* __restrict__ out) {
    // out[s][w][k32][j][lane] = wcat[80 w + 16 j + (lane & 15)][64 s + 32 k32 + 8 (lane >> 4) .. + 7]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FF_C * (FF_C + FF_INNER) / 8) return;
    const int lane = i & 63, r = i >> 6;                    // r = ((s * 4 + w) * 2 + k32) * 5 + j
    const int j = r % 5, r2 = r / 5, k32 = r2 & 1, w = (r2 >> 1) & 3, s = r2 >> 3;
    const int row = 80 * w + 16 * j + (lane & 15), k = 64 * s + 32 * k32 + 8 * (lane >> 4);
    out[i] = *reinterpret_cast<const u32x4*>(wcat + (int64_t)row * (FF_C + FF_INNER) + k);
}

}  // namespace

extern "C" int seer_ff_fused_pack_w1(const void* w1, void* out, void* stream) {
    if (!w1 || !out || ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(out)) & 15)) return SEER_EINVAL;
    const int n = 2 * FF_INNER * FF_C / 8;
    hipLaunchKernelGGL(ff_pack_w1_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(w1), reinterpret_cast<u32x4*>(out));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_ff_fused_pack_wcat(const void* wcat, void* out, void* stream) {
    if (!wcat || !out || ((reinterpret_cast<uintptr_t>(wcat) | reinterpret_cast<uintptr_t>(out)) & 15)) return SEER_EINVAL;
    const int n = FF_C * (FF_C + FF_INNER) / 8;
    hipLaunchKernelGGL(ff_pack_wcat_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(wcat), reinterpret_cast<u32x4*>(out));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_ff_fused_c320(const void* h, int32_t ldh, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t M,
                                  const float* gamma, const float* beta, float eps, const void* w1f, const float* b1, const void* wcf,
                                  const float* bcat, int64_t* colsum_fx, int64_t fx_rows, int32_t fx_reps, float* colsum_tiles,
                                  void* stream) {
    return seer_ff_fused_c320_dt(h, ldh, x, ldx, y, ldy, M, gamma, beta, eps, w1f, b1, wcf, bcat, colsum_fx, fx_rows, fx_reps, colsum_tiles,
                                 SEER_DT_BF16, stream);
}
extern "C" int seer_ff_fused_c320_dt(const void* h, int32_t ldh, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t M,
                                     const float* gamma, const float* beta, float eps, const void* w1f, const float* b1, const void* wcf,
                                     const float* bcat, int64_t* colsum_fx, int64_t fx_rows, int32_t fx_reps, float* colsum_tiles,
                                     int32_t dtype, void* stream) {
    return seer_ff_fused_c320_pre(nullptr, 0, nullptr, nullptr, h, ldh, x, ldx, y, ldy, M, gamma, beta, eps, w1f, b1, wcf, bcat, colsum_fx, fx_rows,
                                  fx_reps, colsum_tiles, dtype, stream);
}
extern "C" int seer_ff_fused_c320_pre(const void* a_in, int32_t lda, const void* wof, const float* bo, const void* h, int32_t ldh, const void* x,
                                      int32_t ldx, void* y, int32_t ldy, int64_t M, const float* gamma, const float* beta, float eps,
                                      const void* w1f, const float* b1, const void* wcf, const float* bcat, int64_t* colsum_fx, int64_t fx_rows,
                                      int32_t fx_reps, float* colsum_tiles, int32_t dtype, void* stream) {
    if (a_in && (!wof || !bo || lda % 8 || lda < FF_C || ldh % 4 ||
                 ((reinterpret_cast<uintptr_t>(a_in) | reinterpret_cast<uintptr_t>(wof) | reinterpret_cast<uintptr_t>(bo)) & 15))) return SEER_EINVAL;
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!h || !x || !y || !gamma || !beta || !w1f || !b1 || !wcf || !bcat) return SEER_EINVAL;
    if (M <= 0 || M >= ((int64_t)1 << 31) - FF_BM || ldh % 8 || ldx % 8 || ldy % 8 || ldh < FF_C || ldx < FF_C || ldy < FF_C) return SEER_EINVAL;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w1f) |
         reinterpret_cast<uintptr_t>(wcf) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(bcat) |
         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) return SEER_EINVAL;
    if (colsum_fx && (fx_rows < FF_BM || fx_rows % 16 || M % fx_rows || fx_reps <= 0)) return SEER_EINVAL;
    if (colsum_tiles && M % FF_BM) return SEER_EINVAL;
    std::call_once(g_ff_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_ff_fused_c320_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_ff_fused_c320_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_ff_fused_c320_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_ff_fused_c320_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
    });
    FfArgs a;
    a.h = reinterpret_cast<const bf16*>(h); a.x = reinterpret_cast<const bf16*>(x); a.y = reinterpret_cast<bf16*>(y);
    a.ldh = ldh; a.ldx = ldx; a.ldy = ldy; a.M = (int)M;
    a.gamma = gamma; a.beta = beta; a.eps = eps;
    a.w1f = reinterpret_cast<const unsigned char*>(w1f); a.b1 = b1; a.wcf = reinterpret_cast<const unsigned char*>(wcf); a.bcat = bcat;
    a.colsum_fx = colsum_fx; a.fx_rows = (int)fx_rows; a.fx_reps = fx_reps; a.colsum_tiles = colsum_tiles;
    a.a = reinterpret_cast<const bf16*>(a_in); a.lda = lda; a.wof = reinterpret_cast<const unsigned char*>(wof); a.bo = bo;
    const dim3 grid((unsigned)((M + FF_BM - 1) / FF_BM));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (a_in) {
        if (dtype == SEER_DT_F16) hipLaunchKernelGGL((seer_ff_fused_c320_kernel<true, true>), grid, dim3(256), FF_LDS, st, a);
        else hipLaunchKernelGGL((seer_ff_fused_c320_kernel<false, true>), grid, dim3(256), FF_LDS, st, a);
    } else {
        if (dtype == SEER_DT_F16) hipLaunchKernelGGL((seer_ff_fused_c320_kernel<true, false>), grid, dim3(256), FF_LDS, st, a);
        else hipLaunchKernelGGL((seer_ff_fused_c320_kernel<false, false>), grid, dim3(256), FF_LDS, st, a);
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
