// The row-local chains in front of the attention launches of a transformer block at the 320-channel level of the Seer UNet, ONE launch each
// on gfx950 (MI355X) -- seer/models/attention.py:129-145 (GroupNorm -> proj_in), :231-240 / :308-318 (norm1 -> to_q | to_k | to_v, the
// rotary embedding of the temporal block :649-651), :316-322 (attn1.to_out + residual -> norm2 -> attn2.to_q):
//
//     h   = [GroupNorm(in)] W1^T + b1 [+ res]                      stored (the block's residual stream)
//     out = LayerNorm(h) [W2_0 | W2_1 | ...]^T                     stored; rotary on the first rot_thirds, a column scale on the first
//
// e.g.  in = x, GroupNorm on, W1 = proj_in, W2 = to_q | to_k | to_v        (replaces gn_apply + proj_in + the q|k|v projection), or
//       in = attention output, res = h, W1 = attn1.to_out, W2 = attn2.to_q  (replaces to_out + the q projection of the cross attention).
//
// Same shape as ff_fused.hip, whose pieces it reuses: rows are independent, so a workgroup OWNS 96 rows in LDS for the whole launch
// (24 576 rows = 256 workgroups = one per CU, one round); T = the tile as five K panels [96 rows][128 B], 16-byte chunks XOR-swizzled
// by row & 7, filled by LDS-DMA; the GroupNorm scale / shift and the LayerNorm run IN PLACE on it (a wave owns 24 rows, 8 lanes a row);
// every 320 x 320 product is the same loop -- wave w owns 80 output columns, its weight fragments STREAM from L2 straight into
// registers in a host-packed fragment order (one contiguous KiB per wave and load, two K steps ahead, every wait a compile-time
// count), the activation fragments come from T one sub step ahead -- and leaves through LDS as whole rows.  What never reaches memory:
// the normalised copy of x (31 MB written + read per block at 24 576 rows), and h is read back by nobody in this launch.
// LDS: 61 440 (T) + 61 440 (residual in / staged output) + 8 960 (constants) + 10 240 (accumulated sums) = 142 080 bytes.
#include "seer_common.h"
#include <mutex>

namespace {

constexpr int RC_C = 320;
constexpr int RC_BM = 96;
constexpr int RC_KS = RC_C / 64;                     // 5 K steps
constexpr int RC_PANEL = RC_BM * 128;                // 12 288
constexpr int RC_T_BYTES = RC_KS * RC_PANEL;         // 61 440
constexpr int RC_CONST_FLOATS = 7 * RC_C;            // gn scale | gn shift (first batch element of the tile) | ln gamma | ln beta | b1 | gn scale | gn shift (second)
constexpr int RC_FX_BYTES = 2 * RC_C * 16;           // (sum, sum of squares) per channel as int64, two batch elements: the accumulated statistics of the input
constexpr int RC_LDS = 2 * RC_T_BYTES + RC_CONST_FLOATS * 4 + RC_FX_BYTES;
constexpr int RC_W_BLOCK = 10 * 1024;                // one wave's fragments of a K step of 64: [k32 2][5 column fragments][64 lanes][16 B]
constexpr int RC_MAT_BYTES = RC_C * RC_C * 2;        // one packed 320 x 320 matrix

struct RcArgs {
    const bf16* in; int ld_in;
    const float* gn_stats; const int64_t* gn_fx; int gn_fx_reps; float gn_inv_count, gn_eps; const float* gn_gamma; const float* gn_beta; int rows_per_batch, groups;
    const unsigned char* w1f; const float* b1; const bf16* res; int ldr; bf16* h; int ldh;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const unsigned char* w2f; int n2; bf16* out; int ldo;
    float col_scale; int scale_thirds;
    const float* rot_table; int rot_tokens_per_batch, rot_pos_offset, rot_head_dim, rot_dim, rot_thirds;
    int M;
};

__device__ __forceinline__ unsigned lds_u32(const void* ptr) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)ptr;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// weight fragments: global -> registers, asynchronously (as ff_fused.hip: the destination registers are outputs of the REQUEST and
// in-out operands of the WAIT; between the two nothing may name them -- asm_check.py::check_async_vregs walks the assembly for that)
// (into ACCUMULATION registers: in flight across a whole epilogue they are the register allocator's first spill candidates in the V
// file -- and a spill of an in-flight register copies stale data; MFMAs take their weight operand from there directly)
struct W10 { u32x4 r[10]; };
__device__ __forceinline__ void req10(W10& w, unsigned voff, const unsigned char* base) {
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&a"(w.r[0]), "=&a"(w.r[1]), "=&a"(w.r[2]), "=&a"(w.r[3]) : "v"(voff), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&a"(w.r[4]), "=&a"(w.r[5]), "=&a"(w.r[6]), "=&a"(w.r[7]) : "v"(voff), "s"(base + 4096) : "memory");
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&a"(w.r[8]), "=&a"(w.r[9]) : "v"(voff), "s"(base + 8192) : "memory");
}
template <int N> __device__ __forceinline__ void got10(W10& w) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+a"(w.r[0]), "+a"(w.r[1]), "+a"(w.r[2]), "+a"(w.r[3]), "+a"(w.r[4]), "+a"(w.r[5]), "+a"(w.r[6]), "+a"(w.r[7]),
                   "+a"(w.r[8]), "+a"(w.r[9])
                 : "n"(N) : "memory");
}
struct AFrag { u32x4 r[6]; };              // rows frow, 16 + frow, ... 80 + frow of a panel (2 KiB apart)
__device__ __forceinline__ void a_req(AFrag& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:2048\n\tds_read_b128 %2, %6 offset:4096\n\t"
                 "ds_read_b128 %3, %6 offset:6144\n\tds_read_b128 %4, %6 offset:8192\n\tds_read_b128 %5, %6 offset:10240"
                 : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]), "=&v"(f.r[3]), "=&v"(f.r[4]), "=&v"(f.r[5]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void a_got(AFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]), "+v"(f.r[4]), "+v"(f.r[5])::"memory");
}
__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// FULL: M is a multiple of 96 and h is stored -- every store of every tile is issued by every wave, so the head of a product can wait for its fragments by
// COUNT (the 15 row stores of the epilogue in between are younger and stay in flight) instead of draining the queue
template <bool F16, bool FULL>
__global__ void __launch_bounds__(256, 1) seer_rowchain_c320_kernel(const RcArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const T = smem;
    unsigned char* const U = smem + RC_T_BYTES;                                  // residual tile in, staged output tiles out
    float* const cst = reinterpret_cast<float*>(smem + 2 * RC_T_BYTES);          // gn scale | gn shift | ln gamma | ln beta | b1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * RC_BM;

    // ---- LDS-DMA of an activation tile: a piece = 8 rows x 128 B; lane l -> row l >> 3, LDS position l & 7 holds source chunk
    // (l & 7) ^ (row & 7); 60 pieces, 15 per wave
    const int drow = lane >> 3;
    const int dchunk = ((lane & 7) ^ (drow & 7)) * 8;
    auto load_tile = [&](unsigned char* dst, const bf16* src, int ld) {
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
            const int m = min(m0 + rg * 8 + drow, p.M - 1);            // (a ragged last tile reads its last row again)
            const bf16* s = src + (int64_t)m * ld + pnl * 64 + dchunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(dst + pnl * RC_PANEL + rg * 1024), 16, 0, 0);
        }
    };
    // whole rows of a staged tile out: the inverse (LDS position l & 7 of row r holds chunk (l & 7) ^ (r & 7))
    auto store_tile = [&](const unsigned char* src, bf16* dst, int ld) {
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const int q = wave * 15 + i, pnl = q / 12, rg = q - pnl * 12;
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + pnl * RC_PANEL + rg * 1024 + lane * 16);
            if (FULL || m0 + rg * 8 + drow < p.M) store16_out(dst + (int64_t)(m0 + rg * 8 + drow) * ld + pnl * 64 + dchunk, v);
        }
    };

    const unsigned voff = (unsigned)lane * 16u;
    W10 wa, wb;                             // two K steps of the stream in flight
    f32x4 Y[6][5];
    const unsigned swz = (unsigned)(frow & 7);
    const unsigned fo0 = (unsigned)(frow * 128 + (((0 + fq) ^ swz) * 16)), fo1 = (unsigned)(frow * 128 + (((4 + fq) ^ swz) * 16));
    const unsigned T0 = lds_u32(T);

    auto mfma_y = [&](const u32x4* w5, const AFrag& a) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) Y[i][j] = mma16<F16>(as_bf(w5[j]), as_bf(a.r[i]), Y[i][j]);
    };
    // Y = T W^T for one packed 320 x 320 matrix.  On entry K steps 0 and 1 of `w` are requested (wa, wb); before it returns the first
    // two K steps of `next` are -- the requests of the next product ride under this one's last MFMAs and its epilogue.  They are
    // UNCONDITIONAL (the last product asks for its own first steps again, 20 KB nobody reads): a request under a branch would make
    // the fragment registers a merge of two definitions, and the copy at the join would read registers whose load is in flight.
    // Every wait is a count of the requests issued behind the one waited for (older operations -- tile loads, row stores -- retire first).
    auto gemm320 = [&](const unsigned char* w, const unsigned char* next) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        AFrag a0, a1;
        a_req(a0, T0 + fo0);
        a_got(a0);
        // S1 was requested before S0 (see the tail of this loop), the epilogue's 15 row stores behind both
        // (ONE pair of wait statements on every path -- two pairs under a branch would make the fragment registers a merge of two
        // definitions, and the copy at the join would read registers still in flight -- in front of it, where nothing can be counted, a drain)
        if (!FULL) wait_vm<0>();            // (FULL: 15 row stores behind the requests -- or, at the first product, nothing at all: phase 0 drained the queue)
        got10<15>(wb);
        got10<15>(wa);
        // S0 (wa)
        a_req(a1, T0 + fo1);                             mfma_y(&wa.r[0], a0); a_got(a1);
        a_req(a0, T0 + RC_PANEL + fo0);                  mfma_y(&wa.r[5], a1); a_got(a0);
        req10(wa, voff, w + 2 * 4 * RC_W_BLOCK);                               // S2 -> wa
        // S1 (wb)
        a_req(a1, T0 + RC_PANEL + fo1);                  mfma_y(&wb.r[0], a0); a_got(a1);
        a_req(a0, T0 + 2 * RC_PANEL + fo0);              mfma_y(&wb.r[5], a1); a_got(a0);
        req10(wb, voff, w + 3 * 4 * RC_W_BLOCK);                               // S3 -> wb; behind S2: S3
        // S2
        a_req(a1, T0 + 2 * RC_PANEL + fo1); got10<10>(wa); mfma_y(&wa.r[0], a0); a_got(a1);
        a_req(a0, T0 + 3 * RC_PANEL + fo0);              mfma_y(&wa.r[5], a1); a_got(a0);
        req10(wa, voff, w + 4 * 4 * RC_W_BLOCK);                               // S4 -> wa; behind S3: S4
        // S3
        a_req(a1, T0 + 3 * RC_PANEL + fo1); got10<10>(wb); mfma_y(&wb.r[0], a0); a_got(a1);
        a_req(a0, T0 + 4 * RC_PANEL + fo0);              mfma_y(&wb.r[5], a1); a_got(a0);
        req10(wb, voff, next + 1 * 4 * RC_W_BLOCK);                            // the next product's S1 -> wb; behind S4: that
        // S4
        a_req(a1, T0 + 4 * RC_PANEL + fo1); got10<10>(wa); mfma_y(&wa.r[0], a0); a_got(a1);
        mfma_y(&wa.r[5], a1);
        req10(wa, voff, next);                                                 // ... and its S0 -> wa
    };

    // ================= phase 0: the tile, the residual tile, the constants; the first product's first two K steps =================
    load_tile(T, p.in, p.ld_in);
    if (p.res) load_tile(U, p.res, p.ldr);
    const unsigned char* const w1_wave = p.w1f + (int64_t)wave * RC_W_BLOCK;
    const unsigned char* const w2_wave = p.w2f ? p.w2f + (int64_t)wave * RC_W_BLOCK : nullptr;
    req10(wa, voff, w1_wave);
    req10(wb, voff, w1_wave + 4 * RC_W_BLOCK);
    {
        // GroupNorm of the input as a per-channel scale / shift of the (at most two: rows_per_batch >= 96) batch elements this tile's
        // rows belong to -- from (sum, sum of squares) per group (gn_stats), or from the fixed-point column sums
        // its producer ACCUMULATED (gn_fx [reps][batch][2][320] int64, seer_gemm_desc::colsum_fx: no statistics launch at all; the
        // replicas and a group's channels are added as integers, one conversion per group in double, as gn_apply_cs_kernel<FX>);
        // LayerNorm affine; bias
        const bool gn = p.gn_stats || p.gn_fx;
        const int nb = gn ? p.M / p.rows_per_batch : 1;
        const int b0 = gn ? m0 / p.rows_per_batch : 0;          // a tile spans at most two batch elements (rows_per_batch >= 96)
        const int cpg = gn ? RC_C / p.groups : 1;
        long long* fxs = reinterpret_cast<long long*>(smem + 2 * RC_T_BYTES + RC_CONST_FLOATS * 4);
        if (p.gn_fx) {
            for (int i = tid; i < 2 * RC_C; i += 256) {
                const int which = i / RC_C, c = i - which * RC_C;
                const int b = min(b0 + which, nb - 1);
                long long sm = 0, sq = 0;
                for (int r = 0; r < p.gn_fx_reps; ++r) {
                    const long long* q = reinterpret_cast<const long long*>(p.gn_fx) + (int64_t)((r * nb + b) * 2) * RC_C + c;
                    sm += q[0];
                    sq += q[RC_C];
                }
                fxs[2 * i] = sm;
                fxs[2 * i + 1] = sq;
            }
            __syncthreads();
        }
        for (int i = tid; i < 2 * RC_C; i += 256) {
            const int which = i / RC_C, c = i - which * RC_C;
            float sc = 1.f, sh = 0.f;
            if (gn) {
                const int b = min(b0 + which, nb - 1);
                float mean, var;
                if (p.gn_fx) {
                    long long sm = 0, sq = 0;
                    const int c0 = which * RC_C + (c / cpg) * cpg;
                    for (int e = 0; e < cpg; ++e) { sm += fxs[2 * (c0 + e)]; sq += fxs[2 * (c0 + e) + 1]; }
                    const double k = (double)p.gn_inv_count / (double)(1 << SEER_GN_FX_SHIFT);
                    const double md = (double)sm * k;
                    double vd = (double)sq * k - md * md;
                    mean = (float)md;
                    var = (float)(vd > 0.0 ? vd : 0.0);
                } else {
                    const f32x2 st = *reinterpret_cast<const f32x2*>(p.gn_stats + ((int64_t)b * p.groups + c / cpg) * 2);
                    mean = st[0] * p.gn_inv_count;
                    var = st[1] * p.gn_inv_count - mean * mean;
                    var = var > 0.f ? var : 0.f;
                }
                sc = rsqrtf(var + p.gn_eps) * p.gn_gamma[c];
                sh = p.gn_beta[c] - mean * sc;
            }
            cst[(which ? 5 : 0) * RC_C + c] = sc;
            cst[(which ? 6 : 1) * RC_C + c] = sh;
            if (!which) {
                cst[2 * RC_C + c] = p.ln_gamma ? p.ln_gamma[c] : 1.f;
                cst[3 * RC_C + c] = p.ln_beta ? p.ln_beta[c] : 0.f;
                cst[4 * RC_C + c] = p.b1 ? p.b1[c] : 0.f;
            }
        }
    }
    wait_vm<0>();
    __syncthreads();                        // T, U and the constants complete

    // lane l: row (l >> 3) of each pass of 8 rows, LDS position l & 7 of every panel = logical chunk (l & 7) ^ (row & 7)
    const int lc = (lane & 7) ^ (lane >> 3);
    if (p.gn_stats || p.gn_fx) {
        // ---- GroupNorm apply in place (wave w: rows 24 w .. 24 w + 23)
        const int first_b = m0 / p.rows_per_batch;
        for (int pass = 0; pass < 3; ++pass) {
            const int row = wave * 24 + pass * 8 + (lane >> 3);
            const bool second = (m0 + row) / p.rows_per_batch != first_b;       // the row belongs to the tile's second batch element
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                unsigned char* cell = T + q * RC_PANEL + row * 128 + (lane & 7) * 16;
                float v[8];
                unpack8t<F16>(*reinterpret_cast<const u32x4*>(cell), v);
                const f32x4* sp = reinterpret_cast<const f32x4*>(cst + (second ? 5 : 0) * RC_C + q * 64 + lc * 8);
                const f32x4* hp = reinterpret_cast<const f32x4*>(cst + (second ? 6 : 1) * RC_C + q * 64 + lc * 8);
                const f32x4 s0 = sp[0], s1 = sp[1], h0 = hp[0], h1 = hp[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * s0[e] + h0[e];
                    v[4 + e] = v[4 + e] * s1[e] + h1[e];
                }
                *reinterpret_cast<u32x4*>(cell) = pack8t<F16>(v);
            }
        }
        barrier_lds();
    }

    // ONE site of the product loop: the fragments of the next product are in flight across this product's epilogue, i.e. across the
    // loop's back edge -- the registers they land in must be the same ones on both sides of it (asm_check.py::check_async_vregs
    // verifies that nothing names them in between), which two inlined copies of the loop do not give
    const unsigned char* wcur = w1_wave;
#pragma unroll 1
    for (int g = 0; g <= p.n2; ++g) {
        const unsigned char* wnext = g < p.n2 ? w2_wave + (int64_t)g * RC_MAT_BYTES : wcur;
        gemm320(wcur, wnext);               // (FULL: every epilogue stores its 15 pieces -- h included, the host sees to that)
        wcur = wnext;
        if (g == 0) {
            // ================= h = T W1^T + b1 (+ res), through T; stored =================
            barrier_lds();                  // every wave has finished reading the input tile
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = 80 * wave + 16 * j + 4 * fq;          // this lane's 4 consecutive output columns
                const f32x4 bb = *reinterpret_cast<const f32x4*>(cst + 4 * RC_C + n);
                const int pnl = n >> 6, ch = (n & 63) >> 3;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int row = 16 * i + frow;
                    const int off = pnl * RC_PANEL + row * 128 + ((ch ^ (row & 7)) * 16) + (n & 7) * 2;
                    f32x4 v = Y[i][j] + bb;
                    if (p.res) {
                        const u32x2 rv = *reinterpret_cast<const u32x2*>(U + off);
                        const f32x2 r01 = unpack2t<F16>(rv[0]), r23 = unpack2t<F16>(rv[1]);
                        v[0] += r01[0]; v[1] += r01[1]; v[2] += r23[0]; v[3] += r23[1];
                    }
                    u32x2 o;
                    o[0] = pack2t<F16>(v[0], v[1]);
                    o[1] = pack2t<F16>(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(T + off) = o;
                }
                __builtin_amdgcn_sched_barrier(0);          // one column fragment at a time: the scheduler would hoist all 30 loads
            }
        } else {
            // ================= out third t = T W2_t^T, rotary / column scale, through U; stored =================
            const int t = g - 1;
            if (t > 0) barrier_lds();       // every wave has read its pieces of the previous third out of U (the residual's last reader sits two barriers back)
            const bool rot = t < p.rot_thirds;
            const float sc = t < p.scale_thirds ? p.col_scale : 1.f;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = 80 * wave + 16 * j + 4 * fq;
                const int pnl = n >> 6, ch = (n & 63) >> 3;
                const int hc = n % p.rot_head_dim;                               // channel inside its head (heads are whole quads)
                const bool rq = rot && hc < p.rot_dim;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int row = 16 * i + frow;
                    f32x4 v = Y[i][j];
                    if (rq) {
                        const int pos = (m0 + row) % p.rot_tokens_per_batch + p.rot_pos_offset;
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rot_table + ((int64_t)pos * (p.rot_dim / 2) + hc / 2) * 2);
                        const f32x2 r01 = rot_pair(f32x2{v[0], v[1]}, cs[0], cs[1]);
                        const f32x2 r23 = rot_pair(f32x2{v[2], v[3]}, cs[2], cs[3]);
                        v = f32x4{r01[0], r01[1], r23[0], r23[1]};
                    }
                    u32x2 o;
                    o[0] = pack2t<F16>(v[0] * sc, v[1] * sc);
                    o[1] = pack2t<F16>(v[2] * sc, v[3] * sc);
                    *reinterpret_cast<u32x2*>(U + pnl * RC_PANEL + row * 128 + ((ch ^ (row & 7)) * 16) + (n & 7) * 2) = o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the staged tile leaves as whole rows.  ONE store site behind the branch: in a FULL launch its 15 pieces are issued on
        // every path around the loop, which is what the counted wait at the head of the next product stands on (and what
        // asm_check.py can see: the structurised branch above has paths through neither epilogue in its flow graph)
        barrier_lds();                      // the tile (h in T / the third in U) complete
        {
            const unsigned char* ssrc = g == 0 ? T : U;
            bf16* sdst = g == 0 ? p.h : p.out + (g - 1) * RC_C;
            const int sld = g == 0 ? p.ldh : p.ldo;
            if (FULL || sdst) store_tile(ssrc, sdst, sld);
        }
        if (g == 0 && p.n2 > 0) {
            if (p.ln_gamma) {
                // ================= LayerNorm of h in place (two-pass statistics in registers, as seer_layernorm) =================
                barrier_lds();              // the store read pieces of (panel, row group); the normalisation owns whole rows
                for (int pass = 0; pass < 3; ++pass) {
                    const int row = wave * 24 + pass * 8 + (lane >> 3);
                    float v[5][8];
                    float sm = 0.f;
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        unpack8t<F16>(*reinterpret_cast<const u32x4*>(T + q * RC_PANEL + row * 128 + (lane & 7) * 16), v[q]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sm += v[q][e];
                    }
                    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                    const float mean = sm * (1.0f / RC_C);
                    float sq = 0.f;
#pragma unroll
                    for (int q = 0; q < 5; ++q)
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = v[q][e] - mean; sq += d * d; }
                    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
                    const float rstd = rsqrtf(sq * (1.0f / RC_C) + p.ln_eps);
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const f32x4* gp = reinterpret_cast<const f32x4*>(cst + 2 * RC_C + q * 64 + lc * 8);
                        const f32x4* bp = reinterpret_cast<const f32x4*>(cst + 3 * RC_C + q * 64 + lc * 8);
                        const f32x4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = (v[q][e] - mean) * rstd * g0[e] + b0[e];
                            o[4 + e] = (v[q][4 + e] - mean) * rstd * g1[e] + b1[e];
                        }
                        *reinterpret_cast<u32x4*>(T + q * RC_PANEL + row * 128 + (lane & 7) * 16) = pack8t<F16>(o);
                    }
                }
            }
            barrier_lds();                  // T = LN(h)
        }
    }
}

std::once_flag g_rc_once;

// fragment-order packing of n_mats 320 x 320 matrices (rows n0 + 320 t .. of W [.., ld]), one thread per 16 bytes:
// out[t][s][w][k32][j][lane] = W[320 t + 80 w + 16 j + (lane & 15)][64 s + 32 k32 + 8 (lane >> 4) .. + 7]
__global__ void rc_pack_kernel(const bf16* __restrict__ W, int ld, int n_mats, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = RC_C * RC_C / 8;
    if (i >= n_mats * per) return;
    const int t = i / per, ii = i - t * per;
    const int lane = ii & 63, r = ii >> 6;                  // r = ((s * 4 + w) * 2 + k32) * 5 + j
    const int j = r % 5, r2 = r / 5, k32 = r2 & 1, w = (r2 >> 1) & 3, s = r2 >> 3;
    const int row = RC_C * t + 80 * w + 16 * j + (lane & 15), k = 64 * s + 32 * k32 + 8 * (lane >> 4);
    out[i] = *reinterpret_cast<const u32x4*>(W + (int64_t)row * ld + k);
}

}  // namespace

extern "C" int seer_rowchain_pack(const void* W, int32_t ld, int32_t n_mats, void* out, void* stream) {
    if (!W || !out || n_mats < 1 || ld < RC_C || ld % 8 || ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(out)) & 15)) return SEER_EINVAL;
    const int n = n_mats * RC_C * RC_C / 8;
    hipLaunchKernelGGL(rc_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(W), ld, n_mats, reinterpret_cast<u32x4*>(out));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_rowchain_c320(const seer_rowchain_desc* d, void* stream) {
    if (!d || !d->inp || !d->w1f || d->M <= 0 || d->M >= (1ll << 31) - RC_BM) return SEER_EINVAL;
    if (d->dtype != SEER_DT_BF16 && d->dtype != SEER_DT_F16) return SEER_EINVAL;
    if (d->ld_in % 8 || d->ld_in < RC_C) return SEER_EINVAL;
    if (!d->h && !d->w2f) return SEER_EINVAL;
    if (d->h && (d->ldh % 8 || d->ldh < RC_C)) return SEER_EINVAL;
    if (d->res && (d->ldr % 8 || d->ldr < RC_C)) return SEER_EINVAL;
    if (d->gn_stats && d->gn_fx) return SEER_EINVAL;
    if (d->gn_fx && d->gn_fx_reps < 1) return SEER_EINVAL;
    if (d->gn_stats || d->gn_fx) {
        if (!d->gn_gamma || !d->gn_beta || d->gn_count <= 0 || d->groups <= 0 || RC_C % d->groups) return SEER_EINVAL;
        if (d->rows_per_batch <= 0 || d->M % d->rows_per_batch) return SEER_EINVAL;
        if (d->rows_per_batch < RC_BM) return SEER_ENOSYS;       // a 96-row tile would span more than two batch elements
    }
    if ((d->ln_gamma == nullptr) != (d->ln_beta == nullptr)) return SEER_EINVAL;
    if (d->w2f) {
        if (d->n2 < 1 || d->n2 > 3 || !d->out || d->ldo % 8 || d->ldo < d->n2 * RC_C) return SEER_EINVAL;
        if (d->scale_thirds < 0 || d->scale_thirds > d->n2 || d->rot_thirds < 0 || d->rot_thirds > d->n2) return SEER_EINVAL;
        if (d->rot_thirds > 0) {
            if (!d->rot_table || d->rot_head_dim <= 0 || d->rot_head_dim % 4 || RC_C % d->rot_head_dim || d->rot_dim <= 0 || d->rot_dim % 4 ||
                d->rot_dim > d->rot_head_dim || d->rot_tokens_per_batch <= 0)
                return SEER_EINVAL;
        }
    }
    uintptr_t al = reinterpret_cast<uintptr_t>(d->inp) | reinterpret_cast<uintptr_t>(d->w1f) | reinterpret_cast<uintptr_t>(d->h) |
                   reinterpret_cast<uintptr_t>(d->res) | reinterpret_cast<uintptr_t>(d->w2f) | reinterpret_cast<uintptr_t>(d->out);
    if (al & 15) return SEER_EINVAL;
    std::call_once(g_rc_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_rowchain_c320_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_rowchain_c320_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_rowchain_c320_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_rowchain_c320_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS);
    });
    RcArgs a;
    a.in = reinterpret_cast<const bf16*>(d->inp); a.ld_in = d->ld_in;
    const bool gn = d->gn_stats || d->gn_fx;
    a.gn_stats = d->gn_stats; a.gn_fx = d->gn_fx; a.gn_fx_reps = d->gn_fx_reps; a.gn_inv_count = gn ? (float)(1.0 / d->gn_count) : 0.f; a.gn_eps = d->gn_eps;
    a.gn_gamma = d->gn_gamma; a.gn_beta = d->gn_beta; a.rows_per_batch = gn ? (int)d->rows_per_batch : 1; a.groups = gn ? d->groups : 1;
    a.w1f = reinterpret_cast<const unsigned char*>(d->w1f); a.b1 = d->b1; a.res = reinterpret_cast<const bf16*>(d->res); a.ldr = d->ldr;
    a.h = reinterpret_cast<bf16*>(d->h); a.ldh = d->ldh;
    a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta; a.ln_eps = d->ln_eps;
    a.w2f = reinterpret_cast<const unsigned char*>(d->w2f); a.n2 = d->w2f ? d->n2 : 0; a.out = reinterpret_cast<bf16*>(d->out); a.ldo = d->ldo;
    a.col_scale = d->col_scale; a.scale_thirds = d->w2f ? d->scale_thirds : 0;
    a.rot_table = d->rot_table; a.rot_tokens_per_batch = d->rot_tokens_per_batch > 0 ? d->rot_tokens_per_batch : 1; a.rot_pos_offset = d->rot_pos_offset;
    a.rot_head_dim = d->rot_head_dim > 0 ? d->rot_head_dim : RC_C; a.rot_dim = d->rot_dim; a.rot_thirds = d->w2f ? d->rot_thirds : 0;
    a.M = (int)d->M;
    const dim3 grid((unsigned)((d->M + RC_BM - 1) / RC_BM));
    const bool full = d->M % RC_BM == 0 && d->h != nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == SEER_DT_F16) {
        if (full) hipLaunchKernelGGL((seer_rowchain_c320_kernel<true, true>), grid, dim3(256), RC_LDS, st, a);
        else hipLaunchKernelGGL((seer_rowchain_c320_kernel<true, false>), grid, dim3(256), RC_LDS, st, a);
    } else {
        if (full) hipLaunchKernelGGL((seer_rowchain_c320_kernel<false, true>), grid, dim3(256), RC_LDS, st, a);
        else hipLaunchKernelGGL((seer_rowchain_c320_kernel<false, false>), grid, dim3(256), RC_LDS, st, a);
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
