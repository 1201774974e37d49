// Weight-stationary persistent bf16 GEMM for the SHORT-K projections of the Seer UNet on gfx950 (MI355X):
//     C[m, n] = epilogue( sum_k A(m, k) * W[n, k] ),   K = 320 or 640   (q|k|v, to_out, proj_in / proj_out, GEGLU ff.net.0 and
//     the 1x1 shortcut convs of the two upper levels: attention.py:484-489,742,783, resnet.py:172)
// Same descriptor and epilogue semantics as seer_gemm_bf16's tile kernel (gemm.hip), which keeps every other shape.
//
// Why.  With K = 320 / 640 a 128x128 output tile has 5 / 10 K steps.  The tile kernel spends 5 us of CU time per tile on it
// (profiles/r02_lab_gemm_base.log: 79 us for the 3840 tiles of the level-0 GEGLU projection) against 1.3 us of MFMA work: the
// block's first operand fetch, its short refill chain and its epilogue are exposed once per tile, and both operands of every
// tile come through the LDS fill path, which gives a CU 50-70 GB/s (profiles/r02_lab_fill.log) -- 160 KB per tile = 2.7 us.
// This kernel removes the per-tile start-up and most of the fill bytes:
//   * one 512-thread workgroup per CU, resident for the whole launch.  Its 8 waves own 8 adjacent column slices of W
//     (64 columns at K = 320, 32 at K = 640) and keep them in REGISTERS as MFMA fragments (160 VGPRs) -- loaded once;
//   * A streams through an LDS ring of three 40 KB slots (64 rows x 320 or 32 rows x 640) filled by global_load_lds; every
//     wave multiplies the SAME rows by its own columns, so a CU pulls each A row once per 512 / 256 columns instead of once
//     per 128: the fill bytes of the level-0 GEGLU projection drop from 630 MB to 160 MB;
//   * ONE s_barrier per slot (160 / 80 MFMAs per wave); inside a slot every wave runs its K loop and its epilogue on its own
//     (no shared output, no shared fragment), so one wave's epilogue VALU overlaps its SIMD neighbour's MFMAs;
//   * counted s_waitcnt vmcnt computed from the wave's own issue count (loads, LDS-DMA and stores retire in order), so the
//     epilogue's residual loads and stores never make a wave wait for ring bytes it does not need yet;
//   * the epilogue leaves through wave-private LDS rows: every global store is a whole 64- or 128-byte row segment.
// Workgroups that stream the same rows of A (different column panels) are placed behind one XCD's L2.
// History (profiles/r02_lab_gemm_ws*.log): W panel in LDS + 16 KB ring = latency-bound, 2.3x slower than the tile kernel;
// chunk c on XCD c % 8 = three XCDs busy; one barrier per K step with a block-wide epilogue = 1.2x slower.
#include "seer_common.h"
#include <mutex>
#include <type_traits>

namespace {

constexpr int WS_BK = 64;
constexpr int WS_NUM_CU = 256;
constexpr int WS_SLOT_BYTES = 40 * 1024;
constexpr int WS_NSLOT = 3;

template <int T, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {          // f(std::integral_constant<int, T>) for T = first .. N - 1
    if constexpr (T < N) {
        f(std::integral_constant<int, T>{});
        static_for<T + 1, N>(f);
    }
}

__device__ __forceinline__ void ws_wait_vmcnt(int n) {       // n is wave-uniform; smaller = stricter, so round DOWN
    if (n >= 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); return; }
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: case 17: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 18: case 19: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 20: case 21: case 22: case 23: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
}

// CW: columns of W per wave (8 waves side by side: a workgroup covers 8 * CW columns).  NK = K / 64 is a template parameter:
// the W fragments are a register array and must be indexed statically (the K loop is fully unrolled).
// LDS (dynamic): [3 slots: NK x (ROWS rows x 128 B), 16-byte chunks XOR-swizzled by row & 7 as in gemm.hip]
//                [8 wave-private areas: a 32-row staging tile + the wave's CW bias values]
template <int CW, int NK, bool GEGLU, bool F16 = false>
__global__ void __launch_bounds__(512) seer_gemm_ws_kernel(const seer_gemm_desc p, const int n_panels, const int wpp) {
    constexpr int BK = WS_BK;
    constexpr int ROWS = WS_SLOT_BYTES / (NK * BK * 2);      // rows of A per slot: 64 (K = 320) or 32 (K = 640)
    constexpr int HALVES = ROWS / 32;                        // a wave multiplies 32 rows at a time
    constexpr int TN = CW / 16;
    constexpr int CWO = GEGLU ? CW / 2 : CW;                 // output columns of a wave
    constexpr int TNO = GEGLU ? TN / 2 : TN;
    constexpr int SPITCH = CWO * 2 + 16;                     // staging row pitch (bytes)
    constexpr int DPW = NK * (ROWS / 8) / 8;                 // LDS-DMA instructions per wave and slot
    static_assert(NK * (ROWS / 8) % 8 == 0 && DPW == 5, "a slot is 40 wave instructions of 1 KB: five per wave");
    static_assert(!GEGLU || TN % 2 == 0, "GEGLU needs value / gate n-tile pairs inside one wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- block -> (panel, chunk of slots).  The dispatcher deals blocks round-robin over the 8 XCDs (blocks L and L+8 share
    // one), so XCD x takes the x-th eighth of the work items in chunk-major order: the panels that stream the same rows of A sit
    // behind the same L2, and every XCD gets the same number of resident workgroups.
    const int total_items = n_panels * wpp;
    const int per_xcd = (total_items + 7) >> 3;
    const int slot_ = blockIdx.x >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + slot_;
    if (slot_ >= per_xcd || item >= total_items) return;   // padding blocks of the placement (no barrier is shared with them)
    const int chunk = item / n_panels;
    const int panel = item - chunk * n_panels;
    const int slots_m = (p.M + ROWS - 1) / ROWS;
    const int s_begin = (int)((int64_t)slots_m * chunk / wpp), s_end = (int)((int64_t)slots_m * (chunk + 1) / wpp);
    const int nslots = s_end - s_begin;
    if (nslots <= 0) return;
    const int n0 = (panel * 8 + wave) * CW;                 // this wave's first GEMM column
    const bool active = n0 < p.N;                           // a wave past the last column only helps to fill the ring

    unsigned char* const ring = smem;
    unsigned char* const stage = smem + WS_NSLOT * WS_SLOT_BYTES + wave * (32 * SPITCH + CW * 4);
    float* const bias_l = reinterpret_cast<float*>(stage + 32 * SPITCH);   // this wave's CW bias values (registers are full of W)

    const bf16* __restrict__ A = reinterpret_cast<const bf16*>(p.A);
    const bf16* __restrict__ A2 = reinterpret_cast<const bf16*>(p.A2);
    const bf16* __restrict__ W = reinterpret_cast<const bf16*>(p.W);
    bf16* __restrict__ Cb = reinterpret_cast<bf16*>(p.C);

    int issued = 0;                                         // vector-memory instructions this wave has issued (counted ones)
    int mark[WS_NSLOT];                                     // `issued` right after the LDS-DMA of the slot last put there

    // ---- LDS-DMA of slot s: 40 instructions of 8 rows x 128 B; wave w issues q = 5 w .. 5 w + 4,  q -> (K step, row group)
    const int drow = lane >> 3;
    const int dchunk = ((lane & 7) ^ (drow & 7)) * 8;       // element offset inside the 64-wide K step (source-side swizzle)
    auto issue_slot = [&](int s) {
        if (s < nslots) {
            const int m_base = (s_begin + s) * ROWS;
            unsigned char* dst = ring + (s % WS_NSLOT) * WS_SLOT_BYTES;
            // source = wave-uniform 64-bit base (slot row, K step) + a 32-bit per-lane byte offset: one VGPR of address per piece
            const int last = p.M - 1 - m_base;               // rows past M re-read the last row (their outputs are not stored)
#pragma unroll
            for (int i = 0; i < DPW; ++i) {
                const int q = wave * DPW + i;
                const int kt = q / (ROWS / 8), rg = q - kt * (ROWS / 8);
                const int kbase = kt * BK;
                const int row = min(rg * 8 + drow, last);
                const bool second = kbase >= p.K1;
                const unsigned char* base = second ? reinterpret_cast<const unsigned char*>(A2 + (int64_t)m_base * p.lda2 + (kbase - p.K1))
                                                   : reinterpret_cast<const unsigned char*>(A + (int64_t)m_base * p.lda + kbase);
                const unsigned voff = (unsigned)(row * (second ? p.lda2 : p.lda) + dchunk) * 2u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                                 (__attribute__((address_space(3))) void*)(dst + kt * (ROWS * 128) + rg * 1024), 16, 0, 0);
            }
            issued += DPW;
        }
#pragma unroll
        for (int k = 0; k < WS_NSLOT; ++k)
            if (k == s % WS_NSLOT) mark[k] = issued;
    };
    issue_slot(0);
    issue_slot(1);

    // ---- W fragments of this wave's columns, for the whole launch (swapped MFMA: W is the A operand, row = output column)
    bf16x8 wf[2 * NK][TN];
    // (j outer, k inner: the two 64-byte halves of a 128-byte line are requested back to back -- with k outer the second half
    //  came 4 instructions x 8 waves later, after the 32 KB L1 had dropped the line: 10 us instead of 5 for the 320 KB)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int gn = n0 + j * 16 + frow;
        gn = gn < p.N ? gn : p.N - 1;
        const bf16* wrow = W + (int64_t)gn * p.K + fq * 8;
#pragma unroll
        for (int kk = 0; kk < 2 * NK; ++kk) wf[kk][j] = *reinterpret_cast<const bf16x8*>(wrow + kk * 32);
    }
    if (lane < CW) bias_l[lane] = (p.bias && n0 + lane < p.N) ? p.bias[n0 + lane] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // W fragments, bias (and the first two slots) landed: once per launch

    const bool do_cs = (p.epilogue & SEER_EPI_COLSCALE) != 0;
    const int n_out = GEGLU ? (p.N >> 1) : p.N;
    const int n0o = GEGLU ? (n0 >> 1) : n0;
    const bool cols_full = n0o + CWO <= n_out;

    // ---- one 32-row block of the slot: bias into the accumulators, residual request, K loop
    f32x4 acc[2][TN];
    auto k_block = [&](const unsigned char* slot, int h, int m0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + j * 16 + fq * 4);
            acc[0][j] = bv;
            acc[1][j] = bv;
        }
        // A fragments one 32-wide k-step ahead of the MFMAs that use them (two waves per SIMD do not hide an LDS round trip per
        // k-step: without the prefetch the K loop ran at half the MFMA rate, profiles/r02_ws_stamps.log).  The reads are inline
        // asm: hipcc sinks a C++ prefetch back under the MFMAs and waits for it at once.
        const unsigned a0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(slot) + (h * 32 + frow) * 128;
        const unsigned a_ks0 = a0 + ((fq ^ (frow & 7)) * 16), a_ks1 = a0 + (((4 + fq) ^ (frow & 7)) * 16);
        bf16x8 af[2][2];
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048" : "=&v"(af[0][0]), "=&v"(af[0][1]) : "v"(a_ks0) : "memory");
        static_for<0, 2 * NK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t + 1 < 2 * NK) {
                constexpr int off = ((t + 1) >> 1) * (ROWS * 128);
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                             : "=&v"(af[(t + 1) & 1][0]), "=&v"(af[(t + 1) & 1][1])
                             : "v"(((t + 1) & 1) ? a_ks1 : a_ks0), "n"(off), "n"(off + 2048) : "memory");
            }
            // the fragments of step t: everything but the two reads just issued has returned
            if constexpr (t + 1 < 2 * NK) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[t & 1][0]), "+v"(af[t & 1][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[t & 1][0]), "+v"(af[t & 1][1]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mma16<F16>(wf[t][j], af[t & 1][i], acc[i][j]);
        });
    };
    // ---- epilogue of the block at rows m0 (swapped MFMA: the lane holds 4 consecutive columns n = .. + 4 fq + r of row frow)
    auto epilogue = [&](int m0) {
        __builtin_amdgcn_s_setprio(2);                // VALU wave ahead of its SIMD neighbour's MFMAs (the matrix pipe needs one issue slot in four)
        // stores are counted only when the whole 32 x CWO block is in range: then every lane stores and every instruction
        // issues (an uncounted store only makes a later wait stricter)
        const bool full = cols_full && (m0 + 32 <= p.M);
        unsigned char* const cbase = reinterpret_cast<unsigned char*>(Cb + (int64_t)m0 * p.ldc + n0o);   // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned char* const stg = stage + i * (16 * SPITCH);
#pragma unroll
            for (int j = 0; j < TN; j += (GEGLU ? 2 : 1)) {
                const int n = n0 + j * 16 + fq * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
                int jo = j;
                if constexpr (GEGLU) {
                    const f32x2 ge0 = gelu_erf_f2(f32x2{acc[i][j + 1][0], acc[i][j + 1][1]});
                    const f32x2 ge1 = gelu_erf_f2(f32x2{acc[i][j + 1][2], acc[i][j + 1][3]});
                    v[0] *= ge0[0]; v[1] *= ge0[1]; v[2] *= ge1[0]; v[3] *= ge1[1];
                    jo = j >> 1;
                }
                if (!GEGLU && do_cs && n < p.col_scale_cols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= p.col_scale;
                }
                u32x2 o;
                o[0] = pack2t<F16>(v[0], v[1]);
                o[1] = pack2t<F16>(v[2], v[3]);
                *reinterpret_cast<u32x2*>(stg + frow * SPITCH + (jo * 16 + fq * 4) * 2) = o;
            }
            // the wave's 16 rows x CWO columns leave as whole row segments, 16 B per lane (wave-private LDS: no barrier)
            constexpr int CPR = CWO / 8;                 // 16-byte chunks per row
            constexpr int NST = (16 * CPR + 63) / 64;
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int c = lane + 64 * it;
                const int row = c / CPR, ch = c - row * CPR;
                const int mr = m0 + i * 16 + row;
                const int nc = n0o + ch * 8;
                if ((16 * CPR) % 64 == 0 || c < 16 * CPR) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * SPITCH + ch * 16);
                    const unsigned voff = (unsigned)((i * 16 + row) * p.ldc + ch * 8) * 2u;
                    if (mr < p.M && nc < n_out) store16_out(cbase, voff, val);
                }
            }
            if (full && (16 * CPR) % 64 == 0) issued += NST;
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // The two waves of a SIMD (w and w + 4) run half a period apart: waves 4..7 defer the epilogue of their last block past
    // the next barrier, so that while one wave of a SIMD issues MFMAs the other is in its epilogue (VALU, LDS, stores) -- in
    // step, both multiply and then both run the epilogue, and the stamps show each taking as long as the K loop.
    // (waves w and w + 4 of a workgroup share a SIMD: HW_ID read back in the kernel, profiles/r02_ws_stamps.log)
    const int late = wave >= 4 ? 1 : 0;
    bool pending = false;                            // the accumulators hold a block whose epilogue has not run
    int pend_m0 = 0;
    for (int s = 0; s <= nslots; ++s) {
        const bool drain = s == nslots;              // one more pass: the late waves' last epilogue (no barrier: every wave skips it)
        if (!drain) {
            int mk = 0;
#pragma unroll
            for (int k = 0; k < WS_NSLOT; ++k)
                if (k == s % WS_NSLOT) mk = mark[k];
            ws_wait_vmcnt(issued - mk);              // this wave's five pieces of slot s
            __builtin_amdgcn_s_barrier();            // slot s complete; every wave is done with slot s - 1
            issue_slot(s + 2);                       // -> refill the slot of s - 1
        }
        if (!active) continue;
        const unsigned char* slot = ring + (s % WS_NSLOT) * WS_SLOT_BYTES;
        const int m_slot = (s_begin + s) * ROWS;
        // phases of a slot: early waves  K(h0) E(h0) [K(h1) E(h1)],  late waves  E(previous) K(h0) [E(h0) K(h1)]  -- ONE K site
        // and ONE epilogue site in the code (two inlined copies of either cost 30-60 spilled registers)
#pragma unroll 1
        for (int ph = 0; ph < 2 * HALVES; ++ph) {
            if (((ph + late) & 1) == 0) {
                const int m0 = m_slot + (ph >> 1) * 32;
                if (!drain && m0 < p.M) {
                    k_block(slot, ph >> 1, m0);
                    pending = true;
                    pend_m0 = m0;
                }
            } else if (pending) {
                epilogue(pend_m0);
                pending = false;
            }
        }
    }
}

template <int CW, int NK, bool GEGLU, bool F16>
std::once_flag g_ws_lds_once;        // the kernel's dynamic-LDS opt-in has run (namespace scope: no function-local statics)

template <int CW, int NK, bool GEGLU, bool F16 = false>
int ws_launch(const seer_gemm_desc& d, hipStream_t st) {
    constexpr int ROWS = WS_SLOT_BYTES / (NK * WS_BK * 2);
    constexpr int CWO = GEGLU ? CW / 2 : CW;
    const int n_panels = (d.N + 8 * CW - 1) / (8 * CW);
    const int slots_m = (d.M + ROWS - 1) / ROWS;
    int wpp = WS_NUM_CU / n_panels;
    if (wpp < 1) return SEER_ENOSYS;
    if (wpp > slots_m) wpp = slots_m;
    const size_t lds = (size_t)WS_NSLOT * WS_SLOT_BYTES + 8 * (32 * (CWO * 2 + 16) + CW * 4);
    std::call_once(g_ws_lds_once<CW, NK, GEGLU, F16>, [lds] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&seer_gemm_ws_kernel<CW, NK, GEGLU, F16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const int grid = 8 * ((n_panels * wpp + 7) / 8);
    hipLaunchKernelGGL((seer_gemm_ws_kernel<CW, NK, GEGLU, F16>), dim3(grid), dim3(512), lds, st, d, n_panels, wpp);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

}  // namespace

// Can this (validated, normalised) descriptor run on the weight-stationary kernel?  Called by seer_gemm_bf16 (gemm.hip).
// Built for the two contraction lengths of the upper UNet levels: K = 320 (64 columns per wave) and K = 640 (32 columns per
// wave); either way a wave holds 160 registers of W fragments.
bool seer_gemm_ws_eligible(const seer_gemm_desc& d) {
    if (d.mode != SEER_GEMM_PLAIN || d.batch > 1) return false;
    // (rotary: its table lookups are ordinary loads whose first use would drain the LDS-DMA ring -- the tile kernel keeps those)
    if (d.epilogue & (SEER_EPI_OUT_F32 | SEER_EPI_SILU | SEER_EPI_TRANS_OUT | SEER_EPI_ROTARY)) return false;
    if (d.rowvec) return false;
    if (d.K != 320 && d.K != 640) return false;
    if (d.M < 1024) return false;                                   // too few rows to stream
    if (d.ldc % 8 || (reinterpret_cast<uintptr_t>(d.C) & 15)) return false;
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (d.N % (geglu ? 64 : 8)) return false;
    if (d.residual) return false;     // (none of the wide short-K projections has one; its registers are better spent on W)
    const int cw = d.K == 320 ? 64 : 32;
    if ((d.N + 8 * cw - 1) / (8 * cw) > WS_NUM_CU) return false;
    return true;
}

// Is it also the faster kernel?  Measured on the shapes of a denoising step (profiles/r02_lab_gemm_ws8.log): yes for the wide
// K = 320 projections (level-0 GEGLU ff.net.0 79 -> 60 us, q|k|v 37 -> 30 us); a tie at K = 640, where a wave covers only 32
// columns; slower for N <= 640, where the 5-10 us it takes a CU to pull its W panel is not amortised over enough rows.
// N = 960 (the q|k|v projection of the 320-wide level) went to the 160-wide tiles once their epilogue got cheap: 26.4 vs 32.7 us
// (profiles/r02_tile_sweep_fastepi.log); the GEGLU projection (N = 2560) stays here, 58.7 vs 65.5
bool seer_gemm_ws_profitable(const seer_gemm_desc& d) { return d.K == 320 && d.N >= 1280 && d.M >= 8192; }

int seer_gemm_ws_launch(const seer_gemm_desc& d, hipStream_t st) {
    const bool geglu = (d.epilogue & SEER_EPI_GEGLU) != 0;
    if (d.epilogue & SEER_EPI_F16) {          // IEEE-half operands (the fp16 engine): the same kernel, other MFMA opcode and pack
        if (d.K == 320) return geglu ? ws_launch<64, 5, true, true>(d, st) : ws_launch<64, 5, false, true>(d, st);
        return geglu ? ws_launch<32, 10, true, true>(d, st) : ws_launch<32, 10, false, true>(d, st);
    }
    if (d.K == 320) return geglu ? ws_launch<64, 5, true>(d, st) : ws_launch<64, 5, false>(d, st);
    return geglu ? ws_launch<32, 10, true>(d, st) : ws_launch<32, 10, false>(d, st);
}
