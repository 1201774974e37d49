// Flash-style attention forward for gfx950 (MI355X): head_dim 40 / 80 / 160 (UNet) and 96 (FSTextTransformer), bf16 in/out.
//
// One kernel serves the three attention sites of the Seer UNet (include/seer_hip.h, seer_attn_fwd):
//   spatial self-attention  [B*F*8, HW, d]  x [.., HW, d]      non-causal
//   text cross-attention    [B*F*8, HW, d]  x [.., 77, d]      non-causal, Sk tail masked
//   temporal window attn    [nW*B*8, F*ws^2, d]                causal (LowerTriangularMask), window gather folded
//                                                              into the token addressing
//
// Structure (v1): 256 threads = 4 waves, each wave owns 32 queries; the block shares 64-key K/V tiles in LDS.
//   S^T = K Q^T  by v_mfma_f32_32x32x16_bf16 with K as the A operand (rows = keys) and Q^T as the B operand, so a lane
//   holds ONE query column: the online-softmax max/sum are in-lane over 16 registers plus one exchange with lane^32.
//   P^T (the accumulator) is converted to bf16 in place and used directly as the B operand of
//   O^T += V^T P^T  (accumulator-as-operand, k order 16s + 8(j>>2) + 4h + (j&3)); the V^T A-fragments come from the
//   row-major [key][d] LDS image through ds_read_b64_tr_b16 (hardware transpose), so V is never transposed in memory.
//   No S x S matrix, no P round trip through LDS.
// LDS images: K rows padded to an odd number of 16-byte chunks (conflict-free ds_read_b128 by key row);
//             V row stride == 64 or 192 (mod 256) bytes (conflict-free transposed reads of 4 key rows).
#include "seer_common.h"

int seer_attn40_launch(const seer_attn_desc& d, int ws_log2, hipStream_t st);   // attention40.hip

namespace {

// measurement-only builds (scripts/probe_attn.sh; results are WRONG): bit 0 = no v_exp (the scaled score is used as is),
// bit 1 = no PV MFMAs, bit 2 = no QK^T MFMAs, bit 3 = K/V tile staged once (no per-tile loads / commits / barriers).
#ifndef SEER_ATTN_PROBE
#define SEER_ATTN_PROBE 0
#endif

// head dims up to this value issue both QK^T products of an LDS tile before the first softmax (see the K loop).
// Measured on MI355X (profiles/r01_attention_pipe.log): 5-8 % SLOWER at d = 40 -- the second score tile costs 30 VGPRs and
// with them the fourth wave per SIMD, and the waves of a SIMD already overlap each other's MFMA and VALU phases -> off.
#ifndef SEER_ATTN_PIPE_MAXD
#define SEER_ATTN_PIPE_MAXD 0
#endif

constexpr int KT = 64;            // keys per LDS tile
constexpr float kNegInf = -__builtin_inff();

template <int D>
struct AttnCfg {
    static constexpr int DP = (D + 15) / 16 * 16;      // contraction length of QK^T (zero padded)
    static constexpr int KSTEPS = DP / 16;
    static constexpr int NDT = (D + 31) / 32;          // 32-row d tiles of O^T
    static constexpr int DV = NDT * 32;
    static constexpr int KRS = DP + 8;                 // K row stride (elements): odd number of 16-B chunks
    static constexpr int VRS = DV < 96 ? 96 : DV;      // V row stride (elements): 192 B or 320 B
    static constexpr int CH = D / 8;                   // 16-byte chunks per row in global memory
    static constexpr int NCH = (KT * CH + 255) / 256;  // chunks per thread per tile (K or V)
    static constexpr int BUF = KT * (KRS + VRS);          // elements per K|V buffer (two buffers: ping-pong)
    static constexpr size_t LDS_BYTES = (size_t)2 * BUF * 2;
};

// v_permlane32_swap a, b:  a' = [a.lo | b.lo], b' = [a.hi | b.hi]; with a = b = v every lane sees both halves' values.
// Inline asm, not __builtin_amdgcn_permlane32_swap: hipcc (ROCm 7.2) returns the builtin's FIRST result for both elements of
// its result pair (scripts/lab_probe3.cpp prints it), so max(r[0], r[1]) silently dropped the other half's 16 keys from the
// running maximum.  Mild scores hide that (any reference near the maximum is fine); scores 130+ log2 units apart
// overflowed exp2 and gave NaN rows -- found by the sharp-softmax cases of scripts/lab_attn.cpp / tests/test_gpu_kernels.py.
// The s_nop pair covers the VALU-write -> permlane-read hazard, which nobody pads inside an asm statement.
__device__ __forceinline__ float xhalf_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float lower_half_value(float v) {      // value of lane (l & 31) in every lane l
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a;
}

struct TokMap {
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

// DBUF: ping-pong K|V LDS buffers (one barrier per tile) vs one buffer (two barriers, half the LDS).
// defer_thr: the running max of a query is only raised when a tile's max exceeds it by more than defer_thr (log2 units), so
//            p <= 2^defer_thr instead of <= 1 and the O rescale (skipped when no lane moved its max) becomes rare; bf16 P keeps
//            its relative precision at any magnitude and l/O accumulate in fp32, so the result is unchanged to rounding.
// F16 (SEER_ATTN_F16): Q, K, V and O hold IEEE half instead of bf16 -- the same 16-bit loads, LDS images and transposed reads (the
// bf16 vector types below are containers of bits); only the MFMA opcode, the constant 1.0 of the denominator column, the conversion
// of P and the output pack differ.  P <= 2^defer_thr = 16 is far inside the half range; probabilities below 6e-8 flush to zero.
template <int D, bool DBUF, bool F16 = false>
__global__ void __launch_bounds__(256) seer_attn_kernel(const seer_attn_desc p, const int ws_log2, const float defer_thr) {
    using C = AttnCfg<D>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* const lds = reinterpret_cast<bf16*>(smem);    // 2 x { K [KT][KRS], V [KT][VRS] }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: tile / mask decisions are wave-uniform branches
    const int lq = lane & 31, lh = lane >> 5;
    // D = 40 / 80 leave unused rows in the last 32-row d tile of O^T: V's first pad column holds 1.0, so row D of O^T is
    // sum_k P[k][q] -- the softmax denominator comes out of the PV MFMA (rescaled with O) instead of 16 VALU adds per tile
    constexpr bool LSUM_MFMA = C::DV > D;
    constexpr int L_TT = D / 32, L_ROW = D % 32;                     // d tile / row inside it that holds the denominator
    constexpr int L_REG = (L_ROW & 3) + 4 * (L_ROW >> 3);            // 32x32 layout: row = (reg&3) + 8*(reg>>2) + 4*lh
    static_assert(!LSUM_MFMA || ((L_ROW & 4) == 0), "denominator row must live in the lh = 0 half");

    // ---- batch decode
    const int y = blockIdx.y;
    const int head = y % p.heads;
    int b = y / p.heads;
    TokMap tok;
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

    const int qblk0 = blockIdx.x * 128;
    const int q0 = qblk0 + wave * 32;             // first query of this wave
    const bool wave_active = q0 < p.Sq;
    const int qi = q0 + lq;                       // this lane's query (sequence position)
    const int qi_c = qi < p.Sq ? qi : p.Sq - 1;

    // ---- zero the K pad columns once (D=40: elements 40..47 take part in the contraction)
    if constexpr (C::DP != D) {
        for (int r = tid; r < (DBUF ? 2 : 1) * KT; r += 256) {
            bf16* kb_ = lds + (r / KT) * C::BUF + (r % KT) * C::KRS + (C::DP - 8);
            *reinterpret_cast<u32x4*>(kb_) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // ---- V pad column D = 1.0 (never overwritten: the tile commits only write columns < D)
    if constexpr (LSUM_MFMA) {
        for (int r = tid; r < (DBUF ? 2 : 1) * KT; r += 256)
            lds[(r / KT) * C::BUF + KT * C::KRS + (r % KT) * C::VRS + D] = to16<F16>(1.0f);
    }

    // ---- Q fragments (B operand: col = query, k = 8h + j inside each 16-wide step)
    bf16x8 qf[C::KSTEPS];
    {
        const bf16* qrow = Qg + (int64_t)tok(qi_c) * p.q_ss;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int e0 = 16 * s + 8 * lh;
            if (e0 + 8 <= D) {
                qf[s] = *reinterpret_cast<const bf16x8*>(qrow + e0);
            } else {
                u32x4 zz = {0u, 0u, 0u, 0u};
                qf[s] = __builtin_bit_cast(bf16x8, zz);
            }
        }
    }

    f32x16 oacc[C::NDT];
#pragma unroll
    for (int t = 0; t < C::NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float m_run = kNegInf;      // running max (log2 domain, already scaled)
    float l_run = 0.f;          // partial row sum over this lane-half's keys
    // SEER_ATTN_Q_PRESCALED: q already carries scale * log2(e)
    const float cscale = (p.flags & SEER_ATTN_Q_PRESCALED) ? 1.0f : p.scale * 1.4426950408889634f;

    // keys this block needs: causal -> up to the last query of the block
    // causal: key j is visible to query i iff j <= i + q_off (q_off = sequence position of this shard's first query)
    const int q_off = p.causal_offset;
    int k_end = p.Sk;
    if (p.causal) {
        const int lastq = min(qblk0 + 127, p.Sq - 1);
        k_end = min(p.Sk, lastq + q_off + 1);
    }
    const int ntiles = (k_end + KT - 1) / KT;

    u32x4 kreg[C::NCH], vreg[C::NCH];
    auto prefetch = [&](int t) {
        const int kt0 = t * KT;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KT * C::CH) {
                const int key = idx / C::CH, ch = idx - key * C::CH;
                int kg = kt0 + key;
                kg = kg < p.Sk ? kg : p.Sk - 1;
                const int64_t tk = tok(kg);
                kreg[i] = *reinterpret_cast<const u32x4*>(Kg + tk * p.k_ss + ch * 8);
                vreg[i] = *reinterpret_cast<const u32x4*>(Vg + tk * p.v_ss + ch * 8);
            }
        }
    };
    auto commit = [&](int buf) {
        bf16* Ks = lds + buf * C::BUF;
        bf16* Vs = Ks + KT * C::KRS;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KT * C::CH) {
                const int key = idx / C::CH, ch = idx - key * C::CH;
                *reinterpret_cast<u32x4*>(Ks + key * C::KRS + ch * 8) = kreg[i];
                *reinterpret_cast<u32x4*>(Vs + key * C::VRS + ch * 8) = vreg[i];
            }
        }
    };

    // ping-pong: tile t is computed from buffer t&1 while tile t+1 travels global -> registers; it is written to the other
    // buffer after the compute (its last readers passed the previous barrier).  One barrier per tile.
    prefetch(0);
    if constexpr (DBUF) {
        commit(0);
        __syncthreads();
    }

    for (int t = 0; t < ntiles; ++t) {
#if SEER_ATTN_PROBE & 8
        if (t == 0) { commit(0); __syncthreads(); }
#else
        if constexpr (!DBUF) {
            __syncthreads();          // every wave finished reading the previous tile
            commit(0);
            __syncthreads();
        }
        if (t + 1 < ntiles) prefetch(t + 1);
#endif
        const bf16* Ks = lds + (DBUF ? (t & 1) : 0) * C::BUF;
        const bf16* Vs = Ks + KT * C::KRS;
        const int kt0 = t * KT;
        // The two 32-key sub tiles of the LDS tile, one after the other -- or (SEER_ATTN_PIPE_MAXD, off by default) both QK^T
        // products first so that the matrix pipe works on sub tile 1 under the softmax of sub tile 0.
        auto sub_valid = [&](int sub) {
            const int kb = kt0 + sub * 32;                 // first key of this 32-key sub tile
            // wave-uniform: past the last key / fully above the causal diagonal
            return wave_active && kb < k_end && !(p.causal && kb > q0 + 31 + q_off);
        };
        auto qk = [&](int sub) {
            // ---- S^T = K Q^T  (rows = keys, cols = queries)
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const bf16* krow = Ks + (sub * 32 + lq) * C::KRS + 8 * lh;
            bf16x8 kf[C::KSTEPS];            // all K fragments first: one exposed LDS latency per sub tile, not one per step
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(krow + 16 * s);
#if SEER_ATTN_PROBE & 4
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s)
#pragma unroll
                for (int r = 0; r < 8; ++r) sacc[r] += (float)kf[s][r] + (float)qf[s][r];
#else
#pragma unroll
            for (int s = 0; s < C::KSTEPS; ++s) sacc = mma32<F16>(kf[s], qf[s], sacc);
#endif

            return sacc;
        };
        auto softmax_pv = [&](int sub, f32x16 sacc) {
            const int kb = kt0 + sub * 32;
            // ---- scale (+ mask on boundary tiles)
            const bool need_mask = (kb + 31 >= p.Sk) || (p.causal && (kb + 31 > q0 + q_off));
            // raw scores stay unscaled: max in the raw domain (cscale > 0), then p = exp2(fma(s, cscale, -m)) -- one FMA
            // and one v_exp per score
            float mx = kNegInf;
            if (need_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = (key < p.Sk) && (!p.causal || key <= qi + q_off);
                    const float x = ok ? sacc[r] : kNegInf;
                    sacc[r] = x;
                    mx = fmaxf(mx, x);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            }
            mx = xhalf_max(mx) * cscale;      // the other 16 keys of this query live in lane ^ 32
            const float m_new = (mx > m_run + defer_thr) ? mx : m_run;
            const float m_use = (m_new == kNegInf) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run == -inf -> 0
            m_run = m_new;
            {
                const f32x2 c2 = {cscale, cscale}, nm2 = {-m_use, -m_use};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {       // v_pk_fma_f32: two scores per instruction
                    const f32x2 t = __builtin_elementwise_fma(f32x2{sacc[r], sacc[r + 1]}, c2, nm2);
#if SEER_ATTN_PROBE & 1
                    sacc[r] = t[0];
                    sacc[r + 1] = t[1];
#else
                    sacc[r] = __builtin_amdgcn_exp2f(t[0]);
                    sacc[r + 1] = __builtin_amdgcn_exp2f(t[1]);
#endif
                }
            }
            if constexpr (!LSUM_MFMA) {
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += sacc[r];
                l_run = l_run * alpha + psum;
            }
            // the running max of most queries stops moving after the first tiles: skip the O rescale when no lane needs it
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int tt = 0; tt < C::NDT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[tt][r] *= alpha;
            }

            // ---- P^T -> bf16 B fragments (k-step s2 uses accumulator registers 8*s2 .. 8*s2+7)
            bf16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[s2][j] = to16<F16>(sacc[8 * s2 + j]);

            // ---- O^T += V^T P^T ; V^T fragments by transposed LDS reads
            // lane group of 16: i = lane & 15 -> (q = i >> 2: key row of the 4x16 block, pcol = i & 3: 4-column group)
            const int li = lane & 15;
            const int g16 = (lane >> 4) & 1;
            const bf16* vbase = Vs + (sub * 32 + 4 * lh + (li >> 2)) * C::VRS + 16 * g16 + 4 * (li & 3);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int tt = 0; tt < C::NDT; ++tt) {
                    const bf16* a0 = vbase + (16 * s2) * C::VRS + 32 * tt;
                    const bf16* a1 = a0 + 8 * C::VRS;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(a0));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(a1));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#if SEER_ATTN_PROBE & 2
                    oacc[tt][s2] += (float)vf[0] + (float)pf[s2][tt];
#else
                    oacc[tt] = mma32<F16>(vf, pf[s2], oacc[tt]);
#endif
                }
            }
        };
        const bool v0 = sub_valid(0), v1 = sub_valid(1);
        if constexpr (D <= SEER_ATTN_PIPE_MAXD) {
            f32x16 s0, s1;
            if (v0) s0 = qk(0);
            if (v1) s1 = qk(1);
            if (v0) softmax_pv(0, s0);
            if (v1) softmax_pv(1, s1);
        } else {                       // larger head dims: the second score tile costs a wave of occupancy
            if (v0) softmax_pv(0, qk(0));
            if (v1) softmax_pv(1, qk(1));
        }
        if constexpr (DBUF) {
            if (t + 1 < ntiles) commit((t + 1) & 1);
            __syncthreads();
        }
    }

    // ---- finalize: O[q][d] = O^T[d][q] / l
    if (!wave_active) return;
    float l_tot;
    if constexpr (LSUM_MFMA) l_tot = lower_half_value(oacc[L_TT][L_REG]);   // row D of O^T sits in the lh = 0 lanes
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    // training: log2-domain log-sum-exp of the scaled scores, read back by seer_attn_bwd
    if (p.lse && lh == 0 && qi < p.Sq) p.lse[(int64_t)y * p.Sq + qi] = m_run + __builtin_amdgcn_logf(l_tot);
    if (qi < p.Sq) {
        bf16* orow = Og + (int64_t)tok(qi) * p.o_ss;
#pragma unroll
        for (int tt = 0; tt < C::NDT; ++tt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * tt + 8 * g + 4 * lh;
                if (d0 < D) {
                    u32x2 o;
                    if constexpr (F16) {
                        o[0] = pack2h(oacc[tt][4 * g + 0] * inv, oacc[tt][4 * g + 1] * inv);
                        o[1] = pack2h(oacc[tt][4 * g + 2] * inv, oacc[tt][4 * g + 3] * inv);
                    } else {
                        o[0] = pack2(oacc[tt][4 * g + 0] * inv, oacc[tt][4 * g + 1] * inv);
                        o[1] = pack2(oacc[tt][4 * g + 2] * inv, oacc[tt][4 * g + 3] * inv);
                    }
                    *reinterpret_cast<u32x2*>(orow + d0) = o;
                }
            }
        }
    }
}

template <int D, bool DBUF, bool F16 = false>
int launch_attn2(const seer_attn_desc& d, int ws_log2, float thr, hipStream_t st) {
    int nbatch = d.batch;
    if (ws_log2 >= 0) nbatch *= (d.H >> ws_log2) * (d.W >> ws_log2);
    dim3 grid((d.Sq + 127) / 128, nbatch * d.heads, 1);
    constexpr size_t lds = AttnCfg<D>::LDS_BYTES / (DBUF ? 1 : 2);
    static_assert(lds <= 64 * 1024, "above the default dynamic-LDS limit: would need a hipFuncSetAttribute opt-in");
    hipLaunchKernelGGL((seer_attn_kernel<D, DBUF, F16>), grid, dim3(256), lds, st, d, ws_log2, thr);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

// measured on one MI355X (profiles/r01_attention_ab.log): deferred max +7-8 %; ping-pong LDS within noise -> the single
// buffer (half the LDS) and a threshold of 4 are the defaults; desc.variant = 6 selects the ping-pong form for A/B runs
template <int D>
int launch_attn(const seer_attn_desc& d, int ws_log2, hipStream_t st) {
    constexpr float thr = 4.0f;
    if (d.flags & SEER_ATTN_F16) return launch_attn2<D, false, true>(d, ws_log2, thr, st);      // IEEE-half operands: the generic kernel
    if constexpr (AttnCfg<D>::LDS_BYTES <= 64 * 1024) {
        if (d.variant == 6) return launch_attn2<D, true>(d, ws_log2, thr, st);
    }
    return launch_attn2<D, false>(d, ws_log2, thr, st);
}

}  // namespace

extern "C" int seer_attn_fwd(const seer_attn_desc* desc, void* stream) {
    if (!desc) return SEER_EINVAL;
    const seer_attn_desc d = *desc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!d.Q || !d.K || !d.V || !d.O) return SEER_EINVAL;
    if (d.batch <= 0 || d.heads <= 0 || d.Sq <= 0 || d.Sk <= 0) return SEER_EINVAL;
    if ((d.q_ss | d.k_ss | d.v_ss) % 8 || d.o_ss % 4) return SEER_EINVAL;
    if ((d.q_bs | d.k_bs | d.v_bs) % 8 || d.o_bs % 4) return SEER_EINVAL;
    if ((d.q_hs | d.k_hs | d.v_hs) % 8 || d.q_hs < 0 || d.k_hs < 0 || d.v_hs < 0) return SEER_EINVAL;
    int ws_log2 = -1;
    if (d.window_ws > 0) {
        if (d.window_ws != 4 && d.window_ws != 8) return SEER_EINVAL;
        ws_log2 = d.window_ws == 4 ? 2 : 3;
        if (d.H % d.window_ws || d.W % d.window_ws) return SEER_EINVAL;
        if (d.Fq <= 0 || d.Sq != d.Fq * d.window_ws * d.window_ws || d.Sk != d.F * d.window_ws * d.window_ws) return SEER_EINVAL;
    }
    if (d.variant < 0 || d.variant > 7 || d.variant == 4) return SEER_EINVAL;
    if ((d.variant == 2 || d.variant == 3 || d.variant == 5 || d.variant == 7) && d.head_dim != 40) return SEER_EINVAL;
    if (d.causal_offset < 0 || (d.causal && d.Sq + d.causal_offset > d.Sk)) return SEER_EINVAL;
    if ((d.flags & SEER_ATTN_F16) && (d.lse || (d.variant != 0 && d.variant != 1 && d.variant != 5))) return SEER_EINVAL;      // inference
    switch (d.head_dim) {
        case 40:
            // IEEE-half operands: the d = 40 kernel's TRACKED form from 256 keys up (its fast path lives off bf16's exponent range),
            // the generic kernel below that and for variant 1
            if (d.flags & SEER_ATTN_F16)
                return (d.variant == 1 || (d.variant == 0 && d.Sk < 256)) ? launch_attn<40>(d, ws_log2, st) : seer_attn40_launch(d, ws_log2, st);
            // the d = 40 kernel pays ~3 us of set-up (constant region, LDS-DMA plan, reference pre-pass) that short key
            // sequences do not earn back (text cross-attention, Sk = 77: 18.7 vs 16.6 us, profiles/r02_attn40_variants.log)
            if (d.variant == 1 || d.variant == 6 || (d.variant == 0 && d.Sk < 256)) return launch_attn<40>(d, ws_log2, st);
            // training forward (lse requested) on an UN-prescaled q: the d = 40 kernel would round q * scale * log2(e) to bf16
            // before Q K^T, and its lse would belong to those rounded scores, while seer_attn_bwd rebuilds P = exp2(q.k * scale *
            // log2(e) - lse) from the unrounded q in fp32 -- probabilities that no longer sum to 1 (~1 % off).  The generic kernel
            // scales the fp32 scores, exactly as the backward does
            if (d.variant == 0 && d.lse && !(d.flags & SEER_ATTN_Q_PRESCALED)) return launch_attn<40>(d, ws_log2, st);
            return seer_attn40_launch(d, ws_log2, st);
        case 80: return launch_attn<80>(d, ws_log2, st);
        case 96: return launch_attn<96>(d, ws_log2, st);      // FSTextTransformer (768 channels, 8 heads)
        case 160: return launch_attn<160>(d, ws_log2, st);
        default: return SEER_ENOSYS;
    }
}
