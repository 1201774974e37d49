// Attention backward for gfx950 (MI355X): the training step's counterpart of attention.hip (SURVEY 8(f) rank 1;
// train.py:380-381 back-propagates through xformers.ops.memory_efficient_attention at attention.py:622-630, 632-703).
//
// Two launches of ONE kernel template, no float atomics (deterministic):
//   MODE 0 (dQ):    a block owns 128 queries and streams 64-key K|V tiles     -> dQ, and delta[q] = <dO[q], O[q]>
//   MODE 1 (dK|dV): a block owns 128 keys    and streams 64-query Q|dO tiles  -> dK, dV
// With "own" rows i (one per lane column, fragments in registers) and streamed rows j (LDS image [j][d]):
//   T1^T[j][i] = <A1[j], own1[i]>      MODE 0: A1 = K, own1 = Q  (S^T)      MODE 1: A1 = Q,  own1 = K  (S)
//   T2^T[j][i] = <A2[j], own2[i]>      MODE 0: A2 = V, own2 = dO (dP^T)     MODE 1: A2 = dO, own2 = V  (dP)
//   P  = exp2(T1 * scale * log2 e - lse2[query])           (lse2 = the forward's log2-domain log-sum-exp)
//   dS = P * (T2 - delta[query]) * scale
//   acc1^T[d][i] += sum_j A1[j][d] dS[j][i]      MODE 0: dQ^T = K^T dS^T      MODE 1: dK^T = Q^T dS
//   acc2^T[d][i] += sum_j A2[j][d] P[j][i]       MODE 1 only: dV^T = dO^T P
// The T products use v_mfma_f32_32x32x16_bf16 with the streamed rows as the A operand (row reads from LDS) and the own
// fragments as B; P / dS stay in the accumulator registers and become the B operand of the second pair of products
// (accumulator-as-operand, as in the forward kernel), whose A operand A^T comes from the same [j][d] LDS image through
// ds_read_b64_tr_b16.
#include "seer_common.h"

namespace {

constexpr int KT = 64;    // streamed rows per LDS tile

template <int D>
struct BwdCfg {
    static constexpr int DP = (D + 15) / 16 * 16;
    static constexpr int KSTEPS = DP / 16;
    static constexpr int NDT = (D + 31) / 32;
    static constexpr int DV = NDT * 32;
    static constexpr int RS = DP + 8;                  // row stride (elements): odd number of 16-byte chunks
    static constexpr int CH = D / 8;
    static constexpr int NCH = (KT * CH + 255) / 256;
    static constexpr int IMG = KT * RS + 64;           // one operand image (+ slack: the transposed reads of the last rows
                                                       // touch columns up to DV, whose products are discarded)
    static constexpr size_t LDS_BYTES = (size_t)2 * IMG * 2 + 2 * KT * sizeof(float);
};

struct TokMapB {
    int ws_log2, HW, W_, wy0, wx0;
    __device__ __forceinline__ int operator()(int pos) const {
        if (ws_log2 < 0) return pos;
        const int ws2 = 2 * ws_log2;
        const int f = pos >> ws2;
        const int rem = pos & ((1 << ws2) - 1);
        const int wy = rem >> ws_log2, wx = rem & ((1 << ws_log2) - 1);
        return f * HW + (wy0 + wy) * W_ + wx0 + wx;
    }
};

template <int D, int MODE>
__global__ void __launch_bounds__(256) seer_attn_bwd_kernel(const seer_attn_bwd_desc pd, const int ws_log2) {
    using C = BwdCfg<D>;
    const seer_attn_desc& p = pd.fwd;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* const img1 = reinterpret_cast<bf16*>(smem);
    bf16* const img2 = img1 + C::IMG;
    float* const Ls = reinterpret_cast<float*>(img2 + C::IMG);     // MODE 1: lse2 / delta of the streamed queries
    float* const Ds = Ls + KT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;

    const int y = blockIdx.y;
    const int head = y % p.heads;
    int b = y / p.heads;
    TokMapB tok;
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
    const bf16* __restrict__ Qg = reinterpret_cast<const bf16*>(p.Q) + (int64_t)b * p.q_bs + head * D;
    const bf16* __restrict__ Kg = reinterpret_cast<const bf16*>(p.K) + (int64_t)b * p.k_bs + head * D;
    const bf16* __restrict__ Vg = reinterpret_cast<const bf16*>(p.V) + (int64_t)b * p.v_bs + head * D;
    const bf16* __restrict__ Og = reinterpret_cast<const bf16*>(p.O) + (int64_t)b * p.o_bs + head * D;
    const bf16* __restrict__ dOg = reinterpret_cast<const bf16*>(pd.dO) + (int64_t)b * pd.do_bs + head * D;
    const float* __restrict__ lse_y = p.lse + (int64_t)y * p.Sq;
    float* __restrict__ delta_y = pd.delta + (int64_t)y * p.Sq;

    const int own_n = MODE == 0 ? p.Sq : p.Sk;
    const int str_n = MODE == 0 ? p.Sk : p.Sq;
    const int blk0 = blockIdx.x * 128;
    const int i0 = blk0 + wave * 32;
    const bool wave_active = i0 < own_n;
    const int ii = i0 + lq;
    const int ii_c = ii < own_n ? ii : own_n - 1;
    const int q_off = p.causal_offset;

    // ---- zero the contraction pad columns of both images once (the tile commits only write columns < D)
    if constexpr (C::DP != D) {
        for (int r = tid; r < 2 * KT; r += 256) {
            bf16* row = (r < KT ? img1 + r * C::RS : img2 + (r - KT) * C::RS) + (C::DP - 8);
            *reinterpret_cast<u32x4*>(row) = u32x4{0u, 0u, 0u, 0u};
        }
    }

    // ---- own fragments (B operand: column = own row, k = 8*lh + j inside each 16-wide step)
    bf16x8 own1[C::KSTEPS], own2[C::KSTEPS];
    float Li = 0.f, Di = 0.f;
    {
        const int64_t trow = tok(ii_c);
        const bf16* r1 = MODE == 0 ? Qg + trow * p.q_ss : Kg + trow * p.k_ss;
        const bf16* r2 = MODE == 0 ? dOg + trow * pd.do_ss : Vg + trow * p.v_ss;
        const bf16* ro = Og + trow * p.o_ss;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < C::KSTEPS; ++s) {
            const int e0 = 16 * s + 8 * lh;
            if (e0 + 8 <= D) {
                own1[s] = *reinterpret_cast<const bf16x8*>(r1 + e0);
                own2[s] = *reinterpret_cast<const bf16x8*>(r2 + e0);
                if constexpr (MODE == 0) {
                    const bf16x8 o = *reinterpret_cast<const bf16x8*>(ro + e0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) part = fmaf((float)o[j], (float)own2[s][j], part);
                }
            } else {
                const u32x4 zz = {0u, 0u, 0u, 0u};
                own1[s] = __builtin_bit_cast(bf16x8, zz);
                own2[s] = __builtin_bit_cast(bf16x8, zz);
            }
        }
        if constexpr (MODE == 0) {
            Di = part + __shfl_xor(part, 32, 64);
            Li = lse_y[ii_c];
            if (lh == 0 && ii < p.Sq) delta_y[ii] = Di;
        }
    }

    f32x16 acc1[C::NDT], acc2[MODE == 1 ? C::NDT : 1];
#pragma unroll
    for (int t = 0; t < C::NDT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 0; t < C::NDT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    }
    const float cscale = p.scale * 1.4426950408889634f;

    // ---- streamed range
    int j_begin = 0, j_end = str_n;
    if (p.causal) {
        if constexpr (MODE == 0) {
            const int lastq = min(blk0 + 127, p.Sq - 1);
            j_end = min(p.Sk, lastq + q_off + 1);
        } else {
            j_begin = max(0, blk0 - q_off) / KT * KT;       // the first query that sees the block's first key
        }
    }
    const int t_begin = j_begin / KT, t_end = (j_end + KT - 1) / KT;

    u32x4 r1reg[C::NCH], r2reg[C::NCH];
    float lreg = 0.f, dreg = 0.f;
    auto prefetch = [&](int t) {
        const int j0 = t * KT;
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KT * C::CH) {
                const int row = idx / C::CH, ch = idx - row * C::CH;
                int jg = j0 + row;
                jg = jg < str_n ? jg : str_n - 1;
                const int64_t tj = tok(jg);
                if constexpr (MODE == 0) {
                    r1reg[i] = *reinterpret_cast<const u32x4*>(Kg + tj * p.k_ss + ch * 8);
                    r2reg[i] = *reinterpret_cast<const u32x4*>(Vg + tj * p.v_ss + ch * 8);
                } else {
                    r1reg[i] = *reinterpret_cast<const u32x4*>(Qg + tj * p.q_ss + ch * 8);
                    r2reg[i] = *reinterpret_cast<const u32x4*>(dOg + tj * pd.do_ss + ch * 8);
                }
            }
        }
        if constexpr (MODE == 1) {
            if (tid < KT) {
                int jg = j0 + tid;
                jg = jg < str_n ? jg : str_n - 1;
                lreg = lse_y[jg];
                dreg = delta_y[jg];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int idx = tid + 256 * i;
            if (idx < KT * C::CH) {
                const int row = idx / C::CH, ch = idx - row * C::CH;
                *reinterpret_cast<u32x4*>(img1 + row * C::RS + ch * 8) = r1reg[i];
                *reinterpret_cast<u32x4*>(img2 + row * C::RS + ch * 8) = r2reg[i];
            }
        }
        if constexpr (MODE == 1) {
            if (tid < KT) {
                Ls[tid] = lreg;
                Ds[tid] = dreg;
            }
        }
    };

    if (t_begin < t_end) prefetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        commit();
        __syncthreads();
        if (t + 1 < t_end) prefetch(t + 1);
        const int j0 = t * KT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + sub * 32;
            bool valid = wave_active && jb < j_end;
            if (p.causal) {
                if constexpr (MODE == 0) valid = valid && !(jb > i0 + 31 + q_off);        // keys all after the wave's queries
                else valid = valid && !(jb + 31 + q_off < i0);                            // queries all before the wave's keys
            }
            if (!valid) continue;
            // ---- T1, T2
            f32x16 t1, t2;
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[r] = t2[r] = 0.f;
            {
                const bf16* arow = img1 + (sub * 32 + lq) * C::RS + 8 * lh;
                bf16x8 af[C::KSTEPS];
#pragma unroll
                for (int s = 0; s < C::KSTEPS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
#pragma unroll
                for (int s = 0; s < C::KSTEPS; ++s) t1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], own1[s], t1, 0, 0, 0);
            }
            {
                const bf16* arow = img2 + (sub * 32 + lq) * C::RS + 8 * lh;
                bf16x8 af[C::KSTEPS];
#pragma unroll
                for (int s = 0; s < C::KSTEPS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(arow + 16 * s);
#pragma unroll
                for (int s = 0; s < C::KSTEPS; ++s) t2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], own2[s], t2, 0, 0, 0);
            }
            // ---- P and dS (rows = streamed j, column = this lane's own row)
            bool need_mask;
            if constexpr (MODE == 0) need_mask = (jb + 31 >= p.Sk) || (p.causal && (jb + 31 > i0 + q_off));
            else need_mask = (jb + 31 >= p.Sq) || (p.causal && (jb + q_off < i0 + 31));
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float L, Dl;
                if constexpr (MODE == 0) { L = Li; Dl = Di; }
                else { L = Ls[sub * 32 + jr]; Dl = Ds[sub * 32 + jr]; }
                float pr = __builtin_amdgcn_exp2f(fmaf(t1[r], cscale, -L));
                if (need_mask) {
                    const int j = jb + jr;
                    bool ok;
                    if constexpr (MODE == 0) ok = (j < p.Sk) && (!p.causal || j <= ii + q_off);
                    else ok = (j < p.Sq) && (!p.causal || ii <= j + q_off);
                    pr = ok ? pr : 0.f;
                }
                const float ds = pr * (t2[r] - Dl) * p.scale;
                pf[r >> 3][r & 7] = (bf16)pr;
                dsf[r >> 3][r & 7] = (bf16)ds;
            }
            // ---- acc1^T += A1^T dS ; acc2^T += A2^T P   (A^T fragments by transposed LDS reads of the [j][d] images)
            const int li = lane & 15;
            const int g16 = (lane >> 4) & 1;
            const int toff = (sub * 32 + 4 * lh + (li >> 2)) * C::RS + 16 * g16 + 4 * (li & 3);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int tt = 0; tt < C::NDT; ++tt) {
                    const int o0 = toff + (16 * s2) * C::RS + 32 * tt;
                    {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(img1 + o0));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(img1 + o0 + 8 * C::RS));
                        bf16x8 af;
                        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                        af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
                        acc1[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf[s2], acc1[tt], 0, 0, 0);
                    }
                    if constexpr (MODE == 1) {
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(img2 + o0));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(img2 + o0 + 8 * C::RS));
                        bf16x8 af;
                        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                        af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
                        acc2[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf[s2], acc2[tt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- write the own rows' gradients: G[i][d] = acc^T[d][i]
    if (!wave_active || ii >= own_n) return;
    const int64_t trow = tok(ii);
    auto store = [&](bf16* grow, const f32x16 (&acc)[C::NDT]) {
#pragma unroll
        for (int tt = 0; tt < C::NDT; ++tt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * tt + 8 * g + 4 * lh;
                if (d0 < D) {
                    u32x2 o;
                    o[0] = pack2(acc[tt][4 * g + 0], acc[tt][4 * g + 1]);
                    o[1] = pack2(acc[tt][4 * g + 2], acc[tt][4 * g + 3]);
                    *reinterpret_cast<u32x2*>(grow + d0) = o;
                }
            }
        }
    };
    if constexpr (MODE == 0) {
        store(reinterpret_cast<bf16*>(pd.dQ) + (int64_t)b * pd.dq_bs + head * D + trow * pd.dq_ss, acc1);
    } else {
        store(reinterpret_cast<bf16*>(pd.dK) + (int64_t)b * pd.dk_bs + head * D + trow * pd.dk_ss, acc1);
        store(reinterpret_cast<bf16*>(pd.dV) + (int64_t)b * pd.dv_bs + head * D + trow * pd.dv_ss, acc2);
    }
}

template <int D, int MODE>
int launch_one(const seer_attn_bwd_desc& d, int ws_log2, hipStream_t st) {
    int nbatch = d.fwd.batch;
    if (ws_log2 >= 0) nbatch *= (d.fwd.H >> ws_log2) * (d.fwd.W >> ws_log2);
    const int own_n = MODE == 0 ? d.fwd.Sq : d.fwd.Sk;
    dim3 grid((own_n + 127) / 128, nbatch * d.fwd.heads, 1);
    constexpr size_t lds = BwdCfg<D>::LDS_BYTES;
    hipLaunchKernelGGL((seer_attn_bwd_kernel<D, MODE>), grid, dim3(256), lds, st, d, ws_log2);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

template <int D>
int launch_bwd(const seer_attn_bwd_desc& d, int ws_log2, hipStream_t st) {
    const int rc = launch_one<D, 0>(d, ws_log2, st);        // dQ + delta first: the dK|dV launch reads delta
    if (rc != SEER_OK) return rc;
    return launch_one<D, 1>(d, ws_log2, st);
}

}  // namespace

extern "C" int seer_attn_bwd(const seer_attn_bwd_desc* desc, void* stream) {
    if (!desc) return SEER_EINVAL;
    const seer_attn_bwd_desc d = *desc;
    const seer_attn_desc& f = d.fwd;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!f.Q || !f.K || !f.V || !f.O || !f.lse || !d.dO || !d.dQ || !d.dK || !d.dV || !d.delta) return SEER_EINVAL;
    if (f.batch <= 0 || f.heads <= 0 || f.Sq <= 0 || f.Sk <= 0) return SEER_EINVAL;
    if (f.q_hs || f.k_hs || f.v_hs) return SEER_ENOSYS;        // the backward reads token-major operands only
    if (f.flags & SEER_ATTN_Q_PRESCALED) return SEER_ENOSYS;   // ... and an un-prescaled q (it applies scale * log2(e) itself)
    if ((f.q_ss | f.k_ss | f.v_ss | f.o_ss | d.do_ss) % 8 || (d.dq_ss | d.dk_ss | d.dv_ss) % 4) return SEER_EINVAL;
    if ((f.q_bs | f.k_bs | f.v_bs | f.o_bs | d.do_bs) % 8 || (d.dq_bs | d.dk_bs | d.dv_bs) % 4) return SEER_EINVAL;
    int ws_log2 = -1;
    if (f.window_ws > 0) {
        if (f.window_ws != 4 && f.window_ws != 8) return SEER_EINVAL;
        ws_log2 = f.window_ws == 4 ? 2 : 3;
        if (f.H % f.window_ws || f.W % f.window_ws) return SEER_EINVAL;
        if (f.Fq != f.F || f.Sq != f.F * f.window_ws * f.window_ws || f.Sk != f.Sq) return SEER_EINVAL;
    }
    if (f.causal_offset < 0 || (f.causal && f.Sq + f.causal_offset > f.Sk)) return SEER_EINVAL;
    switch (f.head_dim) {
        case 40: return launch_bwd<40>(d, ws_log2, st);
        case 80: return launch_bwd<80>(d, ws_log2, st);
        case 96: return launch_bwd<96>(d, ws_log2, st);
        case 160: return launch_bwd<160>(d, ws_log2, st);
        default: return SEER_ENOSYS;
    }
}
