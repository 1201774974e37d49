// HBM-bound normalisation kernels for gfx950: GroupNorm over (C/G, F, H, W) on channels-last data (statistics and
// apply, optional SiLU, optional two-source channel concat), LayerNorm per token row, row softmax.
// All loads/stores are 16 bytes per lane (8 bf16); statistics are fp32.
#include "seer_common.h"

namespace {

constexpr int GN_MAX_RPB = 64;   // rows per block (upper bound)

struct GnGeom {
    int ncols;          // 16-byte chunk columns = (C1+C2)/8
    int ncols1;         // columns that come from x1
    int C1, C2, cpg;
    int rpb;            // rows per block
    int nblk;           // blocks per batch element
};

// thread -> (row lane rl, chunk column) with a FIXED column per pass so that per-column constants / partial sums stay in
// registers.  cols_per_pass = min(ncols, 256); rows_par = 256 / cols_per_pass.
//
// Statistics are DETERMINISTIC (no float atomics): every thread parks its per-column partial sums in LDS, `groups`
// threads then add the contributions of their group in a fixed order and the block writes its partial
// (sum, sumsq)[groups] to `partials[b][blk]`; gn_finalize_kernel adds the per-block partials in block order.
template <bool F16>
__global__ void __launch_bounds__(256) gn_stats_kernel(const bf16* __restrict__ x1, const bf16* __restrict__ x2,
                                                       GnGeom g, int64_t rows_per_batch, int groups,
                                                       float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float slots[];   // [rows_par][ncols][4] = (sa, qa, sb, qb)
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * g.rpb;
    const int64_t r1 = min(r0 + g.rpb, rows_per_batch);
    const int cpp = g.ncols < 256 ? g.ncols : 256;
    const int rows_par = 256 / cpp;
    const int rl = tid / cpp;
    const int cl = tid - rl * cpp;
    if (rl < rows_par) {
        for (int cb = 0; cb < g.ncols; cb += cpp) {
            const int col = cb + cl;
            if (col >= g.ncols) break;
            const bool first = col < g.ncols1;
            const bf16* src = first ? x1 + col * 8 : x2 + (col - g.ncols1) * 8;
            const int ld = first ? g.C1 : g.C2;
            const int c0 = col * 8;
            const int ga = c0 / g.cpg;
            const int split = (ga + 1) * g.cpg - c0;   // elements [0, split) belong to ga, the rest to ga+1
            float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll 4
            for (int64_t r = r0 + rl; r < r1; r += rows_par) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((int64_t)b * rows_per_batch + r) * ld);
                float f[8];
                unpack8t<F16>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < split) { sa += f[e]; qa += f[e] * f[e]; }
                    else { sb += f[e]; qb += f[e] * f[e]; }
                }
            }
            *reinterpret_cast<f32x4*>(slots + ((int64_t)rl * g.ncols + col) * 4) = f32x4{sa, qa, sb, qb};
        }
    }
    __syncthreads();
    if (tid < groups) {
        const int c_lo = tid * g.cpg, c_hi = c_lo + g.cpg - 1;
        float s = 0.f, q = 0.f;
        for (int col = c_lo / 8; col <= c_hi / 8; ++col) {
            const bool is_a = (col * 8) / g.cpg == tid;          // this group is the column's first group
            for (int r = 0; r < rows_par; ++r) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(slots + ((int64_t)r * g.ncols + col) * 4);
                s += is_a ? v[0] : v[2];
                q += is_a ? v[1] : v[3];
            }
        }
        float* o = partials + (((int64_t)b * g.nblk + blockIdx.x) * groups + tid) * 2;
        o[0] = s;
        o[1] = q;
    }
}

// stats[b][g][2] = sum over blocks.  One WAVE per (b, statistic): lane l adds blocks l, l+64, ... in order, then the 64 lane
// sums are added by a fixed butterfly -- the same order for every (b, statistic) and every run (deterministic, no atomics), and
// 2 * batch * groups waves instead of `batch` blocks each walking a 64-deep dependent chain (4.7 -> ~2 us per launch: the
// kernel is pure latency, r02_rocprof_summary_v1.md)
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partials, int nblk, int groups, int batch,
                                                          float* __restrict__ stats) {
    const int n = groups * 2;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);       // wave index = b * n + statistic
    if (w >= batch * n) return;
    const int lane = threadIdx.x & 63;
    const int b = w / n, sidx = w - b * n;
    const float* p = partials + ((int64_t)b * nblk) * n + sidx;
    float acc = 0.f;
    for (int k = lane; k < nblk; k += 64) acc += p[(int64_t)k * n];
    acc = wave_sum(acc);                                     // xor butterfly 32, 16, ..., 1: one fixed order
    if (lane == 0) stats[w] = acc;
}

// stats[b][g][2] from per-tile column sums (seer_gemm_desc::colsum): one BLOCK per (b, g); thread t adds the (tile, channel)
// partials t, t + 256, ... of source 1, then of source 2 (four independent loads in flight per thread), the 64 lane sums of a
// wave meet in a fixed butterfly and wave 0 adds the four wave sums in order: deterministic, no atomics
struct GnColsumSrc {
    const float* cs;
    int C, phases, tiles;
};
__global__ void __launch_bounds__(256) gn_colsum_finalize_kernel(GnColsumSrc s1, GnColsumSrc s2, int batch, int groups,
                                                                 float* __restrict__ stats) {
    __shared__ float wsum[4][2];
    const int w = blockIdx.x;                                // b * groups + g
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = w / groups, g = w - b * groups;
    const int cpg = (s1.C + s2.C) / groups;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const GnColsumSrc& s = k == 0 ? s1 : s2;
        const int base = k == 0 ? 0 : s1.C;                  // first concat channel of this source
        const int a = max(c_lo, base) - base, e = min(c_hi, base + s.C) - base;
        if (s.C == 0 || e <= a) continue;
        const int nch = e - a, tpb = s.tiles / batch;
        const int items = s.phases * tpb * nch;
        auto item = [&](int i) -> f32x2 {
            const int t = i / nch, ch = i - t * nch;
            const int ph = t / tpb, tl = t - ph * tpb;
            return *reinterpret_cast<const f32x2*>(s.cs + (((int64_t)ph * s.tiles + b * tpb + tl) * s.C + a + ch) * 2);
        };
        int i = tid;
        for (; i + 768 < items; i += 1024) {
            const f32x2 v0 = item(i), v1 = item(i + 256), v2 = item(i + 512), v3 = item(i + 768);
            sm += v0[0]; sq += v0[1];
            sm += v1[0]; sq += v1[1];
            sm += v2[0]; sq += v2[1];
            sm += v3[0]; sq += v3[1];
        }
        for (; i < items; i += 256) {
            const f32x2 v = item(i);
            sm += v[0];
            sq += v[1];
        }
    }
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if (lane == 0) {
        wsum[tid >> 6][0] = sm;
        wsum[tid >> 6][1] = sq;
    }
    __syncthreads();
    if (tid == 0) {
        stats[(int64_t)w * 2] = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
        stats[(int64_t)w * 2 + 1] = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
    }
}

template <bool F16>
__global__ void __launch_bounds__(256) gn_apply_kernel(const bf16* __restrict__ x1, const bf16* __restrict__ x2,
                                                       GnGeom g, int64_t rows_per_batch, int groups,
                                                       const float* __restrict__ stats, float inv_count, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int silu, bf16* __restrict__ y) {
    __shared__ float mean_s[64], rstd_s[64];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * g.rpb;
    const int64_t r1 = min(r0 + g.rpb, rows_per_batch);
    const int cpp = g.ncols < 256 ? g.ncols : 256;
    const int rows_par = 256 / cpp;
    const int rl = tid / cpp;
    const int cl = tid - rl * cpp;
    // the statistics, the first column's affine parameters and (below) the first rows are all requested before anything
    // waits: one memory round trip at the head of the block instead of three dependent ones
    float s_in = 0.f, q_in = 0.f;
    if (tid < groups) {
        s_in = stats[((int64_t)b * groups + tid) * 2];
        q_in = stats[((int64_t)b * groups + tid) * 2 + 1];
    }
    f32x4 gpre[2] = {}, bpre[2] = {};
    if (rl < rows_par && cl < g.ncols) {
        gpre[0] = *reinterpret_cast<const f32x4*>(gamma + cl * 8);
        gpre[1] = *reinterpret_cast<const f32x4*>(gamma + cl * 8 + 4);
        bpre[0] = *reinterpret_cast<const f32x4*>(beta + cl * 8);
        bpre[1] = *reinterpret_cast<const f32x4*>(beta + cl * 8 + 4);
    }
    if (tid < groups) {
        const float mean = s_in * inv_count;
        float var = q_in * inv_count - mean * mean;
        var = var > 0.f ? var : 0.f;
        mean_s[tid] = mean;
        rstd_s[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    if (rl >= rows_par) return;
    const int C = g.C1 + g.C2;
    for (int cb = 0; cb < g.ncols; cb += cpp) {
        const int col = cb + cl;
        if (col >= g.ncols) break;
        const bool first = col < g.ncols1;
        const bf16* src = first ? x1 + col * 8 : x2 + (col - g.ncols1) * 8;
        const int ld = first ? g.C1 : g.C2;
        const int c0 = col * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int grp = (c0 + e) / g.cpg;
            const float gm_ = cb == 0 ? gpre[e >> 2][e & 3] : gamma[c0 + e];
            const float bt_ = cb == 0 ? bpre[e >> 2][e & 3] : beta[c0 + e];
            const float a = rstd_s[grp] * gm_;
            sc[e] = a;
            sh[e] = bt_ - mean_s[grp] * a;
        }
#pragma unroll 4
        for (int64_t r = r0 + rl; r < r1; r += rows_par) {
            const int64_t row = (int64_t)b * rows_per_batch + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + row * ld);
            float f[8];
            unpack8t<F16>(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = f[e] * sc[e] + sh[e];
                if (silu) o = silu_f(o);
                f[e] = o;
            }
            store16_out(y + row * C + c0, pack8t<F16>(f));
        }
    }
}

// GroupNorm apply with the statistics taken from the producers' column sums INSIDE the launch (no seer_groupnorm_stats_from_colsums
// launch in front: 75 of the 77 GroupNorms of a step paid a 5.6 us latency-bound launch + a graph-node boundary for a 512-byte
// result).  A block owns a SLICE of whole groups (sc = 80 or 120 channels: >= 128 B of every row it touches) and a row range; it
// first re-derives the (mean, rstd) of its own groups from the (tile, channel) partials of that slice -- thread <-> one channel of
// the slice and one tile lane, fixed order, parked in LDS, one thread per group adds them in order: deterministic, no atomics --
// and then streams its rows.  The partials of a slice are tpb * sc * 8 bytes (82 KB at the 32x32 level, L2-resident) and every
// row block of the slice re-reads them: the grid is sized to ~2 blocks per CU so that the redundancy stays ~40 MB per launch.
struct GnCsGeom {
    int C1, C2, cpg, groups;
    int sc;             // channels per slice (a whole number of groups, a multiple of 8)
    int nslice, nrowblk;
    int rpb;            // rows per row block
};
template <bool FX, bool F16 = false>
__global__ void __launch_bounds__(256) gn_apply_cs_kernel(const bf16* __restrict__ x1, const bf16* __restrict__ x2, GnColsumSrc s1,
                                                          GnColsumSrc s2, GnCsGeom g, int batch, int64_t rows_per_batch,
                                                          float inv_count, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int silu, bf16* __restrict__ y,
                                                          float* __restrict__ stats_out) {
    __shared__ float part[16][128][2];                  // [tile lane][channel of the slice][sum, sumsq]: sc >= 64 -> at most 256 / 16 tile lanes
    __shared__ float mean_s[16], rstd_s[16];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int slice = blockIdx.x % g.nslice, rb = blockIdx.x / g.nslice;
    const int c_lo = slice * g.sc;                      // first concat channel of the slice
    // ---- statistics of the slice's groups.  thread <-> (tile lane, quad of 4 channels): 32 contiguous bytes of a partial row per
    // item, 8..12 tile lanes, four items (eight 16-byte loads) in flight -- three dependent rounds at the 32x32 level instead of
    // the eleven a channel-per-thread form needs
    const int nq = g.sc / 4;                            // channel quads in the slice (20 or 30)
    const int ntl = 256 / nq;                           // tile lanes (12 or 8 at the UNet's widths; 16 at sc = 64, 14 at sc = 72)
    if constexpr (FX) {
        // the producers ACCUMULATED the sums per (batch element, channel) in 64-bit fixed point (seer_gemm_desc::colsum_fx): one
        // 16-byte load per channel, the group's channels added as integers (exact, order-free), one conversion per group
        long long* fxs = reinterpret_cast<long long*>(&part[0][0][0]);      // [channel of the slice][sum, sumsq]
        if (tid < g.sc) {
            const int c = c_lo + tid;
            const GnColsumSrc& s = c < s1.C ? s1 : s2;
            const int cl = c < s1.C ? c : c - s1.C;
            // [rep][b][2][C]: s.phases = replicas (added here: integers, any order), s.tiles = batch elements
            long long sm = 0, sq = 0;
            for (int r = 0; r < s.phases; ++r) {
                const long long* q = reinterpret_cast<const long long*>(s.cs) + (int64_t)((r * s.tiles + b) * 2) * s.C + cl;
                sm += q[0];
                sq += q[s.C];
            }
            fxs[tid * 2] = sm;
            fxs[tid * 2 + 1] = sq;
        }
        __syncthreads();
        const int gs = g.sc / g.cpg;
        const int lane = tid & 63, wv = tid >> 6;
        for (int gi = wv; gi < gs; gi += 4) {
            long long sm = 0, sq = 0;
            for (int ch = lane; ch < g.cpg; ch += 64) {
                sm += fxs[(gi * g.cpg + ch) * 2];
                sq += fxs[(gi * g.cpg + ch) * 2 + 1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                sm += __shfl_xor(sm, o);
                sq += __shfl_xor(sq, o);
            }
            if (lane == 0) {
                const double k = (double)inv_count / (double)(1 << SEER_GN_FX_SHIFT);
                const double mean = (double)sm * k;
                double var = (double)sq * k - mean * mean;
                var = var > 0.0 ? var : 0.0;
                mean_s[gi] = (float)mean;
                rstd_s[gi] = rsqrtf((float)var + eps);
                // (sum, sumsq) per (batch element, group) for a backward pass (seer_groupnorm_bwd): the first row block writes them
                if (stats_out && rb == 0) {
                    const double u = 1.0 / (double)(1 << SEER_GN_FX_SHIFT);
                    float* o = stats_out + ((int64_t)b * g.groups + slice * gs + gi) * 2;
                    o[0] = (float)((double)sm * u);
                    o[1] = (float)((double)sq * u);
                }
            }
        }
    } else {
    {
        const int tl = tid / nq, cq = tid - tl * nq;
        if (tl < ntl) {
            const int c = c_lo + cq * 4;
            const GnColsumSrc& s = c < s1.C ? s1 : s2;
            const int cl = c < s1.C ? c : c - s1.C;
            const int tpb = s.tiles / batch;
            const int items = s.phases * tpb;
            const float* base = s.cs + ((int64_t)b * tpb * s.C + cl) * 2;
            f32x4 a01 = {0.f, 0.f, 0.f, 0.f}, a23 = {0.f, 0.f, 0.f, 0.f};     // (s, q) of channels 0, 1 | 2, 3
            auto at = [&](int i) -> const float* {
                const int ph = i / tpb, t = i - ph * tpb;
                return base + ((int64_t)ph * s.tiles + t) * s.C * 2;
            };
            int i = tl;
            for (; i + 3 * ntl < items; i += 4 * ntl) {
                const float *p0 = at(i), *p1 = at(i + ntl), *p2 = at(i + 2 * ntl), *p3 = at(i + 3 * ntl);
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(p0), w0 = *reinterpret_cast<const f32x4*>(p0 + 4);
                const f32x4 u1 = *reinterpret_cast<const f32x4*>(p1), w1 = *reinterpret_cast<const f32x4*>(p1 + 4);
                const f32x4 u2 = *reinterpret_cast<const f32x4*>(p2), w2 = *reinterpret_cast<const f32x4*>(p2 + 4);
                const f32x4 u3 = *reinterpret_cast<const f32x4*>(p3), w3 = *reinterpret_cast<const f32x4*>(p3 + 4);
                a01 += u0; a23 += w0;
                a01 += u1; a23 += w1;
                a01 += u2; a23 += w2;
                a01 += u3; a23 += w3;
            }
            for (; i < items; i += ntl) {
                const float* p0 = at(i);
                a01 += *reinterpret_cast<const f32x4*>(p0);
                a23 += *reinterpret_cast<const f32x4*>(p0 + 4);
            }
            *reinterpret_cast<f32x4*>(&part[tl][cq * 4][0]) = a01;
            *reinterpret_cast<f32x4*>(&part[tl][cq * 4 + 2][0]) = a23;
        }
    }
    __syncthreads();
    const int gs = g.sc / g.cpg;                        // groups in the slice
    // tile lanes first (one thread per channel, fixed order), then one WAVE per group: lane l adds channels l, l + 64 of the
    // group and the 64 lane sums meet in the fixed butterfly of wave_sum -- the serial form (one thread walking cpg x tile-lane
    // LDS entries) cost 12 us on its own at cpg = 40..80
    if (tid < g.sc) {
        float sm = 0.f, sq = 0.f;
        for (int tl = 0; tl < ntl; ++tl) {
            sm += part[tl][tid][0];
            sq += part[tl][tid][1];
        }
        part[0][tid][0] = sm;
        part[0][tid][1] = sq;
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int gi = wv; gi < gs; gi += 4) {
            float sm = 0.f, sq = 0.f;
            for (int ch = lane; ch < g.cpg; ch += 64) {
                sm += part[0][gi * g.cpg + ch][0];
                sq += part[0][gi * g.cpg + ch][1];
            }
            sm = wave_sum(sm);
            sq = wave_sum(sq);
            if (lane == 0) {
                const float mean = sm * inv_count;
                float var = sq * inv_count - mean * mean;
                var = var > 0.f ? var : 0.f;
                mean_s[gi] = mean;
                rstd_s[gi] = rsqrtf(var + eps);
            }
        }
    }
    }
    __syncthreads();
    // ---- apply: thread -> (row lane, 16-byte chunk of the slice)
    const int cps = g.sc / 8;
    const int rows_par = 256 / cps;
    const int rl = tid / cps, cl = tid - rl * cps;
    if (rl >= rows_par) return;
    const int C = g.C1 + g.C2;
    const int c0 = c_lo + cl * 8;                       // first concat channel of this thread's chunk
    const bool first = c0 < g.C1;
    const bf16* src = first ? x1 + c0 : x2 + (c0 - g.C1);
    const int ld = first ? g.C1 : g.C2;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int grp = (cl * 8 + e) / g.cpg;
        const float a = rstd_s[grp] * gamma[c0 + e];
        sc[e] = a;
        sh[e] = beta[c0 + e] - mean_s[grp] * a;
    }
    const int64_t r0 = (int64_t)rb * g.rpb;
    const int64_t r1 = min(r0 + g.rpb, rows_per_batch);
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += rows_par) {
        const int64_t row = (int64_t)b * rows_per_batch + r;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + row * ld);
        float f[8];
        unpack8t<F16>(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = f[e] * sc[e] + sh[e];
            if (silu) o = silu_f(o);
            f[e] = o;
        }
        store16_out(y + row * C + c0, pack8t<F16>(f));
    }
}

// slice geometry for gn_apply_cs_kernel; false: this GroupNorm keeps the two-launch form
#ifndef SEER_GN_FX_BLOCKS
#define SEER_GN_FX_BLOCKS 1024
#endif
bool gn_cs_geom(int C1, int C2, int groups, int batch, int64_t rows_per_batch, GnCsGeom* g, int target_blocks = 512) {
    const int C = C1 + C2;
    if (C1 <= 0 || C2 < 0 || groups <= 0 || groups > 64 || C % groups || C1 % 8 || C2 % 8) return false;
    g->C1 = C1; g->C2 = C2; g->groups = groups; g->cpg = C / groups;
    int sc = 0;
    for (int k = 1; k * g->cpg <= 128; ++k)
        if (k * g->cpg >= 64 && (k * g->cpg) % 8 == 0 && groups % k == 0) { sc = k * g->cpg; break; }
    if (!sc || sc / g->cpg > 16) return false;
    g->sc = sc;
    g->nslice = C / sc;
    int nrb = target_blocks / (batch * g->nslice);
    if (nrb < 1) nrb = 1;
    const int rows_par = 256 / (sc / 8);
    int64_t rpb = (rows_per_batch + nrb - 1) / nrb;
    rpb = (rpb + rows_par - 1) / rows_par * rows_par;
    if (rpb < rows_par) rpb = rows_par;
    g->rpb = (int)rpb;
    g->nrowblk = (int)((rows_per_batch + rpb - 1) / rpb);
    return true;
}

// LayerNorm: one wave per row, the row lives in registers (<= 3 chunks of 8 per lane: C <= 1536), two-pass statistics.
template <int MAXC, bool F16 = false>
__global__ void __launch_bounds__(256) layernorm_kernel(const bf16* __restrict__ x, int64_t rows, int C, int ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, bf16* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nch = C / 8;
    float gm[MAXC][8], bt[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { gm[i][e] = gamma[ch * 8 + e]; bt[i][e] = beta[ch * 8 + e]; }
        }
    }
    const float invC = 1.0f / (float)C;
    for (int64_t r = wave_global; r < rows; r += nwaves) {
        float f[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ldx + ch * 8);
                unpack8t<F16>(v, f[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[i][e];
            }
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
                store16_out(y + r * ldy + ch * 8, pack8t<F16>(o));
            }
        }
    }
}

// row softmax(x * scale): one wave per row, n <= 64*8*MAXC
template <int MAXC, bool IN_F32, bool F16 = false>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const void* __restrict__ xv, int64_t rows, int n, int ld,
                                                           float scale, bf16* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int nch = n / 8;
    float f[MAXC][8];
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            if constexpr (IN_F32) {
                const float* x = reinterpret_cast<const float*>(xv) + r * ld + ch * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(x), b = *reinterpret_cast<const f32x4*>(x + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[i][e] = a[e]; f[i][4 + e] = b[e]; }
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16*>(xv) + r * ld + ch * 8);
                unpack8t<F16>(v, f[i]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[i][e] *= scale; mx = fmaxf(mx, f[i][e]); }
        }
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[i][e] = __expf(f[i][e] - mx); s += f[i][e]; }
        }
    }
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f[i][e] * inv;
            *reinterpret_cast<u32x4*>(y + r * ldy + ch * 8) = pack8t<F16>(o);
        }
    }
}

bool gn_geom(int C1, int C2, int groups, int batch, int64_t rows_per_batch, GnGeom* g) {
    const int C = C1 + C2;
    if (C1 <= 0 || C2 < 0 || groups <= 0 || groups > 64) return false;
    if (C % groups || C1 % 8 || C2 % 8) return false;
    g->cpg = C / groups;
    if (g->cpg < 4 || (g->cpg < 8 && 8 % g->cpg)) return false;
    g->C1 = C1; g->C2 = C2;
    g->ncols = C / 8;
    g->ncols1 = C1 / 8;
    // rows per block: aim at ~1024 blocks so that small (deep-level) tensors still fill the 256 CUs
    const int cpp = g->ncols < 256 ? g->ncols : 256;
    const int rows_par = 256 / cpp;
    int64_t rpb = (rows_per_batch * batch + 1023) / 1024;
    rpb = (rpb + rows_par - 1) / rows_par * rows_par;
    if (rpb < rows_par) rpb = rows_par;
    if (rpb > GN_MAX_RPB) rpb = GN_MAX_RPB / rows_par * rows_par;
    if (rpb < 1) rpb = 1;
    g->rpb = (int)rpb;
    g->nblk = (int)((rows_per_batch + rpb - 1) / rpb);
    return true;
}

}  // namespace

extern "C" int64_t seer_groupnorm_workspace_floats(int32_t C, int32_t batch, int64_t rows_per_batch, int32_t groups) {
    GnGeom g;
    if (!gn_geom(C, 0, groups, batch, rows_per_batch, &g)) return SEER_EINVAL;
    return (int64_t)batch * g.nblk * groups * 2;
}

extern "C" int seer_groupnorm_stats_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                       int64_t rows_per_batch, int32_t groups, float* stats, float* workspace,
                                       int32_t dtype, void* stream);
extern "C" int seer_groupnorm_stats(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                    int64_t rows_per_batch, int32_t groups, float* stats, float* workspace,
                                    void* stream) {
    return seer_groupnorm_stats_dt(x1, C1, x2, C2, batch, rows_per_batch, groups, stats, workspace, SEER_DT_BF16, stream);
}
extern "C" int seer_groupnorm_stats_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                       int64_t rows_per_batch, int32_t groups, float* stats, float* workspace,
                                       int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    GnGeom g;
    if (!x1 || !stats || !workspace || batch <= 0 || rows_per_batch <= 0) return SEER_EINVAL;
    if (!x2) C2 = 0;
    if (!gn_geom(C1, C2, groups, batch, rows_per_batch, &g)) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int cpp = g.ncols < 256 ? g.ncols : 256;
    const size_t lds = (size_t)(256 / cpp) * g.ncols * 4 * sizeof(float);
    dim3 grid((unsigned)g.nblk, batch);
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL(gn_stats_kernel<true>, grid, dim3(256), lds, st, reinterpret_cast<const bf16*>(x1),
                           reinterpret_cast<const bf16*>(x2), g, rows_per_batch, groups, workspace);
    else
        hipLaunchKernelGGL(gn_stats_kernel<false>, grid, dim3(256), lds, st, reinterpret_cast<const bf16*>(x1),
                           reinterpret_cast<const bf16*>(x2), g, rows_per_batch, groups, workspace);
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((batch * groups * 2 + 3) / 4)), dim3(256), 0, st, workspace, g.nblk, groups,
                       batch, stats);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_groupnorm_stats_from_colsums(const float* cs1, int32_t C1, int32_t phases1, int32_t tiles1,
                                                 const float* cs2, int32_t C2, int32_t phases2, int32_t tiles2,
                                                 int32_t batch, int32_t groups, float* stats, void* stream) {
    if (!cs1 || !stats || batch <= 0 || groups <= 0 || C1 <= 0 || phases1 <= 0 || tiles1 <= 0 || tiles1 % batch) return SEER_EINVAL;
    if (!cs2) C2 = 0;
    if (C2 < 0 || (C2 > 0 && (phases2 <= 0 || tiles2 <= 0 || tiles2 % batch))) return SEER_EINVAL;
    if ((C1 + C2) % groups) return SEER_EINVAL;
    const GnColsumSrc s1{cs1, C1, phases1, tiles1}, s2{cs2, C2, C2 ? phases2 : 0, C2 ? tiles2 : 0};
    hipLaunchKernelGGL(gn_colsum_finalize_kernel, dim3((unsigned)(batch * groups)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), s1, s2, batch, groups, stats);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_groupnorm_apply_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                       int64_t rows_per_batch, int32_t groups, const float* stats, double count,
                                       float eps, const float* gamma, const float* beta, int32_t silu, void* y,
                                       int32_t dtype, void* stream);
extern "C" int seer_groupnorm_apply(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                    int64_t rows_per_batch, int32_t groups, const float* stats, double count,
                                    float eps, const float* gamma, const float* beta, int32_t silu, void* y,
                                    void* stream) {
    return seer_groupnorm_apply_dt(x1, C1, x2, C2, batch, rows_per_batch, groups, stats, count, eps, gamma, beta, silu, y,
                                   SEER_DT_BF16, stream);
}
extern "C" int seer_groupnorm_apply_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                       int64_t rows_per_batch, int32_t groups, const float* stats, double count,
                                       float eps, const float* gamma, const float* beta, int32_t silu, void* y,
                                       int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    GnGeom g;
    if (!x1 || !stats || !gamma || !beta || !y || batch <= 0 || rows_per_batch <= 0 || count <= 0) return SEER_EINVAL;
    if (!x2) C2 = 0;
    if (!gn_geom(C1, C2, groups, batch, rows_per_batch, &g)) return SEER_EINVAL;
    dim3 grid((unsigned)g.nblk, batch);
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), g, rows_per_batch, groups,
                           stats, (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y));
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), g, rows_per_batch, groups,
                           stats, (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_groupnorm_apply_from_colsums(const void* x1, int32_t C1, const void* x2, int32_t C2, const float* cs1,
                                                 int32_t phases1, int32_t tiles1, const float* cs2, int32_t phases2,
                                                 int32_t tiles2, int32_t batch, int64_t rows_per_batch, int32_t groups,
                                                 double count, float eps, const float* gamma, const float* beta,
                                                 int32_t silu, void* y, void* stream) {
    return seer_groupnorm_apply_from_colsums_dt(x1, C1, x2, C2, cs1, phases1, tiles1, cs2, phases2, tiles2, batch, rows_per_batch, groups,
                                                count, eps, gamma, beta, silu, y, SEER_DT_BF16, stream);
}
extern "C" int seer_groupnorm_apply_from_colsums_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, const float* cs1,
                                                    int32_t phases1, int32_t tiles1, const float* cs2, int32_t phases2,
                                                    int32_t tiles2, int32_t batch, int64_t rows_per_batch, int32_t groups,
                                                    double count, float eps, const float* gamma, const float* beta,
                                                    int32_t silu, void* y, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x1 || !cs1 || !gamma || !beta || !y || batch <= 0 || rows_per_batch <= 0 || count <= 0) return SEER_EINVAL;
    if (phases1 <= 0 || tiles1 <= 0 || tiles1 % batch) return SEER_EINVAL;
    if (!x2) C2 = 0;
    if (C2 > 0 && (!cs2 || phases2 <= 0 || tiles2 <= 0 || tiles2 % batch)) return SEER_EINVAL;
    GnCsGeom g;
    if (!gn_cs_geom(C1, C2, groups, batch, rows_per_batch, &g)) return SEER_ENOSYS;
    // Where the one-launch form wins (profiles/r04_gn_fused.log, inside a hipGraph): few partials per batch element and a small
    // tensor -- the 16x16 level down.  At the 32x32 level every one of ~512 blocks re-reads 82 KB of partials (42 MB per launch)
    // and the two launches are faster (11.3 vs 13.3 us); the caller keeps them there.
    {
        const int64_t parts = (int64_t)phases1 * (tiles1 / batch) > (C2 ? (int64_t)phases2 * (tiles2 / batch) : 0)
                                  ? (int64_t)phases1 * (tiles1 / batch) : (int64_t)phases2 * (tiles2 / batch);
        if (parts > 32 || rows_per_batch * (int64_t)(C1 + C2) > (int64_t)4200000) return SEER_ENOSYS;
    }
    const GnColsumSrc s1{cs1, C1, phases1, tiles1}, s2{cs2, C2, C2 ? phases2 : 0, C2 ? tiles2 : 0};
    dim3 grid((unsigned)(g.nslice * g.nrowblk), batch);
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL((gn_apply_cs_kernel<false, true>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), s1, s2, g, batch, rows_per_batch,
                           (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y), (float*)nullptr);
    else
        hipLaunchKernelGGL((gn_apply_cs_kernel<false, false>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), s1, s2, g, batch, rows_per_batch,
                           (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y), (float*)nullptr);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_groupnorm_apply_fx(const void* x1, int32_t C1, const void* x2, int32_t C2, const int64_t* fx1, int32_t reps1,
                                       const int64_t* fx2, int32_t reps2, int32_t batch, int64_t rows_per_batch, int32_t groups,
                                       double count, float eps, const float* gamma, const float* beta, int32_t silu, void* y,
                                       float* stats_out, void* stream) {
    return seer_groupnorm_apply_fx_dt(x1, C1, x2, C2, fx1, reps1, fx2, reps2, batch, rows_per_batch, groups, count, eps, gamma, beta, silu,
                                      y, stats_out, SEER_DT_BF16, stream);
}
extern "C" int seer_groupnorm_apply_fx_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, const int64_t* fx1, int32_t reps1,
                                          const int64_t* fx2, int32_t reps2, int32_t batch, int64_t rows_per_batch, int32_t groups,
                                          double count, float eps, const float* gamma, const float* beta, int32_t silu, void* y,
                                          float* stats_out, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x1 || !fx1 || reps1 < 1 || !gamma || !beta || !y || batch <= 0 || rows_per_batch <= 0 || count <= 0) return SEER_EINVAL;
    if (!x2) C2 = 0;
    if (C2 > 0 && (!fx2 || reps2 < 1)) return SEER_EINVAL;
    GnCsGeom g;
    if (!gn_cs_geom(C1, C2, groups, batch, rows_per_batch, &g, SEER_GN_FX_BLOCKS)) return SEER_ENOSYS;
    const GnColsumSrc s1{reinterpret_cast<const float*>(fx1), C1, reps1, batch}, s2{reinterpret_cast<const float*>(fx2), C2, C2 ? reps2 : 0, batch};
    dim3 grid((unsigned)(g.nslice * g.nrowblk), batch);
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL((gn_apply_cs_kernel<true, true>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), s1, s2, g, batch, rows_per_batch,
                           (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y), stats_out);
    else
        hipLaunchKernelGGL((gn_apply_cs_kernel<true, false>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x1), reinterpret_cast<const bf16*>(x2), s1, s2, g, batch, rows_per_batch,
                           (float)(1.0 / count), eps, gamma, beta, silu, reinterpret_cast<bf16*>(y), stats_out);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

// (sum, sum of squares) per (batch element, column) in the 64-bit fixed point of seer_gemm_desc::colsum_fx, from the activations:
// every ELEMENT is rounded on its own (v * 2^20 and v * v * 2^20: a bf16 value and the fp32 product of two are exact, the rounding
// only cuts what lies below 2^-21), so the totals are sums of integers -- exact, independent of the order of the additions, of the
// block decomposition and, across frame shards, of which rank held which rows.  The statistics pass of a frame-sharded step for
// the GroupNorm sources whose producer leaves no accumulated sums (conv_in's output; tensors above the producers' row limit).
namespace {
template <bool F16>
__global__ void __launch_bounds__(256) gn_stats_fx_kernel(const bf16* __restrict__ x, int C, int64_t rows_per_batch, int rows_per_block,
                                                          long long* __restrict__ fx) {
    __shared__ long long part[4][64][16];                // [row lane][column chunk][8 sums | 8 sums of squares]
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int b = blockIdx.z;
    const int cq = blockIdx.x * 64 + cl;                 // 8-column chunk
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows_per_batch ? r1 : rows_per_batch;
    long long sm[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[j] = sq[j] = 0;
    if (cq * 8 < C) {
        const bf16* base = x + ((int64_t)b * rows_per_batch) * C + cq * 8;
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(base + r * C);
            float f[8];
            if constexpr (F16) unpack8t<true>(raw, f); else unpack8(raw, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sm[j] += __float2ll_rn(f[j] * (float)(1 << SEER_GN_FX_SHIFT));
                sq[j] += __float2ll_rn(f[j] * f[j] * (float)(1 << SEER_GN_FX_SHIFT));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        part[rl][cl][j] = sm[j];
        part[rl][cl][8 + j] = sq[j];
    }
    __syncthreads();
    // thread t < 64 * 4: column chunk cl, quarter rl of its 16 totals -> 4 atomic adds each
    if (cq * 8 < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = rl * 4 + k;                    // 0..15: 0..7 sums, 8..15 sums of squares
            const long long v = part[0][cl][e] + part[1][cl][e] + part[2][cl][e] + part[3][cl][e];
            long long* dst = fx + ((int64_t)(b * 2 + (e >> 3))) * C + cq * 8 + (e & 7);
            if (v) atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)v);
        }
    }
}
}  // namespace

extern "C" int seer_groupnorm_stats_fx(const void* x, int32_t C, int32_t batch, int64_t rows_per_batch, int64_t* fx, int32_t dtype,
                                       void* stream) {
    if (!x || !fx || C <= 0 || C % 8 || batch <= 0 || rows_per_batch <= 0) return SEER_EINVAL;
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    // ~2 blocks per CU at least, at most 64 adds per address
    int rpb = 256;
    while (rpb > 16 && (int64_t)batch * ((rows_per_batch + rpb - 1) / rpb) * ((C / 8 + 63) / 64) < 512) rpb >>= 1;
    while ((rows_per_batch + rpb - 1) / rpb > 64) rpb <<= 1;
    dim3 grid((unsigned)((C / 8 + 63) / 64), (unsigned)((rows_per_batch + rpb - 1) / rpb), (unsigned)batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL(gn_stats_fx_kernel<true>, grid, dim3(256), 0, st, reinterpret_cast<const bf16*>(x), C, rows_per_batch, rpb,
                           reinterpret_cast<long long*>(fx));
    else
        hipLaunchKernelGGL(gn_stats_fx_kernel<false>, grid, dim3(256), 0, st, reinterpret_cast<const bf16*>(x), C, rows_per_batch, rpb,
                           reinterpret_cast<long long*>(fx));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_layernorm(const void* x, int64_t rows, int32_t C, int32_t ldx, const float* gamma,
                              const float* beta, float eps, void* y, int32_t ldy, void* stream) {
    return seer_layernorm_dt(x, rows, C, ldx, gamma, beta, eps, y, ldy, SEER_DT_BF16, stream);
}
extern "C" int seer_layernorm_dt(const void* x, int64_t rows, int32_t C, int32_t ldx, const float* gamma,
                                 const float* beta, float eps, void* y, int32_t ldy, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || C % 8 || ldx % 8 || ldy % 8) return SEER_EINVAL;
    if (C > 64 * 8 * 3) return SEER_ENOSYS;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    bf16* yb = reinterpret_cast<bf16*>(y);
#define SEER_LN_LAUNCH(MC, H) hipLaunchKernelGGL((layernorm_kernel<MC, H>), dim3((unsigned)blocks), dim3(256), 0, st, xb, rows, C, ldx, gamma, beta, eps, yb, ldy)
    if (dtype == SEER_DT_F16) {
        if (C <= 512) SEER_LN_LAUNCH(1, true);
        else if (C <= 1024) SEER_LN_LAUNCH(2, true);
        else SEER_LN_LAUNCH(3, true);
    } else {
        if (C <= 512) SEER_LN_LAUNCH(1, false);
        else if (C <= 1024) SEER_LN_LAUNCH(2, false);
        else SEER_LN_LAUNCH(3, false);
    }
#undef SEER_LN_LAUNCH
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_softmax_rows_dt(const void* x, int32_t x_is_f32, int64_t rows, int32_t n, int32_t ld, float scale,
                                    void* y, int32_t ldy, int32_t dtype, void* stream);
extern "C" int seer_softmax_rows(const void* x, int32_t x_is_f32, int64_t rows, int32_t n, int32_t ld, float scale,
                                 void* y, int32_t ldy, void* stream) {
    return seer_softmax_rows_dt(x, x_is_f32, rows, n, ld, scale, y, ldy, SEER_DT_BF16, stream);
}
extern "C" int seer_softmax_rows_dt(const void* x, int32_t x_is_f32, int64_t rows, int32_t n, int32_t ld, float scale,
                                    void* y, int32_t ldy, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !y || rows <= 0 || n <= 0 || n % 8 || ld % 8 || ldy % 8) return SEER_EINVAL;
    if (n > 64 * 8 * 8) return SEER_ENOSYS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bf16* yb = reinterpret_cast<bf16*>(y);
    dim3 grid((unsigned)((rows + 3) / 4));
#define SEER_SM(MC)                                                                                                   \
    do {                                                                                                              \
        if (dtype == SEER_DT_F16) {                                                                                   \
            if (x_is_f32) hipLaunchKernelGGL((softmax_rows_kernel<MC, true, true>), grid, dim3(256), 0, st, x, rows, n, ld, scale, yb, ldy); \
            else hipLaunchKernelGGL((softmax_rows_kernel<MC, false, true>), grid, dim3(256), 0, st, x, rows, n, ld, scale, yb, ldy);         \
        } else if (x_is_f32) hipLaunchKernelGGL((softmax_rows_kernel<MC, true>), grid, dim3(256), 0, st, x, rows, n, ld, scale, yb, ldy); \
        else hipLaunchKernelGGL((softmax_rows_kernel<MC, false>), grid, dim3(256), 0, st, x, rows, n, ld, scale, yb, ldy);         \
    } while (0)
    if (n <= 1024) SEER_SM(2);
    else if (n <= 2048) SEER_SM(4);
    else SEER_SM(8);
#undef SEER_SM
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
