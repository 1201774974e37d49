// Training-step kernels for gfx950 (MI355X): the backward / optimizer side of the hot path (SURVEY 8(f) rank 1,
// train.py:319-389).  Everything here is HBM-bound elementwise / reduction work; the MFMA work of the backward pass reuses
// seer_gemm_bf16 (dX = dY W through a transposed weight copy, dW = dY^T X through seer_transpose_bf16) and seer_attn_bwd.
// All reductions are two-stage (per-block partials in a caller workspace, added in block order): no float atomics, so a
// step is bit-reproducible.
#include "seer_common.h"
#include <mutex>

namespace {

constexpr int CS_ROWS_MIN = 8;    // fewest rows a block of the column partial-sum kernels covers (bounds the workspace)

__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------------------------------
// transpose: y[c][r] = x[r][c] for r < rows, c < cols; columns rows .. rows_pad-1 of y are zero filled (the dW GEMM's
// contraction length must be a multiple of 64)
__global__ void __launch_bounds__(256) transpose_kernel(const bf16* __restrict__ x, int64_t rows, int cols, int ldx,
                                                        bf16* __restrict__ y, int64_t ldy, int64_t rows_pad) {
    __shared__ unsigned short tile[64][66];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
    unsigned short* ys = reinterpret_cast<unsigned short*>(y);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        unsigned short v = 0;
        if (r0 + r < rows && c0 + c < cols) v = xs[(r0 + r) * ldx + c0 + c];
        tile[r][c] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < cols && r0 + r < rows_pad) ys[(int64_t)(c0 + c) * ldy + r0 + r] = tile[r][c];
    }
}

// 16-byte version (cols % 8 == 0, ldx % 8 == 0, ldy % 64 == 0): a 64x64 tile goes global -> LDS by rows (8 lanes read 128
// contiguous bytes of a row) and LDS -> global by columns (8 lanes write 128 contiguous bytes of an output row)
__global__ void __launch_bounds__(256) transpose_vec_kernel(const bf16* __restrict__ x, int64_t rows, int cols, int ldx,
                                                            bf16* __restrict__ y, int64_t ldy) {
    __shared__ unsigned int tile[64][33];            // [r][c/2]: row pitch 33 words -> conflict-free column reads
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < rows && c0 + 8 * ch < cols) v = *reinterpret_cast<const u32x4*>(x + (r0 + r) * ldx + c0 + 8 * ch);
#pragma unroll
        for (int w = 0; w < 4; ++w) tile[r][4 * ch + w] = v[w];
    }
    __syncthreads();
    const unsigned short* t16 = reinterpret_cast<const unsigned short*>(&tile[0][0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int rc = idx & 7, c = idx >> 3;            // output row c0 + c, columns r0 + 8*rc .. +7
        if (c0 + c < cols) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = t16[(8 * rc + j) * 66 + c];
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) o[w] = (unsigned int)e[2 * w] | ((unsigned int)e[2 * w + 1] << 16);
            *reinterpret_cast<u32x4*>(y + (int64_t)(c0 + c) * ldy + r0 + 8 * rc) = o;
        }
    }
}

// many matrices in ONE launch (the W^T refresh of every trainable matrix after an optimizer step: ~190 launches of 5 us each
// otherwise): block b finds its matrix by binary search over the items' first-tile indices, then works as transpose_vec_kernel
__global__ void __launch_bounds__(256) transpose_batched_kernel(const seer_transpose_item* __restrict__ items, int n_items) {
    __shared__ unsigned int tile[64][33];
    int lo = 0, hi = n_items - 1;                    // last item whose tile0 <= blockIdx.x (tile0 ascending, items[0].tile0 == 0)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].tile0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const seer_transpose_item it = items[lo];
    const int64_t local = (int64_t)blockIdx.x - it.tile0;
    const int64_t r0 = (local % it.tiles_r) * 64;
    const int c0 = (int)(local / it.tiles_r) * 64;
    const bf16* __restrict__ x = reinterpret_cast<const bf16*>(it.x);
    bf16* __restrict__ y = reinterpret_cast<bf16*>(it.y);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < it.rows && c0 + 8 * ch < it.cols) v = *reinterpret_cast<const u32x4*>(x + (r0 + r) * it.ldx + c0 + 8 * ch);
#pragma unroll
        for (int w = 0; w < 4; ++w) tile[r][4 * ch + w] = v[w];
    }
    __syncthreads();
    const unsigned short* t16 = reinterpret_cast<const unsigned short*>(&tile[0][0]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int rc = idx & 7, c = idx >> 3;
        if (c0 + c < it.cols) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = t16[(8 * rc + j) * 66 + c];
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) o[w] = (unsigned int)e[2 * w] | ((unsigned int)e[2 * w + 1] << 16);
            *reinterpret_cast<u32x4*>(y + (int64_t)(c0 + c) * it.ldy + r0 + 8 * rc) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// column partial sums: the block's 256 threads are cpp column owners (8 adjacent columns each) x rows_par row lanes over a
// chunk of g.chunk_rows rows; row lanes are added through LDS in lane order; NV values per element.
// ws layout: [batch][chunk][NV][C]
struct ColGeom {
    int C, nchunks, chunk_rows, cpp, rows_par;
    int64_t rows_per_batch;
};

ColGeom col_geom(int C, int64_t rows_per_batch, int batch = 1) {
    ColGeom g;
    g.C = C;
    const int nc8 = C / 8;
    g.cpp = nc8 < 256 ? nc8 : 256;
    g.rows_par = 256 / g.cpp;
    // rows per block: aim at ~1024 blocks so that every level fills the 256 CUs, at least two rows per row lane and
    // CS_ROWS_MIN rows (the workspace bound), at most 64
    const int colblocks = (nc8 + g.cpp - 1) / g.cpp;
    int64_t cr = rows_per_batch * colblocks * batch / 1024;
    if (cr < 2 * g.rows_par) cr = 2 * g.rows_par;
    if (cr < CS_ROWS_MIN) cr = CS_ROWS_MIN;
    if (cr > 64) cr = 64;
    g.chunk_rows = (int)cr;
    g.nchunks = (int)((rows_per_batch + g.chunk_rows - 1) / g.chunk_rows);
    g.rows_per_batch = rows_per_batch;
    return g;
}
int col_blocks(const ColGeom& g) { return (g.C / 8 + g.cpp - 1) / g.cpp; }

// KIND 0: sum x          (bias gradients)
// KIND 1: LayerNorm      v0 = dy, v1 = dy * xhat        (rowstats = mean, rstd per row)
// KIND 2: GroupNorm      v0 = gy, v1 = gy * xhat        gy = dy * act'(z)   (x = concat x1|x2)
struct ColArgs {
    const bf16* x1; const bf16* x2; const bf16* dy;
    int C1, ldx, lddy;
    const float* rowstats;       // KIND 1
    const float* stats;          // KIND 2: (sum, sumsq) [batch][groups][2] of the forward
    float inv_count, eps;
    const float* gamma; const float* beta;
    int cpg, groups, silu;
};

// (sum, sumsq) -> (mean, rstd): the arithmetic of gn_apply_kernel
__device__ __forceinline__ void gn_mean_rstd(const ColArgs& a, int b, int grp, float& mean, float& rstd) {
    const float* st = a.stats + ((int64_t)b * a.groups + grp) * 2;
    mean = st[0] * a.inv_count;
    const float var = fmaxf(st[1] * a.inv_count - mean * mean, 0.f);
    rstd = rsqrtf(var + a.eps);
}

template <int KIND>
__global__ void __launch_bounds__(256) colpartial_kernel(const ColArgs a, const ColGeom g, float* __restrict__ ws) {
    constexpr int NV = KIND == 0 ? 1 : 2;
    extern __shared__ float red[];                   // [rows_par][cpp][NV*8]
    const int tc = threadIdx.x % g.cpp, rl = threadIdx.x / g.cpp;
    const int c0 = (blockIdx.y * g.cpp + tc) * 8;
    const bool active = c0 < g.C && rl < g.rows_par;
    const int b = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * g.chunk_rows;
    const int64_t r1 = min(r0 + (int64_t)g.chunk_rows, g.rows_per_batch);
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (active) {
        float sc[8], sh[8], gm[8], bt[8];
        if constexpr (KIND == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float mean, rstd;
                gn_mean_rstd(a, b, (c0 + e) / a.cpg, mean, rstd);
                sc[e] = rstd;
                sh[e] = -mean * rstd;
                gm[e] = a.gamma[c0 + e];
                bt[e] = a.beta[c0 + e];
            }
        }
        const bf16* src = a.x1;
        int ld = a.ldx, cc = c0;
        if constexpr (KIND == 2) {
            if (c0 >= a.C1) { src = a.x2; ld = g.C - a.C1; cc = c0 - a.C1; }
            else ld = a.C1;
        }
        for (int64_t r = r0 + rl; r < r1; r += g.rows_par) {
            const int64_t row = (int64_t)b * g.rows_per_batch + r;
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(src + row * ld + cc), f);
            if constexpr (KIND == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s0[e] += f[e];
            } else {
                float d[8];
                unpack8(*reinterpret_cast<const u32x4*>(a.dy + row * a.lddy + c0), d);
                if constexpr (KIND == 1) {
                    const float mean = a.rowstats[row * 2], rstd = a.rowstats[row * 2 + 1];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s0[e] += d[e];
                        s1[e] += d[e] * (f[e] - mean) * rstd;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xh = f[e] * sc[e] + sh[e];
                        float gy = d[e];
                        if (a.silu) gy *= silu_grad_f(xh * gm[e] + bt[e]);
                        s0[e] += gy;
                        s1[e] += gy * xh;
                    }
                }
            }
        }
    }
    if (g.rows_par > 1) {
        if (rl < g.rows_par) {
            float* o = red + ((size_t)rl * g.cpp + tc) * (NV * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = s0[e];
            if constexpr (NV == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[8 + e] = s1[e];
            }
        }
        __syncthreads();
        if (rl == 0) {
            for (int l = 1; l < g.rows_par; ++l) {
                const float* o = red + ((size_t)l * g.cpp + tc) * (NV * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s0[e] += o[e];
                if constexpr (NV == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s1[e] += o[8 + e];
                }
            }
        }
    }
    if (rl == 0 && c0 < g.C) {
        float* o = ws + (((int64_t)b * g.nchunks + blockIdx.x) * NV) * g.C + c0;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = s0[e];
        if constexpr (NV == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[g.C + e] = s1[e];
        }
    }
}

// out_v[z][c] = sum over k < nblocks of ws[z][k][v][c], in a fixed order: 64 columns x 16 k-lanes per block
// grid (ceil(C/64), NV, nz); outputs out0 / out1 are [nz][C] (either may be NULL)
__global__ void __launch_bounds__(1024) colfinal_kernel(const float* __restrict__ ws, int nblocks, int NV, int C,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int v = blockIdx.y, z = blockIdx.z;
    float s = 0.f;
    if (c < C) {
        const float* w = ws + ((int64_t)z * nblocks * NV + v) * C + c;
        for (int k = kl; k < nblocks; k += 16) s += w[(int64_t)k * NV * C];
    }
    red[kl][cl] = s;
    __syncthreads();
    if (kl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][cl];
        float* o = v == 0 ? out0 : out1;
        if (o) o[(int64_t)z * C + c] = t;
    }
}

// the same for many (ws, out) pairs in one launch: the d gamma / d beta finals of every LayerNorm a backward walk passed, deferred to
// its end (nothing reads them before the optimizer).  Table by value in the kernel arguments; block -> (item, column block, v)
constexpr int CF_GROUP = 64;
struct CfItem {
    const float* ws; float* out0; float* out1;
    int nblocks, NV, C, blk0;
};
struct CfGroup {
    CfItem it[CF_GROUP];
    int n;
};
__global__ void __launch_bounds__(1024) colfinal_grouped_kernel(const CfGroup g) {
    __shared__ float red[16][64];
    const int blk = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && blk >= g.it[i + 1].blk0) ++i;
    const CfItem& q = g.it[i];
    const int local = blk - q.blk0;
    const int cblocks = (q.C + 63) / 64;
    const int v = local / cblocks;
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = (local % cblocks) * 64 + cl;
    float s = 0.f;
    if (c < q.C) {
        const float* w = q.ws + (int64_t)v * q.C + c;
        for (int k = kl; k < q.nblocks; k += 16) s += w[(int64_t)k * q.NV * q.C];
    }
    red[kl][cl] = s;
    __syncthreads();
    if (kl == 0 && c < q.C) {
        float t = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += red[l][cl];
        float* o = v == 0 ? q.out0 : q.out1;
        if (o) o[c] = t;
    }
}

// GroupNorm backward, middle stage: the chunk partials of colpartial_kernel<2> (per (b, c): A = sum gy, Bv = sum gy xhat) -> per
// (b, g) projections s1 = sum_c gamma_c A_bc, s2 = sum_c gamma_c B_bc (scaled by 1/count), and dgamma_c = sum_b B_bc,
// dbeta_c = sum_b A_bc.  One block per group: chunk lanes x channel lanes add the partials (then a tree over the chunk lanes: a fixed
// order), wave 0 forms the projections; cpg <= 128.  (This was a colfinal_kernel launch + a per-group kernel.)
// CL = channel lanes (16 / 32 for groups of <= 16 / <= 32 channels, else 64 lanes x 2 channels): the other 1024 / CL lanes split the
// chunks, so the narrow groups of the 32x32 level (10 channels, 1024 chunks) take 16 trips over the partials instead of 64
template <int CL>
__global__ void __launch_bounds__(1024) gn_bwd_group_fused_kernel(const float* __restrict__ ws, int nchunks, int batch, int C,
                                                                  int cpg, int groups, const float* __restrict__ gamma,
                                                                  float inv_count, float* __restrict__ proj,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int KL = 1024 / CL, NK2 = CL == 64 ? 2 : 1;
    __shared__ float red[2 * NK2][KL][CL];           // [v + 2 k2][chunk lane][channel lane]
    const int grp = blockIdx.x;
    const int cl = threadIdx.x % CL, kl = threadIdx.x / CL;
    float dg[NK2], db[NK2];
#pragma unroll
    for (int k2 = 0; k2 < NK2; ++k2) dg[k2] = db[k2] = 0.f;
    // grid.y = batch when no d gamma / d beta is asked for (nothing crosses batch items then), 1 otherwise
    for (int b = blockIdx.y; b < batch; b += gridDim.y) {
        float s[2 * NK2];                            // the sums of this lane at once: independent loads in flight
#pragma unroll
        for (int j = 0; j < 2 * NK2; ++j) s[j] = 0.f;
        const float* w = ws + ((int64_t)b * nchunks) * 2 * C + grp * cpg + cl;
        const bool has0 = cl < cpg, has1 = NK2 == 2 && cl + 64 < cpg;
        for (int k = kl; k < nchunks; k += KL) {
            const float* wk = w + (int64_t)k * 2 * C;
            if (has0) { s[0] += wk[0]; s[1] += wk[C]; }
            if constexpr (NK2 == 2) {
                if (has1) { s[2] += wk[64]; s[3] += wk[C + 64]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * NK2; ++j) red[j][kl][cl] = s[j];
        __syncthreads();
#pragma unroll
        for (int h = KL / 2; h >= 1; h >>= 1) {      // chunk lanes: a tree in a fixed order
            if (kl < h) {
#pragma unroll
                for (int j = 0; j < 2 * NK2; ++j) red[j][kl][cl] += red[j][kl + h][cl];
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            float pa = 0.f, pb = 0.f;
            if (threadIdx.x < CL) {
#pragma unroll
                for (int k2 = 0; k2 < NK2; ++k2) {
                    const int c_local = cl + 64 * k2;
                    if (c_local < cpg) {
                        const float A = red[2 * k2][0][cl], Bv = red[2 * k2 + 1][0][cl];
                        const float gm = gamma[grp * cpg + c_local];
                        pa += gm * A;
                        pb += gm * Bv;
                        dg[k2] += Bv;
                        db[k2] += A;
                    }
                }
            }
            pa = wave_sum(pa);
            pb = wave_sum(pb);
            if (threadIdx.x == 0) {
                proj[((int64_t)b * groups + grp) * 2] = pa * inv_count;
                proj[((int64_t)b * groups + grp) * 2 + 1] = pb * inv_count;
            }
        }
        __syncthreads();
    }
    if (dgamma && threadIdx.x < CL) {
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
            const int c_local = cl + 64 * k2;
            if (c_local < cpg) {
                dgamma[grp * cpg + c_local] = dg[k2];
                dbeta[grp * cpg + c_local] = db[k2];
            }
        }
    }
}
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const ColArgs a, const ColGeom g, const float* __restrict__ proj,
                                                           const bf16* __restrict__ dres1, const bf16* __restrict__ dres2,
                                                           bf16* __restrict__ dx1, bf16* __restrict__ dx2) {
    const int tc = threadIdx.x % g.cpp, rl = threadIdx.x / g.cpp;
    const int c0 = (blockIdx.y * g.cpp + tc) * 8;
    if (c0 >= g.C || rl >= g.rows_par) return;
    const int b = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.x * g.chunk_rows;
    const int64_t r1 = min(r0 + (int64_t)g.chunk_rows, g.rows_per_batch);
    float sc[8], sh[8], gm[8], bt[8], p1[8], p2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int grp = (c0 + e) / a.cpg;
        const int64_t gi = ((int64_t)b * a.groups + grp) * 2;
        float mean, rstd;
        gn_mean_rstd(a, b, grp, mean, rstd);
        sc[e] = rstd;
        sh[e] = -mean * rstd;
        gm[e] = a.gamma[c0 + e];
        bt[e] = a.beta[c0 + e];
        p1[e] = proj[gi];
        p2[e] = proj[gi + 1];
    }
    const bool second = c0 >= a.C1;
    const bf16* src = second ? a.x2 : a.x1;
    const bf16* dres = second ? dres2 : dres1;
    bf16* dst = second ? dx2 : dx1;
    const int ld = second ? g.C - a.C1 : a.C1;
    const int cc = second ? c0 - a.C1 : c0;
    for (int64_t r = r0 + rl; r < r1; r += g.rows_par) {
        const int64_t row = (int64_t)b * g.rows_per_batch + r;
        float f[8], d[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(src + row * ld + cc), f);
        unpack8(*reinterpret_cast<const u32x4*>(a.dy + row * a.lddy + c0), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = f[e] * sc[e] + sh[e];
            float gy = d[e];
            if (a.silu) gy *= silu_grad_f(xh * gm[e] + bt[e]);
            o[e] = sc[e] * (gy * gm[e] - p1[e] - xh * p2[e]);
        }
        if (dres) {
            float q[8];
            unpack8(*reinterpret_cast<const u32x4*>(dres + row * ld + cc), q);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += q[e];
        }
        *reinterpret_cast<u32x4*>(dst + row * ld + cc) = pack8(o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward rows: one wave per row (C <= 1536)
// PART: the launch also leaves d beta / d gamma partial sums, one [2][C] slab per block in `partials` (each lane adds dy and
// dy * xhat of its columns over the rows of its wave, the four waves of the block are added through LDS in wave order;
// colfinal_kernel adds the slabs): no second pass over x and dy, no row statistics in memory
template <int MAXC, bool PART>
__global__ void __launch_bounds__(256) ln_bwd_rows_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, int64_t rows,
                                                          int C, int ldx, int lddy, const float* __restrict__ gamma, float eps,
                                                          const bf16* __restrict__ dres, int ldres, bf16* __restrict__ dx,
                                                          int lddx, float* __restrict__ partials) {
    extern __shared__ float part_lds[];              // PART: [4 waves][2][C]
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nch = C / 8;
    float gm[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[i][e] = gamma[ch * 8 + e];
        }
    }
    const float invC = 1.0f / (float)C;
    [[maybe_unused]] f32x2 pb[MAXC][4], pg[MAXC][4];     // PART: d beta, d gamma of this lane's columns (pairs)
    if constexpr (PART) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) pb[i][e] = pg[i][e] = f32x2{0.f, 0.f};
    }
    for (int64_t r = wave_global; r < rows; r += nwaves) {
        float f[MAXC][8], d[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + ch * 8), f[i]);
                unpack8(*reinterpret_cast<const u32x4*>(dy + r * lddy + ch * 8), d[i]);
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
                for (int e = 0; e < 8; ++e) { const float t = f[i][e] - mean; q += t * t; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + eps);
        // the two row sums as whole packed pairs over (even, odd) elements: written with scalar accumulators, the SLP vectoriser
        // pairs c1 with c2 and reads the products half-swapped (v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]), a form that
        // dropped a term next to a co-tenant process on MI355X (profiles/r03_flake_root_cause.md; asm_check.py refuses it)
        f32x2 c1v = {0.f, 0.f}, c2v = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 xh = (f32x2{f[i][e], f[i][e + 1]} - f32x2{mean, mean}) * f32x2{rstd, rstd};
                    const f32x2 dyv = f32x2{d[i][e], d[i][e + 1]};
                    if constexpr (PART) {
                        pb[i][e >> 1] += dyv;
                        pg[i][e >> 1] = __builtin_elementwise_fma(dyv, xh, pg[i][e >> 1]);
                    }
                    const f32x2 gg = dyv * f32x2{gm[i][e], gm[i][e + 1]};
                    f[i][e] = xh[0]; f[i][e + 1] = xh[1];
                    d[i][e] = gg[0]; d[i][e + 1] = gg[1];
                    c1v += gg;
                    c2v = __builtin_elementwise_fma(gg, xh, c2v);
                }
            }
        }
        const float c1 = wave_sum(c1v[0] + c1v[1]) * invC;
        const float c2 = wave_sum(c2v[0] + c2v[1]) * invC;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (d[i][e] - c1 - f[i][e] * c2);
                if (dres) {
                    float qv[8];
                    unpack8(*reinterpret_cast<const u32x4*>(dres + r * ldres + ch * 8), qv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += qv[e];
                }
                *reinterpret_cast<u32x4*>(dx + r * lddx + ch * 8) = pack8(o);
            }
        }
    }
    if constexpr (PART) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
                float* o = part_lds + (size_t)wave * 2 * C + ch * 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[2 * e] = pb[i][e][0]; o[2 * e + 1] = pb[i][e][1];
                    o[C + 2 * e] = pg[i][e][0]; o[C + 2 * e + 1] = pg[i][e][1];
                }
            }
        }
        __syncthreads();
        float* out = partials + (int64_t)blockIdx.x * 2 * C;
        for (int j = threadIdx.x; j < 2 * C; j += 256)
            out[j] = ((part_lds[j] + part_lds[2 * C + j]) + part_lds[4 * C + j]) + part_lds[6 * C + j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEGLU on the interleaved projection layout of the GEMM epilogue: columns [32g, 32g+16) values, [32g+16, 32g+32) gates
__global__ void __launch_bounds__(256) geglu_fwd_kernel(const bf16* __restrict__ pre, int64_t n_out8, int inner8, int ldp,
                                                        bf16* __restrict__ out, int ldo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out8) return;
    const int64_t row = i / inner8;
    const int c8 = (int)(i - row * inner8);          // 8-column chunk of the output
    const int g = c8 >> 1, half = c8 & 1;
    const bf16* p = pre + row * ldp + 32 * g + 8 * half;
    float v[8], gt[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(p), v);
    unpack8(*reinterpret_cast<const u32x4*>(p + 16), gt);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] * gelu_erf_f(gt[e]);
    *reinterpret_cast<u32x4*>(out + row * ldo + c8 * 8) = pack8(o);
}

__global__ void __launch_bounds__(256) geglu_bwd_kernel(const bf16* __restrict__ pre, const bf16* __restrict__ dout, int64_t n_out8,
                                                        int inner8, int ldp, int lddo, bf16* __restrict__ dpre, int lddp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out8) return;
    const int64_t row = i / inner8;
    const int c8 = (int)(i - row * inner8);
    const int g = c8 >> 1, half = c8 & 1;
    const bf16* p = pre + row * ldp + 32 * g + 8 * half;
    float v[8], gt[8], d[8], dv[8], dg[8];
    unpack8(*reinterpret_cast<const u32x4*>(p), v);
    unpack8(*reinterpret_cast<const u32x4*>(p + 16), gt);
    unpack8(*reinterpret_cast<const u32x4*>(dout + row * lddo + c8 * 8), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = gt[e];
        const float cdf = 0.5f * (1.0f + erff(x * 0.7071067811865476f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        dv[e] = d[e] * x * cdf;
        dg[e] = d[e] * v[e] * (cdf + x * pdf);
    }
    bf16* q = dpre + row * lddp + 32 * g + 8 * half;
    *reinterpret_cast<u32x4*>(q) = pack8(dv);
    *reinterpret_cast<u32x4*>(q + 16) = pack8(dg);
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) add_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y,
                                                       int64_t rows, int cols8, int lda, int ldb, int ldy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols8) return;
    const int64_t r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    float f[8], g[8];
    unpack8(*reinterpret_cast<const u32x4*>(a + r * lda + c), f);
    unpack8(*reinterpret_cast<const u32x4*>(b + r * ldb + c), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
    *reinterpret_cast<u32x4*>(y + r * ldy + c) = pack8(f);
}

// nearest-2x upsample backward: dx[img, y, x, :] = sum of the 2x2 block of du[img, 2y.., 2x.., :]
__global__ void __launch_bounds__(256) sumpool2x_kernel(const bf16* __restrict__ du, int n_img, int H, int W, int C8,
                                                        bf16* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_img * H * W * C8;
    if (i >= total) return;
    const int c = (int)(i % C8) * 8;
    int64_t pix = i / C8;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const int64_t img = pix / H;
    const int C = C8 * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
            float f[8];
            unpack8(*reinterpret_cast<const u32x4*>(du + ((img * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dxx) * C + c), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
    *reinterpret_cast<u32x4*>(dx + ((img * H + y) * W + x) * C + c) = pack8(s);
}

// stride-2 conv backward helper: z[img, 2y, 2x, :] = d[img, y, x, :], zero elsewhere (z is [n_img, 2H, 2W, C])
__global__ void __launch_bounds__(256) zero_insert2x_kernel(const bf16* __restrict__ d, int n_img, int H, int W, int C8,
                                                            bf16* __restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_img * 2 * H * 2 * W * C8;
    if (i >= total) return;
    const int c = (int)(i % C8) * 8;
    int64_t pix = i / C8;
    const int x = (int)(pix % (2 * W));
    pix /= 2 * W;
    const int y = (int)(pix % (2 * H));
    const int64_t img = pix / (2 * H);
    const int C = C8 * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const u32x4*>(d + ((img * H + (y >> 1)) * W + (x >> 1)) * C + c);
    *reinterpret_cast<u32x4*>(z + i * 8) = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// epsilon-MSE (train.py:380): loss = mean((pred[:, :, cond:] - target)^2); dpred = 2 (pred - target) / N, zero on the
// conditioning frames.  Block partials -> ordered final sum.
__global__ void __launch_bounds__(256) mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, int BC, int F_total,
                                                  int cond_f, int HW, float inv_n, float* __restrict__ dpred,
                                                  float* __restrict__ partial) {
    const int64_t total = (int64_t)BC * F_total * HW;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const int64_t t = i / HW;
        const int f = (int)(t % F_total);
        const int64_t bc = t / F_total;
        float g = 0.f;
        if (f >= cond_f) {
            const float d = pred[i] - target[(bc * (F_total - cond_f) + (f - cond_f)) * HW + hw];
            s += d * d;
            g = 2.0f * d * inv_n;
        }
        dpred[i] = g;
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void sum_partials_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += partial[i];
        *out = s * scale;
    }
}

// conv_out backward (input gradient only; conv_out is frozen): dx[b,f,y,x,ci] = sum_{co,ky,kx} dpred[b,co,f,y+1-ky,x+1-kx]
// * w[co,ci,ky,kx].  Wt fp32 [Cout][3][3][C0] (the direct conv_out layout).  One block per (image row), weights in LDS.
template <int COUT>
__global__ void __launch_bounds__(256) conv_out_bwd_kernel(const float* __restrict__ dpred, int B, int C0, int F, int H, int W_,
                                                           const float* __restrict__ Wt, bf16* __restrict__ dx) {
    extern __shared__ float sm[];
    float* ws = sm;                                   // [COUT*9][C0]
    float* patch = sm + COUT * 9 * C0;                // [COUT][3][W_+2]
    for (int i = threadIdx.x; i < COUT * 9 * C0; i += 256) ws[i] = Wt[i];
    const int y = blockIdx.x % H;
    const int img = blockIdx.x / H;                   // b*F + f
    const int b = img / F, f = img % F;
    const int PW = W_ + 2;
    for (int i = threadIdx.x; i < COUT * 3 * PW; i += 256) {
        const int px = i % PW, r = (i / PW) % 3, co = i / (3 * PW);
        const int yy = y + 1 - r, xx = px - 1;        // r = ky: source row y + 1 - ky
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W_) v = dpred[((((int64_t)b * COUT + co) * F + f) * H + yy) * W_ + xx];
        patch[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W_ * C0; i += 256) {
        const int ci = i % C0, x = i / C0;
        float s = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    s += patch[(co * 3 + ky) * PW + (x + 1 - kx) + 1] * ws[((co * 3 + ky) * 3 + kx) * C0 + ci];
        dx[(((int64_t)img * H + y) * W_ + x) * C0 + ci] = (bf16)s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += g[i] * g[i];
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// torch.optim.AdamW (train.py:226-232) on flat fp32 buffers; the gradient is first scaled by
// min(1, max_norm / (sqrt(*sumsq) + 1e-6)) when sumsq != NULL (torch.nn.utils.clip_grad_norm_, train.py:384).
// Also refreshes the bf16 working copy of the parameters.
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, const float* __restrict__ sumsq,
                                                    float max_norm, bf16* __restrict__ p_bf16) {
    float coef = 1.0f;
    if (sumsq) coef = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (p_bf16) p_bf16[i] = (bf16)pi;
    }
}


// text loss of the FSTextTransformer initialisation stage (train.py:346-347, `--text_loss`): loss_text = mean_{b,l,c}
// (mean_f y[b,f,l,c] - t[b,l,c])^2; its gradient 2 (mean_f y - t) / (b l C F) is ADDED to dy for every frame.
// y, dy: bf16 [b][F][LC]; t: fp32 [b][LC].  One thread per 8 channels.
__global__ void __launch_bounds__(256) text_loss_kernel(const bf16* __restrict__ y, const float* __restrict__ t, int b, int F, int64_t LC8,
                                                        float gscale, bf16* __restrict__ dy, float* __restrict__ partial) {
    const int64_t total = (int64_t)b * LC8;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bi = i / LC8, c8 = i - bi * LC8;
        const int64_t LC = LC8 * 8;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = 0.f;
        for (int f = 0; f < F; ++f) {
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(y + ((bi * F + f) * LC) + c8 * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] += v[e];
        }
        float g[8];
        const float invF = 1.0f / (float)F;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = m[e] * invF - t[bi * LC + c8 * 8 + e];
            s += d * d;
            g[e] = d * gscale;
        }
        for (int f = 0; f < F; ++f) {
            bf16* p = dy + ((bi * F + f) * LC) + c8 * 8;
            float v[8];
            unpack8(*reinterpret_cast<const u32x4*>(p), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += g[e];
            *reinterpret_cast<u32x4*>(p) = pack8(v);
        }
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// y = beta * y + alpha * x (gradient accumulation over micro-batches, train.py:321 `accelerator.accumulate`)
__global__ void __launch_bounds__(256) axpby_kernel(float* y, const float* x /* may alias y */, float alpha, float beta,
                                                    int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 yv = {0.f, 0.f, 0.f, 0.f};
        if (beta != 0.f) yv = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[e] = beta * yv[e] + alpha * xv[e];
        reinterpret_cast<f32x4*>(y)[i] = yv;
    }
}

}  // namespace

extern "C" int seer_transpose_bf16(const void* x, int64_t rows, int32_t cols, int32_t ldx, void* y, int64_t ldy,
                                   void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < rows) return SEER_EINVAL;
    const int64_t rows_pad = ldy;
    dim3 grid((unsigned)((rows_pad + 63) / 64), (unsigned)((cols + 63) / 64));
    if (cols % 8 == 0 && ldx % 8 == 0 && ldy % 64 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
        hipLaunchKernelGGL(transpose_vec_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           reinterpret_cast<const bf16*>(x), rows, cols, ldx, reinterpret_cast<bf16*>(y), ldy);
        SEER_LAUNCH_CHECK();
        return SEER_OK;
    }
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(x), rows, cols, ldx, reinterpret_cast<bf16*>(y), ldy, rows_pad);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_transpose_batched_bf16(const seer_transpose_item* items, int32_t n_items, int64_t total_tiles, void* stream) {
    if (!items || n_items <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return SEER_EINVAL;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       items, n_items);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int64_t seer_colsum_workspace_floats(int64_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return SEER_EINVAL;
    return ((rows + CS_ROWS_MIN - 1) / CS_ROWS_MIN) * 2 * (int64_t)cols + 2 * rows;
}

extern "C" int seer_colsum_bf16(const void* x, int64_t rows, int32_t cols, int32_t ldx, float* out, float* workspace,
                                void* stream) {
    if (!x || !out || !workspace || rows <= 0 || cols <= 0 || cols % 8 || ldx % 8) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const ColGeom g = col_geom(cols, rows);
    ColArgs a{};
    a.x1 = reinterpret_cast<const bf16*>(x);
    a.ldx = ldx;
    const size_t lds = (size_t)g.rows_par * g.cpp * 8 * sizeof(float);
    hipLaunchKernelGGL(colpartial_kernel<0>, dim3(g.nchunks, col_blocks(g), 1), dim3(256), lds, st, a, g, workspace);
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(colfinal_kernel, dim3((cols + 63) / 64, 1, 1), dim3(1024), 0, st, workspace, g.nchunks, 1, cols, out,
                       (float*)nullptr);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

namespace {
int64_t ln_bwd_slabs(int64_t rows) {
    const int64_t blocks = (rows + 7) / 8;
    return blocks > 1024 ? 1024 : blocks;
}
}  // namespace

extern "C" int64_t seer_layernorm_bwd_slabs(int64_t rows) { return rows > 0 ? ln_bwd_slabs(rows) : SEER_EINVAL; }

static int layernorm_bwd_impl(const void* x, const void* dy, int64_t rows, int32_t C, int32_t ldx, int32_t lddy,
                              const float* gamma, float eps, const void* dres, int32_t ldres, void* dx, int32_t lddx,
                              float* dgamma, float* dbeta, float* workspace, bool partials_only, void* stream) {
    if (!x || !dy || !gamma || !dx || rows <= 0 || C <= 0 || C % 8 || (ldx | lddy | lddx) % 8) return SEER_EINVAL;
    if (dres && ldres % 8) return SEER_EINVAL;
    if ((dgamma != nullptr) != (dbeta != nullptr)) return SEER_EINVAL;
    if (dgamma && !workspace) return SEER_EINVAL;
    if (C > 64 * 8 * 3) return SEER_ENOSYS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    const bf16* dyb = reinterpret_cast<const bf16*>(dy);
    const bf16* rb = reinterpret_cast<const bf16*>(dres);
    bf16* dxb = reinterpret_cast<bf16*>(dx);
    // plain: one row per wave; with d gamma / d beta: two rows per wave, so that the per-block partial slabs stay few
    // (<= ceil(rows / 8) slabs of [2][C] floats = the first region of seer_colsum_workspace_floats(rows, C))
    int64_t blocks = dgamma ? (rows + 7) / 8 : (rows + 3) / 4;
    const int64_t cap = dgamma ? 1024 : 4096;
    if (blocks > cap) blocks = cap;
    const size_t lds = dgamma ? (size_t)8 * C * sizeof(float) : 0;
#define SEER_LNB(MC)                                                                                                              \
    do {                                                                                                                          \
        if (dgamma)                                                                                                               \
            hipLaunchKernelGGL((ln_bwd_rows_kernel<MC, true>), dim3((unsigned)blocks), dim3(256), lds, st, xb, dyb, rows, C, ldx,  \
                               lddy, gamma, eps, rb, ldres, dxb, lddx, workspace);                                               \
        else                                                                                                                      \
            hipLaunchKernelGGL((ln_bwd_rows_kernel<MC, false>), dim3((unsigned)blocks), dim3(256), 0, st, xb, dyb, rows, C, ldx,   \
                               lddy, gamma, eps, rb, ldres, dxb, lddx, (float*)nullptr);                                         \
    } while (0)
    if (C <= 512) SEER_LNB(1);
    else if (C <= 1024) SEER_LNB(2);
    else SEER_LNB(3);
#undef SEER_LNB
    SEER_LAUNCH_CHECK();
    if (dgamma && !partials_only) {
        hipLaunchKernelGGL(colfinal_kernel, dim3((C + 63) / 64, 2, 1), dim3(1024), 0, st, workspace, (int)blocks, 2, C, dbeta, dgamma);
        SEER_LAUNCH_CHECK();
    }
    return SEER_OK;
}

extern "C" int seer_layernorm_bwd(const void* x, const void* dy, int64_t rows, int32_t C, int32_t ldx, int32_t lddy,
                                  const float* gamma, float eps, const void* dres, int32_t ldres, void* dx, int32_t lddx,
                                  float* dgamma, float* dbeta, float* workspace, void* stream) {
    return layernorm_bwd_impl(x, dy, rows, C, ldx, lddy, gamma, eps, dres, ldres, dx, lddx, dgamma, dbeta, workspace, false, stream);
}

extern "C" int seer_layernorm_bwd_partials(const void* x, const void* dy, int64_t rows, int32_t C, int32_t ldx, int32_t lddy,
                                           const float* gamma, float eps, const void* dres, int32_t ldres, void* dx, int32_t lddx,
                                           float* workspace, void* stream) {
    if (!workspace) return SEER_EINVAL;
    return layernorm_bwd_impl(x, dy, rows, C, ldx, lddy, gamma, eps, dres, ldres, dx, lddx, workspace, workspace, workspace, true, stream);
}

extern "C" int seer_colfinal_grouped(const seer_colfinal_item* items, int32_t n_items, void* stream) {
    if (!items || n_items <= 0) return SEER_EINVAL;
    for (int i = 0; i < n_items; ++i)
        if (!items[i].ws || items[i].nblocks <= 0 || items[i].C <= 0 || items[i].NV < 1 || items[i].NV > 2) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int base = 0; base < n_items; base += CF_GROUP) {
        CfGroup g{};
        g.n = n_items - base < CF_GROUP ? n_items - base : CF_GROUP;
        int blk = 0;
        for (int j = 0; j < g.n; ++j) {
            const seer_colfinal_item& it = items[base + j];
            g.it[j] = CfItem{it.ws, it.out0, it.out1, it.nblocks, it.NV, it.C, blk};
            blk += ((it.C + 63) / 64) * it.NV;
        }
        hipLaunchKernelGGL(colfinal_grouped_kernel, dim3(blk), dim3(1024), 0, st, g);
        SEER_LAUNCH_CHECK();
    }
    return SEER_OK;
}

extern "C" int64_t seer_groupnorm_bwd_workspace_floats(int32_t C, int32_t batch, int64_t rows_per_batch, int32_t groups) {
    if (C <= 0 || batch <= 0 || rows_per_batch <= 0 || groups <= 0) return SEER_EINVAL;
    const int64_t nchunks = (rows_per_batch + CS_ROWS_MIN - 1) / CS_ROWS_MIN;  // upper bound over the chunk sizes in use
    return (int64_t)batch * nchunks * 2 * C + 2 * (int64_t)batch * C + 2 * (int64_t)batch * groups;
}

extern "C" int seer_groupnorm_bwd(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                                  int64_t rows_per_batch, int32_t groups, const float* stats, double count, float eps,
                                  const float* gamma, const float* beta, int32_t silu, const void* dy, const void* dres1,
                                  const void* dres2, void* dx1, void* dx2, float* dgamma, float* dbeta, float* workspace,
                                  void* stream) {
    if (!x1 || !stats || !gamma || !beta || !dy || !dx1 || !workspace || batch <= 0 || rows_per_batch <= 0 || count <= 0)
        return SEER_EINVAL;
    if (!x2) C2 = 0;
    if (x2 && !dx2) return SEER_EINVAL;
    const int C = C1 + C2;
    if (C1 <= 0 || C2 < 0 || C1 % 8 || C2 % 8 || groups <= 0 || C % groups) return SEER_EINVAL;
    if ((dgamma != nullptr) != (dbeta != nullptr)) return SEER_EINVAL;
    if (C / groups > 128) return SEER_ENOSYS;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const ColGeom g = col_geom(C, rows_per_batch, batch);
    float* sums = workspace + (int64_t)batch * ((rows_per_batch + CS_ROWS_MIN - 1) / CS_ROWS_MIN) * 2 * C;      // [2][batch][C]
    float* proj = sums + 2 * (int64_t)batch * C;                                                      // [batch][groups][2]
    ColArgs a{};
    a.x1 = reinterpret_cast<const bf16*>(x1);
    a.x2 = reinterpret_cast<const bf16*>(x2);
    a.dy = reinterpret_cast<const bf16*>(dy);
    a.C1 = C1;
    a.lddy = C;
    a.stats = stats;
    a.inv_count = (float)(1.0 / count);
    a.eps = eps;
    a.gamma = gamma;
    a.beta = beta;
    a.cpg = C / groups;
    a.groups = groups;
    a.silu = silu;
    dim3 grid(g.nchunks, col_blocks(g), batch);
    const size_t lds = (size_t)g.rows_par * g.cpp * 16 * sizeof(float);
    hipLaunchKernelGGL(colpartial_kernel<2>, grid, dim3(256), lds, st, a, g, workspace);
    SEER_LAUNCH_CHECK();
#define SEER_GNB_GROUP(CL)                                                                                                          \
    hipLaunchKernelGGL(gn_bwd_group_fused_kernel<CL>, dim3(groups, dgamma ? 1 : batch), dim3(1024), 0, st, workspace, g.nchunks, batch, \
                       C, a.cpg, groups, gamma, a.inv_count, proj, dgamma, dbeta)
    if (a.cpg <= 16) SEER_GNB_GROUP(16);
    else if (a.cpg <= 32) SEER_GNB_GROUP(32);
    else SEER_GNB_GROUP(64);
#undef SEER_GNB_GROUP
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, dim3(256), 0, st, a, g, proj, reinterpret_cast<const bf16*>(dres1),
                       reinterpret_cast<const bf16*>(dres2), reinterpret_cast<bf16*>(dx1), reinterpret_cast<bf16*>(dx2));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_geglu_fwd(const void* pre, int64_t rows, int32_t inner, int32_t ldp, void* out, int32_t ldo, void* stream) {
    if (!pre || !out || rows <= 0 || inner <= 0 || inner % 16 || ldp % 8 || ldo % 8) return SEER_EINVAL;
    const int64_t n = rows * (inner / 8);
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(pre), n, inner / 8, ldp, reinterpret_cast<bf16*>(out), ldo);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_geglu_bwd(const void* pre, const void* dout, int64_t rows, int32_t inner, int32_t ldp, int32_t lddo,
                              void* dpre, int32_t lddp, void* stream) {
    if (!pre || !dout || !dpre || rows <= 0 || inner <= 0 || inner % 16 || (ldp | lddo | lddp) % 8) return SEER_EINVAL;
    const int64_t n = rows * (inner / 8);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(pre), reinterpret_cast<const bf16*>(dout), n, inner / 8, ldp, lddo,
                       reinterpret_cast<bf16*>(dpre), lddp);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_add_bf16(const void* a, int32_t lda, const void* b, int32_t ldb, void* y, int32_t ldy, int64_t rows,
                             int32_t cols, void* stream) {
    if (!a || !b || !y || rows <= 0 || cols <= 0 || cols % 8 || (lda | ldb | ldy) % 8) return SEER_EINVAL;
    const int64_t n = rows * (cols / 8);
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(a), reinterpret_cast<const bf16*>(b), reinterpret_cast<bf16*>(y), rows,
                       cols / 8, lda, ldb, ldy);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_sumpool2x_bf16(const void* du, int32_t n_img, int32_t H, int32_t W_, int32_t C, void* dx, void* stream) {
    if (!du || !dx || n_img <= 0 || H <= 0 || W_ <= 0 || C <= 0 || C % 8) return SEER_EINVAL;
    const int64_t n = (int64_t)n_img * H * W_ * (C / 8);
    hipLaunchKernelGGL(sumpool2x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(du), n_img, H, W_, C / 8, reinterpret_cast<bf16*>(dx));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_zero_insert2x_bf16(const void* d, int32_t n_img, int32_t H, int32_t W_, int32_t C, void* z, void* stream) {
    if (!d || !z || n_img <= 0 || H <= 0 || W_ <= 0 || C <= 0 || C % 8) return SEER_EINVAL;
    const int64_t n = (int64_t)n_img * 4 * H * W_ * (C / 8);
    hipLaunchKernelGGL(zero_insert2x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16*>(d), n_img, H, W_, C / 8, reinterpret_cast<bf16*>(z));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_mse_loss_grad(const float* pred, const float* target, int32_t B, int32_t C, int32_t F_total, int32_t cond_f,
                                  int32_t HW, float* loss, float* dpred, float* workspace /* 1024 floats */, void* stream) {
    if (!pred || !target || !loss || !dpred || !workspace || B <= 0 || C <= 0 || HW <= 0 || cond_f < 0 || cond_f >= F_total)
        return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * C * F_total * HW;
    const double n = (double)B * C * (F_total - cond_f) * HW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, st, pred, target, B * C, F_total, cond_f, HW, (float)(1.0 / n),
                       dpred, workspace);
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, (float)(1.0 / n), loss);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

namespace { std::once_flag g_conv_out_bwd_lds_once; }      // dynamic-LDS opt-in of conv_out_bwd_kernel (no function-local statics)

extern "C" int seer_conv_out_bwd(const float* dpred, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                                 int32_t Cout, void* dx, void* stream) {
    if (!dpred || !Wt || !dx || B <= 0 || C0 <= 0 || F <= 0 || H <= 0 || W_ <= 0) return SEER_EINVAL;
    if (Cout != 4) return SEER_ENOSYS;
    const size_t lds = ((size_t)Cout * 9 * C0 + (size_t)Cout * 3 * (W_ + 2)) * sizeof(float);
    if (lds > 160 * 1024) return SEER_ENOSYS;
    if (lds > 64 * 1024) {
        std::call_once(g_conv_out_bwd_lds_once, [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_out_bwd_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    }
    hipLaunchKernelGGL(conv_out_bwd_kernel<4>, dim3((unsigned)(B * F * H)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       dpred, B, C0, F, H, W_, Wt, reinterpret_cast<bf16*>(dx));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_sumsq_f32(const float* g, int64_t n, float* out, float* workspace /* 1024 floats */, void* stream) {
    if (!g || !out || !workspace || n <= 0) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, st, g, n, workspace);
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, 1.0f, out);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int32_t step, const float* grad_sumsq, float max_norm,
                               void* p_bf16, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return SEER_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, n,
                       lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_sumsq, max_norm,
                       reinterpret_cast<bf16*>(p_bf16));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_axpby_f32(float* y, const float* x, float alpha, float beta, int64_t n, void* stream) {
    if (!y || !x || n <= 0 || n % 4 || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(x) & 15)) return SEER_EINVAL;
    int64_t blocks = (n / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y, x, alpha, beta,
                       n / 4);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_text_loss_grad(const void* y, const float* target, int32_t b, int32_t F, int64_t LC, void* dy, float* loss,
                                   float* workspace /* 1024 floats */, void* stream) {
    if (!y || !target || !dy || !loss || !workspace || b <= 0 || F <= 0 || LC <= 0 || LC % 8) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)b * (LC / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    const double n = (double)b * (double)LC;
    hipLaunchKernelGGL(text_loss_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const bf16*>(y), target, b, F, LC / 8,
                       (float)(2.0 / (n * F)), reinterpret_cast<bf16*>(dy), workspace);
    SEER_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, (float)(1.0 / n), loss);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
