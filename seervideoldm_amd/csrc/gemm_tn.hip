// Weight-gradient GEMM for gfx950 (MI355X): C[n][k] = sum_m A[m][n] * B[m][k]  (dW = dY^T X of the training step,
// train.py:382 through every trainable nn.Linear), bf16 operands as they sit in memory -- token-major [rows, channels] -- fp32
// result.  Both operands are "transposed" with respect to the MFMA fragment layouts (the contraction index m is the row),
// so instead of materialising dY^T and X^T (seer_transpose_bf16 + the forward GEMM) the 64-row operand tiles go to LDS as
// they are and BOTH fragments come out of ds_read_b64_tr_b16 -- the idiom of the attention kernels' V^T operand.
//
// Tile 128 (n) x 128 (k) per block, 4 waves (2 x 2) of 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 accumulators; the long
// contraction (rows of the activation: 12 288 at the 32x32 level) is split across grid.z, slices write fp32 partial tiles to a
// workspace and a second kernel adds them in slice order (deterministic, no atomics).
#include "seer_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

constexpr int TM = 64;            // contraction rows per LDS tile
constexpr int TRS = 160;          // LDS row stride (elements): 320 B == 64 (mod 256) -> conflict-free transposed reads
constexpr int NCH = 4;            // 16-byte chunks per thread per operand per tile (64 rows x 16 chunks / 256 threads)

// one 128 x 128 tile of one K slice: tile (tn, tk), contraction rows [z * m_chunk, min(M, (z + 1) * m_chunk)); Cz / cz = where this
// slice's tile and column sums go (the caller's C / colsum, or its slice of the workspace)
__device__ __forceinline__ void tn_tile(bf16* imgA, bf16* imgB, const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                        int M, int N, int K, int m_chunk, int tn, int tk, int z, float* __restrict__ Cz,
                                        float* __restrict__ cz /* [N], or NULL */) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int lh = lane >> 5;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = z * m_chunk;
    const int m_end = min(M, m_begin + m_chunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // column sums of A (the bias gradient sum_m dY[m][n]) ride along as one more MFMA against a fragment of ones, in the
    // waves that own the first 64 output columns of the first column tile
    const bool do_colsum = cz != nullptr && tk == 0 && wk == 0;
    f32x16 cacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[i][r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    u32x4 ra[NCH], rb[NCH];
    auto prefetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, ch = idx & 15;
            const int m = m0 + row;
            const u32x4 z = {0u, 0u, 0u, 0u};
            ra[i] = (m < m_end && n0 + 8 * ch < N) ? *reinterpret_cast<const u32x4*>(A + (int64_t)m * lda + n0 + 8 * ch) : z;
            rb[i] = (m < m_end && k0 + 8 * ch < K) ? *reinterpret_cast<const u32x4*>(B + (int64_t)m * ldb + k0 + 8 * ch) : z;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, ch = idx & 15;
            *reinterpret_cast<u32x4*>(imgA + row * TRS + 8 * ch) = ra[i];
            *reinterpret_cast<u32x4*>(imgB + row * TRS + 8 * ch) = rb[i];
        }
    };

    const int li = lane & 15, g16 = (lane >> 4) & 1;
    // transposed-read base: contraction rows 4*lh + (li >> 2) (+8), columns 16*g16 + 4*(li & 3) of a 32-wide operand tile
    const int toff = (4 * lh + (li >> 2)) * TRS + 16 * g16 + 4 * (li & 3);
    auto frag = [&](const bf16* img, int col0, int s) -> bf16x8 {
        const bf16* a0 = img + toff + (16 * s) * TRS + col0;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(a0 + 8 * TRS));
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    if (m_begin < m_end) prefetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += TM) {
        __syncthreads();
        commit();
        __syncthreads();
        if (m0 + TM < m_end) prefetch(m0 + TM);
#pragma unroll
        for (int s = 0; s < TM / 16; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag(imgA, 64 * wn + 32 * i, s);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = frag(imgB, 64 * wk + 32 * j, s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) cacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, cacc[i], 0, 0, 0);
            }
        }
    }
    if (do_colsum && (lane & 31) == 0) {          // every output column of cacc holds the same sums: lane column 0 writes them
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < N) cz[n] = cacc[i][r];
            }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + 64 * wk + 32 * j + (lane & 31);
            if (k < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (n < N) Cz[(int64_t)n * K + k] = acc[i][j][r];
                }
            }
        }
}

__global__ void __launch_bounds__(256, 3) seer_gemm_tn_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ B, int ldb,
                                                           int M, int N, int K, int m_chunk, float* __restrict__ C, int64_t slice_stride,
                                                           float* __restrict__ colsum /* [N] per slice, or NULL */, int64_t colsum_stride) {
    __shared__ __attribute__((aligned(16))) bf16 imgA[TM * TRS];
    __shared__ __attribute__((aligned(16))) bf16 imgB[TM * TRS];
    tn_tile(imgA, imgB, A, lda, B, ldb, M, N, K, m_chunk, blockIdx.x, blockIdx.y, blockIdx.z, C + (int64_t)blockIdx.z * slice_stride,
            colsum ? colsum + (int64_t)blockIdx.z * colsum_stride : nullptr);
}

// ---- grouped form: the weight gradients of MANY layers in one launch (the deferred dW products of a backward pass: each is a small
// launch on its own -- 9 to 800 tiles, a contraction of 192 to 12 288 rows -- and none depends on another).  The problem table
// travels BY VALUE in the kernel arguments (a captured launch keeps it; no device table to build or keep alive): <= TN_GROUP problems
// per launch, sorted by the host longest contraction first; workgroup -> (problem, slice, tile) by a scan over the first-workgroup
// column.
constexpr int TN_GROUP = 48;
struct TnProb {
    const bf16* A; const bf16* B; float* dst; float* cdst;     // dst / cdst: C and colsum, or slice 0 of their workspace
    int64_t slice;                                               // floats between slices (0: unsplit)
    int lda, ldb, M, N, K, m_chunk, tiles_k, tiles;              // tiles = tiles_n * tiles_k
    int wg0, pad;                                                // first workgroup of the problem
};
struct TnGroup {
    TnProb p[TN_GROUP];
    int n;
};

__global__ void __launch_bounds__(256, 3) seer_gemm_tn_grouped_kernel(const TnGroup g) {
    __shared__ __attribute__((aligned(16))) bf16 imgA[TM * TRS];
    __shared__ __attribute__((aligned(16))) bf16 imgB[TM * TRS];
    const int wg = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && wg >= g.p[i + 1].wg0) ++i;
    const TnProb& q = g.p[i];
    const int local = wg - q.wg0;
    const int z = local / q.tiles, t = local % q.tiles;
    tn_tile(imgA, imgB, q.A, q.lda, q.B, q.ldb, q.M, q.N, q.K, q.m_chunk, t / q.tiles_k, t % q.tiles_k, z, q.dst + (int64_t)z * q.slice,
            q.cdst ? q.cdst + (int64_t)z * q.slice : nullptr);
}

struct TnRed {
    const float* ws; float* out; float* out2;
    int64_t slice4, count4, nk4;
    int splits, blk0;
};
struct TnRedGroup {
    TnRed r[TN_GROUP];
    int n;
};

// slices are [N*K (+ N column sums)] floats; float4 i of the sum goes to out (i < nk4) or to out2 (the column sums)
__device__ __forceinline__ void tn_reduce(int64_t i, const float* __restrict__ ws, int splits, int64_t slice4, int64_t count4,
                                          int64_t nk4, float* __restrict__ out, float* __restrict__ out2) {
    if (i >= count4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
    for (int z = 1; z < splits; ++z) {
        const f32x4 v = reinterpret_cast<const f32x4*>(ws + (int64_t)z * slice4 * 4)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
    if (i < nk4) reinterpret_cast<f32x4*>(out)[i] = s;
    else reinterpret_cast<f32x4*>(out2)[i - nk4] = s;
}

__global__ void __launch_bounds__(256) tn_reduce_kernel(const float* __restrict__ ws, int splits, int64_t slice4, int64_t count4,
                                                        int64_t nk4, float* __restrict__ out, float* __restrict__ out2) {
    tn_reduce((int64_t)blockIdx.x * 256 + threadIdx.x, ws, splits, slice4, count4, nk4, out, out2);
}

__global__ void __launch_bounds__(256) tn_reduce_grouped_kernel(const TnRedGroup g) {
    const int blk = blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && blk >= g.r[i + 1].blk0) ++i;
    const TnRed& r = g.r[i];
    tn_reduce((int64_t)(blk - r.blk0) * 256 + threadIdx.x, r.ws, r.splits, r.slice4, r.count4, r.nk4, r.out, r.out2);
}

int tn_splits(int M, int N, int K) {
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    int s = (256 + tiles - 1) / tiles;
    const int max_s = (M + 4 * TM - 1) / (4 * TM);          // at least four LDS tiles of contraction per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

extern "C" int64_t seer_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return SEER_EINVAL;
    const int s = tn_splits(M, N, K);
    return s > 1 ? (int64_t)s * ((int64_t)N * K + N) * sizeof(float) : 0;
}

extern "C" int seer_gemm_tn_f32(const void* A, int32_t lda, const void* B, int32_t ldb, int32_t M, int32_t N, int32_t K, float* C,
                                float* colsum, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return SEER_EINVAL;
    if (N % 8 || K % 8 || lda % 8 || ldb % 8 || lda < N || ldb < K) return SEER_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C) |
         reinterpret_cast<uintptr_t>(colsum)) & 15) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int s = tn_splits(M, N, K);
    const int64_t slice = (int64_t)N * K + N;
    if (s > 1 && (!workspace || workspace_bytes < s * slice * (int64_t)sizeof(float))) return SEER_EINVAL;
    int m_chunk = ((M + s - 1) / s + TM - 1) / TM * TM;
    s = (M + m_chunk - 1) / m_chunk;
    dim3 grid((N + 127) / 128, (K + 127) / 128, s);
    float* ws = reinterpret_cast<float*>(workspace);
    float* dst = s > 1 ? ws : C;
    float* cdst = !colsum ? nullptr : (s > 1 ? ws + (int64_t)N * K : colsum);
    hipLaunchKernelGGL(seer_gemm_tn_kernel, grid, dim3(256), 0, st, reinterpret_cast<const bf16*>(A), lda,
                       reinterpret_cast<const bf16*>(B), ldb, M, N, K, m_chunk, dst, s > 1 ? slice : 0, cdst, s > 1 ? slice : 0);
    SEER_LAUNCH_CHECK();
    if (s > 1) {
        const int64_t nk4 = (int64_t)N * K / 4;
        const int64_t count4 = colsum ? slice / 4 : nk4;           // without column sums the tail of a slice is not read
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, ws, s, slice / 4, count4,
                           nk4, C, colsum);
        SEER_LAUNCH_CHECK();
    }
    return SEER_OK;
}

// ---- grouped entry points
namespace {

// rows of contraction one workgroup of a grouped launch takes (the group fills the chip by its number of problems, so a problem is
// split only to bound the longest workgroup, not to make workgroups)
int tn_group_rows() {
    static const int rows = [] {
        const char* e = getenv("SEER_TN_GROUP_ROWS");
        const int v = e ? atoi(e) : 0;
        return v >= TM ? (v + TM - 1) / TM * TM : 16384;
    }();
    return rows;
}

int tn_group_splits(int M) { return (M + tn_group_rows() - 1) / tn_group_rows(); }

bool tn_item_ok(const seer_tn_item& it) {
    if (!it.A || !it.B || !it.C || it.M <= 0 || it.N <= 0 || it.K <= 0) return false;
    if (it.N % 8 || it.K % 8 || it.lda % 8 || it.ldb % 8 || it.lda < it.N || it.ldb < it.K) return false;
    return !((reinterpret_cast<uintptr_t>(it.A) | reinterpret_cast<uintptr_t>(it.B) | reinterpret_cast<uintptr_t>(it.C) |
              reinterpret_cast<uintptr_t>(it.colsum)) & 15);
}

}  // namespace

extern "C" int64_t seer_gemm_tn_grouped_workspace_bytes(const seer_tn_item* items, int32_t n_items) {
    if (!items || n_items <= 0) return SEER_EINVAL;
    int64_t floats = 0;
    for (int i = 0; i < n_items; ++i) {
        if (!tn_item_ok(items[i])) return SEER_EINVAL;
        const int s = tn_group_splits(items[i].M);
        if (s > 1) floats += (int64_t)s * ((int64_t)items[i].N * items[i].K + items[i].N);
    }
    return floats * (int64_t)sizeof(float);
}

extern "C" int seer_gemm_tn_grouped_f32(const seer_tn_item* items, int32_t n_items, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
    if (!items || n_items <= 0) return SEER_EINVAL;
    const int64_t need = seer_gemm_tn_grouped_workspace_bytes(items, n_items);
    if (need < 0) return (int)need;
    if (need > 0 && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))) return SEER_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // longest contraction first: the workgroups of a launch start in index order
    std::vector<int> order(n_items);
    for (int i = 0; i < n_items; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const int sa = tn_group_splits(items[a].M), sb = tn_group_splits(items[b].M);
        return (items[a].M + sa - 1) / sa > (items[b].M + sb - 1) / sb;
    });
    float* ws = reinterpret_cast<float*>(workspace);
    for (int base = 0; base < n_items; base += TN_GROUP) {
        const int n = std::min(TN_GROUP, n_items - base);
        TnGroup g{};
        TnRedGroup rg{};
        int wg = 0, blk = 0;
        g.n = n;
        for (int j = 0; j < n; ++j) {
            const seer_tn_item& it = items[order[base + j]];
            TnProb& q = g.p[j];
            int s = tn_group_splits(it.M);
            q.m_chunk = ((it.M + s - 1) / s + TM - 1) / TM * TM;
            s = (it.M + q.m_chunk - 1) / q.m_chunk;
            const int64_t slice = (int64_t)it.N * it.K + it.N;
            q.A = reinterpret_cast<const bf16*>(it.A);
            q.B = reinterpret_cast<const bf16*>(it.B);
            q.lda = it.lda; q.ldb = it.ldb; q.M = it.M; q.N = it.N; q.K = it.K;
            q.tiles_k = (it.K + 127) / 128;
            q.tiles = ((it.N + 127) / 128) * q.tiles_k;
            q.wg0 = wg;
            wg += q.tiles * s;
            if (s > 1) {
                q.dst = ws;
                q.cdst = it.colsum ? ws + (int64_t)it.N * it.K : nullptr;
                q.slice = slice;
                TnRed& r = rg.r[rg.n++];
                r.ws = ws; r.out = it.C; r.out2 = it.colsum;
                r.splits = s; r.slice4 = slice / 4; r.nk4 = (int64_t)it.N * it.K / 4;
                r.count4 = it.colsum ? slice / 4 : r.nk4;
                r.blk0 = blk;
                blk += (int)((r.count4 + 255) / 256);
                ws += (int64_t)tn_group_splits(it.M) * slice;        // the layout seer_gemm_tn_grouped_workspace_bytes sized
            } else {
                q.dst = it.C;
                q.cdst = it.colsum;
                q.slice = 0;
            }
        }
        hipLaunchKernelGGL(seer_gemm_tn_grouped_kernel, dim3(wg), dim3(256), 0, st, g);
        SEER_LAUNCH_CHECK();
        if (rg.n) {
            hipLaunchKernelGGL(tn_reduce_grouped_kernel, dim3(blk), dim3(256), 0, st, rg);
            SEER_LAUNCH_CHECK();
        }
    }
    return SEER_OK;
}
