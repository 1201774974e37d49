/*
 * seer_hip.h -- C ABI of libseer_hip.so: the MI355X (gfx950) device kernels behind the
 * Seer DDIM denoising hot path (SeerUNet forward + CFG + DDIM update + VAE decode), the two steps that feed it
 * (FSTextTransformer, VAE encode) and the fine-tuning step of train.py (backward kernels, AdamW).
 *
 * Boundary rules (all entry points):
 *   - extern "C", plain pointers and sizes; every pointer is DEVICE memory unless it says "host".
 *   - explicit stream (a hipStream_t passed as void*); nothing here synchronises, allocates or
 *     keeps global state, so every call is hipGraph-capturable and thread-safe.
 *   - returns 0 on success, a negative SEER_E* code otherwise (never throws).
 *   - activations are token-major / channels-last bf16: [B*F, H*W, C]  (a "token row" = one latent pixel
 *     of one frame; C contiguous).  Weights are bf16 [N][K] with K contiguous
 *     (nn.Linear [out,in] as is; conv [Co,Ci,3,3] repacked to [Co][ky][kx][Ci]).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference repo).
 */
#ifndef SEER_HIP_H
#define SEER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEER_OK 0
#define SEER_EINVAL (-22)   /* bad shape / alignment / flag combination */
#define SEER_ENOSYS (-38)   /* shape class not built (e.g. head dim) */
#define SEER_ELAUNCH (-5)   /* hipLaunchKernel reported an error */

/* storage type of 16-bit activations for the *_dt entry points: bf16 everywhere on the UNet path, IEEE half on the VAE path (the
 * reference decodes in fp32: inference_img.py:118) */
#define SEER_DT_BF16 0
#define SEER_DT_F16 1

/* library / device info ------------------------------------------------------------------- */
int seer_abi_version(void);                 /* bumps when a struct below changes */
const char* seer_strerror(int code);        /* host string */
const char* seer_build_arch(void);          /* "gfx950" */

/* ---- GEMM / implicit-GEMM convolution --------------------------------------------------- */
/* Replaces: nn.Linear (seer/models/attention.py:484-489,742,783), InflatedConv3d 1x1
 * (attention.py:111,126; resnet.py:172) and InflatedConv3d 3x3 incl. stride-2 Downsample3D and the
 * nearest-2x Upsample3D that precedes a conv (resnet.py:8-16,39,52-57,82,144,153).
 *   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
 * A(m,k):  mode PLAIN  : A[m*lda + k] for k < K1, A2[m*lda2 + (k-K1)] for k >= K1 (channel concat of two
 *                        tensors: the skip concat of unet_3d_blocks.py:596,712 without materialising it)
 *          mode CONV3X3: m -> (img, oy, ox); k -> (ky, kx, ci);  X[img, oy*stride+ky-1, ox*stride+kx-1, ci]
 *                        (zero outside), read through a nearest 2x upsample when `upsample` == 1.
 *                        `upsample` == 2: the same Upsample3D conv (resnet.py:52-57) as its four 2x2 PHASE convs -- output
 *                        pixel (2y+a, 2x+b) only ever sees source rows y+a-1, y+a and columns x+b-1, x+b, so phase (a, b)
 *                        is a 4-tap conv over the SOURCE grid with taps summed per source pixel (16 tap-products per
 *                        source pixel instead of 36).  M = n_img*Hin*Win, K = 4*Cin with k -> (ty, tx, ci),
 *                        W = [4 phases (a*2+b)][N][K] (seervideoldm_amd.weights.pack_conv3x3_up_phases), Hout = 2 Hin,
 *                        Wout = 2 Win; the launch runs the four phases as its grid.z and scatters the rows into
 *                        C [n_img*Hout*Wout, ldc].  bias and col_scale only (no residual / rowvec / fp32 / transposed out).
 * epilogue: + bias[n] ; + rowvec[(m / rows_per_batch), n] (the time-embedding add of resnet.py:191-193);
 *           + residual[m*ldr + n] ; GEGLU (attention.py:791-793): W rows are interleaved in groups of 16
 *           (16 value rows, then their 16 gate rows), output is N/2 wide: val * gelu_erf(gate).
 * K (and K1) must be multiples of 64; N a multiple of 4 (32 for GEGLU); lda/ldc/ldr multiples of 8.
 */
#define SEER_GEMM_PLAIN 0
#define SEER_GEMM_CONV3X3 1
#define SEER_EPI_GEGLU 1u      /* fused a * gelu(g) */
#define SEER_EPI_OUT_F32 2u    /* C is fp32 instead of bf16 */
#define SEER_EPI_SILU 4u       /* C = silu(acc + bias) (time_embedding.linear_1) */
#define SEER_EPI_TRANS_OUT 8u  /* store C transposed: Ct[n*ldc + m] (used for V^T in the VAE attention) */
#define SEER_EPI_ROTARY 16u    /* rotate the q|k columns (n < rot_cols) of a fused q|k|v projection (attention.py:649-651) */
#define SEER_EPI_COLSCALE 32u  /* multiply the output columns n < col_scale_cols by col_scale (after bias / rotary): the q columns
                                * of a projection leave the GEMM as q * scale * log2(e), rounded to bf16 ONCE, for
                                * SEER_ATTN_Q_PRESCALED (attention.py:622-630 applies the scale inside the attention op) */

#define SEER_EPI_F16 64u       /* A, A2, W, residual and C hold IEEE half (fp16) instead of bf16; fp32 accumulation as always.  For the
                                * VAE, which the reference never autocasts (inference_img.py:118, ddim_sampling_utils.py:37-41):
                                * 11 significand bits at the bf16 MFMA rate.  Plain and conv launches, unsplit; not with GEGLU,
                                * rotary or colsum */

typedef struct seer_gemm_desc {
    const void* A;          /* bf16 */
    const void* A2;         /* bf16 or NULL */
    const void* W;          /* bf16 [N][K] */
    const float* bias;      /* fp32 [N] or NULL */
    const void* residual;   /* bf16 [M][ldr] or NULL */
    const float* rowvec;    /* fp32 [M/rows_per_batch][rowvec_ld] or NULL */
    void* C;                /* bf16 (or fp32) [M][ldc] */
    int32_t M, N, K, K1;
    int32_t lda, lda2, ldr, ldc;
    int32_t rows_per_batch, rowvec_ld;
    int32_t mode;           /* SEER_GEMM_* */
    uint32_t epilogue;      /* SEER_EPI_* flags */
    /* conv geometry (mode CONV3X3): input NHWC [n_img, Hin, Win, Cin] (pre-upsample size) */
    int32_t Hin, Win, Cin, Hout, Wout, stride, upsample;
    /* batched GEMM (grid.z): element strides; batch<=1 means a single problem */
    int32_t batch;
    int64_t strideA, strideW, strideC;
    /* tile selection: 0 = auto, else SEER_TILE_* */
    int32_t tile;
    /* split-K (small M, long K: the 4x4 / 8x8 level convs): 0 = auto, 1 = off, n = n slices of the K loop.  Slices write
     * fp32 partial tiles to `workspace` ([splits][M][N] floats) and a second kernel adds them IN SLICE ORDER (deterministic,
     * no atomics) and applies the epilogue.  Ignored (no split) when workspace is NULL or too small. */
    int32_t splits;
    void* workspace;
    int64_t workspace_bytes;
    /* SEER_EPI_ROTARY: cos/sin table of seer_rotary_table ([pos][rot_dim/2][2] fp32); position of row m is
     * m % rot_tokens_per_batch + rot_pos_offset; columns n < rot_cols are heads of rot_head_dim channels of which the first
     * rot_dim are rotated (interleaved pairs) */
    const float* rot_table;
    int32_t rot_tokens_per_batch, rot_pos_offset, rot_head_dim, rot_dim, rot_cols;
    /* CONV3X3 padding: 0 = one pixel on every side (every conv of the UNet and the VAE decoder); 1 = only after the last row
     * and column, X[img, oy*stride+ky, ox*stride+kx, ci] -- the VAE encoder's Downsample, F.pad(x, (0,1,0,1)) + conv(stride 2,
     * padding 0) (ldm/modules/diffusionmodules/model.py:60-78) */
    int32_t pad_after_only;
    /* SEER_EPI_COLSCALE */
    int32_t col_scale_cols;
    float col_scale;
    /* optional: per-tile column sums of C as stored (bf16-rounded), colsum[z][ceil(M / rows)][N][2] = (sum, sum of squares) over
     * the tile's rows, rows = seer_gemm_colsum_rows(desc), z = the phase of an upsample == 2 conv (else 0).  The GroupNorm that
     * consumes C (ResnetBlock3D.norm1/norm2, SpatialTransformer3D.norm: resnet.py / attention.py of the reference run
     * F.group_norm on it) takes its statistics from these sums instead of a pass over C: seer_groupnorm_stats_from_colsums.
     * A split-K launch leaves them through its reduce pass (rows = 4 or 16).  A launch that cannot produce them (GEGLU, fp32 /
     * transposed output, the weight-stationary kernel) fails with SEER_EINVAL when colsum is set: ask seer_gemm_colsum_rows
     * first. */
    float* colsum;
    /* in-launch split-K (the 256 x 320 tile kernel, SEER_TILE_T256x320): two 32-bit counters per output tile,
     * seer_gemm_sync_bytes(desc) bytes.  ZERO before the first launch that uses them; every launch leaves them zero again, so
     * one buffer serves all launches that are ordered on one stream (launches that may overlap in time need their own).  NULL:
     * the launch falls back to the two-launch split-K of the smaller tiles. */
    void* sync;
    int64_t sync_bytes;
    /* the same column sums, ACCUMULATED per batch element in 64-bit fixed point instead of written per tile:
     * colsum_fx[rep][b][2][N] (planes: sum, sum of squares) += round(partial * 2^SEER_GN_FX_SHIFT) by integer atomic adds, b = (first
     * row of the partial) / colsum_fx_rows, rep = (index of the partial) % colsum_fx_reps (replicas keep the adds per address
     * low: an atomic on one address retires every ~12 ns).  Integer addition commutes, so the totals do not depend on the
     * order the tiles finish in (bit-identical from run to run, like the per-tile form).  ZERO the buffer before the first launch
     * that adds to it; every launch whose rows belong to the same tensor (the four phases of an upsample == 2 conv) adds to the
     * same buffer.  colsum_fx_reps and the row granularity come from seer_gemm_colsum_fx_layout; set colsum OR colsum_fx.  The
     * consumer is seer_groupnorm_apply_fx: one launch, no statistics pass and no finalize pass.  Range: |sum|, sum of squares
     * < 2^43 per (batch element, column). */
    int64_t* colsum_fx;
    int32_t colsum_fx_rows, colsum_fx_reps;
    /* ---- LayerNorm folded into the GEMM that consumes it (BasicTransformerBlock: norm1 -> to_q|k|v, norm2 -> attn2.to_q,
     * norm3 -> ff.net.0, attention.py:198-200, 231-246, 275-277, 308-327).  With W' = gamma (.) W (scaled along K) and
     * x = the UN-normalised rows,  LN(x) W^T = rstd * (x W'^T - mean * wsum) + (beta W^T + b),  wsum[n] = sum_k W'[n][k]:
     * the LayerNorm launch, its read and its write of the activations disappear.
     * Producer side (the GEMM that writes x): rowstat[m][2] += round((sum, sum of squares) of the wave's columns of row m
     * * 2^SEER_LN_FX_SHIFT), 64-bit integer atomic adds like colsum_fx (one pair per row and wave column; integer adds commute:
     * bit-identical from run to run).  ZERO the buffer before the launch.  The sums are taken from the fp32 values the
     * epilogue is about to round to bf16.  Valid when seer_gemm_rowstat_ok(desc). */
    int64_t* rowstat;
    /* Consumer side: A = x [M][K] (K = the normalised width, A2 NULL), W = W', bias = beta W^T + b; ln_rowstat = the producer's
     * rowstat, ln_wsum fp32 [N] (sums of the bf16-ROUNDED W' rows, so that the mean cancels exactly), ln_eps the LayerNorm's
     * epsilon.  The term is applied to the accumulators before bias / GEGLU / rotary / column scale / residual.  Valid when
     * seer_gemm_lnfold_ok(desc) (unsplit tile-kernel launches with a staged bf16 output; a launch AUTO would give to the
     * weight-stationary kernel says no: LayerNorm + that kernel is the faster pair there). */
    const int64_t* ln_rowstat;
    const float* ln_wsum;
    float ln_eps;
} seer_gemm_desc;
#define SEER_GN_FX_SHIFT 20
#define SEER_LN_FX_SHIFT 24

#define SEER_TILE_AUTO 0
#define SEER_TILE_128x128 1
#define SEER_TILE_64x64 2
#define SEER_TILE_128x64 3
/* LDS-direct (global_load_lds) multi-stage variants: tile _ stages */
#define SEER_TILE_G128x128_2 5
#define SEER_TILE_G128x128_3 6
#define SEER_TILE_G128x64_3 7
#define SEER_TILE_G64x64_3 8
#define SEER_TILE_G64x64_4 9
#define SEER_TILE_G64x64_5 10
#define SEER_TILE_G128x64_4 11
/* 160-wide tiles: N = 320 / 640 (C of the two upper levels) in 2 / 4 column tiles instead of 5 / 10 */
#define SEER_TILE_G128x160_2 12
#define SEER_TILE_G64x160_3 13
/* 8 waves (4 x 2), 256-row tiles */
#define SEER_TILE_G256x128_2 14
#define SEER_TILE_G256x64_3 15
/* 96-row tiles: 24 576 rows (the 32x32 level at CFG batch 2) in 256 row tiles = a whole number of tiles per CU */
#define SEER_TILE_G96x160_2 16
#define SEER_TILE_G96x160_3 17
#define SEER_TILE_G96x128_2 18
/* weight-stationary persistent kernel (gemm_ws.hip): one column panel of W resident in LDS per CU, A streamed through a ring;
 * AUTO picks it for plain GEMMs with K <= 768 and M >= 1024, this value asks for it (falls back to AUTO when not eligible) */
#define SEER_TILE_G256x256_2 21 /* 8 waves, 64x128 wave tiles, 2 x 64 KB stages: the only tile whose FLOPs per LDS-fill byte (128) reach the MFMA roof */
#define SEER_TILE_WS 19
#define SEER_TILE_T256x320 22  /* 8 waves (4 x 2), 64 x 160 wave tiles, 142 FLOP per LDS-fill byte; N % 320 == 0; K slices reduced inside the launch (desc.sync) */
#define SEER_TILE_AUTO_TILED 20   /* AUTO restricted to the tile kernel (A/B runs against the weight-stationary kernel) */

int seer_gemm_bf16(const seer_gemm_desc* desc /* host */, void* stream);
/* bytes of workspace the call would use for split-K with this descriptor (0: it will not split) */
int64_t seer_gemm_workspace_bytes(const seer_gemm_desc* desc /* host */);
/* rows per partial of the column sums this exact launch (same tile / splits / workspace fields) would write to desc->colsum,
 * or 0 when it cannot produce them; the buffer is [z][ceil(M / rows)][N][2] floats */
int32_t seer_gemm_colsum_rows(const seer_gemm_desc* desc /* host */);
/* colsum_fx form of this exact launch for tensors of rows_per_batch rows per batch element: returns the rows one partial covers
 * (rows_per_batch must be a multiple; 0: the launch cannot accumulate column sums) and sets *reps = the replica count to
 * allocate and pass as colsum_fx_reps */
int32_t seer_gemm_colsum_fx_layout(const seer_gemm_desc* desc /* host */, int32_t rows_per_batch, int32_t* reps);
/* 1 when this exact launch can accumulate the row statistics of its output (desc->rowstat), else 0 */
int32_t seer_gemm_rowstat_ok(const seer_gemm_desc* desc /* host */);
/* 1 when this exact launch (ln_rowstat / ln_wsum set) applies the folded LayerNorm, 0 when the caller has to run
 * seer_layernorm and the unfolded weights instead */
int32_t seer_gemm_lnfold_ok(const seer_gemm_desc* desc /* host */);
/* bytes of zeroed counter memory (desc->sync) the call would use to reduce its K slices inside the launch (0: none) */
int64_t seer_gemm_sync_bytes(const seer_gemm_desc* desc /* host */);

/* ---- attention -------------------------------------------------------------------------- */
/* Replaces xformers.ops.memory_efficient_attention as called from CrossAttention
 * (attention.py:622-630: spatial self / text cross, attn_bias None) and from WindowSTempAttention
 * (attention.py:632-703: LowerTriangularMask over window tokens ordered (f, wy, wx)).
 *   O[b, sq, h, :] = softmax_j( scale * <Q[b,sq,h,:], K[b,j,h,:]> (+ causal mask j<=i) ) V[b,j,h,:]
 * Q/K/V/O are addressed as ptr + b*bs + s*ss + h*d (+ element), so they can be column slices of a fused
 * [tokens, 3C] projection output.  window_ws > 0 selects the temporal window form: the batch index runs
 * over (window, b) and sequence position p = (f, wy, wx) maps to token f*H*W + (win_y*ws+wy)*W + win_x*ws+wx
 * (window_partition/window_reverse of attention.py:42-69 folded into the addressing).
 * head_dim in {40, 80, 160} (SeerUNet levels) or 96 (FSTextTransformer, 768 / 8; its attention over frames uses
 * ss = tokens-per-frame rows and bs = one row); bf16 in/out, fp32 softmax statistics and accumulation.
 */
typedef struct seer_attn_desc {
    const void* Q; const void* K; const void* V; void* O;   /* bf16 */
    int64_t q_bs, k_bs, v_bs, o_bs;     /* batch strides (elements) */
    int32_t q_ss, k_ss, v_ss, o_ss;     /* sequence(token) strides (elements) */
    int32_t batch, heads, head_dim;
    int32_t Sq, Sk;
    int32_t causal;
    float scale;
    /* temporal windows (0 = off): K/V hold F frames per batch element, Q/O hold Fq (== F unless frame-sharded) */
    int32_t window_ws, F, H, W;
    int32_t Fq;
    /* causal: key j visible to query i iff j <= i + causal_offset (0 unless the queries are a frame shard whose first
     * query sits at sequence position causal_offset of the key sequence) */
    int32_t causal_offset;
    /* optional (training): fp32 [batch' * heads][Sq] with batch' = batch (x windows); receives log2(sum_j 2^(scale*log2(e)*s_ij))
     * per query, the statistic seer_attn_bwd needs to rebuild the probabilities.  NULL at inference. */
    float* lse;
    /* SEER_ATTN_* flags */
    uint32_t flags;
    /* kernel selection, a descriptor field so that A/B runs need no global state: 0 = auto; 1 = generic kernel, one K|V LDS
     * buffer; 6 = generic kernel, ping-pong buffers; head_dim 40 only: 3 = the d = 40 kernel (fast path with the in-launch
     * fallback; 32 queries per wave), 2 = the same with 64 queries per wave (what 0 picks from four rounds of resident workgroups
     * up), 5 = the kernel running its tracked-reference form directly (what lse != NULL selects), 7 = the 64-query form on a
     * three-stage K|V ring in LDS (Sq % 256 == 0, Sk % 128 == 0, not causal, not windowed: what 0 picks for such launches) */
    int32_t variant;
    /* head strides (elements): head h of Q / K / V starts h * q_hs / k_hs / v_hs elements after the batch element's base.
     * 0 = head_dim: the heads are adjacent column groups of token-major rows (the layout of a fused [tokens, 3C] projection).
     * A HEAD-MAJOR operand -- [batch][head][token][head_dim], token stride = head_dim, head stride = tokens * head_dim, what
     * seer_gemm_bf16 writes with SEER_EPI_HEADMAJOR -- makes every K / V tile of a head one dense run of memory: at head_dim 40
     * the LDS-DMA of a 128-key tile then moves 80 whole 128-byte lines instead of ~200 partial ones (80-byte pieces at a
     * 1920-byte pitch), which is what bounds the d = 40 kernel on token-major operands.  O is always token-major. */
    int64_t q_hs, k_hs, v_hs;
} seer_attn_desc;

/* Q already holds q * scale * log2(e) (the producing GEMM's epilogue multiplied it in, SEER_EPI_COLSCALE): the kernel takes
 * exp2 of the raw dot products and ignores `scale`.  Without the flag the d = 40 kernel multiplies Q itself (one more
 * bf16 rounding of q) and the generic kernel scales the fp32 scores. */
#define SEER_ATTN_Q_PRESCALED 1u
/* Q, K, V and O hold IEEE half (fp16) instead of bf16 -- the UNet engine under fp16 autocast (every reference yaml ships
 * mixed_precision: "fp16").  fp32 scores, statistics and accumulation as always; P is rounded to fp16 for the PV product.  head_dim 40
 * from 256 keys up runs the d = 40 kernel's TRACKED form (its fast path needs bf16's exponent range), everything else the generic
 * kernel (variant 0, 1 or 5; no lse: inference only). */
#define SEER_ATTN_F16 2u

int seer_attn_fwd(const seer_attn_desc* desc /* host */, void* stream);

/* Backward of the call above (the training step, train.py:380-381 through attention.py:622-630 / 632-703):
 * given dO, writes dQ, dK, dV (bf16, addressed like Q/K/V with their own strides, so they can be the column slices of one
 * [tokens, 3C] gradient buffer that feeds a single dX GEMM).  fwd must be the descriptor of the forward call with O and
 * lse filled in by it; delta is a [batch' * heads][Sq] fp32 scratch (<dO, O> per query).  Frame-sharded queries (Fq != F)
 * are not supported.  Deterministic: two launches (dQ; dK|dV), no atomics. */
typedef struct seer_attn_bwd_desc {
    seer_attn_desc fwd;
    const void* dO; void* dQ; void* dK; void* dV;     /* bf16 */
    int64_t do_bs, dq_bs, dk_bs, dv_bs;
    int32_t do_ss, dq_ss, dk_ss, dv_ss;
    float* delta;
} seer_attn_bwd_desc;
int seer_attn_bwd(const seer_attn_bwd_desc* desc /* host */, void* stream);

/* Rotary embedding on q and k in place (rotary-embedding-torch 0.1.5 rotate_queries_or_keys as called at
 * attention.py:649-651): first rot_dim channels of every head, interleaved pairs (x0,x1) -> (x0 c - x1 s, x1 c + x0 s),
 * angle = fp32(pos) * freqs[j] with freqs the module's `rotary_emb.freqs` buffer (10000^(-2j/rot_dim)),
 * pos = token index inside its batch element (f*H*W + y*W + x) + pos_offset (pos_offset = first frame * H*W under
 * frame sharding).  seer_rotary_table fills cos_sin fp32 [T][half][2] once per (level, F); seer_rotary_inplace
 * rotates x: bf16 [rows, ld] where head h of q occupies columns [col0_q + h*head_dim, ...) and of k [col0_k + ...). */
int seer_rotary_table(const float* freqs, int32_t T, int32_t half, float* cos_sin, void* stream);
int seer_rotary_inplace(void* x, int64_t rows, int32_t ld, int32_t col0_q, int32_t col0_k, int32_t heads,
                        int32_t head_dim, int32_t rot_dim, int32_t tokens_per_batch, int32_t pos_offset,
                        const float* cos_sin, void* stream);

/* ---- normalisation ---------------------------------------------------------------------- */
/* GroupNorm over (C/G, F, H, W) per (b, g) on channels-last data -- torch.nn.GroupNorm applied to the 5-D
 * tensor (resnet.py:179,197; attention.py:133; unet_3d_condition.py:368).  Two sources = channel concat.
 * stats: writes (sum, sumsq) per (b, g) to stats[b][g][2] (fp32).  Deterministic: per-block partials go to `workspace`
 * (seer_groupnorm_workspace_floats(...) floats) and are added in block order -- no float atomics, so two runs, or two
 * identical batch elements, give bit-identical statistics.  Under frame sharding the caller all-reduces `stats`
 * between the two calls.
 * apply: y = (x-mean)*rstd*gamma+beta, optional SiLU, bf16 out [rows, C1+C2]; count = elements per group over
 * ALL shards (so mean = sum/count). */
int64_t seer_groupnorm_workspace_floats(int32_t C, int32_t batch, int64_t rows_per_batch, int32_t groups);
int seer_groupnorm_stats(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                         int64_t rows_per_batch, int32_t groups, float* stats, float* workspace, void* stream);
/* The same stats[b][g][2] from the column sums the producers of x1 / x2 left behind (seer_gemm_desc::colsum) -- no pass over
 * the activations.  Source i: C_i channels, partial rows [phases_i][tiles_i][C_i][2]; the tiles of a phase cover the batch
 * elements in order, tiles_i / batch each (tiles_i % batch == 0).  One wave per (b, g) adds its channels' partials in a fixed
 * order: deterministic; equal to seer_groupnorm_stats up to the fp32 order of additions. */
int seer_groupnorm_stats_from_colsums(const float* cs1, int32_t C1, int32_t phases1, int32_t tiles1, const float* cs2,
                                      int32_t C2, int32_t phases2, int32_t tiles2, int32_t batch, int32_t groups,
                                      float* stats, void* stream);
int seer_groupnorm_apply(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                         int64_t rows_per_batch, int32_t groups, const float* stats, double count, float eps,
                         const float* gamma, const float* beta, int32_t silu, void* y, void* stream);
/* seer_groupnorm_stats_from_colsums + seer_groupnorm_apply in ONE launch: every block re-derives the statistics of the groups it
 * normalises from the producers' column sums (same arguments as the two calls; bf16 activations).  Single-process runs only: a
 * frame-sharded run has to all-reduce the statistics between the two steps and keeps the two calls.  SEER_ENOSYS when the
 * channel layout does not slice into whole groups of 64..128 channels (the caller then makes the two calls). */
int seer_groupnorm_apply_from_colsums(const void* x1, int32_t C1, const void* x2, int32_t C2, const float* cs1, int32_t phases1,
                                      int32_t tiles1, const float* cs2, int32_t phases2, int32_t tiles2, int32_t batch,
                                      int64_t rows_per_batch, int32_t groups, double count, float eps, const float* gamma,
                                      const float* beta, int32_t silu, void* y, void* stream);
/* GroupNorm apply whose statistics are the fixed-point column sums the producers ACCUMULATED (seer_gemm_desc::colsum_fx):
 * fx1 [reps1][batch][2][C1], fx2 [reps2][batch][2][C2] int64 (NULL with C2 = 0).  One launch per GroupNorm; every block converts the sums of
 * the groups it normalises (double precision) and streams its rows.  stats_out (or NULL): receives (sum, sum of squares) per
 * (batch element, group) as fp32 [batch][groups][2], what seer_groupnorm_bwd takes.  SEER_ENOSYS as above. */
int seer_groupnorm_apply_fx(const void* x1, int32_t C1, const void* x2, int32_t C2, const int64_t* fx1, int32_t reps1,
                            const int64_t* fx2, int32_t reps2, int32_t batch, int64_t rows_per_batch, int32_t groups, double count, float eps, const float* gamma,
                            const float* beta, int32_t silu, void* y, float* stats_out, void* stream);
/* The same accumulated sums from the ACTIVATIONS: fx [batch][2][C] int64 (one replica; ADDED to: zero it first) receives
 * sum_r round(x[r][c] * 2^20) and sum_r round(x[r][c]^2 * 2^20) over the rows of each batch element.  Every element is rounded on
 * its own, so the totals are exact integer sums: independent of the order of the additions and -- frame shards -- of which rank
 * held which rows (an int64 all-reduce of the shards' sums IS the unsharded result).  The statistics pass of a frame-sharded step
 * for GroupNorm sources without accumulated producer sums (resnet.py:179,197, attention.py:133 normalise over all frames). */
int seer_groupnorm_stats_fx(const void* x, int32_t C, int32_t batch, int64_t rows_per_batch, int64_t* fx, int32_t dtype,
                            void* stream);
/* seer_groupnorm_apply_from_colsums / seer_groupnorm_apply_fx with the storage type of x1 / x2 / y chosen by `dtype` (SEER_DT_*):
 * the UNet engine on fp16 storage */
int seer_groupnorm_apply_from_colsums_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, const float* cs1, int32_t phases1,
                                         int32_t tiles1, const float* cs2, int32_t phases2, int32_t tiles2, int32_t batch,
                                         int64_t rows_per_batch, int32_t groups, double count, float eps, const float* gamma,
                                         const float* beta, int32_t silu, void* y, int32_t dtype, void* stream);
int seer_groupnorm_apply_fx_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, const int64_t* fx1, int32_t reps1,
                               const int64_t* fx2, int32_t reps2, int32_t batch, int64_t rows_per_batch, int32_t groups, double count,
                               float eps, const float* gamma, const float* beta, int32_t silu, void* y, float* stats_out, int32_t dtype,
                               void* stream);
/* The feed-forward of a transformer block at the 320-channel level and the transformer's proj_out, ONE launch (csrc/ff_fused.hip):
 *     y = x + [Wp | Wp W2] [h | g] + bcat,   g = GEGLU(LayerNorm(h; gamma, beta, eps) W1^T + b1)
 * i.e. norm3 -> ff.net.0 -> ff.net.2 + residual -> proj_out + residual (seer/models/attention.py:231-248, 308-327, 742-747,
 * 783-793, 126, 141-145).  h, x, y: [M][320] bf16 with row strides ldh, ldx, ldy (elements, multiples of 8; y may alias x);
 * a workgroup owns 96 rows (the last one fewer when M is not a multiple).  w1f, wcf: the two weight matrices in the kernel's FRAGMENT order, made once by the pack entry points below from
 * w1 [2560][320] bf16 (interleaved GEGLU row order: 16 value rows, 16 gate rows, ...) and wcat [320][1600] bf16 = [Wp | Wp W2];
 * b1 [2560] fp32 in the same interleaved order; bcat [320] fp32 = Wp b2 + bp.  colsum_fx (or NULL): [fx_reps][M / fx_rows][2][320]
 * int64, ADDED to, the fixed-point column sums of y as seer_gemm_desc::colsum_fx (fx_rows = rows per batch element, a multiple of
 * 16 and at least 96: a tile may straddle two batch elements).  colsum_tiles (or NULL; M a multiple of 96): [M / 96][320][2] fp32,
 * WRITTEN, (sum, sum of squares) of the stored values per 96-row tile as seer_gemm_desc::colsum.  All pointers 16-byte aligned.  SEER_EINVAL otherwise. */
int seer_ff_fused_c320(const void* h, int32_t ldh, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t M,
                       const float* gamma, const float* beta, float eps, const void* w1f, const float* b1, const void* wcf,
                       const float* bcat, int64_t* colsum_fx, int64_t fx_rows, int32_t fx_reps, float* colsum_tiles, void* stream);
/* ... with h, x, y and the two packed weight matrices in the storage type `dtype` (SEER_DT_*; the pack entry points move 16-bit
 * words and serve both) */
int seer_ff_fused_c320_dt(const void* h, int32_t ldh, const void* x, int32_t ldx, void* y, int32_t ldy, int64_t M,
                          const float* gamma, const float* beta, float eps, const void* w1f, const float* b1, const void* wcf,
                          const float* bcat, int64_t* colsum_fx, int64_t fx_rows, int32_t fx_reps, float* colsum_tiles, int32_t dtype,
                          void* stream);
/* The row-local chains in front of the attention launches of a transformer block at the 320-channel level, ONE launch (csrc/rowchain.hip):
 *     h   = [GroupNorm(inp)] W1^T + b1 [+ res]                         (stored when h != NULL)
 *     out = LayerNorm(h; ln_gamma, ln_beta, ln_eps) [W2_0 | ... ]^T     n2 = 1..3 thirds of 320 columns (skipped when w2f == NULL)
 * GroupNorm -> proj_in -> norm1 -> to_q | to_k | to_v (seer/models/attention.py:129-145, 231-240, 308-318; the temporal block's rotary
 * embedding :649-651 as rot_*), or attn1.to_out + residual -> norm2 -> attn2.to_q (:316-322).  inp, res, h [M][320], out [M][n2 * 320]
 * in the storage type `dtype` (SEER_DT_*), row strides multiples of 8 elements, h may alias res.  GroupNorm: gn_stats [batch][groups][2]
 * fp32 (sum, sum of squares per (batch element, group): what seer_groupnorm_stats* write), gn_count elements per group, rows_per_batch
 * >= 96 (SEER_ENOSYS otherwise: a workgroup's 96 rows span at most two batch elements); NULL = no normalisation of
 * the input; or gn_fx [gn_fx_reps][batch][2][320] int64, the fixed-point column sums the producer of `inp` ACCUMULATED
 * (seer_gemm_desc::colsum_fx): no statistics launch in front.  ln_gamma / ln_beta NULL = no LayerNorm.  w1f / w2f: the matrices in FRAGMENT order (seer_rowchain_pack).  The first
 * rot_thirds thirds are rotated like SEER_EPI_ROTARY (table of seer_rotary_table, position = row % rot_tokens_per_batch +
 * rot_pos_offset, heads of rot_head_dim channels, the first rot_dim rotated), then the first scale_thirds thirds are multiplied by
 * col_scale (the q columns leave as q * scale * log2(e) for SEER_ATTN_Q_PRESCALED).  All pointers 16-byte aligned. */
typedef struct seer_rowchain_desc {
    const void* inp; int32_t ld_in;
    const float* gn_stats; const int64_t* gn_fx; int32_t gn_fx_reps;
    double gn_count; float gn_eps; const float* gn_gamma; const float* gn_beta; int64_t rows_per_batch; int32_t groups;
    const void* w1f; const float* b1; const void* res; int32_t ldr; void* h; int32_t ldh;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const void* w2f; int32_t n2; void* out; int32_t ldo;
    float col_scale; int32_t scale_thirds;
    const float* rot_table; int32_t rot_tokens_per_batch, rot_pos_offset, rot_head_dim, rot_dim, rot_thirds;
    int64_t M;
    int32_t dtype;
} seer_rowchain_desc;
int seer_rowchain_c320(const seer_rowchain_desc* desc /* host */, void* stream);
/* n_mats 320 x 320 matrices -- rows 320 t .. 320 t + 319 of W [n_mats * 320][ld], 16-bit elements -- into the kernel's fragment order:
 * out[t][K step 5][wave 4][k32 2][column fragment 5][lane 64][8], element W[320 t + 80 w + 16 j + (lane & 15)][64 s + 32 k32 + 8 (lane >> 4) + e] */
int seer_rowchain_pack(const void* W, int32_t ld, int32_t n_mats, void* out, void* stream);
/* ... with a prologue: the rows the launch reads as h are  h + a Wo^T + bo  -- the attention's to_out projection and its residual
 * (seer/models/attention.py:237-240, 316-322), computed in the tile and stored nowhere (this launch is their only reader).  a [M][320]
 * (row stride lda), wof = Wo [320][320] in the fragment order of seer_rowchain_pack, bo [320] fp32; a == NULL: no prologue. */
int seer_ff_fused_c320_pre(const void* a, int32_t lda, const void* wof, const float* bo, const void* h, int32_t ldh, const void* x,
                           int32_t ldx, void* y, int32_t ldy, int64_t M, const float* gamma, const float* beta, float eps,
                           const void* w1f, const float* b1, const void* wcf, const float* bcat, int64_t* colsum_fx, int64_t fx_rows,
                           int32_t fx_reps, float* colsum_tiles, int32_t dtype, void* stream);
/* w1 [2560][320] bf16 -> out (same size): [chunk 20][wave 4][K step 5][k32 2][value | gate][lane 64][8 bf16], element
 * w1[128 c + 32 w + 16 f + (lane & 15)][64 ks + 32 k32 + 8 (lane >> 4) + e]: every fragment load of the kernel is one contiguous KiB */
int seer_ff_fused_pack_w1(const void* w1, void* out, void* stream);
/* wcat [320][1600] bf16 -> out (same size): [K step 25][wave 4][k32 2][column fragment 5][lane 64][8 bf16], element
 * wcat[80 w + 16 j + (lane & 15)][64 s + 32 k32 + 8 (lane >> 4) + e] */
int seer_ff_fused_pack_wcat(const void* wcat, void* out, void* stream);
/* the same two with the storage type of x1 / x2 / y chosen by `dtype` (SEER_DT_*): the VAE's nn.GroupNorm(32, eps 1e-6)
 * (ldm/modules/diffusionmodules/model.py:38-40) on fp16 activations */
int seer_groupnorm_stats_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                            int64_t rows_per_batch, int32_t groups, float* stats, float* workspace, int32_t dtype, void* stream);
int seer_groupnorm_apply_dt(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch,
                            int64_t rows_per_batch, int32_t groups, const float* stats, double count, float eps,
                            const float* gamma, const float* beta, int32_t silu, void* y, int32_t dtype, void* stream);

/* nn.LayerNorm(C) per token row (attention.py:198-200,275-277), eps 1e-5; bf16 in/out, fp32 statistics. */
int seer_layernorm(const void* x, int64_t rows, int32_t C, int32_t ldx, const float* gamma, const float* beta,
                   float eps, void* y, int32_t ldy, void* stream);
/* ... with x and y in the storage type `dtype` (SEER_DT_*) */
int seer_layernorm_dt(const void* x, int64_t rows, int32_t C, int32_t ldx, const float* gamma, const float* beta,
                      float eps, void* y, int32_t ldy, int32_t dtype, void* stream);

/* y = softmax(scale * x) over rows of a [rows, n] matrix, x bf16 or fp32, y bf16 (VAE mid attention,
 * ldm/modules/diffusionmodules/model.py:186-197) */
int seer_softmax_rows(const void* x, int32_t x_is_f32, int64_t rows, int32_t n, int32_t ld, float scale, void* y,
                      int32_t ldy, void* stream);
/* ... with 16-bit x and y in the storage type `dtype` (SEER_DT_*) */
int seer_softmax_rows_dt(const void* x, int32_t x_is_f32, int64_t rows, int32_t n, int32_t ld, float scale, void* y,
                         int32_t ldy, int32_t dtype, void* stream);

/* ---- small / boundary kernels ----------------------------------------------------------- */
/* diffusers Timesteps(320, flip_sin_to_cos, freq_shift) (unet_3d_condition.py:97,307): out[b] = [cos | sin](t*f_i)
 * (or [sin|cos] when flip==0), fp32 [B, dim]. t: int64 [B]. */
int seer_timestep_embedding(const int64_t* t, int32_t B, int32_t dim, int32_t flip_sin_to_cos, float freq_shift,
                            float* out, void* stream);

/* small-M linear: y[b, n] = act( sum_k f(x[b,k]) * W[n,k] + bias[n] ), f = SiLU if silu_in.  x fp32 [B,K], W bf16 [N][K],
 * y fp32.  Used for time_embedding (unet_3d_condition.py:308) and all 22 time_emb_proj at once (resnet.py:192). */
int seer_linear_smallm(const float* x, int32_t B, int32_t K, const void* W, const float* bias, int32_t N,
                       int32_t silu_in, int32_t silu_out, float* y, void* stream);
/* ... with W in the storage type `dtype` (SEER_DT_*) */
int seer_linear_smallm_dt(const float* x, int32_t B, int32_t K, const void* W, const float* bias, int32_t N,
                          int32_t silu_in, int32_t silu_out, float* y, int32_t dtype, void* stream);

/* conv_in: InflatedConv3d(4->C0, 3x3, pad 1) reading the reference layout [B, Cin, F, H, W] fp32 and writing
 * channels-last bf16 [B*F, H*W, C0] (unet_3d_condition.py:94,311).  W fp32 repacked to [3][3][Cin][C0]. */
int seer_conv_in(const float* x, int32_t B, int32_t Cin, int32_t F, int32_t H, int32_t W_, const float* Wt,
                 const float* bias, int32_t Cout, void* y, void* stream);
/* conv_out: InflatedConv3d(C0->Cout(4), 3x3, pad 1) reading channels-last bf16 and writing [B, Cout, F, H, W] fp32
 * (unet_3d_condition.py:205,370).  W fp32 [Cout][3][3][C0]. */
int seer_conv_out(const void* x, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                  const float* bias, int32_t Cout, float* y, void* stream);
/* the VAE's conv_in / conv_out (ldm/modules/diffusionmodules/model.py:487-491,556-568) with the channels-last side stored
 * as `dtype` (SEER_DT_*); conv_out: Cout = 3 for SEER_DT_F16 */
int seer_conv_in_dt(const float* x, int32_t B, int32_t Cin, int32_t F, int32_t H, int32_t W_, const float* Wt,
                    const float* bias, int32_t Cout, void* y, int32_t dtype, void* stream);
int seer_conv_out_dt(const void* x, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                     const float* bias, int32_t Cout, float* y, int32_t dtype, void* stream);

/* pointwise channel mix on NCHW fp32: the VAE's post_quant_conv (1x1, 4->4; ldm/models/autoencoder.py:330-333).
 * W fp32 [Cout][Cin]. */
int seer_conv1x1_nchw_f32(const float* x, int32_t N, int32_t Cin, int32_t Cout, int32_t HW, const float* Wt,
                          const float* bias, float* y, void* stream);

/* layout / dtype conversion: fp32 [rows, C] -> bf16 (context, weights) */
int seer_cast_f32_bf16(const float* x, int64_t n, void* y, void* stream);
/* ... to the 16-bit storage type `dtype` (SEER_DT_*) */
int seer_cast_f32_dt(const float* x, int64_t n, void* y, int32_t dtype, void* stream);
/* NHWC bf16 -> NCHW fp32 and back (VAE boundary) */
int seer_nchw_f32_to_nhwc_bf16(const float* x, int32_t N, int32_t C, int32_t HW, void* y, void* stream);
int seer_nhwc_bf16_to_nchw_f32(const void* x, int32_t N, int32_t C, int32_t HW, float* y, void* stream);

/* ---- sampler step ------------------------------------------------------------------------ */
/* CFG combine + DDIM update of DDIMSampler.p_sample_ddim (ldm/models/diffusion/ddim_video.py:209-238), fp32:
 *   e   = e_uc + scale*(e_c - e_uc)            (only frames >= cond_f of the UNet output are used)
 *   x0  = (x - sqrt(1-a_t) e)/sqrt(a_t)
 *   x'  = sqrt(a_prev) x0 + sqrt(1-a_prev-sigma^2) e + sigma*noise
 * eps: UNet output [2b (uc then c), C, F_total, HW] fp32 (or [b,...] with scale==1 / e_uc NULL semantics:
 * pass cfg=0); x, x_prev, pred_x0: [b, C, F_pred, HW].  coef: device fp32 [steps][4] = (a_t, a_prev, sigma, sqrt(1-a_t)),
 * row `index` is used (no host->device scalar copies per step, unlike ddim_video.py:219-222). */
int seer_cfg_ddim_step(const float* eps, int32_t cfg, int32_t b, int32_t C, int32_t F_total, int32_t cond_f,
                       int32_t HW, float scale, const float* coef, int32_t index, const float* x,
                       const float* noise /* may be NULL when sigma==0 */, float* x_prev, float* pred_x0, void* stream);

/* The same step with its schedule index in device memory, for a p_sample_ddim that is captured WHOLE in one hipGraph (kernel
 * arguments are frozen at capture, the step counter must not be).  step: int32[2]; step[0] = index of the next step to run,
 * step[1] = index of the step in flight.
 *   seer_ddim_step_begin    (ddim_video.py:189,201-203) first kernel of a step: sample[reps*b, C, f1 + F_pred, HW] =
 *                           cat([x0_emb, x], frames) repeated `reps` times (2 = the [uc, c] CFG pair), t_out[reps*b] =
 *                           t_table[step[0]], and step[1] = step[0];
 *   seer_cfg_ddim_step_dev  last kernel: seer_cfg_ddim_step with index = step[1]; then step[0] = index - 1.  x_prev may be x.
 * No kernel reads and writes the same word, so replays of the captured step walk the schedule downwards on their own; the host
 * writes step[0] only to start a chain. */
int seer_ddim_step_begin(const float* x0_emb /* NULL iff f1 == 0 */, const float* x, int32_t b, int32_t reps, int32_t C,
                         int32_t f1, int32_t F_pred, int32_t HW, const int64_t* t_table, int32_t* step, float* sample,
                         int64_t* t_out, void* stream);
int seer_cfg_ddim_step_dev(const float* eps, int32_t cfg, int32_t b, int32_t C, int32_t F_total, int32_t cond_f, int32_t HW,
                           float scale, const float* coef, int32_t* step, const float* x, const float* noise, float* x_prev,
                           float* pred_x0, void* stream);

/* decoded image post-process of ddim_sample (utils/ddim_sampling_utils.py:41): clamp((x+1)/2, 0, 1) in place */
int seer_clamp01(float* x, int64_t n, void* stream);

/* VAE encode, the step before the path: DiagonalGaussianDistribution.sample
 * (ldm/modules/distributions/distributions.py:24-37; diffusers AutoencoderKL.encode(x).latent_dist.sample(),
 * inference_img.py:168): moments fp32 [N, 2C, HW] = (mean | logvar) along channels ->
 *   out[n, c, i] = mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[n, c, i]        (noise NULL: the mode) */
int seer_gaussian_sample(const float* moments, int32_t N, int32_t C, int32_t HW, const float* noise, float* out,
                         void* stream);

/* ---- training step (train.py:319-389) --------------------------------------------------------------
 * The backward pass reuses seer_gemm_bf16 for the input gradients and has one more MFMA kernel for the weight gradients:
 *   dX[M,K] = dY[M,N] W[N,K]       -> A = dY, W' = W^T ([K][N], a transposed copy of the weight: seer_transpose_bf16)
 *   dW[N,K] = dY^T[N,M] X[M,K]     -> seer_gemm_tn_f32 (both operands read as they are, fragments by transposed LDS reads)
 *   conv3x3 dX                     -> the CONV3X3 mode with the weight repacked as w'[ci][2-ky][2-kx][co]; a stride-2 conv
 *                                     first spreads dY with seer_zero_insert2x_bf16, a conv behind the nearest-2x upsample
 *                                     folds its dX with seer_sumpool2x_bf16
 * and seer_attn_bwd for attention.  The entry points below are the HBM-bound remainder.  Every reduction is two-stage through
 * a caller workspace (no float atomics). */

/* weight gradient without transposes: C[n*K + k] = sum_m A[m*lda + n] * B[m*ldb + k]   (dW[N,K] = dY[M,N]^T X[M,K]), bf16
 * operands in their token-major layout, fp32 result; colsum (optional) receives sum_m A[m][n] -- the bias gradient -- from
 * one more MFMA against a fragment of ones.  N, K, lda, ldb multiples of 8.  The contraction is split across blocks; slices
 * meet in `workspace` (seer_gemm_tn_workspace_bytes(M, N, K) bytes, 0 = not needed) and are added in slice order. */
int64_t seer_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int seer_gemm_tn_f32(const void* A, int32_t lda, const void* B, int32_t ldb, int32_t M, int32_t N, int32_t K, float* C,
                     float* colsum, void* workspace, int64_t workspace_bytes, void* stream);

/* The weight gradients of MANY layers in one launch (ABI 24).  In the reference every trainable nn.Linear gets its `.grad` from
 * autograd's backward (train.py:382 `accelerator.backward(loss)`), one product per layer wherever the walk reaches it; none of them
 * depends on another and nothing reads them before the optimizer (train.py:383-386), so a backward pass may leave them to its end:
 * `items` is a HOST array (the launch carries its problem table in the kernel arguments, 48 problems per launch: a captured launch keeps
 * it), item i as seer_gemm_tn_f32's arguments.  A problem is split over ceil(M / 16384) workgroups per tile (SEER_TN_GROUP_ROWS: the group fills the chip by its number of problems);
 * slices meet in `workspace` (seer_gemm_tn_grouped_workspace_bytes) and one more launch adds them in slice order.  Results are
 * independent of how problems are grouped: a problem's bits depend on its own (M, N, K) only. */
typedef struct seer_tn_item {
    const void* A;          /* dY [M][N] bf16, row pitch lda */
    const void* B;          /* X  [M][K] bf16, row pitch ldb */
    float* C;               /* dW [N][K] fp32 */
    float* colsum;          /* [N] fp32 or NULL */
    int32_t lda, ldb, M, N, K;
    int32_t reserved;
} seer_tn_item;
int64_t seer_gemm_tn_grouped_workspace_bytes(const seer_tn_item* items /* host */, int32_t n_items);
int seer_gemm_tn_grouped_f32(const seer_tn_item* items /* host */, int32_t n_items, void* workspace, int64_t workspace_bytes,
                             void* stream);

/* y[c*ldy + r] = x[r*ldx + c]; columns rows..ldy-1 of y are zero filled */
int seer_transpose_bf16(const void* x, int64_t rows, int32_t cols, int32_t ldx, void* y, int64_t ldy, void* stream);

/* The same for many matrices in one launch (the W^T refresh of all trainable matrices after an optimizer step).  `items` is a
 * DEVICE array, read by the kernel: matrix i is x [rows, cols] with row pitch ldx -> y [cols, ldy], 64x64 tiles numbered
 * tile0 + (tile row) + tiles_r * (tile column), tiles_r = ldy / 64, tile0 = the running sum of tiles_r * ceil(cols / 64)
 * (ascending, items[0].tile0 = 0); total_tiles = that sum over all items.  Every item: cols % 8 == 0, ldx % 8 == 0,
 * ldy % 64 == 0, ldy >= rows, x and y 16-byte aligned (the caller checks: the table is not readable from the host side). */
typedef struct seer_transpose_item {
    const void* x;
    void* y;
    int64_t rows;
    int64_t ldy;
    int64_t tile0;
    int32_t cols;
    int32_t ldx;
    int32_t tiles_r;
    int32_t reserved;
} seer_transpose_item;
int seer_transpose_batched_bf16(const seer_transpose_item* items, int32_t n_items, int64_t total_tiles, void* stream);

/* out[c] = sum_r x[r][c] (bias gradients).  workspace: seer_colsum_workspace_floats(rows, cols) floats (the same size
 * serves seer_layernorm_bwd). */
int64_t seer_colsum_workspace_floats(int64_t rows, int32_t cols);
int seer_colsum_bf16(const void* x, int64_t rows, int32_t cols, int32_t ldx, float* out, float* workspace, void* stream);

/* nn.LayerNorm backward (attention.py:198-200,275-277): dx = rstd (g - mean(g) - xhat mean(g xhat)) (+ dres), g = dy gamma;
 * dgamma = sum_r dy xhat, dbeta = sum_r dy (both NULL for a frozen norm).  dres: gradient arriving on the residual path
 * that shares x (fused add), or NULL. */
int seer_layernorm_bwd(const void* x, const void* dy, int64_t rows, int32_t C, int32_t ldx, int32_t lddy, const float* gamma,
                       float eps, const void* dres, int32_t ldres, void* dx, int32_t lddx, float* dgamma, float* dbeta,
                       float* workspace, void* stream);
/* The same, leaving d gamma / d beta as partial slabs: workspace[slab][2][C] (v 0 = d beta, v 1 = d gamma), seer_layernorm_bwd_slabs(rows)
 * slabs; seer_colfinal_grouped adds the slabs of MANY norms in one launch (ABI 24: the finals of every LayerNorm a backward walk
 * passed, deferred to its end -- nothing reads them before the optimizer, train.py:383-386).  Bits as seer_layernorm_bwd's. */
int64_t seer_layernorm_bwd_slabs(int64_t rows);
int seer_layernorm_bwd_partials(const void* x, const void* dy, int64_t rows, int32_t C, int32_t ldx, int32_t lddy, const float* gamma,
                                float eps, const void* dres, int32_t ldres, void* dx, int32_t lddx, float* workspace, void* stream);
/* out_v[c] = sum over k < nblocks of ws[(k * NV + v) * C + c], k in a fixed order; `items` is a HOST array (64 per launch, by value) */
typedef struct seer_colfinal_item {
    const float* ws;
    float* out0;            /* v = 0, or NULL */
    float* out1;            /* v = 1, or NULL */
    int32_t nblocks, NV, C, reserved;
} seer_colfinal_item;
int seer_colfinal_grouped(const seer_colfinal_item* items /* host */, int32_t n_items, void* stream);

/* GroupNorm (+ optional SiLU) backward over (C/G, F, H, W) per (b, g); stats/count/eps/gamma/beta/silu as in the forward
 * pair seer_groupnorm_stats / seer_groupnorm_apply.  dy bf16 [rows, C1+C2]; dx1/dx2 are the gradients of the two concat
 * sources (+ dres1/dres2 when given). */
int64_t seer_groupnorm_bwd_workspace_floats(int32_t C, int32_t batch, int64_t rows_per_batch, int32_t groups);
int seer_groupnorm_bwd(const void* x1, int32_t C1, const void* x2, int32_t C2, int32_t batch, int64_t rows_per_batch,
                       int32_t groups, const float* stats, double count, float eps, const float* gamma, const float* beta,
                       int32_t silu, const void* dy, const void* dres1, const void* dres2, void* dx1, void* dx2,
                       float* dgamma, float* dbeta, float* workspace, void* stream);

/* GEGLU (attention.py:785-793) outside the GEMM epilogue, on the interleaved projection layout of SEER_EPI_GEGLU
 * (columns [32g, 32g+16) values, [32g+16, 32g+32) gates -> output columns [16g, 16g+16)): the training forward keeps the
 * pre-activation for the backward. */
int seer_geglu_fwd(const void* pre, int64_t rows, int32_t inner, int32_t ldp, void* out, int32_t ldo, void* stream);
int seer_geglu_bwd(const void* pre, const void* dout, int64_t rows, int32_t inner, int32_t ldp, int32_t lddo, void* dpre,
                   int32_t lddp, void* stream);

/* y = a + b on bf16 [rows, cols] views (gradient fan-in) */
int seer_add_bf16(const void* a, int32_t lda, const void* b, int32_t ldb, void* y, int32_t ldy, int64_t rows, int32_t cols,
                  void* stream);
/* backward of the nearest-2x upsample (resnet.py:39): dx[img,y,x,:] = sum of the 2x2 block of du [n_img, 2H, 2W, C] */
int seer_sumpool2x_bf16(const void* du, int32_t n_img, int32_t H, int32_t W, int32_t C, void* dx, void* stream);
/* z [n_img, 2H, 2W, C]: z[2y, 2x] = d[y, x], zero elsewhere (input-gradient of a stride-2 conv, resnet.py:52-57) */
int seer_zero_insert2x_bf16(const void* d, int32_t n_img, int32_t H, int32_t W, int32_t C, void* z, void* stream);

/* epsilon-MSE of train.py:380: loss = mean((pred[:, :, cond_f:] - target)^2) over [B, C, F_total - cond_f, HW];
 * dpred fp32 [B, C, F_total, HW] = d loss / d pred (zero on the conditioning frames).  workspace: 1024 floats. */
int seer_mse_loss_grad(const float* pred, const float* target, int32_t B, int32_t C, int32_t F_total, int32_t cond_f,
                       int32_t HW, float* loss, float* dpred, float* workspace, void* stream);
/* `--text_loss` of train.py:346-347 (the FSTextTransformer initialisation stage): loss = mean_{b,l,c} (mean_f y - target)^2
 * with y bf16 [b, F, LC] (FSTextTransformer output), target fp32 [b, LC] (the CLIP sequence); its gradient is ADDED to dy
 * (bf16 [b, F, LC], the gradient arriving from the UNet).  workspace: 1024 floats. */
int seer_text_loss_grad(const void* y, const float* target, int32_t b, int32_t F, int64_t LC, void* dy, float* loss,
                        float* workspace, void* stream);
/* input gradient of conv_out (frozen): dpred fp32 [B, Cout, F, H, W] -> dx bf16 [B*F, H*W, C0]; W fp32 [Cout][3][3][C0] */
int seer_conv_out_bwd(const float* dpred, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W, const float* Wt,
                      int32_t Cout, void* dx, void* stream);

/* y = beta*y + alpha*x on fp32 buffers (x may alias y), n % 4 == 0 (gradient accumulation over micro-batches: train.py:321
 * `accelerator.accumulate`, configs/train.yaml gradient_accumulation_steps) */
int seer_axpby_f32(float* y, const float* x, float alpha, float beta, int64_t n, void* stream);
/* out[0] = sum g^2 (workspace: 1024 floats) */
int seer_sumsq_f32(const float* g, int64_t n, float* out, float* workspace, void* stream);
/* torch.optim.AdamW step (train.py:226-232,385) on flat fp32 buffers; when grad_sumsq != NULL the gradient is first scaled
 * by min(1, max_norm / (sqrt(*grad_sumsq) + 1e-6)) (clip_grad_norm_, train.py:384).  step = 1, 2, ...; p_bf16 (optional)
 * receives the bf16 working copy of the updated parameters. */
int seer_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, const float* grad_sumsq, float max_norm, void* p_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEER_HIP_H */
