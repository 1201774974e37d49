// Small / boundary kernels of the Seer hot path on gfx950: rotary embedding, timestep embedding, small-M linear,
// conv_in / conv_out (layout change fused), casts, CFG + DDIM update.
#include "seer_common.h"
#include <atomic>

namespace {

// ---- rotary ------------------------------------------------------------------------------------------------
// table[pos][j] = (cos, sin)(pos * freqs[j]) in fp32, angle rounded once like the reference's einsum
// (rotary-embedding-torch 0.1.5: freqs = einsum('..., f -> ... f', t.float(), freqs)).
__global__ void rotary_table_kernel(const float* __restrict__ freqs, int T, int half, float* __restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * half) return;
    const int pos = i / half, j = i - pos * half;
    const float ang = (float)pos * freqs[j];
    float s, c;
    sincosf(ang, &s, &c);
    cs[2 * i] = c;
    cs[2 * i + 1] = s;
}

// one wave per token row: lane -> (which of q/k, head, 8-element chunk of the rotated prefix)
__global__ void __launch_bounds__(256) rotary_kernel(bf16* __restrict__ x, int64_t rows, int ld, int col0_q, int col0_k,
                                                     int heads, int head_dim, int rot_dim, int tokens_per_batch,
                                                     int pos_offset, const float* __restrict__ cs) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int cpr = rot_dim / 8;                 // chunks per head
    const int per_which = heads * cpr;
    const int half = rot_dim / 2;
    const int pos = (int)(row % tokens_per_batch) + pos_offset;
    for (int item = lane; item < 2 * per_which; item += 64) {
        const int which = item / per_which;
        const int rem = item - which * per_which;
        const int head = rem / cpr, ch = rem - head * cpr;
        bf16* ptr = x + row * ld + (which ? col0_k : col0_q) + head * head_dim + ch * 8;
        u32x4 v = *reinterpret_cast<u32x4*>(ptr);
        float f[8];
        unpack8(v, f);
        const float* t = cs + ((int64_t)pos * half + ch * 4) * 2;
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            // t*cos + rotate_half(t)*sin, rotate_half: (x0, x1) -> (-x1, x0); rot_pair (seer_common.h) says why it is not written out
            const f32x2 r = rot_pair(f32x2{f[2 * pi], f[2 * pi + 1]}, t[2 * pi], t[2 * pi + 1]);
            f[2 * pi] = r[0];
            f[2 * pi + 1] = r[1];
        }
        *reinterpret_cast<u32x4*>(ptr) = pack8(f);
    }
}

// ---- timestep embedding ---------------------------------------------------------------------------------------
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, int B, int dim, int flip, float shift,
                                          float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    // diffusers 0.10.2 get_timestep_embedding: exponent = -ln(10000) * j / (half - shift); emb = t * exp(exponent)
    const float expo = -9.210340371976184f * (float)j / ((float)half - shift);
    const float arg = (float)t[b] * expf(expo);
    float s, c;
    sincosf(arg, &s, &c);
    float* o = out + (int64_t)b * dim;
    if (flip) { o[j] = c; o[half + j] = s; }
    else { o[j] = s; o[half + j] = c; }
}

// ---- small-M linear (time-embedding MLP, B <= 8 rows): act(x) staged once per block in LDS, each wave owns ROWS
// consecutive output features so ROWS independent 16-B weight loads per lane are in flight per K pass (the 51 MB
// time_emb_proj stack is a pure HBM stream) -----------------------------------------------------------------------
template <int MAXB, int ROWS, bool F16 = false>
__global__ void __launch_bounds__(256) linear_smallm_kernel(const float* __restrict__ x, int B, int K,
                                                            const bf16* __restrict__ W, const float* __restrict__ bias,
                                                            int N, int silu_in, int silu_out, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [B][K]
    for (int i = threadIdx.x; i < B * K; i += 256) {
        float v = x[i];
        if (silu_in) v = silu_f(v);
        xs[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (n0 >= N) return;
    float acc[ROWS][MAXB];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int b = 0; b < MAXB; ++b) acc[r][b] = 0.f;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
        u32x4 wv[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int n = n0 + r < N ? n0 + r : N - 1;
            wv[r] = *reinterpret_cast<const u32x4*>(W + (int64_t)n * K + k0);
        }
        float xv[MAXB][8];
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int bb = b < B ? b : 0;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(xs + bb * K + k0);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(xs + bb * K + k0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[b][e] = lo[e]; xv[b][4 + e] = hi[e]; }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float w[8];
            unpack8t<F16>(wv[r], w);
#pragma unroll
            for (int b = 0; b < MAXB; ++b)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r][b] += xv[b][e] * w[e];
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const float v0 = wave_sum(acc[r][b]);
            if (lane == 0 && b < B && n0 + r < N) {
                float v = v0;
                if (bias) v += bias[n0 + r];
                if (silu_out) v = silu_f(v);
                y[(int64_t)b * N + n0 + r] = v;
            }
        }
    }
}

// ---- conv_in: [B, Cin, F, H, W] fp32 -> channels-last bf16 [B*F*H*W, Cout]; weights fp32 [3][3][Cin][Cout] ----
// per block: the 9*Cin*Cout weights and the 3x3xCin input patches of its `ppb` pixels are staged in LDS once (every
// output-channel group needs the same 9*Cin inputs); a thread owns 8 output channels of one pixel lane
template <bool F16>
__global__ void __launch_bounds__(256) conv_in_kernel(const float* __restrict__ x, int B, int Cin, int F, int H, int Wd,
                                                      const float* __restrict__ Wt, const float* __restrict__ bias,
                                                      int Cout, bf16* __restrict__ y, int ppb) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];     // [9*Cin][Cout] then patches [ppb][9*Cin]
    const int nw = 9 * Cin * Cout;
    const int kk = 9 * Cin;
    float* patch = wsm + nw;
    for (int i = threadIdx.x * 4; i < nw; i += 1024) *reinterpret_cast<f32x4*>(wsm + i) = *reinterpret_cast<const f32x4*>(Wt + i);
    const int64_t npix = (int64_t)B * F * H * Wd;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int np = (int)(p0 + ppb < npix ? ppb : npix - p0);
    const int64_t cstride = (int64_t)F * H * Wd;   // channel stride of the NCFHW input
    for (int i = threadIdx.x; i < np * kk; i += 256) {
        const int pp = i / kk, k = i - pp * kk;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int64_t pix = p0 + pp;
        const int ox = (int)(pix % Wd);
        const int oy = (int)((pix / Wd) % H);
        const int f = (int)((pix / ((int64_t)Wd * H)) % F);
        const int b = (int)(pix / ((int64_t)Wd * H * F));
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < Wd)
            v = x[(((int64_t)b * Cin + ci) * F + f) * H * Wd + (int64_t)iy * Wd + ix];
        patch[i] = v;
    }
    __syncthreads();
    const int cgroups = Cout / 8;
    const int npl = 256 / cgroups;                // pixel lanes per block
    const int pl = threadIdx.x / cgroups;
    const int cg = threadIdx.x - pl * cgroups;
    if (pl >= npl) return;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias ? bias[cg * 8 + e] : 0.f;
    for (int pp = pl; pp < np; pp += npl) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bv[e];
        const float* xp = patch + pp * kk;
        const float* wp = wsm + cg * 8;
#pragma unroll 4
        for (int k = 0; k < kk; ++k) {
            const float xv = xp[k];
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + k * Cout);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + k * Cout + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += xv * w0[e]; acc[4 + e] += xv * w1[e]; }
        }
        *reinterpret_cast<u32x4*>(y + (p0 + pp) * Cout + cg * 8) = pack8t<F16>(acc);
    }
}

// ---- conv_out: channels-last bf16 [B*F*H*W, C0] -> [B, Cout, F, H, W] fp32; weights fp32 [Cout][3][3][C0] -------
// weights in LDS once per block; one wave per pixel (lanes over the 9 * C0/8 input chunks), `ppb` pixels per block
template <int COUT, bool F16 = false>
__global__ void __launch_bounds__(256) conv_out_kernel(const bf16* __restrict__ x, int B, int C0, int F, int H, int Wd,
                                                       const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       float* __restrict__ y, int ppb) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];     // [COUT][9][C0]
    const int nw = COUT * 9 * C0;
    for (int i = threadIdx.x * 4; i < nw; i += 1024) *reinterpret_cast<f32x4*>(wsm + i) = *reinterpret_cast<const f32x4*>(Wt + i);
    __syncthreads();
    const int64_t npix = (int64_t)B * F * H * Wd;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int64_t p1 = p0 + ppb < npix ? p0 + ppb : npix;
    const int lane = threadIdx.x & 63;
    const int nch = C0 / 8;
    for (int64_t pix = p0 + (threadIdx.x >> 6); pix < p1; pix += 4) {
        const int ox = (int)(pix % Wd);
        const int oy = (int)((pix / Wd) % H);
        const int64_t img = pix / ((int64_t)Wd * H);       // b*F + f
        // (even, odd) element pairs accumulated as whole packed operations: with scalar accumulators the SLP vectoriser builds
        // packed multiplies that read the high half of their second source (op_sel[1] = 1), the form asm_check.py refuses
        f32x2 acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = f32x2{0.f, 0.f};
        for (int item = lane; item < 9 * nch; item += 64) {
            const int tap = item / nch, ch = item - tap * nch;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + ((img * H + iy) * Wd + ix) * C0 + ch * 8);
            float f[8];
            unpack8t<F16>(v, f);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float* w = wsm + (o * 9 + tap) * C0 + ch * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + 4);
                acc[o] = __builtin_elementwise_fma(f32x2{f[0], f[1]}, f32x2{w0[0], w0[1]}, acc[o]);
                acc[o] = __builtin_elementwise_fma(f32x2{f[2], f[3]}, f32x2{w0[2], w0[3]}, acc[o]);
                acc[o] = __builtin_elementwise_fma(f32x2{f[4], f[5]}, f32x2{w1[0], w1[1]}, acc[o]);
                acc[o] = __builtin_elementwise_fma(f32x2{f[6], f[7]}, f32x2{w1[2], w1[3]}, acc[o]);
            }
        }
        const int f_ = (int)(img % F);
        const int b = (int)(img / F);
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float v = wave_sum(acc[o][0] + acc[o][1]);
            if (lane == 0) y[((((int64_t)b * COUT + o) * F + f_) * H + oy) * Wd + ox] = v + (bias ? bias[o] : 0.f);
        }
    }
}

template <bool F16>
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, int64_t n, bf16* __restrict__ y) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        u32x2 o;
        if constexpr (F16) {
            o[0] = pack2h(v[0], v[1]);
            o[1] = pack2h(v[2], v[3]);
        } else {
            o[0] = pack2(v[0], v[1]);
            o[1] = pack2(v[2], v[3]);
        }
        *reinterpret_cast<u32x2*>(y + i) = o;
    } else {
        for (int64_t j = i; j < n; ++j) {
            if constexpr (F16) reinterpret_cast<_Float16*>(y)[j] = (_Float16)x[j];
            else y[j] = (bf16)x[j];
        }
    }
}

// [N, C, HW] fp32 <-> [N, HW, C] bf16 through a 32x32 LDS tile
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, int C, int HW, bf16* __restrict__ y) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? x[((int64_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (c < C && p < HW) y[((int64_t)n * HW + p) * C + c] = (bf16)tile[tx][i];
    }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16* __restrict__ x, int C, int HW, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (c < C && p < HW) ? (float)x[((int64_t)n * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) y[((int64_t)n * C + c) * HW + p] = tile[tx][i];
    }
}

// ---- CFG + DDIM update (fp32, one thread per latent element) ------------------------------------------------------
__global__ void cfg_ddim_kernel(const float* __restrict__ eps, int cfg, int b, int C, int Ft, int cond_f, int HW,
                                float scale, const float* __restrict__ coef, int index, const float* __restrict__ x,
                                const float* __restrict__ noise, float* __restrict__ x_prev,
                                float* __restrict__ pred_x0) {
    const int Fp = Ft - cond_f;
    const int64_t n = (int64_t)b * C * Fp * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int hw = (int)(i % HW);
    const int f = (int)((i / HW) % Fp);
    const int c = (int)((i / ((int64_t)HW * Fp)) % C);
    const int bi = (int)(i / ((int64_t)HW * Fp * C));
    const int64_t eoff = (((int64_t)bi * C + c) * Ft + (f + cond_f)) * HW + hw;
    float e;
    if (cfg) {
        const float eu = eps[eoff];
        const float ec = eps[eoff + (int64_t)b * C * Ft * HW];
        e = eu + scale * (ec - eu);
    } else {
        e = eps[eoff];
    }
    const float a_t = coef[4 * index], a_prev = coef[4 * index + 1], sigma = coef[4 * index + 2],
                s1m = coef[4 * index + 3];
    const float xv = x[i];
    const float x0 = (xv - s1m * e) / sqrtf(a_t);
    const float dir = sqrtf(1.f - a_prev - sigma * sigma) * e;
    float nz = 0.f;
    if (noise) nz = sigma * noise[i];
    x_prev[i] = sqrtf(a_prev) * x0 + dir + nz;
    if (pred_x0) pred_x0[i] = x0;
}

// ---- the same step with its schedule index in DEVICE memory (a whole p_sample_ddim captured in one hipGraph: the kernel
// arguments of a replay are frozen, the step counter is not).  step[0] = index of the NEXT step to run, step[1] = index of the
// step in flight.  seer_ddim_step_begin (first kernel of a step) reads step[0] everywhere and one thread copies it to step[1];
// seer_cfg_ddim_step_dev (last kernel) reads step[1] everywhere and one thread writes step[0] = index - 1: no kernel both reads
// and writes a word, so a chain of replays walks the schedule downwards without the host touching device memory.
// Input assembly of ddim_video.py:189,201-203: the UNet input [reps*b, C, F, HW] = cat([x0_emb, x], frames), repeated for the CFG
// pair, and t[reps*b] = timesteps[index].
__global__ void ddim_assemble_kernel(const float* __restrict__ x0_emb, const float* __restrict__ x, int b, int reps, int C, int f1,
                                     int Fp, int HW, const int64_t* __restrict__ t_table, int* __restrict__ step,
                                     float* __restrict__ sample, int64_t* __restrict__ t_out) {
    const int F = f1 + Fp;
    const int64_t per = (int64_t)b * C * F * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int index = step[0];
    if (i == 0) step[1] = index;
    if (i < reps * b) t_out[i] = t_table[index];
    if (i >= per) return;
    const int hw = (int)(i % HW);
    const int f = (int)((i / HW) % F);
    const int64_t bc = i / ((int64_t)HW * F);            // bi * C + c
    const float v = f < f1 ? x0_emb[(bc * f1 + f) * HW + hw] : x[(bc * Fp + (f - f1)) * HW + hw];
    for (int r = 0; r < reps; ++r) sample[i + r * per] = v;
}

// x may alias x_prev (the captured step updates its latent in place): every thread reads its element before it writes it
__global__ void cfg_ddim_dev_kernel(const float* __restrict__ eps, int cfg, int b, int C, int Ft, int cond_f, int HW, float scale,
                                    const float* __restrict__ coef, int* __restrict__ step, const float* x,
                                    const float* __restrict__ noise, float* x_prev, float* __restrict__ pred_x0) {
    const int Fp = Ft - cond_f;
    const int64_t n = (int64_t)b * C * Fp * HW;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int index = step[1];
    if (i == 0) step[0] = index - 1;
    if (i >= n) return;
    const int hw = (int)(i % HW);
    const int f = (int)((i / HW) % Fp);
    const int c = (int)((i / ((int64_t)HW * Fp)) % C);
    const int bi = (int)(i / ((int64_t)HW * Fp * C));
    const int64_t eoff = (((int64_t)bi * C + c) * Ft + (f + cond_f)) * HW + hw;
    float e;
    if (cfg) {
        const float eu = eps[eoff];
        const float ec = eps[eoff + (int64_t)b * C * Ft * HW];
        e = eu + scale * (ec - eu);
    } else {
        e = eps[eoff];
    }
    const float a_t = coef[4 * index], a_prev = coef[4 * index + 1], sigma = coef[4 * index + 2], s1m = coef[4 * index + 3];
    const float xv = x[i];
    const float x0 = (xv - s1m * e) / sqrtf(a_t);
    const float dir = sqrtf(1.f - a_prev - sigma * sigma) * e;
    float nz = 0.f;
    if (noise) nz = sigma * noise[i];
    x_prev[i] = sqrtf(a_prev) * x0 + dir + nz;
    if (pred_x0) pred_x0[i] = x0;
}

// pointwise channel mix on NCHW fp32 (VAE post_quant_conv, 4 -> 4): y[n,co,p] = sum_ci W[co,ci] x[n,ci,p] + b[co]
__global__ void conv1x1_nchw_kernel(const float* __restrict__ x, int N, int Cin, int Cout, int HW,
                                    const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Cout * HW) return;
    const int p = (int)(i % HW);
    const int co = (int)((i / HW) % Cout);
    const int n = (int)(i / ((int64_t)HW * Cout));
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) acc += Wt[co * Cin + ci] * x[((int64_t)n * Cin + ci) * HW + p];
    y[i] = acc;
}

__global__ void clamp01_kernel(float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = (x[i] + 1.0f) * 0.5f;
        x[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// DiagonalGaussianDistribution.sample: moments [N, 2C, HW] = (mean | logvar)
__global__ void gaussian_sample_kernel(const float* __restrict__ mom, int C, int HW, const float* __restrict__ noise,
                                       float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t chw = (int64_t)C * HW;
    const int64_t img = i / chw, r = i - img * chw;
    const float mean = mom[img * 2 * chw + r];
    float v = mean;
    if (noise) {
        const float logvar = fminf(fmaxf(mom[img * 2 * chw + chw + r], -30.0f), 20.0f);
        v = fmaf(__expf(0.5f * logvar), noise[i], mean);
    }
    out[i] = v;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" int seer_rotary_table(const float* freqs, int32_t T, int32_t half, float* cos_sin, void* stream) {
    if (!freqs || !cos_sin || T <= 0 || half <= 0) return SEER_EINVAL;
    const int n = T * half;
    hipLaunchKernelGGL(rotary_table_kernel, dim3((n + 255) / 256), dim3(256), 0, S(stream), freqs, T, half, cos_sin);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_rotary_inplace(void* x, int64_t rows, int32_t ld, int32_t col0_q, int32_t col0_k, int32_t heads,
                                   int32_t head_dim, int32_t rot_dim, int32_t tokens_per_batch, int32_t pos_offset,
                                   const float* cos_sin, void* stream) {
    if (!x || !cos_sin || rows <= 0 || heads <= 0 || rot_dim <= 0 || rot_dim % 8 || rot_dim > head_dim) return SEER_EINVAL;
    if (ld % 8 || col0_q % 8 || col0_k % 8 || head_dim % 8 || tokens_per_batch <= 0) return SEER_EINVAL;
    hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S(stream),
                       reinterpret_cast<bf16*>(x), rows, ld, col0_q, col0_k, heads, head_dim, rot_dim, tokens_per_batch,
                       pos_offset, cos_sin);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_timestep_embedding(const int64_t* t, int32_t B, int32_t dim, int32_t flip_sin_to_cos,
                                       float freq_shift, float* out, void* stream) {
    if (!t || !out || B <= 0 || dim <= 0 || dim % 2) return SEER_EINVAL;
    const int n = B * dim / 2;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, S(stream), t, B, dim,
                       flip_sin_to_cos, freq_shift, out);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

// kernels whose dynamic LDS may exceed the 64 KiB default: opt in once per kernel (160 KiB per CU on gfx950)
template <typename K>
int ensure_lds(K kernel, size_t bytes, std::atomic<bool>* done) {      // setting the attribute twice is harmless: the flag only has to be race-free
    if (bytes > 160 * 1024) return SEER_EINVAL;
    if (bytes > 64 * 1024 && !done->load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return SEER_ELAUNCH;
        done->store(true, std::memory_order_release);
    }
    return SEER_OK;
}

extern "C" int seer_linear_smallm(const float* x, int32_t B, int32_t K, const void* W, const float* bias, int32_t N,
                                  int32_t silu_in, int32_t silu_out, float* y, void* stream) {
    return seer_linear_smallm_dt(x, B, K, W, bias, N, silu_in, silu_out, y, SEER_DT_BF16, stream);
}
extern "C" int seer_linear_smallm_dt(const float* x, int32_t B, int32_t K, const void* W, const float* bias, int32_t N,
                                     int32_t silu_in, int32_t silu_out, float* y, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !W || !y || B <= 0 || B > 8 || K <= 0 || K % 8 || N <= 0) return SEER_EINVAL;
    const size_t lds = (size_t)B * K * sizeof(float);
    const bf16* Wb = reinterpret_cast<const bf16*>(W);
    static std::atomic<bool> done[4] = {{false}, {false}, {false}, {false}};
#define SEER_LSM(MB, R, H, GRID, FLAG)                                                                                              \
    do {                                                                                                                            \
        const int rc = ensure_lds(linear_smallm_kernel<MB, R, H>, lds, &done[FLAG]);                                                \
        if (rc != SEER_OK) return rc;                                                                                               \
        hipLaunchKernelGGL((linear_smallm_kernel<MB, R, H>), dim3(GRID), dim3(256), lds, S(stream), x, B, K, Wb, bias, N, silu_in,   \
                           silu_out, y);                                                                                            \
    } while (0)
    if (B <= 2) {
        if (dtype == SEER_DT_F16) SEER_LSM(2, 4, true, (N + 15) / 16, 0);
        else SEER_LSM(2, 4, false, (N + 15) / 16, 1);
    } else {
        if (dtype == SEER_DT_F16) SEER_LSM(8, 2, true, (N + 7) / 8, 2);
        else SEER_LSM(8, 2, false, (N + 7) / 8, 3);
    }
#undef SEER_LSM
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_conv_in_dt(const float* x, int32_t B, int32_t Cin, int32_t F, int32_t H, int32_t W_, const float* Wt,
                               const float* bias, int32_t Cout, void* y, int32_t dtype, void* stream);
extern "C" int seer_conv_in(const float* x, int32_t B, int32_t Cin, int32_t F, int32_t H, int32_t W_, const float* Wt,
                            const float* bias, int32_t Cout, void* y, void* stream) {
    return seer_conv_in_dt(x, B, Cin, F, H, W_, Wt, bias, Cout, y, SEER_DT_BF16, stream);
}
extern "C" int seer_conv_in_dt(const float* x, int32_t B, int32_t Cin, int32_t F, int32_t H, int32_t W_, const float* Wt,
                               const float* bias, int32_t Cout, void* y, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !Wt || !y || B <= 0 || Cin <= 0 || F <= 0 || H <= 0 || W_ <= 0 || Cout <= 0 || Cout % 8 || Cout > 2048)
        return SEER_EINVAL;
    const int ppb = 32;
    const size_t lds = ((size_t)9 * Cin * Cout + (size_t)ppb * 9 * Cin) * sizeof(float);
    static std::atomic<bool> done{false}, done16{false};
    const int rc = dtype == SEER_DT_F16 ? ensure_lds(conv_in_kernel<true>, lds, &done16) : ensure_lds(conv_in_kernel<false>, lds, &done);
    if (rc != SEER_OK) return rc;
    const int64_t npix = (int64_t)B * F * H * W_;
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL(conv_in_kernel<true>, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), lds, S(stream), x, B, Cin, F, H, W_,
                           Wt, bias, Cout, reinterpret_cast<bf16*>(y), ppb);
    else
        hipLaunchKernelGGL(conv_in_kernel<false>, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), lds, S(stream), x, B, Cin, F, H, W_,
                           Wt, bias, Cout, reinterpret_cast<bf16*>(y), ppb);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_conv_out_dt(const void* x, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                                const float* bias, int32_t Cout, float* y, int32_t dtype, void* stream);
extern "C" int seer_conv_out(const void* x, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                             const float* bias, int32_t Cout, float* y, void* stream) {
    return seer_conv_out_dt(x, B, C0, F, H, W_, Wt, bias, Cout, y, SEER_DT_BF16, stream);
}
extern "C" int seer_conv_out_dt(const void* x, int32_t B, int32_t C0, int32_t F, int32_t H, int32_t W_, const float* Wt,
                                const float* bias, int32_t Cout, float* y, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !Wt || !y || B <= 0 || C0 <= 0 || C0 % 8 || F <= 0 || H <= 0 || W_ <= 0) return SEER_EINVAL;
    const int64_t npix = (int64_t)B * F * H * W_;
    const int ppb = 32;
    dim3 grid((unsigned)((npix + ppb - 1) / ppb));
    const size_t lds = (size_t)Cout * 9 * C0 * sizeof(float);
    const bf16* xb = reinterpret_cast<const bf16*>(x);
    if (dtype == SEER_DT_F16) {
        if (Cout != 3) return SEER_ENOSYS;            // the VAE decoder's RGB head is the one fp16 consumer
        static std::atomic<bool> done16{false};
        const int rc = ensure_lds(conv_out_kernel<3, true>, lds, &done16);
        if (rc != SEER_OK) return rc;
        hipLaunchKernelGGL((conv_out_kernel<3, true>), grid, dim3(256), lds, S(stream), xb, B, C0, F, H, W_, Wt, bias, y, ppb);
    } else if (Cout == 4) {
        static std::atomic<bool> done{false};
        const int rc = ensure_lds(conv_out_kernel<4>, lds, &done);
        if (rc != SEER_OK) return rc;
        hipLaunchKernelGGL(conv_out_kernel<4>, grid, dim3(256), lds, S(stream), xb, B, C0, F, H, W_, Wt, bias, y, ppb);
    } else if (Cout == 3) {
        static std::atomic<bool> done{false};
        const int rc = ensure_lds(conv_out_kernel<3>, lds, &done);
        if (rc != SEER_OK) return rc;
        hipLaunchKernelGGL(conv_out_kernel<3>, grid, dim3(256), lds, S(stream), xb, B, C0, F, H, W_, Wt, bias, y, ppb);
    } else {
        return SEER_ENOSYS;
    }
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_cast_f32_bf16(const float* x, int64_t n, void* y, void* stream) { return seer_cast_f32_dt(x, n, y, SEER_DT_BF16, stream); }
extern "C" int seer_cast_f32_dt(const float* x, int64_t n, void* y, int32_t dtype, void* stream) {
    if (dtype != SEER_DT_BF16 && dtype != SEER_DT_F16) return SEER_EINVAL;
    if (!x || !y || n <= 0) return SEER_EINVAL;
    const int64_t nt = (n + 3) / 4;
    if (dtype == SEER_DT_F16)
        hipLaunchKernelGGL(cast_f32_bf16_kernel<true>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, S(stream), x, n, reinterpret_cast<bf16*>(y));
    else
        hipLaunchKernelGGL(cast_f32_bf16_kernel<false>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, S(stream), x, n, reinterpret_cast<bf16*>(y));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_nchw_f32_to_nhwc_bf16(const float* x, int32_t N, int32_t C, int32_t HW, void* y, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return SEER_EINVAL;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, S(stream), x, C, HW, reinterpret_cast<bf16*>(y));
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}
extern "C" int seer_nhwc_bf16_to_nchw_f32(const void* x, int32_t N, int32_t C, int32_t HW, float* y, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return SEER_EINVAL;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, S(stream), reinterpret_cast<const bf16*>(x), C, HW, y);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_cfg_ddim_step(const float* eps, int32_t cfg, int32_t b, int32_t C, int32_t F_total, int32_t cond_f,
                                  int32_t HW, float scale, const float* coef, int32_t index, const float* x,
                                  const float* noise, float* x_prev, float* pred_x0, void* stream) {
    if (!eps || !coef || !x || !x_prev || b <= 0 || C <= 0 || F_total <= cond_f || cond_f < 0 || HW <= 0 || index < 0)
        return SEER_EINVAL;
    const int64_t n = (int64_t)b * C * (F_total - cond_f) * HW;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), eps, cfg, b, C,
                       F_total, cond_f, HW, scale, coef, index, x, noise, x_prev, pred_x0);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_ddim_step_begin(const float* x0_emb, const float* x, int32_t b, int32_t reps, int32_t C, int32_t f1,
                                    int32_t F_pred, int32_t HW, const int64_t* t_table, int32_t* step, float* sample,
                                    int64_t* t_out, void* stream) {
    if (!x || !t_table || !step || !sample || !t_out || b <= 0 || reps <= 0 || C <= 0 || f1 < 0 || F_pred <= 0 || HW <= 0 ||
        (f1 > 0 && !x0_emb))
        return SEER_EINVAL;
    const int64_t n = (int64_t)b * C * (f1 + F_pred) * HW;
    hipLaunchKernelGGL(ddim_assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x0_emb, x, b, reps, C, f1,
                       F_pred, HW, t_table, step, sample, t_out);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_cfg_ddim_step_dev(const float* eps, int32_t cfg, int32_t b, int32_t C, int32_t F_total, int32_t cond_f,
                                      int32_t HW, float scale, const float* coef, int32_t* step, const float* x,
                                      const float* noise, float* x_prev, float* pred_x0, void* stream) {
    if (!eps || !coef || !step || !x || !x_prev || b <= 0 || C <= 0 || F_total <= cond_f || cond_f < 0 || HW <= 0) return SEER_EINVAL;
    const int64_t n = (int64_t)b * C * (F_total - cond_f) * HW;
    hipLaunchKernelGGL(cfg_ddim_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), eps, cfg, b, C, F_total,
                       cond_f, HW, scale, coef, step, x, noise, x_prev, pred_x0);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_conv1x1_nchw_f32(const float* x, int32_t N, int32_t Cin, int32_t Cout, int32_t HW, const float* Wt,
                                     const float* bias, float* y, void* stream) {
    if (!x || !Wt || !y || N <= 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return SEER_EINVAL;
    const int64_t n = (int64_t)N * Cout * HW;
    hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, N, Cin, Cout,
                       HW, Wt, bias, y);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_clamp01(float* x, int64_t n, void* stream) {
    if (!x || n <= 0) return SEER_EINVAL;
    hipLaunchKernelGGL(clamp01_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), x, n);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_gaussian_sample(const float* moments, int32_t N, int32_t C, int32_t HW, const float* noise, float* out,
                                    void* stream) {
    if (!moments || !out || N <= 0 || C <= 0 || HW <= 0) return SEER_EINVAL;
    const int64_t n = (int64_t)N * C * HW;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), moments, C, HW,
                       noise, out, n);
    SEER_LAUNCH_CHECK();
    return SEER_OK;
}

extern "C" int seer_abi_version(void) { return 24; }
extern "C" const char* seer_build_arch(void) { return "gfx950"; }
extern "C" const char* seer_strerror(int code) {
    switch (code) {
        case SEER_OK: return "ok";
        case SEER_EINVAL: return "invalid argument (shape/alignment/flags)";
        case SEER_ENOSYS: return "shape class not built";
        case SEER_ELAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
