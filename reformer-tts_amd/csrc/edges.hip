// Kernels of the convolutional "edges" of the model: encoder prenet convolutions and the mel postnet
// (/root/reference/reformer_tts/model/modules.py:8-61,103-169) and the loss (model/loss.py:28-53).
//
// A Conv1d(k=5, pad=2) on channels-last rows is an IMPLICIT GEMM (rtts_conv1d_k5, csrc/gemm_nt.hip): the activations of a
// convolution stack live in "halo rows" -- (B, L + 4, C) with two zero rows in front of and behind every sequence -- so the
// five taps are five row-shifted reads of the same array (forward, input gradient and weight gradient alike) and no window
// matrix is ever written.  Everything around the GEMMs is here and knows the halo:
//   to_halo                        plain rows (fp32 | bf16) -> bf16 halo rows (the stack's input)
//   conv weight permute + cast     (Cout,Cin,5) fp32 master -> (Cout,5,Cin_pad) bf16, and the adjoint for dW
//   col_stats                      per-channel batch mean / rstd (+ running-stat update), two-stage, deterministic
//   bn_act_fwd                     z = dropout(act(gamma * (y-mean)*rstd + beta))  -> bf16
//   bn_act_bwd (stats + apply)     full BatchNorm(train) backward through act and dropout
//   tts_loss_fwd_bwd               masked MSE/L1 on raw + postnet mel, BCE-with-logits(pos_weight) on stop:
//                                  the three means AND their gradients in one pass
// Dropout masks come from a counter-based hash of (seed, element index): reproducible in the backward
// without storing a mask.  All HBM-streaming; column sums via per-workgroup partial rows (no atomics).
#include "rtts_common.h"

#define ED_THREADS 256
#define ED_PBLOCKS 256

// keep-scale of element idx: 0 (dropped) or 1/(1-p)   (rtts_common.h)
#define ed_drop rtts_drop_keep

// ------------------------------------------------------------------ halo rows
// Row m of a halo array belongs to sequence b = m / P (P = L + 2H) at position t = m % P - H; it carries data iff b < B and
// 0 <= t < L, and is zero otherwise.  A plain array is the case H = 0, P = L.
struct EdHalo {
    int B, L, P, H;
};
__device__ __forceinline__ bool ed_valid(const EdHalo& g, long long m, int& b, int& t) {
    if (m < 0 || m >= (long long)g.B * g.P) return false;
    // m / P without the 64-bit division sequence (~100 instructions in every element of three streaming kernels): the float
    // quotient is within one of the truth for m < 2^31 (B * P rows), two compares put it right
    const int mi = (int)m;
    int q = (int)((float)mi * __builtin_amdgcn_rcpf((float)g.P));
    int r = mi - q * g.P;
    if (r < 0) { --q; r += g.P; }
    if (r >= g.P) { ++q; r -= g.P; }
    b = q;
    t = r - g.H;
    return (unsigned)t < (unsigned)g.L;
}
// element index -> (row, channel) for a row of C channels (the index fits 32 bits: checked on the host); C is a power of two in
// every configuration (shift / mask), anything else takes the 32-bit division
__device__ __forceinline__ void ed_row_col(unsigned e, int C, int cshift, unsigned& row, int& c) {
    if (cshift >= 0) {
        row = e >> cshift;
        c = (int)(e & (unsigned)(C - 1));
    } else {
        row = e / (unsigned)C;
        c = (int)(e - row * (unsigned)C);
    }
}
static inline int ed_cshift(int C) { return (C & (C - 1)) == 0 ? __builtin_ctz((unsigned)C) : -1; }
// tanh(x) = 1 - 2 / (e^{2x} + 1): two fast-math instructions instead of libm's branchy polynomial (the results go to bf16 or
// into a gradient that is rounded to bf16; |error| ~ 1e-6)
__device__ __forceinline__ float ed_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // e^{2x}
    return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
static inline EdHalo ed_halo(int B, int L, int halo) { return EdHalo{B, L, L + 2 * halo, halo}; }

// dst (bf16 halo rows: `rows` rows of C channels, halo row 0 at row `lead`) = src (plain (B*L, ld_src) rows, fp32 or bf16);
// every row outside the valid set is written as zero.  8 channels per thread.
template <bool SRC_F32>
__global__ __launch_bounds__(ED_THREADS) void to_halo_kernel(const void* __restrict__ src, int64_t ld_src, int64_t bs_src, int C_src, EdHalo g, int C,
                                                             int lead, long long rows, bf16_t* __restrict__ dst) {
    const int pieces = C / 8;
    const size_t total = (size_t)rows * pieces;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % pieces);
        const long long r = (long long)(i / pieces);
        int b, t;
        uint4 o = make_uint4(0, 0, 0, 0);
        if (pc * 8 < C_src && ed_valid(g, r - lead, b, t)) {        // channels >= C_src: zero padding up to the GEMM's K granule
            const size_t soff = (size_t)b * bs_src + (size_t)t * ld_src;     // bs_src: elements between samples (a time-sliced view)
            if (SRC_F32) {
                const float* p = reinterpret_cast<const float*>(src) + soff + pc * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
                o.x = pack_bf16x2(v0.x, v0.y); o.y = pack_bf16x2(v0.z, v0.w);
                o.z = pack_bf16x2(v1.x, v1.y); o.w = pack_bf16x2(v1.z, v1.w);
            } else {
                o = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(src) + soff + pc * 8);
            }
        }
        *reinterpret_cast<uint4*>(dst + (size_t)r * C + pc * 8) = o;
    }
}

// ------------------------------------------------------------------ conv weight layouts
// wp[co][k][ci] (bf16, Cin padded to CP with zeros) = w[co][ci][k] (fp32)
__global__ __launch_bounds__(ED_THREADS) void conv_w_perm_kernel(const float* __restrict__ w, int Co, int Ci, int CP, bf16_t* __restrict__ wp) {
    const size_t total = (size_t)Co * 5 * CP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CP), k = (int)((i / CP) % 5), co = (int)(i / ((size_t)CP * 5));
        wp[i] = ci < Ci ? f32_to_bf16(w[((size_t)co * Ci + ci) * 5 + k]) : (bf16_t)0;
    }
}
struct CwGroup {
    rtts_conv_perm_job j[RTTS_CONV_PERM_MAX_GROUP];
    int blk_start[RTTS_CONV_PERM_MAX_GROUP + 1];
    int n;
};
// the same re-layout for up to 8 convolutions in one grid: blockIdx.x picks the job by its block range
__global__ __launch_bounds__(ED_THREADS) void conv_w_perm_grouped_kernel(const CwGroup g) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < RTTS_CONV_PERM_MAX_GROUP; ++i)
        if (i < g.n && (int)blockIdx.x >= g.blk_start[i]) ji = i;
    const rtts_conv_perm_job& J = g.j[ji];
    const int Co = J.Co, Ci = J.Ci, CP = J.CP;
    const float* __restrict__ w = J.w;
    bf16_t* __restrict__ wp = (bf16_t*)J.wp;
    // one (co, ci) pair per thread and pass: its five taps are 20 contiguous bytes of w (a wave reads 1280 contiguous bytes; with a
    // tap per thread every lane touched a different 20-byte group and used 4 of them), and each tap's row of wp takes a wave's 64
    // values as one 128-byte store
    const size_t total = (size_t)Co * CP;
    const size_t nthr = (size_t)(g.blk_start[ji + 1] - g.blk_start[ji]) * blockDim.x;
    for (size_t i = (size_t)((int)blockIdx.x - g.blk_start[ji]) * blockDim.x + threadIdx.x; i < total; i += nthr) {
        const int ci = (int)(i % CP), co = (int)(i / CP);
        float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (ci < Ci) {
            const float* src = w + ((size_t)co * Ci + ci) * 5;
#pragma unroll
            for (int k = 0; k < 5; ++k) t[k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) wp[((size_t)co * 5 + k) * CP + ci] = f32_to_bf16(t[k]);
    }
}
__global__ __launch_bounds__(ED_THREADS) void conv_dw_unperm_grouped_kernel(const CwGroup g) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < RTTS_CONV_PERM_MAX_GROUP; ++i)
        if (i < g.n && (int)blockIdx.x >= g.blk_start[i]) ji = i;
    const rtts_conv_perm_job& J = g.j[ji];
    const int Co = J.Co, Ci = J.Ci, CP = J.CP;
    const float* __restrict__ dwp = J.w;
    float* __restrict__ dw = (float*)J.wp;
    // one (co, ci) pair per thread and pass, as above: five coalesced row reads, 20 contiguous bytes of dw updated
    const size_t total = (size_t)Co * Ci;
    const size_t nthr = (size_t)(g.blk_start[ji + 1] - g.blk_start[ji]) * blockDim.x;
    for (size_t i = (size_t)((int)blockIdx.x - g.blk_start[ji]) * blockDim.x + threadIdx.x; i < total; i += nthr) {
        const int ci = (int)(i % Ci), co = (int)(i / Ci);
        float* dst = dw + i * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] += dwp[((size_t)co * 5 + k) * CP + ci];
    }
}
// dw[co][ci][k] += dwp[co][k][ci]
__global__ __launch_bounds__(ED_THREADS) void conv_dw_unperm_kernel(const float* __restrict__ dwp, int Co, int Ci, int CP, float* __restrict__ dw) {
    const size_t total = (size_t)Co * Ci * 5;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % 5), ci = (int)((i / 5) % Ci), co = (int)(i / ((size_t)Ci * 5));
        dw[i] += dwp[((size_t)co * 5 + k) * CP + ci];
    }
}

// ------------------------------------------------------------------ per-channel partial sums over rows
// partial[blk][0..C) = sum_rows a(row,c); partial[blk][C..2C) = sum_rows b(row,c); thread owns 4 channels
template <int MODE>   // 0: (y, y*y)   1: (g, g*yhat) with g = dz * act'(.) * dropmask
__global__ __launch_bounds__(ED_THREADS) void col_partial_kernel(const float* __restrict__ y, const bf16_t* __restrict__ dz,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                 uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale, EdHalo g,
                                                                 int dz_halo, int C, float* __restrict__ partial) {
    if (seed_dev) seed += seed_dev[0];
    const int M = g.B * g.P;               // rows of y that can carry data (halo rows among them are skipped)
    // block = 64 channel-quads x 4 row-lanes; grid.x = row slabs, grid.y = channel groups of 256
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + cq * 4;
    __shared__ float red[2][4][256];
    float sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    if (c < C) {
        float mu[4] = {0, 0, 0, 0}, rs[4] = {1, 1, 1, 1}, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
        }
        for (int row = blockIdx.x * 4 + rl; row < M; row += gridDim.x * 4) {
            int sb_, st_;
            if (!ed_valid(g, row, sb_, st_)) continue;
            const float4 yv = *reinterpret_cast<const float4*>(y + (size_t)row * C + c);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { sa[j] += yy[j]; sb[j] = __builtin_fmaf(yy[j], yy[j], sb[j]); }
            } else {
                const size_t drow = dz_halo ? (size_t)row : (size_t)sb_ * g.L + st_;
                const uint2 dv = *reinterpret_cast<const uint2*>(dz + drow * C + c);
                const float dd[4] = {__uint_as_float(dv.x << 16), __uint_as_float(dv.x & 0xffff0000u), __uint_as_float(dv.y << 16),
                                     __uint_as_float(dv.y & 0xffff0000u)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float yh = (yy[j] - mu[j]) * rs[j];
                    const float pre = __builtin_fmaf(yh, ga[j], be[j]);
                    float da;
                    if (act == 1) da = pre > 0.f ? 1.f : 0.f;
                    else { const float th = ed_tanh(pre); da = 1.f - th * th; }
                    const float g = dd[j] * da * (thresh ? ed_drop(seed, (uint32_t)((size_t)row * C + c + j), thresh, dscale) : 1.f);
                    sa[j] += g;
                    sb[j] = __builtin_fmaf(g, yh, sb[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cq * 4 + j] = sa[j]; red[1][rl][cq * 4 + j] = sb[j]; }
    __syncthreads();
    const int t = threadIdx.x;   // 256 channels of this group
    if (blockIdx.y * 256 + t < C) {
        float* prow = partial + (size_t)blockIdx.x * 2 * C;
        prow[blockIdx.y * 256 + t] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
        prow[C + blockIdx.y * 256 + t] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
    }
}

// column sums of the two partial planes: a block owns 64 channels, its 4 waves each sum every 4th partial row
// (coalesced 256-B reads), then the 4 wave sums are combined in wave order (deterministic)
#define ED_FIN_WAVES 16     // waves of a finalize block: 8 blocks of 4 waves walked 64 partial rows each in a dependent chain (9 us)
__device__ __forceinline__ void ed_reduce_partials(const float* __restrict__ partial, int nrows, int C, int c, float& s, float& q,
                                                   float (*red)[ED_FIN_WAVES][64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    if (c < C) {
        // up to 16 rows per wave, all loads in flight at once, added in row order (fixed order: deterministic)
        float va[16], vb[16];
        for (int r0 = wave; r0 < nrows; r0 += ED_FIN_WAVES * 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = r0 + u * ED_FIN_WAVES;
                va[u] = r < nrows ? partial[(size_t)r * 2 * C + c] : 0.f;
                vb[u] = r < nrows ? partial[(size_t)r * 2 * C + C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a += va[u];
                b += vb[u];
            }
        }
    }
    red[0][wave][lane] = a;
    red[1][wave][lane] = b;
    __syncthreads();
    s = 0.f;
    q = 0.f;
#pragma unroll
    for (int w = 0; w < ED_FIN_WAVES; ++w) {
        s += red[0][w][lane];
        q += red[1][w][lane];
    }
}

// mean/rstd from the partials (+ running statistics, momentum 0.1, unbiased variance)
__global__ __launch_bounds__(64 * ED_FIN_WAVES) void bn_finalize_stats_kernel(const float* __restrict__ partial, int nrows, int M, int C,
                                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                       const float* __restrict__ mean_shift, long long* __restrict__ num_batches) {
    __shared__ float red[2][ED_FIN_WAVES][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    float s, q;
    ed_reduce_partials(partial, nrows, C, c, s, q, red);
    if (num_batches && blockIdx.x == 0 && threadIdx.x == 0) num_batches[0] += 1;
    if (threadIdx.x >= 64 || c >= C) return;
    const float mu = s / M;
    const float var = fmaxf(q / M - mu * mu, 0.f);
    mean[c] = mu;
    rstd[c] = rsqrtf(var + 1e-5f);
    if (run_mean) {
        run_mean[c] = 0.9f * run_mean[c] + 0.1f * (mu + (mean_shift ? mean_shift[c] : 0.f));
        run_var[c] = 0.9f * run_var[c] + 0.1f * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
    }
}

// sums of the backward partials: sums[0..C) = sum g, sums[C..2C) = sum g*yhat; dbeta += , dgamma +=
__global__ __launch_bounds__(64 * ED_FIN_WAVES) void bn_finalize_bwd_kernel(const float* __restrict__ partial, int nrows, int C,
                                                                     float* __restrict__ sums, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta) {
    __shared__ float red[2][ED_FIN_WAVES][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    float s, q;
    ed_reduce_partials(partial, nrows, C, c, s, q, red);
    if (threadIdx.x >= 64 || c >= C) return;
    sums[c] = s;
    sums[C + c] = q;
    dbeta[c] += s;
    dgamma[c] += q;
}

// ---- data-parallel BatchNorm (SyncBN): the per-rank sums leave the chip between the two stages so that the caller can
//      all-reduce them (2*C floats per BatchNorm layer and direction).  Reference: modules.py:29,127 normalise
//      over the WHOLE batch, which under data parallelism is spread over the ranks.
// moments[0..C) = sum y, moments[C..2C) = sum y^2 over this rank's valid rows
__global__ __launch_bounds__(64 * ED_FIN_WAVES) void bn_moments_kernel(const float* __restrict__ partial, int nrows, int C, float* __restrict__ moments) {
    __shared__ float red[2][ED_FIN_WAVES][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    float s, q;
    ed_reduce_partials(partial, nrows, C, c, s, q, red);
    if (threadIdx.x >= 64 || c >= C) return;
    moments[c] = s;
    moments[C + c] = q;
}
// mean / rstd (+ running statistics) from summed moments over `count` rows (the global batch)
__global__ void bn_from_moments_kernel(const float* __restrict__ moments, float count, int C, float* __restrict__ mean, float* __restrict__ rstd,
                                       float* __restrict__ run_mean, float* __restrict__ run_var, const float* __restrict__ mean_shift,
                                       long long* __restrict__ num_batches) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (num_batches && c == 0) num_batches[0] += 1;
    if (c >= C) return;
    if (count <= 0.f) count = moments[2 * C];         // the row count travelled through the all-reduce with the sums
    const float mu = moments[c] / count;
    const float var = fmaxf(moments[C + c] / count - mu * mu, 0.f);
    mean[c] = mu;
    rstd[c] = rsqrtf(var + 1e-5f);
    if (run_mean) {
        run_mean[c] = 0.9f * run_mean[c] + 0.1f * (mu + (mean_shift ? mean_shift[c] : 0.f));
        run_var[c] = 0.9f * run_var[c] + 0.1f * var * (count / (count > 1.f ? count - 1.f : 1.f));
    }
}

// z = dropout(act(gamma*(y-mean)*rstd + beta)) as bf16; 4 channels per thread.  y: halo (or plain) rows described by g;
// z_halo: z is a halo array of z_rows rows whose row z_lead is halo row 0 (everything outside the valid set is written as
// zero: the next convolution's taps read it); else z has B*L plain rows.  The dropout counter is the element index in y.
__global__ __launch_bounds__(ED_THREADS) void bn_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int act, uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                                uint32_t thresh, float dscale, EdHalo g, int z_halo, int z_lead, size_t n4, int C,
                                                                int cshift, bf16_t* __restrict__ z) {
    if (seed_dev) seed += seed_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        unsigned zrow;
        int c;
        ed_row_col((unsigned)i * 4u, C, cshift, zrow, c);
        const long long zr = (long long)zrow;
        long long ym;
        bool ok = true;
        if (z_halo) {
            int b, t;
            ym = zr - z_lead;
            ok = ed_valid(g, ym, b, t);
        } else {
            const long long b = zr / g.L;
            ym = b * g.P + g.H + (zr - b * g.L);
        }
        uint2 pk = make_uint2(0u, 0u);
        if (ok) {
            const size_t e = (size_t)ym * C + c;
            const float4 yv = *reinterpret_cast<const float4*>(y + e);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
            // the four per-channel vectors as 16-byte loads: scalar loads made this kernel 17 memory instructions per 16 bytes of y
            const float4 m4 = *reinterpret_cast<const float4*>(mean + c), r4 = *reinterpret_cast<const float4*>(rstd + c);
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
            const float mu[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
            const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pre = __builtin_fmaf((yy[j] - mu[j]) * rs[j], ga[j], be[j]);
                float a = act == 1 ? fmaxf(pre, 0.f) : ed_tanh(pre);
                if (thresh) a *= ed_drop(seed, (uint32_t)(e + j), thresh, dscale);
                o[j] = a;
            }
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
        }
        reinterpret_cast<uint2*>(z)[i] = pk;
    }
}

// dy = gamma*rstd * (g - mean(g) - yhat*mean(g*yhat))  as bf16, in y's layout (dy_rows rows, halo row 0 at row dy_lead, zero
// outside the valid set: the transposed convolution and the weight gradient read it with shifted rows); dz halo or plain
__global__ __launch_bounds__(ED_THREADS) void bn_act_bwd_apply_kernel(const float* __restrict__ y, const bf16_t* __restrict__ dz,
                                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                      uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale,
                                                                      const float* __restrict__ sums, float inv_m, EdHalo g, int dz_halo, int dy_lead,
                                                                      size_t n4, int C, int cshift, bf16_t* __restrict__ dy) {
    if (seed_dev) seed += seed_dev[0];
    if (inv_m < 0.f) inv_m = 1.f / sums[2 * C];       // data-parallel form: the global row count came through the all-reduce
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        unsigned drow_;
        int c;
        ed_row_col((unsigned)i * 4u, C, cshift, drow_, c);
        const long long ym = (long long)drow_ - dy_lead;
        int b, t;
        uint2 pk = make_uint2(0u, 0u);
        if (ed_valid(g, ym, b, t)) {
            const size_t e = (size_t)ym * C + c;
            const float4 yv = *reinterpret_cast<const float4*>(y + e);
            const size_t drow = dz_halo ? (size_t)ym : (size_t)b * g.L + t;
            const uint2 dv = *reinterpret_cast<const uint2*>(dz + drow * C + c);
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
            const float dd[4] = {__uint_as_float(dv.x << 16), __uint_as_float(dv.x & 0xffff0000u), __uint_as_float(dv.y << 16),
                                 __uint_as_float(dv.y & 0xffff0000u)};
            const float4 m4 = *reinterpret_cast<const float4*>(mean + c), r4 = *reinterpret_cast<const float4*>(rstd + c);
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
            const float4 s0 = *reinterpret_cast<const float4*>(sums + c), s1 = *reinterpret_cast<const float4*>(sums + C + c);
            const float mu4[4] = {m4.x, m4.y, m4.z, m4.w}, rs4[4] = {r4.x, r4.y, r4.z, r4.w};
            const float ga4[4] = {g4.x, g4.y, g4.z, g4.w}, be4[4] = {b4.x, b4.y, b4.z, b4.w};
            const float sa4[4] = {s0.x, s0.y, s0.z, s0.w}, sb4[4] = {s1.x, s1.y, s1.z, s1.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rs = rs4[j], ga = ga4[j];
                const float yh = (yy[j] - mu4[j]) * rs;
                const float pre = __builtin_fmaf(yh, ga, be4[j]);
                float da;
                if (act == 1) da = pre > 0.f ? 1.f : 0.f;
                else { const float th = ed_tanh(pre); da = 1.f - th * th; }
                const float gg = dd[j] * da * (thresh ? ed_drop(seed, (uint32_t)(e + j), thresh, dscale) : 1.f);
                o[j] = ga * rs * (gg - sa4[j] * inv_m - yh * sb4[j] * inv_m);
            }
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
        }
        reinterpret_cast<uint2*>(dy)[i] = pk;
    }
}

// ------------------------------------------------------------------ TTS loss: values and gradients in one pass
// raw, post (B,L,NM) f32 strided; tgt, mask (B,L,NM) f32 contiguous; stop logits (B,L) stride; tstop (B,L)
// kind 0: MSE, 1: L1.  out[0..2] partial sums per block -> finalize.  Gradients of
//   w_raw*mean(loss(raw*mask, tgt)) + w_post*mean(loss(post*mask, tgt)) + w_stop*mean(BCE(stop, tstop; pos_weight))
__global__ __launch_bounds__(ED_THREADS) void tts_loss_kernel(const float* __restrict__ raw, const float* __restrict__ post, int64_t ld_mel,
                                                              const float* __restrict__ tgt, const float* __restrict__ mask,
                                                              const float* __restrict__ stop, int64_t ld_stop, const float* __restrict__ tstop,
                                                              int rows, int NM, int kind, float pos_weight, float w_raw, float w_post,
                                                              float w_stop, float* __restrict__ d_raw, float* __restrict__ d_post,
                                                              int64_t ld_grad, float* __restrict__ d_stop, float* __restrict__ partial,
                                                              int Lp, int Lv, const float* __restrict__ res, int64_t ld_res, int hp, int dp_lead,
                                                              long long dp_rows, long long tgt_bs, const int* __restrict__ lv_dev,
                                                              long long mask_bs, long long tstop_bs, int vec4) {
    // lv_dev: the loss length as a DEVICE word (a replayed hipGraph serves batches of any length up to the buffers' Lv: the
    // host value is then only the layout of tgt / mask / tstop, whose batch strides are tgt_bs / mask_bs / tstop_bs)
    if (lv_dev) Lv = min(max(lv_dev[0], 1), Lv);
    // res != NULL: the postnet prediction is raw + res, res in halo rows (halo hp) -- the residual add of
    // reformer_tts.py:139-140 happens here; d_post is then written in the same halo rows (dp_rows rows, halo row 0 at row
    // dp_lead, zero outside the valid set): it is the output gradient of the last convolution.
    const int PH = Lp + 2 * hp;
    // predictions / gradients: rows = B*Lp (the decoder's padded length); targets: B*Lv rows (the batch's own length,
    // reformer_tts.py:141-143 crops the predictions to it).  Rows t >= Lv of a sample get zero gradient and no loss.
    float s_raw = 0.f, s_post = 0.f, s_stop = 0.f;
    const size_t nel = (size_t)rows * NM;
    const size_t vrows = (size_t)(rows / Lp) * Lv;
    const float inv_el = 1.f / ((float)vrows * NM), inv_rows = 1.f / (float)vrows;
    if (vec4) {
        // four channels per thread (NM % 4 == 0, 16-byte aligned rows: checked by the launcher): one element per thread cost a
        // 64-bit division by NM = 80 and four scalar loads per element -- 17.7 us for 25 MB
        const unsigned NQ = (unsigned)NM / 4, nq = (unsigned)rows * NQ;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += gridDim.x * blockDim.x) {
            const unsigned row = i / NQ, c = (i % NQ) * 4;
            const unsigned t = row % (unsigned)Lp, bi = row / (unsigned)Lp;
            const size_t hrow = (size_t)bi * PH + hp + t;
            const size_t prow = (size_t)dp_lead + hrow;
            float4 gr = make_float4(0.f, 0.f, 0.f, 0.f), gp = gr;
            if ((int)t < Lv) {
                const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)bi * mask_bs + (size_t)t * NM + c);
                const float4 tg = *reinterpret_cast<const float4*>(tgt + (size_t)bi * tgt_bs + (size_t)t * NM + c);
                const float4 rv = *reinterpret_cast<const float4*>(raw + (size_t)row * ld_mel + c);
                float4 pv;
                if (res) {
                    const float4 rs = *reinterpret_cast<const float4*>(res + hrow * ld_res + c);
                    pv = make_float4(rv.x + rs.x, rv.y + rs.y, rv.z + rs.z, rv.w + rs.w);
                } else {
                    pv = *reinterpret_cast<const float4*>(post + (size_t)row * ld_mel + c);
                }
                const float mkv[4] = {mk.x, mk.y, mk.z, mk.w}, tgv[4] = {tg.x, tg.y, tg.z, tg.w};
                const float rvv[4] = {rv.x, rv.y, rv.z, rv.w}, pvv[4] = {pv.x, pv.y, pv.z, pv.w};
                float grv[4], gpv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r = rvv[e] * mkv[e] - tgv[e], p = pvv[e] * mkv[e] - tgv[e];
                    if (kind == 0) {
                        s_raw = __builtin_fmaf(r, r, s_raw);
                        s_post = __builtin_fmaf(p, p, s_post);
                        grv[e] = w_raw * 2.f * r * mkv[e] * inv_el;
                        gpv[e] = w_post * 2.f * p * mkv[e] * inv_el;
                    } else {
                        s_raw += fabsf(r);
                        s_post += fabsf(p);
                        grv[e] = w_raw * (r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f)) * mkv[e] * inv_el;
                        gpv[e] = w_post * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f)) * mkv[e] * inv_el;
                    }
                }
                gr = make_float4(grv[0], grv[1], grv[2], grv[3]);
                gp = make_float4(gpv[0], gpv[1], gpv[2], gpv[3]);
            }
            *reinterpret_cast<float4*>(d_raw + (size_t)row * ld_grad + c) = gr;
            *reinterpret_cast<float4*>(d_post + prow * ld_grad + c) = gp;
        }
    } else
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nel; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / NM;
        const int c = (int)(i % NM);
        const int t = (int)(row % Lp);
        const size_t hrow = (row / Lp) * PH + hp + t;             // the same row in halo coordinates (== row when hp == 0)
        const size_t prow = (size_t)dp_lead + hrow;
        if (t >= Lv) {
            d_raw[row * ld_grad + c] = 0.f;
            d_post[prow * ld_grad + c] = 0.f;
            continue;
        }
        const float mk = mask[(row / Lp) * (size_t)mask_bs + (size_t)t * NM + c], tg = tgt[(row / Lp) * (size_t)tgt_bs + (size_t)t * NM + c];   // tgt: a time-sliced view of the batch
        const float rv = raw[row * ld_mel + c];
        const float pv = res ? rv + res[hrow * ld_res + c] : post[row * ld_mel + c];
        const float r = rv * mk - tg, p = pv * mk - tg;
        if (kind == 0) {
            s_raw = __builtin_fmaf(r, r, s_raw);
            s_post = __builtin_fmaf(p, p, s_post);
            d_raw[row * ld_grad + c] = w_raw * 2.f * r * mk * inv_el;
            d_post[prow * ld_grad + c] = w_post * 2.f * p * mk * inv_el;
        } else {
            s_raw += fabsf(r);
            s_post += fabsf(p);
            d_raw[row * ld_grad + c] = w_raw * (r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f)) * mk * inv_el;
            d_post[prow * ld_grad + c] = w_post * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f)) * mk * inv_el;
        }
    }
    // columns NM .. ld_grad of the gradient rows (padding of a 128-wide layout) are zeroed here
    const int npad = (int)ld_grad - NM;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * npad; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / npad;
        const int c = NM + (int)(i % npad);
        d_raw[row * ld_grad + c] = 0.f;
        d_post[((size_t)dp_lead + (row / Lp) * PH + hp + row % Lp) * ld_grad + c] = 0.f;
    }
    if (hp > 0) {       // halo rows, lead-in and tail of d_post
        const EdHalo g{rows / Lp, Lp, PH, hp};
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)dp_rows * ld_grad; i += (size_t)gridDim.x * blockDim.x) {
            int b, t;
            if (!ed_valid(g, (long long)(i / ld_grad) - dp_lead, b, t)) d_post[i] = 0.f;
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows; i += (size_t)gridDim.x * blockDim.x) {
        const int tt = (int)(i % Lp);
        if (tt >= Lv) {
            d_stop[i] = 0.f;
            continue;
        }
        const float x = stop[i * ld_stop], t = tstop[(i / Lp) * (size_t)tstop_bs + tt];
        // BCE with logits, pos_weight pw:  (1-t) x + (1 + (pw-1) t) * softplus(-x)
        const float lw = 1.f + (pos_weight - 1.f) * t;
        const float sp = fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));      // softplus(-x)
        s_stop += (1.f - t) * x + lw * sp;
        const float sig = 1.f / (1.f + expf(-x));
        d_stop[i] = w_stop * ((1.f - t) - lw * (1.f - sig)) * inv_rows;
    }
    __shared__ float red[3][ED_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s_raw += __shfl_xor(s_raw, o);
        s_post += __shfl_xor(s_post, o);
        s_stop += __shfl_xor(s_stop, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s_raw;
        red[1][threadIdx.x >> 6] = s_post;
        red[2][threadIdx.x >> 6] = s_stop;
    }
    __syncthreads();
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// losses[0..3] = total, raw, post, stop
__global__ void tts_loss_finalize_kernel(const float* __restrict__ partial, int nblocks, float inv_el, float inv_rows, float w_raw,
                                         float w_post, float w_stop, float* __restrict__ losses, const int* __restrict__ lv_dev, int batch,
                                         int Lv, int NM) {
    if (lv_dev) {
        const float vrows = (float)batch * (float)min(max(lv_dev[0], 1), Lv);
        inv_el = 1.f / (vrows * NM);
        inv_rows = 1.f / vrows;
    }
    // one wave; lane l sums blocks l, l+64, ... in order, then a fixed butterfly: deterministic
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) { a += partial[i * 3]; b += partial[i * 3 + 1]; c += partial[i * 3 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
    }
    if (threadIdx.x != 0) return;
    a *= inv_el; b *= inv_el; c *= inv_rows;
    losses[0] = w_raw * a + w_post * b + w_stop * c;
    losses[1] = a; losses[2] = b; losses[3] = c;
}

// ------------------------------------------------------------------ gradient of the [mel | stop] heads
// dheads (M, W) fp32 = d_raw (plain rows) + d_post (halo rows) + dx0 (halo rows; columns < NM only: the gradient that came back
// through the postnet's first convolution), column NM = d_stop  (reformer_tts.py:65-66,139-140 backward).  W = 128.
__global__ __launch_bounds__(ED_THREADS) void heads_grad_kernel(const float* __restrict__ d_raw, const float* __restrict__ d_post, int dp_lead,
                                                                const float* __restrict__ dx0, int64_t ld_dx0, const float* __restrict__ d_stop,
                                                                EdHalo g, int NM, int W, float* __restrict__ out,
                                                                const float* __restrict__ scale_dev) {
    // scale_dev: the upstream scalar gradient of the total loss; d_raw, d_post and d_stop were stored for an upstream of 1, dx0
    // came back through the postnet from the already scaled d_post
    const float up = scale_dev ? scale_dev[0] : 1.f;
    const size_t total = (size_t)g.B * g.L * (W / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (W / 4)) * 4;
        const size_t row = i / (W / 4);
        const size_t hrow = (row / g.L) * g.P + g.H + row % g.L;
        const float4 a = *reinterpret_cast<const float4*>(d_raw + row * W + c);
        const float4 b = *reinterpret_cast<const float4*>(d_post + ((size_t)dp_lead + hrow) * W + c);
        float o[4] = {(a.x + b.x) * up, (a.y + b.y) * up, (a.z + b.z) * up, (a.w + b.w) * up};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c + j < NM) o[j] += dx0[hrow * ld_dx0 + c + j];
            else if (c + j == NM) o[j] = d_stop[row] * up;
        }
        *reinterpret_cast<float4*>(out + row * W + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------ embedding backward (padding_idx rows get no gradient)
// dE[id] += sum over rows with ids[row] == id of dx[row]: one block per (id, 64-channel group).  Wave w scans the
// 64-row groups w, w+4, ...: the lanes compare 64 ids at once, the ballot lists the matching rows and they are added
// in row order; the 4 wave sums are combined in wave order => deterministic, no atomics
// dx row rr = (b, t), b = rr / L, t = rr % L, lives at dx + b * bstride + t * rstride (floats): the gradient of the convolution
// stack arrives as a strided view of its halo rows (edges.Halo.valid) and is read in place instead of through a contiguous copy
__global__ __launch_bounds__(ED_THREADS) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx, int64_t bstride,
                                                                   int64_t rstride, int L, int rows,
                                                                   int C, int padding_idx, float* __restrict__ dE, uint32_t seed,
                                                                   const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale) {
    __shared__ float red[4][64];
    const int id = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    if (id == padding_idx) return;
    if (thresh && seed_dev) seed += seed_dev[0];
    float acc = 0.f;
    for (int base = wave * 64; base < rows; base += 4 * 64) {
        const int r = base + lane;
        unsigned long long m = __ballot(r < rows && ids[r] == (int64_t)id);
        while (m) {
            const int rr = base + __builtin_ctzll(m);
            m &= m - 1;
            const int bb = rr / L;
            if (c < C) acc += dx[(size_t)bb * bstride + (size_t)(rr - bb * L) * rstride + c] * (thresh ? rtts_drop_keep(seed, (uint32_t)rr * C + c, thresh, dscale) : 1.f);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < C) dE[(size_t)id * C + c] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// out[row] = dropout_p(E[ids[row]]) (fp32; nn.Embedding + the Dropout in front of the encoder prenet's convolutions,
// modules.py:17,22,56): 4 channels per thread, the keep decision of element (row, c) is hash(seed, row * C + c)
__global__ __launch_bounds__(ED_THREADS) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ E, size_t n4, int C,
                                                                   int n_emb, uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                                   float dscale, float* __restrict__ out) {
    if (thresh && seed_dev) seed += seed_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4, row = e / C;
        const int c = (int)(e % C);
        long long id = ids[row];
        id = id < 0 ? 0 : (id >= n_emb ? n_emb - 1 : id);          // ids are validated on the host side of the data pipeline
        float4 v = *reinterpret_cast<const float4*>(E + (size_t)id * C + c);
        if (thresh) {
            v.x *= rtts_drop_keep(seed, (uint32_t)e, thresh, dscale);
            v.y *= rtts_drop_keep(seed, (uint32_t)e + 1, thresh, dscale);
            v.z *= rtts_drop_keep(seed, (uint32_t)e + 2, thresh, dscale);
            v.w *= rtts_drop_keep(seed, (uint32_t)e + 3, thresh, dscale);
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// ------------------------------------------------------------------ small copies / adds of a step in ONE launch
// Up to RTTS_SEGMENTS_MAX (dst, src, count) jobs: the per-step weight refreshes of padded GEMM operands (heads, padded biases,
// K-padded weights) and the additions of padded gradient blocks into their parameters' gradients, which each cost a
// ~5 us launch as separate copy_/add_ calls whatever their size.
struct SgJob {
    void* dst;
    const void* src;
    long long count;
    int kind, blk_start;
};
struct SgGroup {
    SgJob j[RTTS_SEGMENTS_MAX];
    int n;
};
__global__ __launch_bounds__(ED_THREADS) void segments_kernel(const SgGroup grp) {
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.j[i].blk_start) ji = i;
    const SgJob& J = grp.j[ji];
    const int nblk = (ji + 1 < grp.n ? grp.j[ji + 1].blk_start : (int)gridDim.x) - J.blk_start;
    for (long long i = (long long)((int)blockIdx.x - J.blk_start) * ED_THREADS + threadIdx.x; i < J.count; i += (long long)nblk * ED_THREADS) {
        switch (J.kind) {
        case RTTS_SEG_COPY_F32: reinterpret_cast<float*>(J.dst)[i] = reinterpret_cast<const float*>(J.src)[i]; break;
        case RTTS_SEG_COPY_BF16: reinterpret_cast<bf16_t*>(J.dst)[i] = reinterpret_cast<const bf16_t*>(J.src)[i]; break;
        case RTTS_SEG_ADD_F32: reinterpret_cast<float*>(J.dst)[i] += reinterpret_cast<const float*>(J.src)[i]; break;
        default: reinterpret_cast<bf16_t*>(J.dst)[i] = f32_to_bf16(reinterpret_cast<const float*>(J.src)[i]); break;
        }
    }
}

// ------------------------------------------------------------------ scaled positional encoding (modules.py:172-192)
// out[m][c] = y[m][c] + alpha * dropout_p(table[m % T][c]); the mask depends on (t, c) only: shared over the batch
__global__ __launch_bounds__(ED_THREADS) void pe_add_kernel(const bf16_t* __restrict__ y, const float* __restrict__ table,
                                                            const float* __restrict__ alpha, uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                            uint32_t thresh, float dscale, int T, size_t n4, int d, float* __restrict__ out) {
    if (seed_dev) seed += seed_dev[0];
    const float a = alpha[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % d);
        const int t = (int)((e / d) % T);
        const uint2 yv = reinterpret_cast<const uint2*>(y)[i];
        const float4 tv = *reinterpret_cast<const float4*>(table + (size_t)t * d + c);
        const float tt[4] = {tv.x, tv.y, tv.z, tv.w};
        const float yy[4] = {__uint_as_float(yv.x << 16), __uint_as_float(yv.x & 0xffff0000u), __uint_as_float(yv.y << 16),
                             __uint_as_float(yv.y & 0xffff0000u)};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = yy[j] + a * tt[j] * (thresh ? ed_drop(seed, (uint32_t)(t * d + c + j), thresh, dscale) : 1.f);
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// partial[blk] = sum over this block's elements of dy * dropout(table)
__global__ __launch_bounds__(ED_THREADS) void pe_dalpha_partial_kernel(const float* __restrict__ dy, const float* __restrict__ table,
                                                                       uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale, int T,
                                                                       size_t n4, int d, float* __restrict__ partial) {
    if (seed_dev) seed += seed_dev[0];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = (int)(e % d);
        const int t = (int)((e / d) % T);
        const float4 dv = reinterpret_cast<const float4*>(dy)[i];
        const float4 tv = *reinterpret_cast<const float4*>(table + (size_t)t * d + c);
        const float dd[4] = {dv.x, dv.y, dv.z, dv.w}, tt[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s = __builtin_fmaf(dd[j], tt[j] * (thresh ? ed_drop(seed, (uint32_t)(t * d + c + j), thresh, dscale) : 1.f), s);
    }
    __shared__ float red[ED_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void pe_dalpha_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ dalpha) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) dalpha[0] += s;
}

// h = dropout_p(relu(h)) in place (bf16), 8 elements per thread; the backward gate is (h_out > 0) * 1/(1-p)
__global__ __launch_bounds__(ED_THREADS) void relu_drop_kernel(bf16_t* __restrict__ h, uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                               uint32_t thresh, float dscale, size_t n8) {
    if (seed_dev) seed += seed_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        uint4 t = reinterpret_cast<uint4*>(h)[i];
        uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo = fmaxf(__uint_as_float(u[j] << 16), 0.f), hi = fmaxf(__uint_as_float(u[j] & 0xffff0000u), 0.f);
            if (thresh) {
                lo *= ed_drop(seed, (uint32_t)(i * 8 + 2 * j), thresh, dscale);
                hi *= ed_drop(seed, (uint32_t)(i * 8 + 2 * j + 1), thresh, dscale);
            }
            u[j] = pack_bf16x2(lo, hi);
        }
        t.x = u[0]; t.y = u[1]; t.z = u[2]; t.w = u[3];
        reinterpret_cast<uint4*>(h)[i] = t;
    }
}

// ------------------------------------------------------------------ host side
static inline unsigned ed_grid(size_t items) {
    size_t b = (items + ED_THREADS - 1) / ED_THREADS;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}
static inline uint32_t ed_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)((double)p * 4294967296.0); }

extern "C" int rtts_to_halo(const void* src, int64_t ld_src, int64_t src_batch_stride, int C_src, int src_f32, int B, int L, int halo, int C,
                            void* dst, int lead, int64_t rows, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(src && dst && B > 0 && L > 0 && halo >= 0 && C > 0 && C % 8 == 0 && C_src > 0 && C_src % 8 == 0 && C_src <= C && ld_src >= C_src &&
                     ld_src % (src_f32 ? 4 : 8) == 0, "rtts_to_halo: bad arguments");
    if (src_batch_stride == 0) src_batch_stride = (int64_t)L * ld_src;
    RTTS_REQUIRE(src_batch_stride >= (int64_t)L * ld_src && src_batch_stride % (src_f32 ? 4 : 8) == 0,
                 "rtts_to_halo: src_batch_stride must cover L rows and keep 16-byte alignment");
    RTTS_REQUIRE(lead >= 0 && rows >= lead + (int64_t)B * (L + 2 * halo), "rtts_to_halo: dst has %lld rows, needs %lld", (long long)rows,
                 (long long)(lead + (int64_t)B * (L + 2 * halo)));
    const dim3 grid(ed_grid((size_t)rows * (C / 8)));
    if (src_f32)
        hipLaunchKernelGGL(to_halo_kernel<true>, grid, dim3(ED_THREADS), 0, (hipStream_t)stream, src, ld_src, src_batch_stride, C_src, ed_halo(B, L, halo), C,
                           lead, (long long)rows, (bf16_t*)dst);
    else
        hipLaunchKernelGGL(to_halo_kernel<false>, grid, dim3(ED_THREADS), 0, (hipStream_t)stream, src, ld_src, src_batch_stride, C_src, ed_halo(B, L, halo), C,
                           lead, (long long)rows, (bf16_t*)dst);
    RTTS_LAUNCH_CHECK("rtts_to_halo");
    return 0;
}

extern "C" int rtts_heads_grad(const float* d_raw, const float* d_post, int dpost_lead, const float* dx0, int64_t ld_dx0, const float* d_stop,
                               int B, int L, int halo, int n_mels, int width, float* dheads, const float* scale_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(d_raw && d_post && dx0 && d_stop && dheads && B > 0 && L > 0 && halo >= 0 && width % 4 == 0 && n_mels < width && dpost_lead >= 0,
                 "rtts_heads_grad: bad arguments");
    hipLaunchKernelGGL(heads_grad_kernel, dim3(ed_grid((size_t)B * L * (width / 4))), dim3(ED_THREADS), 0, (hipStream_t)stream, d_raw, d_post,
                       dpost_lead, dx0, ld_dx0, d_stop, ed_halo(B, L, halo), n_mels, width, dheads, scale_dev);
    RTTS_LAUNCH_CHECK("rtts_heads_grad");
    return 0;
}

extern "C" int rtts_conv_w_perm(const float* w, int Co, int Ci, int CP, void* wp, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(w && wp && Co > 0 && Ci > 0 && CP >= Ci, "rtts_conv_w_perm: bad arguments");
    hipLaunchKernelGGL(conv_w_perm_kernel, dim3(ed_grid((size_t)Co * 5 * CP)), dim3(ED_THREADS), 0, (hipStream_t)stream, w, Co, Ci, CP, (bf16_t*)wp);
    RTTS_LAUNCH_CHECK("rtts_conv_w_perm");
    return 0;
}

extern "C" int rtts_conv_w_perm_grouped(const rtts_conv_perm_job* jobs, int n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(jobs && n > 0 && n <= RTTS_CONV_PERM_MAX_GROUP, "rtts_conv_w_perm_grouped: 1..%d jobs", RTTS_CONV_PERM_MAX_GROUP);
    CwGroup g;
    g.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        RTTS_REQUIRE(jobs[i].w && jobs[i].wp && jobs[i].Co > 0 && jobs[i].Ci > 0 && jobs[i].CP >= jobs[i].Ci, "rtts_conv_w_perm_grouped: bad job %d", i);
        g.j[i] = jobs[i];
        g.blk_start[i] = blk;
        blk += (int)ed_grid((size_t)jobs[i].Co * jobs[i].CP);
    }
    for (int i = n; i <= RTTS_CONV_PERM_MAX_GROUP; ++i) g.blk_start[i] = blk;
    hipLaunchKernelGGL(conv_w_perm_grouped_kernel, dim3(blk), dim3(ED_THREADS), 0, (hipStream_t)stream, g);
    RTTS_LAUNCH_CHECK("rtts_conv_w_perm_grouped");
    return 0;
}

extern "C" int rtts_conv_dw_unperm_grouped(const rtts_conv_perm_job* jobs, int n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(jobs && n > 0 && n <= RTTS_CONV_PERM_MAX_GROUP, "rtts_conv_dw_unperm_grouped: 1..%d jobs", RTTS_CONV_PERM_MAX_GROUP);
    CwGroup g;
    g.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        RTTS_REQUIRE(jobs[i].w && jobs[i].wp && jobs[i].Co > 0 && jobs[i].Ci > 0 && jobs[i].CP >= jobs[i].Ci, "rtts_conv_dw_unperm_grouped: bad job %d", i);
        g.j[i] = jobs[i];
        g.blk_start[i] = blk;
        blk += (int)ed_grid((size_t)jobs[i].Co * jobs[i].Ci);
    }
    for (int i = n; i <= RTTS_CONV_PERM_MAX_GROUP; ++i) g.blk_start[i] = blk;
    hipLaunchKernelGGL(conv_dw_unperm_grouped_kernel, dim3(blk), dim3(ED_THREADS), 0, (hipStream_t)stream, g);
    RTTS_LAUNCH_CHECK("rtts_conv_dw_unperm_grouped");
    return 0;
}

extern "C" int rtts_conv_dw_unperm(const float* dwp, int Co, int Ci, int CP, float* dw, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dwp && dw && Co > 0 && Ci > 0 && CP >= Ci, "rtts_conv_dw_unperm: bad arguments");
    hipLaunchKernelGGL(conv_dw_unperm_kernel, dim3(ed_grid((size_t)Co * Ci * 5)), dim3(ED_THREADS), 0, (hipStream_t)stream, dwp, Co, Ci, CP, dw);
    RTTS_LAUNCH_CHECK("rtts_conv_dw_unperm");
    return 0;
}

static inline dim3 ed_col_grid(int M, int C) {
    int slabs = (M + 3) / 4;
    if (slabs > ED_PBLOCKS) slabs = ED_PBLOCKS;
    return dim3(slabs, (C + 255) / 256);
}

extern "C" int rtts_bn_stats(const float* y, int B, int L, int halo, int C, float* mean, float* rstd, float* run_mean, float* run_var,
                             const float* mean_shift, int64_t* num_batches, float* partial_ws, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && mean && rstd && partial_ws && B > 0 && L > 0 && halo >= 0 && C > 0 && C % 4 == 0, "rtts_bn_stats: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    const dim3 grid = ed_col_grid(B * g.P, C);
    hipLaunchKernelGGL((col_partial_kernel<0>), grid, dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0u, (const uint32_t*)nullptr, 0u, 1.f, g, 1, C,
                       partial_ws);
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((C + 63) / 64), dim3(64 * ED_FIN_WAVES), 0, (hipStream_t)stream, partial_ws,
                       (int)grid.x, B * L, C, mean, rstd, run_mean, run_var, mean_shift, (long long*)num_batches);
    RTTS_LAUNCH_CHECK("rtts_bn_stats");
    return 0;
}

// the second half of rtts_bn_stats on partial rows somebody else produced (rtts_conv1d_k5_moments: the convolution's own epilogue)
extern "C" int rtts_bn_stats_from_partials(const float* partial, int nrows, int B, int L, int C, float* mean, float* rstd, float* run_mean,
                                           float* run_var, const float* mean_shift, int64_t* num_batches, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(partial && mean && rstd && nrows > 0 && B > 0 && L > 0 && C > 0 && C % 4 == 0, "rtts_bn_stats_from_partials: bad arguments");
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((C + 63) / 64), dim3(64 * ED_FIN_WAVES), 0, (hipStream_t)stream, partial, nrows, B * L, C, mean,
                       rstd, run_mean, run_var, mean_shift, (long long*)num_batches);
    RTTS_LAUNCH_CHECK("rtts_bn_stats_from_partials");
    return 0;
}

extern "C" int rtts_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                               float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C, void* z, int z_halo,
                               int z_lead, int64_t z_rows, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && mean && rstd && gamma && beta && z && B > 0 && L > 0 && halo >= 0 && C % 4 == 0 && (act == 1 || act == 2) && drop_p >= 0.f &&
                     drop_p < 1.f, "rtts_bn_act_fwd: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    RTTS_REQUIRE(z_halo ? (halo > 0 && z_lead >= 0 && z_rows >= z_lead + (int64_t)B * g.P) : (z_rows == (int64_t)B * L && z_lead == 0),
                 "rtts_bn_act_fwd: z must be a halo array with room for B*(L+2*halo) rows behind its lead-in, or B*L plain rows");
    const size_t n4 = (size_t)z_rows * C / 4;
    RTTS_REQUIRE(n4 < (1ull << 30), "rtts_bn_act_fwd: more than 2^32 elements");
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ed_grid(n4)), dim3(ED_THREADS), 0, (hipStream_t)stream, y, mean, rstd, gamma, beta, act, seed, seed_dev,
                       ed_thresh(drop_p), 1.f / (1.f - drop_p), g, z_halo, z_lead, n4, C, ed_cshift(C), (bf16_t*)z);
    RTTS_LAUNCH_CHECK("rtts_bn_act_fwd");
    return 0;
}

extern "C" int rtts_bn_act_bwd(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C,
                               void* dy, int dy_lead, int64_t dy_rows, float* dgamma, float* dbeta, float* partial_ws, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && dz && mean && rstd && gamma && beta && dy && dgamma && dbeta && partial_ws && B > 0 && L > 0 && halo >= 0 && C % 4 == 0 &&
                     (act == 1 || act == 2), "rtts_bn_act_bwd: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    RTTS_REQUIRE(dy_lead >= 0 && dy_rows >= dy_lead + (int64_t)B * g.P && (halo > 0 || dz_halo == 0 || true),
                 "rtts_bn_act_bwd: dy needs room for B*(L+2*halo) rows behind its lead-in");
    const dim3 grid = ed_col_grid(B * g.P, C);
    const uint32_t th = ed_thresh(drop_p);
    const float ds = 1.f / (1.f - drop_p);
    float* sums = partial_ws + (size_t)ED_PBLOCKS * 2 * C;
    const int dzh = (halo > 0 && dz_halo) ? 1 : 0;
    hipLaunchKernelGGL((col_partial_kernel<1>), grid, dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)dz, mean, rstd, gamma, beta, act,
                       seed, seed_dev, th, ds, g, dzh, C, partial_ws);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 63) / 64), dim3(64 * ED_FIN_WAVES), 0, (hipStream_t)stream, partial_ws,
                       (int)grid.x, C, sums, dgamma, dbeta);
    const size_t n4 = (size_t)dy_rows * C / 4;
    RTTS_REQUIRE(n4 < (1ull << 30), "rtts_bn_act_bwd: more than 2^32 elements");
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ed_grid(n4)), dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)dz, mean, rstd,
                       gamma, beta, act, seed, seed_dev, th, ds, sums, 1.f / (float)((size_t)B * L), g, dzh, dy_lead, n4, C, ed_cshift(C), (bf16_t*)dy);
    RTTS_LAUNCH_CHECK("rtts_bn_act_bwd");
    return 0;
}

extern "C" int rtts_bn_moments(const float* y, int B, int L, int halo, int C, float* moments, float* partial_ws, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && moments && partial_ws && B > 0 && L > 0 && halo >= 0 && C > 0 && C % 4 == 0, "rtts_bn_moments: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    const dim3 grid = ed_col_grid(B * g.P, C);
    hipLaunchKernelGGL((col_partial_kernel<0>), grid, dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0u, (const uint32_t*)nullptr, 0u, 1.f, g, 1, C,
                       partial_ws);
    hipLaunchKernelGGL(bn_moments_kernel, dim3((C + 63) / 64), dim3(64 * ED_FIN_WAVES), 0, (hipStream_t)stream, partial_ws, (int)grid.x, C, moments);
    RTTS_LAUNCH_CHECK("rtts_bn_moments");
    return 0;
}

extern "C" int rtts_bn_from_moments(const float* moments, int64_t count, int C, float* mean, float* rstd, float* run_mean, float* run_var,
                                    const float* mean_shift, int64_t* num_batches, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(moments && mean && rstd && count >= 0 && C > 0, "rtts_bn_from_moments: bad arguments");
    hipLaunchKernelGGL(bn_from_moments_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, moments, (float)count, C, mean, rstd,
                       run_mean, run_var, mean_shift, (long long*)num_batches);
    RTTS_LAUNCH_CHECK("rtts_bn_from_moments");
    return 0;
}

extern "C" int rtts_bn_act_bwd_sums(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo,
                                    int C, float* sums, float* dgamma, float* dbeta, float* partial_ws, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && dz && mean && rstd && gamma && beta && sums && dgamma && dbeta && partial_ws && B > 0 && L > 0 && halo >= 0 && C % 4 == 0 &&
                     (act == 1 || act == 2), "rtts_bn_act_bwd_sums: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    const dim3 grid = ed_col_grid(B * g.P, C);
    const int dzh = (halo > 0 && dz_halo) ? 1 : 0;
    hipLaunchKernelGGL((col_partial_kernel<1>), grid, dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)dz, mean, rstd, gamma, beta, act,
                       seed, seed_dev, ed_thresh(drop_p), 1.f / (1.f - drop_p), g, dzh, C, partial_ws);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 63) / 64), dim3(64 * ED_FIN_WAVES), 0, (hipStream_t)stream, partial_ws,
                       (int)grid.x, C, sums, dgamma, dbeta);
    RTTS_LAUNCH_CHECK("rtts_bn_act_bwd_sums");
    return 0;
}

extern "C" int rtts_bn_act_bwd_apply(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                                     const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo,
                                     int C, const float* sums, int64_t count, void* dy, int dy_lead, int64_t dy_rows, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && dz && mean && rstd && gamma && beta && sums && dy && count >= 0 && B > 0 && L > 0 && halo >= 0 && C % 4 == 0 &&
                     (act == 1 || act == 2), "rtts_bn_act_bwd_apply: bad arguments");
    const EdHalo g = ed_halo(B, L, halo);
    RTTS_REQUIRE(dy_lead >= 0 && dy_rows >= dy_lead + (int64_t)B * g.P, "rtts_bn_act_bwd_apply: dy needs room for B*(L+2*halo) rows behind its lead-in");
    const int dzh = (halo > 0 && dz_halo) ? 1 : 0;
    const size_t n4 = (size_t)dy_rows * C / 4;
    RTTS_REQUIRE(n4 < (1ull << 30), "rtts_bn_act_bwd_apply: more than 2^32 elements");
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ed_grid(n4)), dim3(ED_THREADS), 0, (hipStream_t)stream, y, (const bf16_t*)dz, mean, rstd,
                       gamma, beta, act, seed, seed_dev, ed_thresh(drop_p), 1.f / (1.f - drop_p), sums, count > 0 ? 1.f / (float)count : -1.f, g, dzh,
                       dy_lead, n4, C, ed_cshift(C), (bf16_t*)dy);
    RTTS_LAUNCH_CHECK("rtts_bn_act_bwd_apply");
    return 0;
}

extern "C" int rtts_tts_loss(const float* raw, const float* post, int64_t ld_mel, const float* tgt, const float* mask, const float* stop,
                             int64_t ld_stop, const float* tstop, int rows, int NM, int kind, float pos_weight, float w_raw, float w_post,
                             float w_stop, float* d_raw, float* d_post, int64_t ld_grad, float* d_stop, float* losses, float* partial_ws,
                             int padded_len, int valid_len, const float* res, int64_t ld_res, int halo, int dpost_lead, int64_t dpost_rows,
                             int64_t tgt_batch_stride, const int32_t* valid_len_dev, int64_t mask_batch_stride, int64_t tstop_batch_stride,
                             void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(raw && (post || res) && tgt && mask && stop && tstop && d_raw && d_post && d_stop && losses && partial_ws && rows > 0 && NM > 0 &&
                     ld_grad >= NM, "rtts_tts_loss: bad arguments");
    RTTS_REQUIRE(padded_len > 0 && valid_len > 0 && valid_len <= padded_len && rows % padded_len == 0,
                 "rtts_tts_loss: rows must be batch * padded_len and 0 < valid_len <= padded_len");
    RTTS_REQUIRE(kind == 0 || kind == 1, "rtts_tts_loss: Unsupported loss type: %d", kind);
    RTTS_REQUIRE(res ? (halo > 0 && ld_res >= NM && dpost_lead >= 0 &&
                        dpost_rows >= dpost_lead + (int64_t)(rows / padded_len) * (padded_len + 2 * halo))
                     : (halo == 0 && dpost_lead == 0),
                 "rtts_tts_loss: res / d_post in halo rows need halo > 0 and room for B*(L+2*halo) rows; without res: halo = lead = 0");
    if (tgt_batch_stride == 0) tgt_batch_stride = (int64_t)valid_len * NM;
    RTTS_REQUIRE(tgt_batch_stride >= (int64_t)valid_len * NM, "rtts_tts_loss: tgt_batch_stride below valid_len * NM");
    if (mask_batch_stride == 0) mask_batch_stride = (int64_t)valid_len * NM;
    if (tstop_batch_stride == 0) tstop_batch_stride = valid_len;
    RTTS_REQUIRE(mask_batch_stride >= (int64_t)valid_len * NM && tstop_batch_stride >= valid_len,
                 "rtts_tts_loss: mask / tstop batch strides below valid_len rows");
    const int blocks = 512;
    const uintptr_t bits = (uintptr_t)raw | (uintptr_t)post | (uintptr_t)res | (uintptr_t)tgt | (uintptr_t)mask | (uintptr_t)d_raw | (uintptr_t)d_post;
    const int vec4 = NM % 4 == 0 && (bits & 15) == 0 && ld_mel % 4 == 0 && ld_grad % 4 == 0 && (!res || ld_res % 4 == 0) &&
                     tgt_batch_stride % 4 == 0 && mask_batch_stride % 4 == 0 && (int64_t)rows * (NM / 4) < (1ll << 31);
    hipLaunchKernelGGL(tts_loss_kernel, dim3(blocks), dim3(ED_THREADS), 0, (hipStream_t)stream, raw, post, ld_mel, tgt, mask, stop, ld_stop,
                       tstop, rows, NM, kind, pos_weight, w_raw, w_post, w_stop, d_raw, d_post, ld_grad, d_stop, partial_ws, padded_len,
                       valid_len, res, ld_res, halo, dpost_lead, (long long)dpost_rows, (long long)tgt_batch_stride, valid_len_dev,
                       (long long)mask_batch_stride, (long long)tstop_batch_stride, vec4);
    const float vrows = (float)(rows / padded_len) * (float)valid_len;
    hipLaunchKernelGGL(tts_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, blocks, 1.f / (vrows * NM),
                       1.f / vrows, w_raw, w_post, w_stop, losses, valid_len_dev, rows / padded_len, valid_len, NM);
    RTTS_LAUNCH_CHECK("rtts_tts_loss");
    return 0;
}

extern "C" int rtts_embedding_bwd(const int64_t* ids, const float* dx, int rows, int C, int n_embeddings, int padding_idx, float* dE,
                                  float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(ids && dx && dE && rows > 0 && C > 0 && n_embeddings > 0 && drop_p >= 0.f && drop_p < 1.f, "rtts_embedding_bwd: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(n_embeddings, (C + 63) / 64), dim3(ED_THREADS), 0, (hipStream_t)stream,
                       ids, dx, (int64_t)0, (int64_t)C, rows, rows, C, padding_idx, dE, seed, seed_dev, ed_thresh(drop_p), 1.f / (1.f - drop_p));
    RTTS_LAUNCH_CHECK("rtts_embedding_bwd");
    return 0;
}

// the same with dx as a (B, L, C) view of a larger array: batch stride and row stride in floats (rows = B * L)
extern "C" int rtts_embedding_bwd_strided(const int64_t* ids, const float* dx, int64_t batch_stride, int64_t row_stride, int L, int rows, int C,
                                          int n_embeddings, int padding_idx, float* dE, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                          void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(ids && dx && dE && rows > 0 && L > 0 && rows % L == 0 && C > 0 && row_stride >= C && batch_stride >= 0 && n_embeddings > 0 &&
                     drop_p >= 0.f && drop_p < 1.f, "rtts_embedding_bwd_strided: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(n_embeddings, (C + 63) / 64), dim3(ED_THREADS), 0, (hipStream_t)stream,
                       ids, dx, batch_stride, row_stride, L, rows, C, padding_idx, dE, seed, seed_dev, ed_thresh(drop_p), 1.f / (1.f - drop_p));
    RTTS_LAUNCH_CHECK("rtts_embedding_bwd_strided");
    return 0;
}

extern "C" int rtts_embedding_fwd(const int64_t* ids, const float* E, int rows, int C, int n_embeddings, float drop_p, uint32_t seed,
                                  const uint32_t* seed_dev, float* out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(ids && E && out && rows > 0 && C > 0 && C % 4 == 0 && n_embeddings > 0 && drop_p >= 0.f && drop_p < 1.f,
                 "rtts_embedding_fwd: bad arguments");
    const size_t n4 = (size_t)rows * C / 4;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ed_grid(n4)), dim3(ED_THREADS), 0, (hipStream_t)stream, ids, E, n4, C, n_embeddings, seed,
                       seed_dev, ed_thresh(drop_p), 1.f / (1.f - drop_p), out);
    RTTS_LAUNCH_CHECK("rtts_embedding_fwd");
    return 0;
}

extern "C" int rtts_segments(const rtts_segment* jobs, int n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(jobs && n > 0 && n <= RTTS_SEGMENTS_MAX, "rtts_segments: 1..%d jobs", RTTS_SEGMENTS_MAX);
    SgGroup g;
    g.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        RTTS_REQUIRE(jobs[i].dst && jobs[i].src && jobs[i].count > 0 && jobs[i].kind >= 0 && jobs[i].kind <= RTTS_SEG_CAST_F32_BF16,
                     "rtts_segments: bad job %d", i);
        g.j[i].dst = jobs[i].dst;
        g.j[i].src = jobs[i].src;
        g.j[i].count = jobs[i].count;
        g.j[i].kind = jobs[i].kind;
        g.j[i].blk_start = blk;
        long long b = (jobs[i].count + ED_THREADS * 4 - 1) / (ED_THREADS * 4);
        blk += (int)(b > 256 ? 256 : b);
    }
    hipLaunchKernelGGL(segments_kernel, dim3(blk), dim3(ED_THREADS), 0, (hipStream_t)stream, g);
    RTTS_LAUNCH_CHECK("rtts_segments");
    return 0;
}

extern "C" int rtts_pe_add(const void* y, const float* table, const float* alpha, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                           int T, int64_t M, int d, float* out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(y && table && alpha && out && T > 0 && M > 0 && d > 0 && d % 4 == 0 && M % T == 0 && drop_p >= 0.f && drop_p < 1.f,
                 "rtts_pe_add: bad arguments");
    const size_t n4 = (size_t)M * d / 4;
    hipLaunchKernelGGL(pe_add_kernel, dim3(ed_grid(n4)), dim3(ED_THREADS), 0, (hipStream_t)stream, (const bf16_t*)y, table, alpha, seed, seed_dev,
                       ed_thresh(drop_p), 1.f / (1.f - drop_p), T, n4, d, out);
    RTTS_LAUNCH_CHECK("rtts_pe_add");
    return 0;
}

extern "C" int rtts_pe_dalpha(const float* dy, const float* table, float drop_p, uint32_t seed, const uint32_t* seed_dev, int T, int64_t M,
                              int d, float* dalpha, float* partial_ws, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dy && table && dalpha && partial_ws && T > 0 && M > 0 && d % 4 == 0 && M % T == 0, "rtts_pe_dalpha: bad arguments");
    const size_t n4 = (size_t)M * d / 4;
    const int blocks = 512;
    hipLaunchKernelGGL(pe_dalpha_partial_kernel, dim3(blocks), dim3(ED_THREADS), 0, (hipStream_t)stream, dy, table, seed, seed_dev, ed_thresh(drop_p),
                       1.f / (1.f - drop_p), T, n4, d, partial_ws);
    hipLaunchKernelGGL(pe_dalpha_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_ws, blocks, dalpha);
    RTTS_LAUNCH_CHECK("rtts_pe_dalpha");
    return 0;
}

// ---------------------------------------------------------------- what the model derives from a batch before its first layer
// reference reformer_tts.py:119-125: phonemes right-padded with 0 to a multiple of pad_base, phoneme_mask = padded != 0, the frame
// mask (here: loss_mask.mean(-1), wrappers.py:60) padded with 0 and cast to bool.  One launch instead of seven (pad = fill + copy,
// compare, mean, pad, cast, the inverted key mask the cross-attention wants).
__global__ __launch_bounds__(256) void batch_masks_kernel(const int64_t* __restrict__ phonemes, int64_t ph_stride, int B, int Lp, int Lp_pad,
                                                          const float* __restrict__ loss_mask, int64_t lm_bstride, int64_t lm_rstride, int Lm,
                                                          int Lm_pad, int n_mels, int64_t* __restrict__ pad_ph, uint8_t* __restrict__ ph_mask,
                                                          uint8_t* __restrict__ ph_not, uint8_t* __restrict__ sp_mask) {
    const int per_b = Lp_pad + Lm_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * per_b; i += gridDim.x * blockDim.x) {
        const int b = i / per_b, j = i % per_b;
        if (j < Lp_pad) {
            const int64_t v = j < Lp ? phonemes[(int64_t)b * ph_stride + j] : 0;
            pad_ph[(size_t)b * Lp_pad + j] = v;
            ph_mask[(size_t)b * Lp_pad + j] = v != 0;
            ph_not[(size_t)b * Lp_pad + j] = v == 0;
        } else {
            const int t = j - Lp_pad;
            bool on = false;
            if (t < Lm) {
                const float* row = loss_mask + (int64_t)b * lm_bstride + (int64_t)t * lm_rstride;
                float sum = 0.f;
                for (int c = 0; c < n_mels; ++c) sum += row[c];
                on = (sum / (float)n_mels) != 0.f;       // torch: mean(-1) then .to(bool)  (NaN -> true, like torch)
            }
            sp_mask[(size_t)b * Lm_pad + t] = on;
        }
    }
}

extern "C" int rtts_batch_masks(const int64_t* phonemes, int64_t ph_stride, int B, int Lp, int Lp_pad, const float* loss_mask, int64_t lm_bstride,
                                int64_t lm_rstride, int Lm, int Lm_pad, int n_mels, int64_t* pad_phonemes, uint8_t* phoneme_mask,
                                uint8_t* phoneme_pad_mask, uint8_t* frame_mask, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(phonemes && loss_mask && pad_phonemes && phoneme_mask && phoneme_pad_mask && frame_mask, "rtts_batch_masks: null pointer");
    RTTS_REQUIRE(B > 0 && Lp > 0 && Lp_pad >= Lp && Lm > 0 && Lm_pad >= Lm && n_mels > 0 && ph_stride >= Lp, "rtts_batch_masks: bad shape");
    const int n = B * (Lp_pad + Lm_pad);
    hipLaunchKernelGGL(batch_masks_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, phonemes, ph_stride, B, Lp, Lp_pad, loss_mask,
                       lm_bstride, lm_rstride, Lm, Lm_pad, n_mels, pad_phonemes, phoneme_mask, phoneme_pad_mask, frame_mask);
    RTTS_LAUNCH_CHECK("rtts_batch_masks");
    return 0;
}

extern "C" int rtts_relu_drop(void* h, float drop_p, uint32_t seed, const uint32_t* seed_dev, int64_t n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(h && n > 0 && n % 8 == 0 && drop_p >= 0.f && drop_p < 1.f, "rtts_relu_drop: bad arguments");
    hipLaunchKernelGGL(relu_drop_kernel, dim3(ed_grid((size_t)n / 8)), dim3(ED_THREADS), 0, (hipStream_t)stream, (bf16_t*)h, seed, seed_dev,
                       ed_thresh(drop_p), 1.f / (1.f - drop_p), (size_t)n / 8);
    RTTS_LAUNCH_CHECK("rtts_relu_drop");
    return 0;
}
