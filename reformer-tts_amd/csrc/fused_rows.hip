// Row-wise fused kernels around the GEMMs of a reversible block: LayerNorm forward/backward,
// residual epilogues, casts and bias-gradient column sums.  All are HBM-streaming kernels
// (one pass over their operands, 8/16-byte accesses per lane, no atomics: column sums go
// through per-workgroup partial rows and a tiny final reduction, so results are deterministic).
//
// They replace, fused, the ATen chains behind WithNorm / FeedForward / the residual adds of
// the reference (reformer_tts/model/reformer.py:25-45, modules.py:195-207, reversible.py:56-98)
// -- LayerNorm (eps 1e-5, affine), bias add, ReLU, x1 + f(x2) -- and their autograd backward.
//
// Layout: a wave owns a row; lane l owns the column chunks (k*64 + l)*VEC .. +VEC, k < EPL/VEC,
// where EPL = d/64 elements per lane.  d must be a multiple of 128 (VEC = 2) or 256 (VEC = 4);
// bf16-only kernels use VEC = 8 when d % 512 == 0.
#include "rtts_common.h"

#define FR_THREADS 256
#define FR_WAVES 4
#define FR_PARTIAL_BLOCKS 256   // rows of every column-sum partial buffer

// ---------------------------------------------------------------- vector helpers
template <int VEC> struct VecF32;
template <> struct VecF32<2> { typedef float2 T; };
template <> struct VecF32<4> { typedef float4 T; };

template <int EPL, int VEC>
__device__ __forceinline__ void load_row_f32(const float* __restrict__ row, int lane, float* v) {
#pragma unroll
    for (int k = 0; k < EPL / VEC; ++k) {
        const typename VecF32<VEC>::T t = *reinterpret_cast<const typename VecF32<VEC>::T*>(row + (k * 64 + lane) * VEC);
        const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[k * VEC + j] = f[j];
    }
}

template <int EPL, int VEC>
__device__ __forceinline__ void store_row_f32(float* __restrict__ row, int lane, const float* v) {
#pragma unroll
    for (int k = 0; k < EPL / VEC; ++k) {
        typename VecF32<VEC>::T t;
        float* f = reinterpret_cast<float*>(&t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] = v[k * VEC + j];
        *reinterpret_cast<typename VecF32<VEC>::T*>(row + (k * 64 + lane) * VEC) = t;
    }
}

// Rows that leave for HBM (not the LDS staging of block_partial).  RTTS_ROW_WT (A/B builds, scripts/build_ab.sh): 1 = the fp32
// rows as 16-byte write-through stores, 2 = the 8-byte bf16 rows too; 0 = plain stores.
#ifndef RTTS_ROW_WT
#define RTTS_ROW_WT 0
#endif
template <int EPL, int VEC>
__device__ __forceinline__ void store_row_f32_out(float* __restrict__ row, int lane, const float* v) {
#if RTTS_ROW_WT >= 1
    if constexpr (VEC == 4) {
#pragma unroll
        for (int k = 0; k < EPL / VEC; ++k) {
            uint4 t;
            t.x = __float_as_uint(v[k * 4]); t.y = __float_as_uint(v[k * 4 + 1]); t.z = __float_as_uint(v[k * 4 + 2]); t.w = __float_as_uint(v[k * 4 + 3]);
            rtts_store16_out(row + (k * 64 + lane) * 4, t);
        }
        return;
    }
#endif
    store_row_f32<EPL, VEC>(row, lane, v);
}

template <int EPL, int VEC>
__device__ __forceinline__ void load_row_bf16(const bf16_t* __restrict__ row, int lane, float* v) {
#pragma unroll
    for (int k = 0; k < EPL / VEC; ++k) {
        const bf16_t* p = row + (k * 64 + lane) * VEC;
        if constexpr (VEC == 8) {
            const uint4 t = *reinterpret_cast<const uint4*>(p);
            const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[k * 8 + 2 * j] = __uint_as_float(u[j] << 16);
                v[k * 8 + 2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
            }
        } else if constexpr (VEC == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            v[k * 4] = __uint_as_float(t.x << 16);
            v[k * 4 + 1] = __uint_as_float(t.x & 0xffff0000u);
            v[k * 4 + 2] = __uint_as_float(t.y << 16);
            v[k * 4 + 3] = __uint_as_float(t.y & 0xffff0000u);
        } else {
            const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
            v[k * 2] = __uint_as_float(t << 16);
            v[k * 2 + 1] = __uint_as_float(t & 0xffff0000u);
        }
    }
}

template <int EPL, int VEC, bool WT = true>
__device__ __forceinline__ void store_row_bf16(bf16_t* __restrict__ row, int lane, const float* v) {
#pragma unroll
    for (int k = 0; k < EPL / VEC; ++k) {
        bf16_t* p = row + (k * 64 + lane) * VEC;
        if constexpr (VEC == 8) {
            uint4 t;
            t.x = pack_bf16x2(v[k * 8], v[k * 8 + 1]);
            t.y = pack_bf16x2(v[k * 8 + 2], v[k * 8 + 3]);
            t.z = pack_bf16x2(v[k * 8 + 4], v[k * 8 + 5]);
            t.w = pack_bf16x2(v[k * 8 + 6], v[k * 8 + 7]);
            *reinterpret_cast<uint4*>(p) = t;
        } else if constexpr (VEC == 4) {
            uint2 t;
            t.x = pack_bf16x2(v[k * 4], v[k * 4 + 1]);
            t.y = pack_bf16x2(v[k * 4 + 2], v[k * 4 + 3]);
#if RTTS_ROW_WT >= 2
            if (WT) {
                typedef int rv2i __attribute__((ext_vector_type(2)));
                const rv2i w2 = {(int)t.x, (int)t.y};
                asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(w2) : "memory");
                continue;
            }
#endif
            *reinterpret_cast<uint2*>(p) = t;
        } else {
            *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(v[k * 2], v[k * 2 + 1]);
        }
    }
}

// column of register element e of a row held in the load_row_f32 / load_row_bf16 layout
template <int VEC>
__device__ __forceinline__ int row_col(int e, int lane) { return ((e / VEC) * 64 + lane) * VEC + (e % VEC); }

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// per-lane column accumulators of the 4 waves of a block -> one partial row (deterministic order)
template <int EPL, int VEC>
__device__ __forceinline__ void block_partial(const float* acc, float* __restrict__ partial_row, float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    store_row_f32<EPL, VEC>(lds + wave * EPL * 64, lane, acc);
    __syncthreads();
    for (int c = threadIdx.x; c < EPL * 64; c += FR_THREADS)
        partial_row[c] = lds[c] + lds[EPL * 64 + c] + lds[2 * EPL * 64 + c] + lds[3 * EPL * 64 + c];
    __syncthreads();
}

// ---------------------------------------------------------------- LayerNorm forward
template <int EPL, int VEC>
__global__ __launch_bounds__(FR_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ xn,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M) {
    constexpr int D = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * FR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[EPL], g[EPL], b[EPL];
    load_row_f32<EPL, VEC>(x + (size_t)row * D, lane, v);
    load_row_f32<EPL, VEC>(gamma, lane, g);
    load_row_f32<EPL, VEC>(beta, lane, b);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        v[e] -= mu;
        q = __builtin_fmaf(v[e], v[e], q);
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + 1e-5f);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = __builtin_fmaf(v[e] * rs, g[e], b[e]);
    store_row_bf16<EPL, VEC>(xn + (size_t)row * D, lane, v);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// ---------------------------------------------------------------- residual epilogue + the NEXT block's LayerNorm
// x += sign * (g + bias) in place, then xn = LayerNorm(x) * gamma + beta (bf16) with its row statistics: the stream a
// block has just updated (forward) or reconstructed (backward) is exactly the next block's LayerNorm input, so the
// row is normalised while it is still in registers instead of being read back by a separate launch.
template <int EPL, int VEC>
__global__ __launch_bounds__(FR_THREADS) void residual_ln_kernel(const float* x, float* y, const bf16_t* __restrict__ g,
                                                                 const float* __restrict__ bias, float sign,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 bf16_t* __restrict__ xn, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int M, uint32_t seed,
                                                                 const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale) {
    constexpr int D = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * FR_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[EPL], gv[EPL], w[EPL];
    load_row_f32<EPL, VEC>(x + (size_t)row * D, lane, v);
    load_row_bf16<EPL, VEC>(g + (size_t)row * D, lane, gv);
    if (bias) {
        load_row_f32<EPL, VEC>(bias, lane, w);
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] += w[e];
    }
    if (thresh) {          // post-attention dropout on f(x) = g + bias (reformer_pytorch's post_attn_dropout)
        if (seed_dev) seed += seed_dev[0];
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] *= rtts_drop_keep(seed, (uint32_t)row * D + row_col<VEC>(e, lane), thresh, dscale);
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] += sign * gv[e];
    store_row_f32_out<EPL, VEC>(y + (size_t)row * D, lane, v);      // y == x: the stream is updated in place
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s += v[e];
    const float mu = wave_sum(s) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        v[e] -= mu;
        q = __builtin_fmaf(v[e], v[e], q);
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / D) + 1e-5f);
    load_row_f32<EPL, VEC>(gamma, lane, gv);
    load_row_f32<EPL, VEC>(beta, lane, w);
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = __builtin_fmaf(v[e] * rs, gv[e], w[e]);
    store_row_bf16<EPL, VEC>(xn + (size_t)row * D, lane, v);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// ---------------------------------------------------------------- LayerNorm backward
// dx_io += rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dxn * gamma;
// partial_g[blk] = sum_rows dxn * xhat, partial_b[blk] = sum_rows dxn
template <int EPL, int VEC>
__global__ __launch_bounds__(FR_THREADS) void ln_bwd_kernel(const bf16_t* __restrict__ dxn, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* dx_in, const float* __restrict__ addend,
                                                            float* dx_io, float* __restrict__ partial_g, float* __restrict__ partial_b, int M,
                                                            bf16_t* __restrict__ dyb_next, float* __restrict__ partial_next,
                                                            uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                            float dscale) {
    // dyb_next != null: the gradient stream this kernel has just completed is the NEXT block's output gradient, so its
    // bf16 copy (times that block's post-attention dropout keep-scale, if any) and the partial column sums for that
    // block's output bias leave in the same pass -- what a separate rtts_cast_colsum launch would re-read the stream for.
    constexpr int D = EPL * 64;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63;
    float gam[EPL], ag[EPL], ab[EPL], an[EPL];
    load_row_f32<EPL, VEC>(gamma, lane, gam);
    if (thresh && seed_dev) seed += seed_dev[0];
#pragma unroll
    for (int e = 0; e < EPL; ++e) ag[e] = ab[e] = an[e] = 0.f;
    for (int row = blockIdx.x * FR_WAVES + (threadIdx.x >> 6); row < M; row += gridDim.x * FR_WAVES) {
        float dy[EPL], xv[EPL], dx[EPL];
        load_row_bf16<EPL, VEC>(dxn + (size_t)row * D, lane, dy);
        load_row_f32<EPL, VEC>(x + (size_t)row * D, lane, xv);
        load_row_f32<EPL, VEC>(dx_in + (size_t)row * D, lane, dx);      // dx_in == dx_io: in place (a lane reads what it writes)
        if (addend) {       // the last update of a reversible stack's backward: the OTHER gradient stream joins here (dx = g1 + g2)
            float ad[EPL];
            load_row_f32<EPL, VEC>(addend + (size_t)row * D, lane, ad);
#pragma unroll
            for (int e = 0; e < EPL; ++e) dx[e] += ad[e];
        }
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            xv[e] = (xv[e] - mu) * rs;          // xhat
            ag[e] = __builtin_fmaf(dy[e], xv[e], ag[e]);
            ab[e] += dy[e];
            dy[e] *= gam[e];                    // g
            s1 += dy[e];
            s2 = __builtin_fmaf(dy[e], xv[e], s2);
        }
        s1 = wave_sum(s1) * (1.f / D);
        s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int e = 0; e < EPL; ++e) dx[e] += rs * (dy[e] - s1 - xv[e] * s2);
        store_row_f32_out<EPL, VEC>(dx_io + (size_t)row * D, lane, dx);
        if (dyb_next) {
            if (thresh) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) dx[e] *= rtts_drop_keep(seed, (uint32_t)row * D + row_col<VEC>(e, lane), thresh, dscale);
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) an[e] += dx[e];
            store_row_bf16<EPL, VEC>(dyb_next + (size_t)row * D, lane, dx);
        }
    }
    block_partial<EPL, VEC>(ag, partial_g + (size_t)blockIdx.x * D, lds_f);
    block_partial<EPL, VEC>(ab, partial_b + (size_t)blockIdx.x * D, lds_f);
    if (dyb_next) block_partial<EPL, VEC>(an, partial_next + (size_t)blockIdx.x * D, lds_f);
}

// ---------------------------------------------------------------- fp32 -> bf16 cast + column sums
template <int EPL, int VEC>
__global__ __launch_bounds__(FR_THREADS) void cast_colsum_kernel(const float* __restrict__ dy, bf16_t* __restrict__ dyb,
                                                                 float* __restrict__ partial, int M, uint32_t seed,
                                                                 const uint32_t* __restrict__ seed_dev, uint32_t thresh, float dscale,
                                                                 const float* __restrict__ scale_dev) {
    constexpr int D = EPL * 64;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    if (thresh && seed_dev) seed += seed_dev[0];
    const float up = scale_dev ? scale_dev[0] : 1.f;       // an upstream scalar gradient (the loss weight of a backward root)
    for (int row = blockIdx.x * FR_WAVES + (threadIdx.x >> 6); row < M; row += gridDim.x * FR_WAVES) {
        float v[EPL];
        load_row_f32<EPL, VEC>(dy + (size_t)row * D, lane, v);
        if (scale_dev) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] *= up;
        }
        if (thresh) {      // backward of the dropout that sat on this block's output: same (seed, element) decisions
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] *= rtts_drop_keep(seed, (uint32_t)row * D + row_col<VEC>(e, lane), thresh, dscale);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += v[e];
        store_row_bf16<EPL, VEC>(dyb + (size_t)row * D, lane, v);
    }
    block_partial<EPL, VEC>(acc, partial + (size_t)blockIdx.x * D, lds_f);
}

// ---------------------------------------------------------------- bf16 column sums, optionally gated by ReLU
// RELU: dst = dh * (h > 0) * gate_scale (dst == dh: in place), then summed
template <int EPL, int VEC, bool RELU>
__global__ __launch_bounds__(FR_THREADS) void colsum_bf16_kernel(const bf16_t* dh, const bf16_t* __restrict__ h, bf16_t* dst,
                                                                 int64_t ld, float* __restrict__ partial, int M, float gate_scale) {
    constexpr int D = EPL * 64;
    constexpr int AV = (EPL % 4 == 0) ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int row = blockIdx.x * FR_WAVES + (threadIdx.x >> 6); row < M; row += gridDim.x * FR_WAVES) {
        float v[EPL];
        load_row_bf16<EPL, VEC>(dh + (size_t)row * ld, lane, v);
        if (RELU) {
            float hv[EPL];
            load_row_bf16<EPL, VEC>(h + (size_t)row * ld, lane, hv);
#pragma unroll
            for (int e = 0; e < EPL; ++e) v[e] = hv[e] > 0.f ? v[e] * gate_scale : 0.f;
            store_row_bf16<EPL, VEC>(dst + (size_t)row * ld, lane, v);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += v[e];
    }
    // the partial row is laid out in the bf16 chunking (VEC); re-map through LDS in that order
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < EPL / VEC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) lds_f[wave * D + (k * 64 + lane) * VEC + j] = acc[k * VEC + j];
    __syncthreads();
    float* prow = partial + (size_t)blockIdx.x * D;
    for (int c = threadIdx.x; c < D; c += FR_THREADS) prow[c] = lds_f[c] + lds_f[D + c] + lds_f[2 * D + c] + lds_f[3 * D + c];
    (void)AV;
}

// out[c] += sum over the partial rows, in a fixed order: a block owns 64 columns, its 4 waves each
// sum every 4th partial row (coalesced 256-B reads), then the 4 wave sums are added in wave order
// (blockIdx.y == 1 selects a second partial buffer / output: the two sums of a LayerNorm backward in one launch)
__global__ __launch_bounds__(FR_THREADS) void colsum_final_kernel(const float* __restrict__ partial, int nrows, int n,
                                                                  float* __restrict__ out, const float* __restrict__ partial2,
                                                                  float* __restrict__ out2) {
    __shared__ float red[FR_WAVES][64];
    if (blockIdx.y) {
        partial = partial2;
        out = out2;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < n) {
#pragma unroll 8
        for (int r = wave; r < nrows; r += FR_WAVES) s += partial[(size_t)r * n + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < n) out[c] += (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// The same reduction for up to RTTS_COLSUM_MAX_GROUP (partial buffer, output) pairs in ONE launch: bias / LayerNorm
// gradients are leaves of the backward, so the executor queues their finalisation and flushes a layer's worth together
// (a decoder layer has 15 of these 5-us launches otherwise).
struct CsJob {
    const float* partial;
    float* out;
    int nrows, n, ld, blk_start;
};
struct CsGroup {
    CsJob j[RTTS_COLSUM_MAX_GROUP];
    int n;
};
// 16 waves per 64 columns: a job has up to 256 partial rows, i.e. <= 16 loads per wave, all in flight at once (with 4 waves
// and 64 dependent-latency loads each the launch took 14 us for ~10 MB -- latency, not bandwidth)
#define CS_WAVES 16
__global__ __launch_bounds__(64 * CS_WAVES) void colsum_final_grouped_kernel(const CsGroup grp) {
    __shared__ float red[CS_WAVES][64];
    int ji = 0;
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.j[i].blk_start) ji = i;
    const CsJob& J = grp.j[ji];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - J.blk_start) * 64 + lane;
    float s = 0.f;
    if (c < J.n) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = wave + k * CS_WAVES;
            v[k] = r < J.nrows ? J.partial[(size_t)r * J.ld + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; k += 4) s += (v[k] + v[k + 1]) + (v[k + 2] + v[k + 3]);
        for (int r = wave + 16 * CS_WAVES; r < J.nrows; r += CS_WAVES) s += J.partial[(size_t)r * J.ld + c];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < J.n) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CS_WAVES; w += 4) t += (red[w][lane] + red[w + 1][lane]) + (red[w + 2][lane] + red[w + 3][lane]);
        J.out[c] += t;
    }
}

// ---------------------------------------------------------------- elementwise epilogues
// y = x + sign * (g + bias)        (x, y fp32; g bf16; bias fp32 or null), 4 elements per thread
__global__ __launch_bounds__(FR_THREADS) void residual_epilogue_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                                       const float* __restrict__ bias, float sign,
                                                                       float* __restrict__ y, size_t n4, int d, uint32_t seed,
                                                                       const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                                       float dscale) {
    if (thresh && seed_dev) seed += seed_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const uint2 gv = reinterpret_cast<const uint2*>(g)[i];
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + (i * 4) % d);
        float f[4] = {__uint_as_float(gv.x << 16) + bv.x, __uint_as_float(gv.x & 0xffff0000u) + bv.y,
                      __uint_as_float(gv.y << 16) + bv.z, __uint_as_float(gv.y & 0xffff0000u) + bv.w};
        if (thresh) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] *= rtts_drop_keep(seed, (uint32_t)(i * 4 + j), thresh, dscale);
        }
        float4 o;
        o.x = xv.x + sign * f[0];
        o.y = xv.y + sign * f[1];
        o.z = xv.z + sign * f[2];
        o.w = xv.w + sign * f[3];
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// h = relu(h + bias) in place (bf16), 8 elements per thread
__global__ __launch_bounds__(FR_THREADS) void bias_act_kernel(bf16_t* __restrict__ h, const float* __restrict__ bias, size_t n8, int d,
                                                              int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        uint4 t = reinterpret_cast<uint4*>(h)[i];
        const float* b = bias + (i * 8) % d;
        const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
        uint32_t u[4] = {t.x, t.y, t.z, t.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float lo = __uint_as_float(u[j] << 16) + bb[2 * j], hi = __uint_as_float(u[j] & 0xffff0000u) + bb[2 * j + 1];
            if (relu) {
                lo = fmaxf(lo, 0.f);
                hi = fmaxf(hi, 0.f);
            }
            u[j] = pack_bf16x2(lo, hi);
        }
        t.x = u[0]; t.y = u[1]; t.z = u[2]; t.w = u[3];
        reinterpret_cast<uint4*>(h)[i] = t;
    }
}

__global__ __launch_bounds__(FR_THREADS) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y);
        o.y = pack_bf16x2(v.z, v.w);
        reinterpret_cast<uint2*>(dst)[i] = o;
    }
}

// ---------------------------------------------------------------- host side
static inline unsigned stream_grid(size_t work_items) {
    size_t b = (work_items + FR_THREADS - 1) / FR_THREADS;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

#define FR_DISPATCH_D(d, CALL)                                    \
    switch (d) {                                                  \
        case 128: { CALL(2, 2); break; }                          \
        case 256: { CALL(4, 4); break; }                          \
        case 384: { CALL(6, 2); break; }                          \
        case 512: { CALL(8, 4); break; }                          \
        case 768: { CALL(12, 4); break; }                         \
        case 1024: { CALL(16, 4); break; }                        \
        case 2048: { CALL(32, 4); break; }                        \
        default:                                                  \
            rtts_set_error("row kernels: width %d unsupported (128, 256, 384, 512, 768, 1024, 2048)", d); \
            return -1;                                            \
    }

extern "C" int rtts_ln_fwd(const float* x, const float* gamma, const float* beta, void* xn, float* mean, float* rstd, int M, int d,
                           void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(x && gamma && beta && xn && mean && rstd && M > 0, "rtts_ln_fwd: bad arguments");
    const dim3 grid((M + FR_WAVES - 1) / FR_WAVES);
#define CALL(EPL, VEC) hipLaunchKernelGGL((ln_fwd_kernel<EPL, VEC>), grid, dim3(FR_THREADS), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)xn, mean, rstd, M)
    FR_DISPATCH_D(d, CALL)
#undef CALL
    RTTS_LAUNCH_CHECK("rtts_ln_fwd");
    return 0;
}

extern "C" int rtts_ln_bwd_join(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                                const float* addend, float* dx_io, float* dgamma, float* dbeta, float* partial_ws, int M, int d, void* dyb_next,
                                float* partial_next, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);

extern "C" int rtts_ln_bwd(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx_io,
                           float* dgamma, float* dbeta, float* partial_ws, int M, int d, void* dyb_next, float* partial_next,
                           float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    return rtts_ln_bwd_join(dxn, x, mean, rstd, gamma, dx_io, nullptr, dx_io, dgamma, dbeta, partial_ws, M, d, dyb_next, partial_next, drop_p,
                            seed, seed_dev, stream);
}

// dx_out = dx_in + dLN(dxn): the out-of-place form (dx_in == dx_out: rtts_ln_bwd).  The stack executor uses it for the FIRST update
// of each gradient stream, which starts as the caller's dout itself instead of a copy of it.
extern "C" int rtts_ln_bwd_to(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                              float* dx_io, float* dgamma, float* dbeta, float* partial_ws, int M, int d, void* dyb_next, float* partial_next,
                              float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    return rtts_ln_bwd_join(dxn, x, mean, rstd, gamma, dx_in, nullptr, dx_io, dgamma, dbeta, partial_ws, M, d, dyb_next, partial_next, drop_p,
                            seed, seed_dev, stream);
}

// dx_out = dx_in + addend + dLN(dxn) (addend may be NULL: rtts_ln_bwd_to).  The last LayerNorm backward of a reversible stack's
// backward: both streams started as the stack's input, so d(input) = g1 + g2 -- the sum rides in the pass that completes g2
// instead of a separate pass over both streams.
extern "C" int rtts_ln_bwd_join(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                                const float* addend, float* dx_io, float* dgamma, float* dbeta, float* partial_ws, int M, int d, void* dyb_next,
                                float* partial_next, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dx_in, "rtts_ln_bwd_to: null dx_in");
    RTTS_REQUIRE(addend != dx_io, "rtts_ln_bwd_join: the addend must not be the output");
    RTTS_REQUIRE(!dyb_next || partial_next, "rtts_ln_bwd: dyb_next needs partial_next");
    RTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "rtts_ln_bwd: bad drop_p");
    RTTS_REQUIRE(dxn && x && mean && rstd && gamma && dx_io && partial_ws && M > 0 && (!dgamma == !dbeta), "rtts_ln_bwd: bad arguments");
    int blocks = (M + FR_WAVES - 1) / FR_WAVES;
    if (blocks > FR_PARTIAL_BLOCKS) blocks = FR_PARTIAL_BLOCKS;
    float* pg = partial_ws;
    float* pb = partial_ws + (size_t)FR_PARTIAL_BLOCKS * d;
    const size_t lds = (size_t)FR_WAVES * d * sizeof(float);
#define CALL(EPL, VEC) hipLaunchKernelGGL((ln_bwd_kernel<EPL, VEC>), dim3(blocks), dim3(FR_THREADS), lds, (hipStream_t)stream, (const bf16_t*)dxn, x, mean, rstd, gamma, dx_in, addend, dx_io, pg, pb, M, (bf16_t*)dyb_next, partial_next, seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p))
    FR_DISPATCH_D(d, CALL)
#undef CALL
    if (dgamma) {       // NULL: the caller finalises the partial rows itself (rtts_colsum_final_grouped)
        const dim3 g2((d + 63) / 64, 2);
        hipLaunchKernelGGL(colsum_final_kernel, g2, dim3(FR_THREADS), 0, (hipStream_t)stream, pg, blocks, d, dgamma, pb, dbeta);
    }
    RTTS_LAUNCH_CHECK("rtts_ln_bwd");
    return 0;
}

extern "C" int rtts_cast_colsum(const float* dy, void* dyb, float* dbias, float* partial_ws, int M, int d, float drop_p, uint32_t seed,
                                const uint32_t* seed_dev, const float* scale_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dy && dyb && partial_ws && M > 0 && drop_p >= 0.f && drop_p < 1.f, "rtts_cast_colsum: bad arguments");
    int blocks = (M + FR_WAVES - 1) / FR_WAVES;
    if (blocks > FR_PARTIAL_BLOCKS) blocks = FR_PARTIAL_BLOCKS;
    const size_t lds = (size_t)FR_WAVES * d * sizeof(float);
#define CALL(EPL, VEC) hipLaunchKernelGGL((cast_colsum_kernel<EPL, VEC>), dim3(blocks), dim3(FR_THREADS), lds, (hipStream_t)stream, dy, (bf16_t*)dyb, partial_ws, M, seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p), scale_dev)
    FR_DISPATCH_D(d, CALL)
#undef CALL
    if (dbias)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64), dim3(FR_THREADS), 0, (hipStream_t)stream,
                           partial_ws, blocks, d, dbias, (const float*)nullptr, (float*)nullptr);
    RTTS_LAUNCH_CHECK("rtts_cast_colsum");
    return 0;
}

extern "C" int rtts_colsum_bf16(void* dh, const void* h, int64_t ld, float* dbias, float* partial_ws, int M, int d, int relu_gate,
                                float gate_scale, void* gated_out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dh && partial_ws && M > 0 && (!relu_gate || h), "rtts_colsum_bf16: bad arguments");
    bf16_t* dst = (bf16_t*)(gated_out ? gated_out : dh);
    RTTS_REQUIRE(ld >= d && ld % 8 == 0, "rtts_colsum_bf16: bad row stride");
    int blocks = (M + FR_WAVES - 1) / FR_WAVES;
    if (blocks > FR_PARTIAL_BLOCKS) blocks = FR_PARTIAL_BLOCKS;
    const size_t lds = (size_t)FR_WAVES * d * sizeof(float);
#define CALL(EPL, VEC)                                                                                                         \
    if (relu_gate)                                                                                                             \
        hipLaunchKernelGGL((colsum_bf16_kernel<EPL, VEC, true>), dim3(blocks), dim3(FR_THREADS), lds, (hipStream_t)stream,     \
                           (const bf16_t*)dh, (const bf16_t*)h, dst, ld, partial_ws, M, gate_scale);                           \
    else                                                                                                                       \
        hipLaunchKernelGGL((colsum_bf16_kernel<EPL, VEC, false>), dim3(blocks), dim3(FR_THREADS), lds, (hipStream_t)stream,    \
                           (const bf16_t*)dh, (const bf16_t*)h, dst, ld, partial_ws, M, gate_scale)
    FR_DISPATCH_D(d, CALL)
#undef CALL
    if (dbias)
        hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64), dim3(FR_THREADS), 0, (hipStream_t)stream,
                           partial_ws, blocks, d, dbias, (const float*)nullptr, (float*)nullptr);
    RTTS_LAUNCH_CHECK("rtts_colsum_bf16");
    return 0;
}

// out = a + b (the two streams of a reversible stack, reversible.py:155-158 sums their halves): fp32 and / or a bf16 twin
// for the consumer that reads it as a GEMM operand (the heads, the cross attention's keys)
__global__ __launch_bounds__(FR_THREADS) void sum_streams_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n4,
                                                                 float* __restrict__ out, bf16_t* __restrict__ out_bf16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        const float4 o = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        if (out) reinterpret_cast<float4*>(out)[i] = o;
        if (out_bf16) reinterpret_cast<uint2*>(out_bf16)[i] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
    }
}

extern "C" int rtts_sum_streams(const float* a, const float* b, int64_t n, float* out, void* out_bf16, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(a && b && (out || out_bf16) && n > 0 && n % 4 == 0, "rtts_sum_streams: bad arguments");
    hipLaunchKernelGGL(sum_streams_kernel, dim3(stream_grid((size_t)n / 4)), dim3(FR_THREADS), 0, (hipStream_t)stream, a, b, (size_t)n / 4, out,
                       (bf16_t*)out_bf16);
    RTTS_LAUNCH_CHECK("rtts_sum_streams");
    return 0;
}

extern "C" int rtts_colsum_partial_rows(int M) {
    int blocks = (M + FR_WAVES - 1) / FR_WAVES;
    return blocks > FR_PARTIAL_BLOCKS ? FR_PARTIAL_BLOCKS : blocks;
}

extern "C" int rtts_colsum_final_grouped(const rtts_colsum_job* jobs, int n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(jobs && n > 0 && n <= RTTS_COLSUM_MAX_GROUP, "rtts_colsum_final_grouped: 1..%d jobs", RTTS_COLSUM_MAX_GROUP);
    CsGroup grp;
    grp.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        RTTS_REQUIRE(jobs[i].partial && jobs[i].out && jobs[i].nrows > 0 && jobs[i].n > 0, "rtts_colsum_final_grouped: bad job %d", i);
        grp.j[i].partial = jobs[i].partial;
        grp.j[i].out = jobs[i].out;
        grp.j[i].nrows = jobs[i].nrows;
        grp.j[i].n = jobs[i].n;
        RTTS_REQUIRE(jobs[i].ld == 0 || jobs[i].ld >= jobs[i].n, "rtts_colsum_final_grouped: job %d: row stride below the width", i);
        grp.j[i].ld = jobs[i].ld ? jobs[i].ld : jobs[i].n;
        grp.j[i].blk_start = blk;
        blk += (jobs[i].n + 63) / 64;
    }
    hipLaunchKernelGGL(colsum_final_grouped_kernel, dim3(blk), dim3(64 * CS_WAVES), 0, (hipStream_t)stream, grp);
    RTTS_LAUNCH_CHECK("rtts_colsum_final_grouped");
    return 0;
}

extern "C" int rtts_residual_epilogue(const float* x, const void* g, const float* bias, float sign, float* y, int64_t M, int d,
                                      float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(x && g && y && M > 0 && d > 0 && d % 4 == 0 && drop_p >= 0.f && drop_p < 1.f, "rtts_residual_epilogue: bad arguments");
    const size_t n4 = (size_t)M * d / 4;
    hipLaunchKernelGGL(residual_epilogue_kernel, dim3(stream_grid(n4)), dim3(FR_THREADS), 0, (hipStream_t)stream, x, (const bf16_t*)g,
                       bias, sign, y, n4, d, seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p));
    RTTS_LAUNCH_CHECK("rtts_residual_epilogue");
    return 0;
}

extern "C" int rtts_residual_ln(float* x, const void* g, const float* bias, float sign, const float* gamma, const float* beta,
                                void* xn, float* mean, float* rstd, int M, int d, float drop_p, uint32_t seed,
                                const uint32_t* seed_dev, float* y, void* stream) {
    RTTS_ENTER(stream);
    if (!y) y = x;
    RTTS_REQUIRE(x && g && gamma && beta && xn && mean && rstd && M > 0 && drop_p >= 0.f && drop_p < 1.f, "rtts_residual_ln: bad arguments");
    const dim3 grid((M + FR_WAVES - 1) / FR_WAVES);
#define CALL(EPL, VEC) hipLaunchKernelGGL((residual_ln_kernel<EPL, VEC>), grid, dim3(FR_THREADS), 0, (hipStream_t)stream, x, y, (const bf16_t*)g, bias, sign, gamma, beta, (bf16_t*)xn, mean, rstd, M, seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p))
    FR_DISPATCH_D(d, CALL)
#undef CALL
    RTTS_LAUNCH_CHECK("rtts_residual_ln");
    return 0;
}

extern "C" int rtts_bias_act(void* h, const float* bias, int64_t M, int d, int relu, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(h && bias && M > 0 && d > 0 && d % 8 == 0, "rtts_bias_act: bad arguments");
    const size_t n8 = (size_t)M * d / 8;
    hipLaunchKernelGGL(bias_act_kernel, dim3(stream_grid(n8)), dim3(FR_THREADS), 0, (hipStream_t)stream, (bf16_t*)h, bias, n8, d, relu);
    RTTS_LAUNCH_CHECK("rtts_bias_act");
    return 0;
}

extern "C" int rtts_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(src && dst && n > 0 && n % 4 == 0, "rtts_cast_f32_bf16: n must be a positive multiple of 4");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(stream_grid((size_t)n / 4)), dim3(FR_THREADS), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, (size_t)n / 4);
    RTTS_LAUNCH_CHECK("rtts_cast_f32_bf16");
    return 0;
}
