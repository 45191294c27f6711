// SqueezeWave vocoder, inference path: the pieces between the 1x1-convolution GEMMs of a WN block
// (/root/reference/reformer_tts/squeeze_wave/modules.py:88-122 depthwise separable convolution with its BatchNorm folded,
// :10-24 the tanh*sigmoid gate with the nearest-neighbour upsampled mel conditioning :216-225, :353-359 the inverse of
// the affine coupling).  Activations are channels-last rows (B*L, C) -- a 1x1 Conv1d is then a plain GEMM over rows and
// every kernel here streams rows with 8/16-byte accesses; all three are HBM-bound elementwise/stencil kernels.
#include "rtts_common.h"

#define SW_THREADS 256

static inline unsigned sw_grid(size_t items) {
    size_t b = (items + SW_THREADS - 1) / SW_THREADS;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// y[b][l][c] = bias[c] + sum_k w[c][k] * x[b][l + k - 1][c]   (kernel 3, zero padding), x fp32 -> y bf16; 4 channels/thread.
// edge_lo / edge_hi (may be null): per-channel constants subtracted at l = 0 / l = L-1 -- with an eval-mode BatchNorm folded
// into w and bias, the reference's zero padding pads bn(x), so the folded constant must not be counted for the missing tap.
__global__ __launch_bounds__(SW_THREADS) void sw_depthwise_k3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, int L, int C, size_t n4,
                                                                     bf16_t* __restrict__ y, const float* __restrict__ edge_lo,
                                                                     const float* __restrict__ edge_hi) {
    const int c4 = C / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        const size_t row = i / c4;
        const int l = (int)(row % L);
        const float4 mid = *reinterpret_cast<const float4*>(x + row * C + c);
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (l > 0) lo = *reinterpret_cast<const float4*>(x + (row - 1) * C + c);
        if (l + 1 < L) hi = *reinterpret_cast<const float4*>(x + (row + 1) * C + c);
        const float a[4] = {lo.x, lo.y, lo.z, lo.w}, m[4] = {mid.x, mid.y, mid.z, mid.w}, h[4] = {hi.x, hi.y, hi.z, hi.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wc = w + (size_t)(c + j) * 3;
            o[j] = __builtin_fmaf(wc[0], a[j], __builtin_fmaf(wc[1], m[j], __builtin_fmaf(wc[2], h[j], bias[c + j])));
            if (edge_lo && l == 0) o[j] -= edge_lo[c + j];
            if (edge_hi && l + 1 == L) o[j] -= edge_hi[c + j];
        }
        uint2 pk;
        pk.x = pack_bf16x2(o[0], o[1]);
        pk.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(y + row * C + c) = pk;
    }
}

// acts[m][c] = tanh(pw[m][c] + cond[r][off + c]) * sigmoid(pw[m][C + c] + cond[r][off + C + c]),  r = (b, l / up):
// the per-layer slice of the mel conditioning, nearest-neighbour upsampled; 8 channels per thread
__global__ __launch_bounds__(SW_THREADS) void sw_gate_kernel(const bf16_t* __restrict__ pw, const bf16_t* __restrict__ cond, int64_t ld_cond,
                                                             int off, int up, int L, int Lm, int C, size_t n8, bf16_t* __restrict__ acts) {
    const int c8 = C / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8) * 8;
        const size_t row = i / c8;
        const size_t b = row / L;
        const int l = (int)(row % L);
        const size_t crow = b * Lm + (up > 1 ? l / up : l);
        const uint4 pt = *reinterpret_cast<const uint4*>(pw + row * 2 * C + c);
        const uint4 ps = *reinterpret_cast<const uint4*>(pw + row * 2 * C + C + c);
        const uint4 ct = *reinterpret_cast<const uint4*>(cond + crow * ld_cond + off + c);
        const uint4 cs = *reinterpret_cast<const uint4*>(cond + crow * ld_cond + off + C + c);
        const uint32_t a[4] = {pt.x, pt.y, pt.z, pt.w}, bq[4] = {ps.x, ps.y, ps.z, ps.w};
        const uint32_t e[4] = {ct.x, ct.y, ct.z, ct.w}, f[4] = {cs.x, cs.y, cs.z, cs.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t0 = __uint_as_float(a[j] << 16) + __uint_as_float(e[j] << 16);
            const float t1 = __uint_as_float(a[j] & 0xffff0000u) + __uint_as_float(e[j] & 0xffff0000u);
            const float s0 = __uint_as_float(bq[j] << 16) + __uint_as_float(f[j] << 16);
            const float s1 = __uint_as_float(bq[j] & 0xffff0000u) + __uint_as_float(f[j] & 0xffff0000u);
            o[j] = pack_bf16x2(tanhf(t0) / (1.f + __expf(-s0)), tanhf(t1) / (1.f + __expf(-s1)));
        }
        *reinterpret_cast<uint4*>(acts + row * C + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// inverse affine coupling, in place on the second half of the channels: a1 = (a1 - b) / exp(s), wn = [s | b] (fp32)
__global__ __launch_bounds__(SW_THREADS) void sw_coupling_inv_kernel(float* __restrict__ audio, int64_t ld_audio, const float* __restrict__ wn,
                                                                     int half, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / half;
        const int c = (int)(i % half);
        float* a1 = audio + row * ld_audio + half + c;
        const float s = wn[row * 2 * half + c], b = wn[row * 2 * half + half + c];
        *a1 = (*a1 - b) * __expf(-s);
    }
}

extern "C" int rtts_sw_depthwise_k3(const float* x, const float* w, const float* bias, int B, int L, int C, void* y, const float* edge_lo,
                                    const float* edge_hi, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(x && w && bias && y && B > 0 && L > 0 && C > 0 && C % 4 == 0, "rtts_sw_depthwise_k3: bad arguments (C %% 4 == 0)");
    const size_t n4 = (size_t)B * L * C / 4;
    hipLaunchKernelGGL(sw_depthwise_k3_kernel, dim3(sw_grid(n4)), dim3(SW_THREADS), 0, (hipStream_t)stream, x, w, bias, L, C, n4, (bf16_t*)y, edge_lo, edge_hi);
    RTTS_LAUNCH_CHECK("rtts_sw_depthwise_k3");
    return 0;
}

extern "C" int rtts_sw_gate(const void* pw, const void* cond, int64_t ld_cond, int cond_offset, int upsample, int B, int L, int Lm, int C,
                            void* acts, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(pw && cond && acts && B > 0 && L > 0 && Lm > 0 && C > 0 && C % 8 == 0 && cond_offset % 8 == 0 && ld_cond % 8 == 0 &&
                     upsample >= 1 && (int64_t)Lm * upsample == L && ld_cond >= cond_offset + 2 * C,
                 "rtts_sw_gate: bad arguments (C %% 8 == 0, L == Lm * upsample)");
    const size_t n8 = (size_t)B * L * C / 8;
    hipLaunchKernelGGL(sw_gate_kernel, dim3(sw_grid(n8)), dim3(SW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)pw, (const bf16_t*)cond,
                       ld_cond, cond_offset, upsample, L, Lm, C, n8, (bf16_t*)acts);
    RTTS_LAUNCH_CHECK("rtts_sw_gate");
    return 0;
}

extern "C" int rtts_sw_coupling_inv(float* audio, int64_t ld_audio, const float* wn_out, int64_t rows, int half, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(audio && wn_out && rows > 0 && half > 0 && ld_audio >= 2 * half, "rtts_sw_coupling_inv: bad arguments");
    const size_t n = (size_t)rows * half;
    hipLaunchKernelGGL(sw_coupling_inv_kernel, dim3(sw_grid(n)), dim3(SW_THREADS), 0, (hipStream_t)stream, audio, ld_audio, wn_out, half, n);
    RTTS_LAUNCH_CHECK("rtts_sw_coupling_inv");
    return 0;
}

// ------------------------------------------------------------------ inverse coupling + inverse invertible 1x1 convolution
// out (rows, n) fp32 = [a0 | (a1 - b) * exp(-s)] @ Winv^T  -- the tail of one flow of SqueezeWave.infer
// (/root/reference/reformer_tts/squeeze_wave/modules.py:353-361: the affine coupling undone, then InvertibleConv1d's
// W_inverse, :57-85).  The audio path stays fp32 end to end (twelve chained matrix products: bf16 operands would leave ~3
// digits), so this is an fp32 FMA kernel, not an MFMA one: n <= 128 channels, 2*n*n FLOP per row -- 0.4 GFLOP for a
// 2-second utterance.  One workgroup = 32 rows; Winv^T (n x n) and the 32 coupled rows sit in LDS, thread (row, g) owns
// the output columns g, g + 8, ... of its row: the W reads of 8 neighbouring lanes are 8 consecutive words, the x reads
// broadcast.
#define SWI_ROWS 32
#define SWI_MAXN 128
__global__ __launch_bounds__(SW_THREADS) void sw_coupling_inv1x1_kernel(const float* __restrict__ audio, int64_t ld_audio,
                                                                        const float* __restrict__ wn, int64_t ld_wn,
                                                                        const float* __restrict__ winv, int n, int half, long long rows,
                                                                        float* __restrict__ out, int64_t ld_out) {
    extern __shared__ __attribute__((aligned(16))) float swi_smem[];
    float* Wt = swi_smem;                     // [n][n]: Wt[k][c] = Winv[c][k]
    float* X = Wt + n * n;                    // [SWI_ROWS][n + 1]
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += SW_THREADS) {
        const int c = i / n, k = i % n;       // coalesced read of Winv's rows; the transposed store is n-strided (once per block)
        Wt[k * n + c] = winv[i];
    }
    const long long r0 = (long long)blockIdx.x * SWI_ROWS;
    for (int i = tid; i < SWI_ROWS * n; i += SW_THREADS) {
        const int rr = i / n, k = i % n;
        const long long row = r0 + rr;
        float v = 0.f;
        if (row < rows) {
            v = audio[row * ld_audio + k];
            if (k >= half) {
                const float s = wn[row * ld_wn + (k - half)], b = wn[row * ld_wn + k];
                v = (v - b) * __expf(-s);
            }
        }
        X[rr * (n + 1) + k] = v;
    }
    __syncthreads();
    const int rr = tid >> 3, g = tid & 7;
    float acc[SWI_MAXN / 8];
#pragma unroll
    for (int j = 0; j < SWI_MAXN / 8; ++j) acc[j] = 0.f;
    const float* xr = X + rr * (n + 1);
    for (int k = 0; k < n; ++k) {
        const float xv = xr[k];
        const float* wk = Wt + k * n + g;
#pragma unroll
        for (int j = 0; j < SWI_MAXN / 8; ++j)
            if (g + 8 * j < n) acc[j] = __builtin_fmaf(xv, wk[8 * j], acc[j]);
    }
    const long long row = r0 + rr;
    if (row < rows) {
#pragma unroll
        for (int j = 0; j < SWI_MAXN / 8; ++j)
            if (g + 8 * j < n) out[row * ld_out + g + 8 * j] = acc[j];
    }
}

static RttsLdsState g_swi_lds;

extern "C" int rtts_sw_coupling_inv1x1(const float* audio, int64_t ld_audio, const float* wn_out, int64_t ld_wn, const float* winv, int n,
                                       int64_t rows, float* out, int64_t ld_out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(audio && wn_out && winv && out && rows > 0 && n >= 2 && n % 2 == 0 && n <= SWI_MAXN && ld_audio >= n && ld_wn >= n && ld_out >= n,
                 "rtts_sw_coupling_inv1x1: n must be even and <= %d, leading dimensions >= n (got n=%d)", SWI_MAXN, n);
    RTTS_REQUIRE(out != audio, "rtts_sw_coupling_inv1x1: not in place (a block reads rows of audio that another may have rewritten)");
    const size_t lds = ((size_t)n * n + (size_t)SWI_ROWS * (n + 1)) * sizeof(float);
    RTTS_ENSURE_LDS("rtts_sw_coupling_inv1x1", sw_coupling_inv1x1_kernel, lds, g_swi_lds);
    const unsigned blocks = (unsigned)((rows + SWI_ROWS - 1) / SWI_ROWS);
    hipLaunchKernelGGL(sw_coupling_inv1x1_kernel, dim3(blocks), dim3(SW_THREADS), lds, (hipStream_t)stream, audio, ld_audio, wn_out, ld_wn, winv,
                       n, n / 2, (long long)rows, out, ld_out);
    RTTS_LAUNCH_CHECK("rtts_sw_coupling_inv1x1");
    return 0;
}
