// Round combine (forward), delta and partial-gradient reduction (backward) of LSH attention.
// All three are pure HBM-streaming kernels: 8 lanes own one 128-B head row (16 B per lane).
//
// Replaces the unsort + per-round logsumexp weighting of the reference's LSH layer
// (reformer_pytorch 0.19.1 via reformer_tts/model/reformer.py:217; SURVEY.md Appendix B
// steps 10-12); the unsort itself is free because rtts_lsh_attn_fwd already wrote o and lse at
// unsorted positions.
#include "rtts_common.h"
#include <float.h>

#define CB_DH 64
#define CB_MAXR 16

__device__ __forceinline__ void unpack8(const uint4 u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(w[k] << 16);
        f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]);
    u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]);
    u.w = pack_bf16x2(f[6], f[7]);
    return u;
}

__global__ __launch_bounds__(256) void lsh_combine_fwd_kernel(const bf16_t* __restrict__ o, const float* __restrict__ lse, int H,
                                                              int T, int n_hashes, size_t rows, bf16_t* __restrict__ out,
                                                              int64_t ld_out, float* __restrict__ lse_tot) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> 3;   // (bh, t)
    const int piece = gid & 7;
    if (row >= rows) return;
    const int bh = row / T, t = row % T;
    const int b = bh / H, h = bh % H;
    float lv[CB_MAXR];
    float m = -FLT_MAX;
    for (int r = 0; r < n_hashes; ++r) {
        lv[r] = lse[((size_t)bh * n_hashes + r) * T + t];
        m = fmaxf(m, lv[r]);
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wsum = 0.f;
    for (int r = 0; r < n_hashes; ++r) {
        const float w = __expf(lv[r] - m);
        wsum += w;
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(o + (((size_t)bh * n_hashes + r) * T + t) * CB_DH + piece * 8), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_fmaf(w, f[k], acc[k]);
    }
    const float inv = 1.f / wsum;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= inv;
    *reinterpret_cast<uint4*>(out + ((size_t)b * T + t) * ld_out + h * CB_DH + piece * 8) = pack8(acc);
    if (piece == 0) lse_tot[row] = m + logf(wsum);
}

__global__ __launch_bounds__(256) void lsh_bwd_delta_kernel(const bf16_t* __restrict__ out, int64_t ld_out,
                                                            const bf16_t* __restrict__ dout, int64_t ld_do, int H, int T,
                                                            size_t rows, float* __restrict__ delta) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> 3;
    const int piece = gid & 7;
    float s = 0.f;
    if (row < rows) {
        const int bh = row / T, t = row % T;
        const int b = bh / H, h = bh % H;
        float a[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(out + ((size_t)b * T + t) * ld_out + h * CB_DH + piece * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(dout + ((size_t)b * T + t) * ld_do + h * CB_DH + piece * 8), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) s = __builtin_fmaf(a[k], g[k], s);
    }
    s = rtts_sum8(s);
    if (row < rows && piece == 0) delta[row] = s;
}

__global__ __launch_bounds__(256) void lsh_bwd_reduce_kernel(const bf16_t* __restrict__ dqk_part, const bf16_t* __restrict__ dv_part,
                                                             int H, int T, int n_hashes, size_t rows, size_t slot_stride,
                                                             bf16_t* __restrict__ dqk, bf16_t* __restrict__ dv, int64_t ld_d,
                                                             const uint8_t* __restrict__ row_flags) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> 3;
    const int piece = gid & 7;
    if (row >= rows) return;
    const int bh = row / T, t = row % T;
    const int b = bh / H, h = bh % H;
    float aq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < n_hashes; ++r) {
        const size_t rr = ((size_t)bh * n_hashes + r) * T + t;
        const size_t off = rr * CB_DH + piece * 8;
        // row_flags (the walking backward): most slot-0 rows are complete and have no slot-1 partner
        const int nslot = (row_flags == nullptr || row_flags[rr]) ? 2 : 1;
        float f[8];
        for (int s = 0; s < nslot; ++s) {
            unpack8(*reinterpret_cast<const uint4*>(dqk_part + s * slot_stride + off), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) aq[k] += f[k];
            unpack8(*reinterpret_cast<const uint4*>(dv_part + s * slot_stride + off), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) av[k] += f[k];
        }
    }
    const size_t oo = ((size_t)b * T + t) * ld_d + h * CB_DH + piece * 8;
    *reinterpret_cast<uint4*>(dqk + oo) = pack8(aq);
    *reinterpret_cast<uint4*>(dv + oo) = pack8(av);
}

extern "C" int rtts_lsh_combine_fwd(const void* o, const float* lse, int B, int H, int T, int dh, int n_hashes, void* out,
                                    int64_t ld_out, float* lse_tot, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(o && lse && out && lse_tot, "rtts_lsh_combine_fwd: null pointer");
    RTTS_REQUIRE(dh == CB_DH && n_hashes > 0 && n_hashes <= CB_MAXR, "rtts_lsh_combine_fwd: need dh == 64, n_hashes <= 16");
    RTTS_REQUIRE(B > 0 && H > 0 && T > 0 && ld_out >= (int64_t)H * dh && ld_out % 8 == 0, "rtts_lsh_combine_fwd: bad shape/stride");
    RTTS_REQUIRE((((uintptr_t)o | (uintptr_t)out) & 15) == 0, "rtts_lsh_combine_fwd: buffers must be 16-byte aligned");
    const size_t rows = (size_t)B * H * T;
    const unsigned grid = (unsigned)((rows * 8 + 255) / 256);
    hipLaunchKernelGGL(lsh_combine_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, lse, H, T, n_hashes,
                       rows, (bf16_t*)out, ld_out, lse_tot);
    RTTS_LAUNCH_CHECK("rtts_lsh_combine_fwd");
    return 0;
}

extern "C" int rtts_lsh_bwd_delta(const void* out, int64_t ld_out, const void* dout, int64_t ld_dout, int B, int H, int T, int dh,
                                  float* delta, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(out && dout && delta, "rtts_lsh_bwd_delta: null pointer");
    RTTS_REQUIRE(dh == CB_DH && B > 0 && H > 0 && T > 0, "rtts_lsh_bwd_delta: need dh == 64");
    RTTS_REQUIRE(ld_out >= (int64_t)H * dh && ld_out % 8 == 0 && ld_dout >= (int64_t)H * dh && ld_dout % 8 == 0,
                 "rtts_lsh_bwd_delta: bad strides");
    RTTS_REQUIRE((((uintptr_t)out | (uintptr_t)dout) & 15) == 0, "rtts_lsh_bwd_delta: buffers must be 16-byte aligned");
    const size_t rows = (size_t)B * H * T;
    const unsigned grid = (unsigned)((rows * 8 + 255) / 256);
    hipLaunchKernelGGL(lsh_bwd_delta_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)out, ld_out,
                       (const bf16_t*)dout, ld_dout, H, T, rows, delta);
    RTTS_LAUNCH_CHECK("rtts_lsh_bwd_delta");
    return 0;
}

extern "C" int rtts_lsh_bwd_reduce(const void* dqk_part, const void* dv_part, int B, int H, int T, int dh, int n_hashes, void* dqk,
                                   void* dv, int64_t ld_d, const uint8_t* row_flags, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(dqk_part && dv_part && dqk && dv, "rtts_lsh_bwd_reduce: null pointer");
    RTTS_REQUIRE(dh == CB_DH && B > 0 && H > 0 && T > 0 && n_hashes > 0, "rtts_lsh_bwd_reduce: need dh == 64");
    RTTS_REQUIRE(ld_d >= (int64_t)H * dh && ld_d % 8 == 0, "rtts_lsh_bwd_reduce: bad stride");
    RTTS_REQUIRE((((uintptr_t)dqk_part | (uintptr_t)dv_part | (uintptr_t)dqk | (uintptr_t)dv) & 15) == 0,
                 "rtts_lsh_bwd_reduce: buffers must be 16-byte aligned");
    const size_t rows = (size_t)B * H * T;
    const size_t slot_stride = rows * n_hashes * CB_DH;
    const unsigned grid = (unsigned)((rows * 8 + 255) / 256);
    hipLaunchKernelGGL(lsh_bwd_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dqk_part,
                       (const bf16_t*)dv_part, H, T, n_hashes, rows, slot_stride, (bf16_t*)dqk, (bf16_t*)dv, ld_d, row_flags);
    RTTS_LAUNCH_CHECK("rtts_lsh_bwd_reduce");
    return 0;
}
