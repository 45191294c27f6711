// On-box peak probes for bench.py's roofline lines ("peaks must be measured on the box"):
//   rtts_peak_copy   float4 stream copy dst = src, four loads in flight per thread: the HBM rate a plain coalesced kernel reaches
//                    (read + write bytes / time)
//   rtts_peak_mfma   back-to-back v_mfma_f32_16x16x32_bf16 on register operands (random, non-zero: zero operands let the chip
//                    hold a higher clock, cdna_hip_programming.md rule 25), two waves per SIMD on every CU: the dense bf16
//                    matrix rate the chip sustains at the clock it holds under that load
// Neither is on the training path.
#include "rtts_common.h"
#include <stdlib.h>

// FOUR 16-byte loads in flight per thread before the first store (one in flight -- round 3's probe -- reached 4.8 TB/s where
// MI355X_MICROARCH.md measures 6.29 TB/s for a float4 copy: a probe that understates the peak flatters every "fraction of the
// measured peak" derived from it); the tail (n4 not a multiple of 4 x the grid's threads) is copied one float4 at a time
__global__ __launch_bounds__(256) void peak_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    // a workgroup walks 16 KB tiles (4 x 256 float4, contiguous): the four loads of a thread are 4 KB apart inside one tile
    const size_t ntile = n4 / 1024;
    for (size_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const size_t i = tile * 1024 + threadIdx.x;
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a;
        dst[i + 256] = b;
        dst[i + 512] = c;
        dst[i + 768] = d;
    }
    for (size_t i = ntile * 1024 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(512) void peak_mfma_kernel(float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    // operands in [-1, 1), different in every lane
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t h1 = rtts_drop_hash(17u + j, (uint32_t)(blockIdx.x * 512 + threadIdx.x));
        const uint32_t h2 = rtts_drop_hash(91u + j, (uint32_t)(blockIdx.x * 512 + threadIdx.x));
        a[j] = (__bf16)((float)(int)(h1 >> 8) * (1.f / 8388608.f) - 1.f);
        b[j] = (__bf16)((float)(int)(h2 >> 8) * (1.f / 8388608.f) - 1.f);
    }
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 123.456f) sink[lane] = s;          // keeps the chain alive; (practically) never true
}

// rtts_comm_probe: what a ring all-reduce looks like to the CUs -- a SMALL grid of resident workgroups that moves a message
// through the fabric at the pace of a link, not of HBM: each workgroup copies 16 KB pieces (four 16-byte loads in flight per thread,
// +1 on every word, write-through stores) and sleeps `sleep` x ~0.5 us between pieces.  scripts/comm_overlap_probe.py runs it on a second
// stream beside the data-parallel chain of hipGraphs to see how much of such a kernel hides behind the backward and what it costs
// the chain.  Not on the training path.
__global__ __launch_bounds__(256) void comm_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int sleep) {
    for (size_t base = (size_t)blockIdx.x * 1024; base < n16; base += (size_t)gridDim.x * 1024) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = base + k * 256 + threadIdx.x;
            v[k] = i < n16 ? src[i] : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = base + k * 256 + threadIdx.x;
            v[k].x += 1u; v[k].y += 1u; v[k].z += 1u; v[k].w += 1u;
            if (i < n16) rtts_store16_out(dst + i, v[k]);
        }
#pragma unroll 1
        for (int t = 0; t < sleep; ++t) __builtin_amdgcn_s_sleep(16);      // ~1 k cycles = 0.5 us per unit
    }
}

extern "C" int rtts_comm_probe(const void* src, void* dst, int64_t bytes, int workgroups, int sleep, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0 && workgroups > 0 && workgroups <= 256 && sleep >= 0, "rtts_comm_probe: bad arguments");
    hipLaunchKernelGGL(comm_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, (size_t)bytes / 16, sleep);
    RTTS_LAUNCH_CHECK("rtts_comm_probe");
    return 0;
}

// rtts_debug_stamp: one lane writes the 100 MHz wall clock into buf[slot] -- a marker that can be captured into a hipGraph, so that
// the timeline of an UNPROFILED replay can be read back (scripts/replay_stamps.py: when does each branch of the step really start?).
// Not on the training path.
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int slot) {
    if (threadIdx.x == 0) buf[slot] = wall_clock64();
}
extern "C" int rtts_debug_stamp(void* buf, int slot, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(buf && slot >= 0, "rtts_debug_stamp: bad arguments");
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)buf, slot);
    RTTS_LAUNCH_CHECK("rtts_debug_stamp");
    return 0;
}

extern "C" int rtts_peak_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(src && dst && bytes > 0 && bytes % 16 == 0, "rtts_peak_copy: bytes must be a positive multiple of 16");
    static const int grid = [] { const char* e = getenv("RTTS_PEAK_COPY_GRID"); return e ? atoi(e) : 16384; }();     // probe tuning only (1024: 5.27, 8192: 5.59, 32768: 5.67 TB/s on one box)
    hipLaunchKernelGGL(peak_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, (size_t)bytes / 16);
    RTTS_LAUNCH_CHECK("rtts_peak_copy");
    return 0;
}

// FLOP of one launch: workgroups * 8 waves * iters * 8 MFMAs * (16*16*32*2)
extern "C" int rtts_peak_mfma(float* sink, int workgroups, int iters, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(sink && workgroups > 0 && iters > 0, "rtts_peak_mfma: bad arguments");
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, sink, iters);
    RTTS_LAUNCH_CHECK("rtts_peak_mfma");
    return 0;
}
