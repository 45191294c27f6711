// Shared helpers for the gfx950 kernels of librtts_hip.so (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/rtts.h"

typedef uint16_t bf16_t;  // storage type of a bfloat16
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;    // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define RTTS_LDS __attribute__((address_space(3)))

extern "C" void rtts_set_error(const char* fmt, ...);
// test-only overrides of the LSH kernels' run lengths (rtts_debug_set_walk, rtts_api.cpp): -1 = the library's own pick
int rtts_walk_override(int backward);

#define RTTS_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            rtts_set_error(__VA_ARGS__);   \
            return -1;                     \
        }                                  \
    } while (0)

#define RTTS_LAUNCH_CHECK(name)                                              \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            int dev_ = -1;                                                   \
            (void)hipGetDevice(&dev_);                                       \
            rtts_set_error("%s: launch failed: %s (current device %d; the entry points launch on the caller's current device: " \
                           "the stream and every pointer must belong to it)", name, hipGetErrorString(e_), dev_); \
            return -2;                                                       \
        }                                                                    \
    } while (0)

// DEVICE OF A CALL.  An entry point is called from the Python main thread in a forward and from PyTorch's
// autograd worker thread in a backward; HIP's current device is per thread.  Every launching entry point therefore binds the
// calling thread to the device its stream belongs to before it touches the runtime: the explicit `int device` of the ABI is
// carried by the stream handle (hipStreamGetDevice), so a caller cannot pass a stream of one device and the ordinal of another.
// The null stream has no device of its own: the thread's current device is used, as everywhere in HIP.
static inline int rtts_bind_device(void* stream) {
    if (!stream) return 0;
    hipDevice_t dev = -1;
    if (hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess) {
        (void)hipGetLastError();
        rtts_set_error("the stream handle %p does not belong to any device of this process", stream);
        return -3;
    }
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == (int)dev) return 0;
    if (hipSetDevice((int)dev) != hipSuccess) {
        (void)hipGetLastError();
        rtts_set_error("cannot bind the calling thread to device %d (the device of the stream)", (int)dev);
        return -3;
    }
    return 0;
}
#define RTTS_ENTER(stream)                              \
    do {                                                \
        const int rc_enter_ = rtts_bind_device(stream); \
        if (rc_enter_) return rc_enter_;                \
    } while (0)

// The dynamic-LDS limit is an attribute of a LOADED function, i.e. per device: raise it once per (function, device), again when
// a later call needs more, and report a failure at the call that caused it instead of at some later launch.
// Forwards (Python main thread) and backwards (autograd's worker thread) launch concurrently: the per-device record is atomic,
// and two threads that both find it too small both raise the limit (idempotent) -- no launch can run ahead of its attribute.
struct RttsLdsState { std::atomic<size_t> set[64]; };
static inline hipError_t rtts_ensure_lds(const void* func, size_t lds, RttsLdsState& st) {
    if (lds <= 64 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (st.set[dev].load(std::memory_order_acquire) >= lds) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        size_t cur = st.set[dev].load(std::memory_order_relaxed);
        while (cur < lds && !st.set[dev].compare_exchange_weak(cur, lds, std::memory_order_release, std::memory_order_relaxed)) {}
    }
    return e;
}
#define RTTS_ENSURE_LDS(name, func, lds, state)                                                                   \
    do {                                                                                                          \
        const hipError_t e_lds_ = rtts_ensure_lds(reinterpret_cast<const void*>(func), (lds), (state));           \
        RTTS_REQUIRE(e_lds_ == hipSuccess, "%s: cannot raise the dynamic LDS limit to %zu bytes: %s", name, (size_t)(lds), \
                     hipGetErrorString(e_lds_));                                                                  \
    } while (0)

// Output rows of the compute-bound kernels leave as 16-byte WRITE-THROUGH stores (global_store_dwordx4 ... sc1): the bytes go out to
// the fabric while the kernel still computes, instead of sitting in the XCD's L2 as dirty lines that are written back at the kernel
// boundary (B / 6 TB/s on top of every boundary: MI355X_MICROARCH.md price list).  Nobody re-reads them from this L2: the consumer is
// the next launch, whose workgroups sit on all eight XCDs.  -DRTTS_WT_STORES=0 builds plain stores (A/B runs: scripts/ab_attn.py).
#ifndef RTTS_WT_STORES
#define RTTS_WT_STORES 1
#endif
typedef int rtts_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rtts_store16_out(void* p, const uint4 v) {
#if RTTS_WT_STORES
    const rtts_v4i w = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    // s_nop 1 INSIDE the statement: the hardware needs wait states between a store of more than 64 bits and a VALU write of its data
    // registers; hipcc pads its own stores, not an asm statement's (cdna_hip_programming.md 5.7: without it the next instruction
    // may overwrite the data before the store has read it -- seen as NaN rows in the forward's output)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(bf16_t, b);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// two floats -> one dword of bf16 (one v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// eight floats -> one MFMA fragment (four v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x8 cvt_bf16x8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    const f32x8 v = {a0, a1, a2, a3, a4, a5, a6, a7};
    return __builtin_convertvector(v, bf16x8);
}

// Cross-lane reductions without the LDS crossbar (ds_bpermute costs an LDS round trip per step):
// sum over each aligned group of 8 lanes, result in all 8 (two quad_perm DPP adds + one row_half_mirror)
__device__ __forceinline__ float rtts_sum8(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // lane ^ 1
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // lane ^ 2
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // 7 - lane: the other quad
    return x;
}
// lane l and lane l ^ 32 combined: v_permlane32_swap (gfx950) leaves {x[l & 31], x[32 + (l & 31)]} in the two results
__device__ __forceinline__ float rtts_xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float rtts_xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Partial-gradient slots of the LSH attention backward (lsh_attn_bwd.hip writes them, lsh_combine.hip's reduce sums them;
// callers size dqk_part by rtts_lsh_bwd_qk_slots()).
#define RTTS_LSH_BWD_QK_SLOTS 2

// XCD-aware bijective remap of a linear workgroup id: the hardware deals workgroups
// round-robin over the 8 XCDs, so ids congruent mod 8 share an L2.  Give each XCD a
// contiguous range of work items so that neighbours (which share gathered rows) hit in L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t id, uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, x = id & 7u, i = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Counter-based dropout: element idx of a tensor is kept iff hash(seed, idx) >= thresh, thresh = p * 2^32; the same
// (seed, idx) gives the same decision in the forward, in the reversible reconstruction and in the backward, so no
// mask is ever stored.  seed = host constant + a device word the trainer rewrites per step (graph replays draw fresh masks).
// The seed passes through a finalizer of its own and enters the index hash TWICE (added in front, xor-ed in the middle):
// the host's per-site seeds are multiples of the index multiplier, and with a single additive entry the masks of two
// sites were shifted copies of one sequence (mask_{k+1}[i] == mask_k[i+1]); a keyed double entry has no such alias.
__device__ __forceinline__ uint32_t rtts_drop_hash(uint32_t seed, uint32_t idx) {
    uint32_t s = seed;
    s ^= s >> 16; s *= 0x85ebca6bu; s ^= s >> 13; s *= 0xc2b2ae35u; s ^= s >> 16;
    uint32_t x = idx * 0x9E3779B1u + s;
    x ^= x >> 16; x *= 0x7feb352du; x ^= (s << 13) | (s >> 19); x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// keep-scale of element idx: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float rtts_drop_keep(uint32_t seed, uint32_t idx, uint32_t thresh, float scale) {
    return rtts_drop_hash(seed, idx) >= thresh ? scale : 0.f;
}
static inline uint32_t rtts_drop_thresh(float p) { return p <= 0.f ? 0u : (uint32_t)((double)p * 4294967296.0); }

