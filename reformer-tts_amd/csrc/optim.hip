// Flat-buffer optimizer kernels: gradient sum of squares (for the global-norm clip) and the
// AdamW update, one launch each over ALL parameters (they live in one contiguous fp32 buffer).
//
// Replaces, for the training step of SURVEY.md section 8f rank 1: torch.nn.utils.clip_grad_norm_
// (via pytorch-lightning's gradient_clip_val, /root/reference/reformer_tts/training/train.py:77-89)
// and transformers.optimization.AdamW.step as configured at
// /root/reference/reformer_tts/training/wrappers.py:240-256,284-297:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v)+eps)
//   then, for decayed parameters, p -= lr wd p   (decoupled, applied to the updated p).
// Pure HBM streaming: 16 B (+1 B mask) read and 12 B written per parameter.
#include "rtts_common.h"

#define OPT_THREADS 256
#define OPT_MAX_BLOCKS 2048

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    float s = 0.f;
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = g4[i];
        s = __builtin_fmaf(x.x, x.x, s);
        s = __builtin_fmaf(x.y, x.y, s);
        s = __builtin_fmaf(x.z, x.z, s);
        s = __builtin_fmaf(x.w, x.w, s);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float x = g[(n4 << 2) + threadIdx.x];
        s = __builtin_fmaf(x, x, s);
    }
    __shared__ float red[OPT_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// scale[0] = grad_mult * min(1, max_norm / (grad_mult * |g| + 1e-6)) (max_norm <= 0: no clip); scale[1] = grad_mult*|g|
__global__ __launch_bounds__(OPT_THREADS) void clip_scale_kernel(const float* __restrict__ partial, int nblocks, float grad_mult,
                                                                 float max_norm, float* __restrict__ scale) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += OPT_THREADS) s += partial[i];
    __shared__ float red[OPT_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]) * grad_mult;
        float c = 1.f;
        if (max_norm > 0.f) c = fminf(1.f, max_norm / (norm + 1e-6f));
        scale[0] = grad_mult * c;
        scale[1] = norm;
    }
}

// four parameters per thread (16-byte accesses); the bf16 mirror the next forward's GEMMs read is written in the same
// pass (mirror may be null), which saves re-reading the 108 MB of master weights in a separate cast launch
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, const uint8_t* __restrict__ decay, size_t n4,
                                                            const float* __restrict__ scale, const float* __restrict__ hyper,
                                                            float b1, float b2, float eps, float wd, bf16_t* __restrict__ mirror) {
    // hyper = {lr, lr * sqrt(1-b2^t) / (1-b1^t)} lives in device memory so that a captured graph replays with fresh values
    const float gs = scale ? scale[0] : 1.f;
    const float lr = hyper[0], step_size = hyper[1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 g4 = reinterpret_cast<const float4*>(g)[i];
        const float4 m4 = reinterpret_cast<const float4*>(m)[i];
        const float4 v4 = reinterpret_cast<const float4*>(v)[i];
        const float4 p4 = reinterpret_cast<const float4*>(p)[i];
        const uint32_t d4 = reinterpret_cast<const uint32_t*>(decay)[i];
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
        float mo[4], vo[4], po[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = gg[j] * gs;
            mo[j] = b1 * mm[j] + (1.f - b1) * gi;
            vo[j] = b2 * vv[j] + (1.f - b2) * gi * gi;
            float pi = pp[j] - step_size * mo[j] / (sqrtf(vo[j]) + eps);
            if ((d4 >> (8 * j)) & 0xffu) pi -= lr * wd * pi;
            po[j] = pi;
        }
        reinterpret_cast<float4*>(m)[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
        reinterpret_cast<float4*>(p)[i] = make_float4(po[0], po[1], po[2], po[3]);
        if (mirror) {
            uint2 o;
            o.x = pack_bf16x2(po[0], po[1]);
            o.y = pack_bf16x2(po[2], po[3]);
            reinterpret_cast<uint2*>(mirror)[i] = o;
        }
    }
}

extern "C" int rtts_grad_clip_scale(const float* grads, int64_t n, float grad_mult, float max_norm, float* partial_ws,
                                    float* scale_out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(grads && partial_ws && scale_out && n > 0, "rtts_grad_clip_scale: bad arguments");
    RTTS_REQUIRE(((uintptr_t)grads & 15) == 0, "rtts_grad_clip_scale: grads must be 16-byte aligned");
    int blocks = (int)((n / 4 + OPT_THREADS - 1) / OPT_THREADS);
    if (blocks < 1) blocks = 1;
    if (blocks > OPT_MAX_BLOCKS) blocks = OPT_MAX_BLOCKS;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, grads, (size_t)n, partial_ws);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, partial_ws, blocks, grad_mult,
                       max_norm, scale_out);
    RTTS_LAUNCH_CHECK("rtts_grad_clip_scale");
    return 0;
}

extern "C" int rtts_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                               int64_t n, const float* scale, const float* hyper, float beta1, float beta2, float eps,
                               float weight_decay, void* bf16_mirror, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(params && grads && exp_avg && exp_avg_sq && decay_mask && hyper && n > 0, "rtts_adamw_step: bad arguments");
    RTTS_REQUIRE(n % 4 == 0, "rtts_adamw_step: n must be a multiple of 4 (pad the flat buffers)");
    RTTS_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                     ((uintptr_t)decay_mask & 3) == 0 && ((uintptr_t)bf16_mirror & 7) == 0,
                 "rtts_adamw_step: buffers must be 16-byte aligned (decay mask 4, mirror 8)");
    int blocks = (int)((n / 4 + OPT_THREADS - 1) / OPT_THREADS);
    if (blocks > OPT_MAX_BLOCKS) blocks = OPT_MAX_BLOCKS;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                       decay_mask, (size_t)n / 4, scale, hyper, beta1, beta2, eps, weight_decay, (bf16_t*)bf16_mirror);
    RTTS_LAUNCH_CHECK("rtts_adamw_step");
    return 0;
}
