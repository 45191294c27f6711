// Weight-gradient GEMM with split-K over the token dimension:
//     C[N][K] (fp32) += sum_m A[m][N] * B[m][K]        A = dY (M x N, bf16), B = X (M x K, bf16)
// i.e. dW = dY^T X for a Linear layer y = x W^T (the autograd backward of the projections in
// /root/reference/reformer_tts/model/modules.py:195-207, reformer.py:161-217).
//
// Why hand-written: here M = B*T = 12288 tokens is the CONTRACTION and N, K are only 512..2048, so a
// plain tiling has 32..128 output tiles for 256 CUs and a 12288-deep serial loop.  This kernel splits
// the token range over SPLIT workgroups per 128x128 output tile (grid ~ 512), each accumulating a
// partial tile on v_mfma_f32_32x32x16_bf16; partial tiles go to fp32 slabs and a second tiny kernel
// adds them into C in a fixed order (deterministic; no atomics).
//
// Both operands are contracted over their ROW index, so fragments come from row-major LDS images by
// ds_read_b64_tr_b16 (hardware transpose), exactly like the dQ phase of lsh_attn_bwd.hip.
// Staging is register double-buffered: the next 64-row stage is in flight while the current one feeds
// the MFMAs (issue early / write late).
#include "rtts_common.h"

#define GT_BN 128      // output rows  (N)
#define GT_BK 128      // output cols  (K)
#define GT_BM 64       // contraction rows per stage
#define GT_ROWB 272    // LDS row stride in bytes: 128 bf16 + 16 B pad
#define GT_THREADS 256

typedef __attribute__((ext_vector_type(8))) short gt_short8;

__device__ __forceinline__ bf16x8 gt_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p1);
    const gt_short8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

__global__ __launch_bounds__(GT_THREADS, 2) void gemm_tn_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                                const bf16_t* __restrict__ b, int64_t ldb, int M, int N, int K,
                                                                int split, float* __restrict__ out, int64_t ldo,
                                                                size_t slab_stride, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                          // [2][64][272]
    unsigned char* Bs = smem + 2 * GT_BM * GT_ROWB;    // [2][64][272]

    const int tiles_k = K / GT_BK;
    const int tile = blockIdx.x / split, sp = blockIdx.x % split;
    const int n0 = (tile / tiles_k) * GT_BN, k0 = (tile % tiles_k) * GT_BK;
    const int rows_per = M / split;                    // multiple of 64 (checked on the host)
    const int m_begin = sp * rows_per, nstage = rows_per / GT_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;           // 2 x 2 waves, 64 x 64 outputs each
    const int hh = lane >> 5;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;

    // staging map: 64 rows x 16 pieces of 16 B per operand; thread -> (row = it*16 + tid/16, piece = tid%16)
    const int srow = tid >> 4, spiece = tid & 15;
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GT_LOAD_STAGE(stage)                                                                       \
    do {                                                                                           \
        const size_t m_ = (size_t)m_begin + (size_t)(stage) * GT_BM + srow;                        \
        ra0 = *reinterpret_cast<const uint4*>(a + (m_ + 0) * lda + n0 + spiece * 8);               \
        ra1 = *reinterpret_cast<const uint4*>(a + (m_ + 16) * lda + n0 + spiece * 8);              \
        ra2 = *reinterpret_cast<const uint4*>(a + (m_ + 32) * lda + n0 + spiece * 8);              \
        ra3 = *reinterpret_cast<const uint4*>(a + (m_ + 48) * lda + n0 + spiece * 8);              \
        rb0 = *reinterpret_cast<const uint4*>(b + (m_ + 0) * ldb + k0 + spiece * 8);               \
        rb1 = *reinterpret_cast<const uint4*>(b + (m_ + 16) * ldb + k0 + spiece * 8);              \
        rb2 = *reinterpret_cast<const uint4*>(b + (m_ + 32) * ldb + k0 + spiece * 8);              \
        rb3 = *reinterpret_cast<const uint4*>(b + (m_ + 48) * ldb + k0 + spiece * 8);              \
    } while (0)
#define GT_STORE_STAGE(buf)                                                                        \
    do {                                                                                           \
        unsigned char* pa_ = As + ((buf) * GT_BM + srow) * GT_ROWB + spiece * 16;                  \
        unsigned char* pb_ = Bs + ((buf) * GT_BM + srow) * GT_ROWB + spiece * 16;                  \
        *reinterpret_cast<uint4*>(pa_) = ra0;                                                      \
        *reinterpret_cast<uint4*>(pa_ + 16 * GT_ROWB) = ra1;                                       \
        *reinterpret_cast<uint4*>(pa_ + 32 * GT_ROWB) = ra2;                                       \
        *reinterpret_cast<uint4*>(pa_ + 48 * GT_ROWB) = ra3;                                       \
        *reinterpret_cast<uint4*>(pb_) = rb0;                                                      \
        *reinterpret_cast<uint4*>(pb_ + 16 * GT_ROWB) = rb1;                                       \
        *reinterpret_cast<uint4*>(pb_ + 32 * GT_ROWB) = rb2;                                       \
        *reinterpret_cast<uint4*>(pb_ + 48 * GT_ROWB) = rb3;                                       \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    GT_LOAD_STAGE(0);
    GT_STORE_STAGE(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) GT_LOAD_STAGE(s + 1);      // in flight during the MFMAs below
        const unsigned char* Ab = As + buf * GT_BM * GT_ROWB;
        const unsigned char* Bb = Bs + buf * GT_BM * GT_ROWB;
#pragma unroll
        for (int ks = 0; ks < GT_BM / 16; ++ks) {
            const int mr = ks * 16 + 8 * hh + trq;
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ca = (wn * 64 + t * 32 + 16 * trc + 4 * trp) * 2;
                const int cb = (wk * 64 + t * 32 + 16 * trc + 4 * trp) * 2;
                af[t] = gt_tr_frag(Ab + mr * GT_ROWB + ca, Ab + (mr + 4) * GT_ROWB + ca);
                bfr[t] = gt_tr_frag(Bb + mr * GT_ROWB + cb, Bb + (mr + 4) * GT_ROWB + cb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nstage) GT_STORE_STAGE(buf ^ 1);   // other buffer: its readers passed the previous barrier
        __syncthreads();
    }

    // epilogue: C tile rows n = (i&3) + 8*(i>>2) + 4*hh, cols k = lane&31
    float* dst = out + (split > 1 ? (size_t)sp * slab_stride : 0);
    const bool add = (split == 1) && accumulate;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = n0 + wn * 64 + i * 32 + 4 * hh;
            const int kc = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* p = dst + (size_t)(nb + (e & 3) + 8 * (e >> 2)) * ldo + kc;
                *p = add ? *p + acc[i][j][e] : acc[i][j][e];
            }
        }
}

// c[n][k] (+)= sum over slabs, fixed order; 4 floats per thread
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, size_t slab_stride, int N, int K,
                                                          float* __restrict__ c, int64_t ldc, int accumulate) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total4 = (size_t)N * K / 4;
    if (i4 >= total4) return;
    const size_t e = i4 * 4;
    const int n = (int)(e / K), k = (int)(e % K);
    float4 s = *reinterpret_cast<const float4*>(slabs + e);
    for (int t = 1; t < nslabs; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)t * slab_stride + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(c + (size_t)n * ldc + k);
    if (accumulate) {
        const float4 o = *dst;
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *dst = s;
}

static bool g_gt_attr = false;

extern "C" int rtts_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int M, int N, int K, float* c, int64_t ldc,
                            int accumulate, float* slab_ws, int64_t slab_ws_floats, void* stream) {
    RTTS_REQUIRE(a && b && c, "rtts_gemm_tn: null pointer");
    RTTS_REQUIRE(N > 0 && K > 0 && M > 0 && N % GT_BN == 0 && K % GT_BK == 0 && M % GT_BM == 0,
                 "rtts_gemm_tn: need N %% 128 == 0, K %% 128 == 0, M %% 64 == 0 (got M=%d N=%d K=%d)", M, N, K);
    RTTS_REQUIRE(lda >= N && ldb >= K && ldc >= K && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "rtts_gemm_tn: bad leading dimensions");
    RTTS_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0, "rtts_gemm_tn: buffers must be 16-byte aligned");
    const int tiles = (N / GT_BN) * (K / GT_BK);
    // split the token range so that the grid is ~2 workgroups per CU, each with >= 4 stages
    int split = 1;
    const int stages = M / GT_BM;
    while (tiles * split < 512 && split * 2 <= stages / 4 && stages % (split * 2) == 0) split *= 2;
    const size_t slab = (size_t)N * K;
    if (split > 1 && (!slab_ws || (size_t)slab_ws_floats < slab * split)) {
        RTTS_REQUIRE(slab_ws != nullptr, "rtts_gemm_tn: split-K needs a slab workspace");
        while (split > 1 && (size_t)slab_ws_floats < slab * split) split /= 2;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 4 * GT_BM * GT_ROWB;
    if (!g_gt_attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_gt_attr = true;
    }
    if (split == 1) {
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles), dim3(GT_THREADS), lds, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, M, N, K, 1,
                           c, ldc, (size_t)0, accumulate);
    } else {
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * split), dim3(GT_THREADS), lds, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, M, N,
                           K, split, slab_ws, (int64_t)K, slab, 0);
        const unsigned blocks = (unsigned)((slab / 4 + 255) / 256);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab_ws, split, slab, N, K, c, ldc, accumulate);
    }
    RTTS_LAUNCH_CHECK("rtts_gemm_tn");
    return 0;
}
