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
//
// GROUPED launches: the weight gradients of one reversible layer (7 of them in a decoder layer) are leaves of the
// backward -- nothing reads them before the all-reduce/optimizer -- so the executor defers them and launches them
// together.  One grid then holds 240 tiles instead of 16..64, the split factor drops from 8..32 to 2, and the fp32
// slab traffic (split x output size, written and read back) drops ~7x; per-launch ramp-up and tails are paid once.
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

struct GtProb {
    const bf16_t* a;
    const bf16_t* b;
    float* out;          // split == 1: C itself; else this problem's slab area
    float* c;
    int64_t lda, ldb, ldo, ldc;
    size_t slab_stride;  // N*K
    int M, N, K, split, accumulate;
    int wg_start;        // first workgroup of this problem in the main grid
    int rb_start;        // first block of this problem in the slab-reduce grid
};
struct GtGroup {
    GtProb p[RTTS_GEMM_TN_MAX_GROUP];
    int n;
};

__global__ __launch_bounds__(GT_THREADS, 2) void gemm_tn_kernel(const GtGroup grp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                          // [2][64][272]
    unsigned char* Bs = smem + 2 * GT_BM * GT_ROWB;    // [2][64][272]

    int pi = 0;
#pragma unroll
    for (int i = 1; i < RTTS_GEMM_TN_MAX_GROUP; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.p[i].wg_start) pi = i;
    const GtProb& P = grp.p[pi];
    const bf16_t* __restrict__ a = P.a;
    const bf16_t* __restrict__ b = P.b;
    float* __restrict__ out = P.out;
    const int64_t lda = P.lda, ldb = P.ldb, ldo = P.ldo;
    const int M = P.M, K = P.K, split = P.split, accumulate = P.accumulate;
    const size_t slab_stride = P.slab_stride;
    const int blk = (int)blockIdx.x - P.wg_start;

    const int tiles_k = K / GT_BK;
    const int tile = blk / split, sp = blk % split;
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

// c[n][k] (+)= sum over slabs, fixed order; 4 floats per thread; one launch for every split problem of the group
__global__ __launch_bounds__(256) void slab_reduce_kernel(const GtGroup grp) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < RTTS_GEMM_TN_MAX_GROUP; ++i)
        if (i < grp.n && (int)blockIdx.x >= grp.p[i].rb_start) pi = i;
    const GtProb& P = grp.p[pi];
    if (P.split == 1) return;                  // wrote C directly (its rb range is empty anyway)
    const size_t i4 = (size_t)((int)blockIdx.x - P.rb_start) * blockDim.x + threadIdx.x;
    const size_t total4 = P.slab_stride / 4;
    if (i4 >= total4) return;
    const size_t e = i4 * 4;
    const int n = (int)(e / P.K), k = (int)(e % P.K);
    const float* __restrict__ slabs = P.out;
    float4 s = *reinterpret_cast<const float4*>(slabs + e);
    for (int t = 1; t < P.split; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)t * P.slab_stride + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(P.c + (size_t)n * P.ldc + k);
    if (P.accumulate) {
        const float4 o = *dst;
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *dst = s;
}

static bool g_gt_attr = false;

extern "C" int rtts_gemm_tn_grouped(const rtts_gemm_tn_problem* problems, int n, float* slab_ws, int64_t slab_ws_floats, void* stream) {
    RTTS_REQUIRE(problems && n > 0 && n <= RTTS_GEMM_TN_MAX_GROUP, "rtts_gemm_tn_grouped: 1..%d problems", RTTS_GEMM_TN_MAX_GROUP);
    GtGroup grp;
    grp.n = n;
    int total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        const rtts_gemm_tn_problem& q = problems[i];
        RTTS_REQUIRE(q.a && q.b && q.c, "rtts_gemm_tn: null pointer");
        RTTS_REQUIRE(q.N > 0 && q.K > 0 && q.M > 0 && q.N % GT_BN == 0 && q.K % GT_BK == 0 && q.M % GT_BM == 0,
                     "rtts_gemm_tn: need N %% 128 == 0, K %% 128 == 0, M %% 64 == 0 (got M=%d N=%d K=%d)", q.M, q.N, q.K);
        RTTS_REQUIRE(q.lda >= q.N && q.ldb >= q.K && q.ldc >= q.K && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0,
                     "rtts_gemm_tn: bad leading dimensions");
        RTTS_REQUIRE((((uintptr_t)q.a | (uintptr_t)q.b | (uintptr_t)q.c) & 15) == 0, "rtts_gemm_tn: buffers must be 16-byte aligned");
        total_tiles += (q.N / GT_BN) * (q.K / GT_BK);
    }
    // one split target for the group: ~2 workgroups per CU over all problems, each workgroup with >= 4 stages
    int want = 1;
    while (total_tiles * want * 2 <= 640) want *= 2;
    int wg = 0, rb = 0;
    size_t slab_used = 0;
    bool any_split = false;
    for (int i = 0; i < n; ++i) {
        const rtts_gemm_tn_problem& q = problems[i];
        GtProb& P = grp.p[i];
        const int tiles = (q.N / GT_BN) * (q.K / GT_BK), stages = q.M / GT_BM;
        const size_t slab = (size_t)q.N * q.K;
        int split = 1;
        while (split < want && split * 2 <= stages / 4 && stages % (split * 2) == 0) split *= 2;
        while (split > 1 && (!slab_ws || slab_used + slab * split > (size_t)slab_ws_floats)) split /= 2;
        P.a = (const bf16_t*)q.a; P.b = (const bf16_t*)q.b; P.c = q.c;
        P.lda = q.lda; P.ldb = q.ldb; P.ldc = q.ldc;
        P.M = q.M; P.N = q.N; P.K = q.K; P.split = split; P.accumulate = q.accumulate;
        P.slab_stride = slab;
        P.wg_start = wg;
        P.rb_start = rb;
        wg += tiles * split;
        if (split > 1) {
            P.out = slab_ws + slab_used;
            P.ldo = q.K;
            slab_used += slab * split;
            rb += (int)((slab / 4 + 255) / 256);
            any_split = true;
        } else {
            P.out = q.c;
            P.ldo = q.ldc;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 4 * GT_BM * GT_ROWB;
    if (!g_gt_attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_gt_attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(wg), dim3(GT_THREADS), lds, s, grp);
    if (any_split) hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, grp);
    RTTS_LAUNCH_CHECK("rtts_gemm_tn");
    return 0;
}

extern "C" int rtts_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int M, int N, int K, float* c, int64_t ldc,
                            int accumulate, float* slab_ws, int64_t slab_ws_floats, void* stream) {
    rtts_gemm_tn_problem q;
    q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.c = c; q.ldc = ldc;
    q.M = M; q.N = N; q.K = K; q.accumulate = accumulate;
    return rtts_gemm_tn_grouped(&q, 1, slab_ws, slab_ws_floats, stream);
}
