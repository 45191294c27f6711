// Dense encoder-decoder ("cross") attention, forward and backward, for short key sequences
// (T_k = 128 or 256 keys held entirely on chip; the text side of Reformer-TTS is padded to 256).
//
// Replaces nn.MultiheadAttention as wrapped by MultiheadAttentionWrapper
// (/root/reference/reformer_tts/model/reformer.py:161-186): scores = (q W_q)(k W_k)^T / sqrt(dh),
// key_padding_mask -> -inf, softmax, P V, per head; the four projections stay GEMMs outside.
//
// Same MFMA dataflow as the LSH chunk kernels (lsh_attn_fwd.hip / lsh_attn_bwd.hip):
//   forward : lane = query, S^T = K Q^T, in-register softmax over all keys, O^T = V^T P^T
//   backward: lane = key, wave w owns keys [64w, 64w+64): dV, dK complete in registers for the
//             workgroup's 128 queries; dS^T crosses LDS once for dQ^T = K^T dS^T.  dK/dV of the
//             T_q/128 query blocks are written as partial slabs and summed by rtts_sum_slabs
//             (deterministic, no atomics).
#include "rtts_common.h"
#include <float.h>

#define XA_DH 64
#define XA_ROWB 144
#define XA_QB 128     // queries per workgroup
#ifndef XA_UNROLL
#define XA_UNROLL 1   // query-tile loop of the backward
#endif

typedef __attribute__((ext_vector_type(8))) short xa_short8;

__device__ __forceinline__ bf16x8 xa_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p1);
    const xa_short8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// ------------------------------------------------------------------------------ forward
// LDS images of the backward (and the forward's V image) (same layouts as lsh_attn_bwd.hip, where the bank arithmetic is written out): rows of 128 B
// without padding, 16-byte pieces XOR-swizzled by the row so that ds_read_b128 of one piece from 16 rows and
// ds_read_b64_tr_b16 of 4 consecutive rows x 64 B are both conflict free; the dS^T image [key][128 queries] swizzles its
// 8-byte granules by the key (conflict-free 8-byte stores of 16 consecutive keys and transposed reads of 4 keys x 64 B;
// the padded 272-byte rows were 4-way conflicted on the read side).
__device__ __forceinline__ int xa_sw(int row) { return ((row >> 1) & 3) | ((((row >> 3) ^ (row >> 1)) & 1) << 2); }
__device__ __forceinline__ int xa_off(int row, int piece) { return row * 128 + ((piece ^ xa_sw(row)) << 4); }
__device__ __forceinline__ int xa_ds_off(int key, int gran) {
    const int k0 = key & 1, k1 = (key >> 1) & 1, k2 = (key >> 2) & 1, k3 = (key >> 3) & 1;
    return key * 256 + ((gran ^ ((k1 << 4) | (k0 << 3) | (k1 << 2) | (k2 << 1) | k3)) << 3);
}

// Row staging of the epilogues (lsh_attn_bwd.hip's ab_stg_w / ab_stg_r, where the bank arithmetic is written out): the
// accumulators hold a row (query / key) per lane and 4 dh values per register group, so a lane's natural store is 8 bytes into
// 16 different 128-byte rows -- 16 store instructions per lane that each touch 32 rows.  Through a swizzled [32][128 B] image a
// wave leaves full rows instead: 4 ds_read_b128 + 4 16-byte write-through stores (8 complete rows per instruction).
// (Round 4 also tried the forward with TWO passes over the key tiles -- maximum first, then Q K^T again, exp, row sum, P V --
//  so that a query's 256 logits need not sit in registers: 168 instead of 288 registers, two workgroups per CU instead of one,
//  bit-identical, and SLOWER: 25.2 against 22.2 us.  The kernel is bound by its vector work, not by exposed latency.)
// (and a vector-instruction diet of the same kernel -- padding-only key tiles skipped, all-valid tiles unmasked, scale and maximum in
//  one fma: 22.1 against 22.0 us.  With one four-wave workgroup per CU the kernel's time is the serial chain "64 KB of K and V
//  rows arrive (every CU at once: ~3 us) -> products and softmax (~2.5 us) -> rows leave"; neither fewer instructions nor more
//  resident waves shorten it.  A third variant -- a workgroup loads the K / V images once and works TWO query blocks -- was 29.7
//  against 21.9 us: the load is not what a block's 7.3 us are spent on either.  All three removed again; profiles/r04_xattn_ab.log.)
#ifndef XA_STAGED
#define XA_STAGED 1    // 0: the 8-byte stores of rounds 1-3, for A/B runs
#endif
__device__ __forceinline__ int xa_stg_w(int r, int piece, int hh) { return r * 128 + ((piece ^ (r & 7)) << 4) + ((hh ^ ((r >> 3) & 1)) << 3); }
__device__ __forceinline__ int xa_stg_r(int i, int srow, int spiece) { return (i * 8 + srow) * 128 + ((spiece ^ srow) << 4); }
// acc[dt][4 g + j] of lane (r, hh) = element (row r, dh = 32 dt + 8 g + 4 hh + j), times `mul`  ->  32 rows of 64 bf16 at
// dst + row * ld (16-byte aligned rows), through the wave-private 4 KB image `stg`
__device__ __forceinline__ void xa_store_rows(unsigned char* stg, const f32x16 (&acc)[2], float mul, bf16_t* dst, int64_t ld, int lane) {
    const int r = lane & 31, hh = lane >> 5, srow = lane >> 3, spiece = lane & 7;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul);
            pk.y = pack_bf16x2(acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(stg + xa_stg_w(r, dt * 4 + g, hh)) = pk;
        }
    __builtin_amdgcn_wave_barrier();
    uint4 rowv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(stg + xa_stg_r(i, srow, spiece));
        rowv[i] = (i & 1) ? uint4{v.z, v.w, v.x, v.y} : v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rtts_store16_out(dst + (size_t)(i * 8 + srow) * ld + spiece * 8, rowv[i]);
    __builtin_amdgcn_wave_barrier();
}

template <int TK, bool MULTI>
__global__ __launch_bounds__(256) void xattn_fwd_kernel(const bf16_t* __restrict__ q, int64_t ld_q, const bf16_t* __restrict__ kv,
                                                        int64_t ld_kv, const uint8_t* __restrict__ kvalid, int H, int Tq, int TKtot_,
                                                        bf16_t* __restrict__ o, int64_t ld_o, float* __restrict__ lse,
                                                        uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                        float dscale) {
    // TK = keys held on chip at a time; TKtot (a multiple of TK) keys are walked chunk by chunk with a running maximum and
    // normaliser.  MULTI = false: TKtot == TK is a compile-time fact -- the loop below runs once, the running state folds away
    // and the kernel is the single-pass softmax (two waves per SIMD; the general form needs one wave's worth of registers more)
    const int TKtot = MULTI ? TKtot_ : TK;
    constexpr int NKT = TK / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                       // [TK][128] swizzled (xa_off), like V
    unsigned char* Vs = Ks + TK * 128;
    int* kval = reinterpret_cast<int*>(Vs + TK * 128);

    const int nqb = Tq / XA_QB;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / nqb, qb = wi % nqb;
    const int b = bh / H, h = bh % H;
    const int d = H * XA_DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // Q fragments straight from global: lane (r, hh) holds Q[q0 + r][16 ks + 8 hh .. +8]
    const int qrow = qb * XA_QB + wave * 32 + r;
    const bf16_t* qptr = q + ((size_t)b * Tq + qrow) * ld_q + (size_t)h * XA_DH;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qptr + ks * 16 + 8 * hh);
    if (thresh && seed_dev) seed += seed_dev[0];

    float m_run = -FLT_MAX, l_run = 0.f;
    f32x16 oacc[2] = {{0}, {0}};
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
#pragma unroll 1
    for (int c0 = 0; c0 < TKtot; c0 += TK) {
        const bf16_t* kbase = kv + ((size_t)b * TKtot + c0) * ld_kv + (size_t)h * XA_DH;
        const bf16_t* vbase = kbase + d;
        if (c0) __syncthreads();        // every wave is done with the previous chunk's images
        // K and V rows go global -> LDS by DMA (no staging registers, no ds_write pass): one wave-instruction fills 8 consecutive
        // 128-byte rows in lane order, so the swizzle goes on the SOURCE side (lsh_attn_bwd.hip's gather)
        constexpr int ITERS = TK * 8 / 256;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rowb = it * 32 + wave * 8;                     // wave-uniform: first row of this instruction
            const int row = rowb + (lane >> 3);
            const int lp = (lane & 7) ^ xa_sw(row);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (size_t)row * ld_kv + lp * 8),
                                             (RTTS_LDS void*)(Ks + rowb * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (size_t)row * ld_kv + lp * 8),
                                             (RTTS_LDS void*)(Vs + rowb * 128), 16, 0, 0);
        }
        for (int j = tid; j < TK; j += 256) kval[j] = kvalid ? (int)kvalid[(size_t)b * TKtot + c0 + j] : 1;
        __syncthreads();

        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 acc = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + xa_off(kt * 32 + r, ks * 2 + hh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
            }
            s[kt] = acc;
        }
        float m = -FLT_MAX;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4 kvv = *reinterpret_cast<const int4*>(kval + kt * 32 + 8 * g + 4 * hh);
                const int vv[4] = {kvv.x, kvv.y, kvv.z, kvv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = vv[j] ? s[kt][4 * g + j] * 0.125f : -FLT_MAX;
                    s[kt][4 * g + j] = x;
                    m = fmaxf(m, x);
                }
            }
        m = fmaxf(rtts_xhalf_max(m), m_run);
        // rescale what the earlier chunks left (first chunk: m_run = -FLT_MAX, l_run = 0, oacc = 0; a chunk of masked keys only
        // leaves m = m_run and alpha = 1)
        const float alpha = (m_run == -FLT_MAX) ? 0.f : __expf(m_run - m);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = s[kt][i] == -FLT_MAX ? 0.f : __expf(s[kt][i] - m);
                s[kt][i] = p;
                l += p;
            }
        l_run = l_run * alpha + rtts_xhalf_sum(l);
        m_run = m;
        if (c0) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
        }
        if (thresh) {
            // nn.MultiheadAttention(dropout=p): dropout on the NORMALISED probabilities; the normaliser is the full one, the
            // keep-scale of element (head, query, key) multiplies the unnormalised p before P V
            const uint32_t base = ((uint32_t)bh * (uint32_t)Tq + (uint32_t)qrow) * (uint32_t)TKtot + (uint32_t)c0;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    s[kt][i] *= rtts_drop_keep(seed, base + (uint32_t)(kt * 32 + 8 * (i >> 2) + 4 * hh + (i & 3)), thresh, dscale);
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int o8 = 8 * s2;
                const bf16x8 pf = cvt_bf16x8(s[kt][o8], s[kt][o8 + 1], s[kt][o8 + 2], s[kt][o8 + 3], s[kt][o8 + 4], s[kt][o8 + 5],
                                             s[kt][o8 + 6], s[kt][o8 + 7]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int blk = (kt * 32 + 16 * s2) * 128;
                    const int t0 = xa_off(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);
                    const int t1 = xa_off(4 * hh + trq, (dt ^ 1) * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);   // row + 8: sw flips bit 2
                    const bf16x8 vf = xa_tr_frag(Vs + blk + t0, Vs + blk + 8 * 128 + t1);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
            }
    }
    const float inv_l = 1.f / l_run;
#if XA_STAGED
    __syncthreads();                 // every wave is done with the K image: it becomes the waves' row staging (4 KB each)
    xa_store_rows(Ks + wave * 4096, oacc, inv_l, o + ((size_t)b * Tq + qb * XA_QB + wave * 32) * ld_o + (size_t)h * XA_DH, ld_o, lane);
#else
    bf16_t* optr = o + ((size_t)b * Tq + qrow) * ld_o + (size_t)h * XA_DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(oacc[dt][4 * g] * inv_l, oacc[dt][4 * g + 1] * inv_l);
            pk.y = pack_bf16x2(oacc[dt][4 * g + 2] * inv_l, oacc[dt][4 * g + 3] * inv_l);
            *reinterpret_cast<uint2*>(optr + dt * 32 + 8 * g + 4 * hh) = pk;
        }
#endif
    if (hh == 0) lse[(size_t)bh * Tq + qrow] = m_run + logf(l_run);
}

// ------------------------------------------------------------------------------ backward
// XA_KT2 = 32-key tiles per wave: 2 = one wave per 64 keys (one wave per SIMD at TK = 256), 1 = one wave per 32 keys (two per SIMD,
// the decomposition of lsh_attn_bwd.hip, which runs the same 128 x 256 tile in under half the cycles)
#ifndef XA_KT2
#define XA_KT2 1
#endif
template <int TK, bool DROP>
__global__ __launch_bounds__(TK * 2 / XA_KT2) void xattn_bwd_kernel(const bf16_t* __restrict__ q, int64_t ld_q, const bf16_t* __restrict__ kv,
                                                       int64_t ld_kv, const uint8_t* __restrict__ kvalid,
                                                       const bf16_t* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, int H, int Tq, int TKtot, bf16_t* __restrict__ dq,
                                                       int64_t ld_dq, size_t dq_chunk_stride, bf16_t* __restrict__ dkv_part, int B,
                                                       uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                       float dscale) {
    // blockIdx.y = key chunk: this workgroup owns keys [kc0, kc0 + TK) of TKtot.  P = exp(s - lse) needs only the forward's
    // global lse, so the chunks are independent: dK/dV of the chunk are complete for this query block, dQ is the chunk's
    // share (dq_chunk_stride elements between the chunks' partial dQ arrays; one chunk: dq itself)
    const int kc0 = blockIdx.y * TK;
    dq += (size_t)blockIdx.y * dq_chunk_stride;
    constexpr int KT2 = XA_KT2;
    constexpr int NTHR = TK * 2 / KT2;       // one wave per 32 * KT2 keys
    constexpr int NW = NTHR / 64;
    if (DROP && seed_dev) seed += seed_dev[0];
    constexpr int DSROW = XA_QB * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                        // [TK][128]   swizzled
    unsigned char* Qs = Ks + TK * 128;               // [128][128]  swizzled
    unsigned char* Os = Qs + XA_QB * 128;            // [128][128]  swizzled
    unsigned char* Ds = Os + XA_QB * 128;            // [TK][256]   dS^T (already * scale), swizzled
    float* qlse = reinterpret_cast<float*>(Ds + TK * DSROW);
    float* qdel = qlse + XA_QB;

    const int nqb = Tq / XA_QB;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / nqb, qb = wi % nqb;
    const int b = bh / H, h = bh % H;
    const int d = H * XA_DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    const bf16_t* kbase = kv + ((size_t)b * TKtot + kc0) * ld_kv + (size_t)h * XA_DH;
    const bf16_t* vbase = kbase + d;
    const bf16_t* qbase = q + ((size_t)b * Tq + (size_t)qb * XA_QB) * ld_q + (size_t)h * XA_DH;
    const bf16_t* dobase = dout + ((size_t)b * Tq + (size_t)qb * XA_QB) * ld_do + (size_t)h * XA_DH;

    // images by LDS-DMA (one wave-instruction = 8 rows of 128 B in lane order, the swizzle on the SOURCE side)
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int it = 0; it < TK * 8 / NTHR; ++it) {
            const int rowb = it * (NTHR / 8) + wv * 8, row = rowb + (lane >> 3);
            const int lp = (lane & 7) ^ xa_sw(row);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (size_t)row * ld_kv + lp * 8),
                                             (RTTS_LDS void*)(Ks + rowb * 128), 16, 0, 0);
        }
        for (int rowb = wv * 8; rowb < XA_QB; rowb += NTHR / 8) {
            const int row = rowb + (lane >> 3);
            const int lp = (lane & 7) ^ xa_sw(row);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (size_t)row * ld_q + lp * 8),
                                             (RTTS_LDS void*)(Qs + rowb * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dobase + (size_t)row * ld_do + lp * 8),
                                             (RTTS_LDS void*)(Os + rowb * 128), 16, 0, 0);
        }
    }
    for (int j = tid; j < XA_QB; j += NTHR) {
        qlse[j] = lse[(size_t)bh * Tq + qb * XA_QB + j] * 1.4426950408889634f;   // base-2 softmax: exp2(s * c - lse * log2 e)
        qdel[j] = delta[(size_t)bh * Tq + qb * XA_QB + j];
    }
    int myrow[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) myrow[k2] = (wave * KT2 + k2) * 32 + r;
    bf16x8 vf[KT2][4], kf[KT2][4];
    int kvl[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            vf[k2][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)myrow[k2] * ld_kv + ks * 16 + 8 * hh);
            kf[k2][ks] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)myrow[k2] * ld_kv + ks * 16 + 8 * hh);
        }
        kvl[k2] = kvalid ? (int)kvalid[(size_t)b * TKtot + kc0 + myrow[k2]] : 1;
    }
    __syncthreads();

    f32x16 dvacc[KT2][2], gacc[KT2][2];
#pragma unroll
    for (int a = 0; a < KT2; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            dvacc[a][c2] = (f32x16){0};
            gacc[a][c2] = (f32x16){0};
        }
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
    int fro[4], tro[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fro[ks] = xa_off(r, ks * 2 + hh);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) tro[dt] = xa_off(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);
    int dso[KT2][4];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)
#pragma unroll
        for (int g = 0; g < 4; ++g) dso[k2][g] = xa_ds_off((wave * KT2 + k2) * 32 + r, 2 * g + hh);

#pragma unroll XA_UNROLL
    for (int qt = 0; qt < XA_QB / 32; ++qt) {
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Qs + qt * (32 * 128) + fro[ks]);
            dof[ks] = *reinterpret_cast<const bf16x8*>(Os + qt * (32 * 128) + fro[ks]);
        }
        bf16x8 qtf[2][2], dotf[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int blk = (qt * 32 + 16 * s2) * 128;     // second read: 8 rows on, where sw flips the piece bit dt toggles
                qtf[s2][dt] = xa_tr_frag(Qs + blk + tro[dt], Qs + blk + 8 * 128 + tro[dt ^ 1]);
                dotf[s2][dt] = xa_tr_frag(Os + blk + tro[dt], Os + blk + 8 * 128 + tro[dt ^ 1]);
            }
#pragma unroll
        for (int k2 = 0; k2 < KT2; ++k2) {
            f32x16 sacc = {0}, pacc = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[k2][ks], sacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[k2][ks], pacc, 0, 0, 0);
            }
            float pp[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qt * 32 + 8 * g + 4 * hh;
                const float4 l4 = *reinterpret_cast<const float4*>(qlse + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(qdel + q0);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * g + j;
                    const float p = kvl[k2] ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], 0.125f * 1.4426950408889634f, -lv[j])) : 0.f;
                    // dropout on P: O = (D*P) V  =>  dV += (D*P)^T dO,  dS = P * (D*dP - delta)   (delta = O.dO as ever);
                    // DROP is a template parameter: nn.MultiheadAttention's dropout is 0 in config/baseline.yml
                    const float keep = DROP ? rtts_drop_keep(seed, ((uint32_t)bh * (uint32_t)Tq + (uint32_t)(qb * XA_QB + q0 + j)) *
                                                                         (uint32_t)TKtot + (uint32_t)(kc0 + myrow[k2]), thresh, dscale)
                                              : 1.f;
                    pp[i] = p * keep;
                    ds[i] = p * (pacc[i] * keep - dl[j]) * 0.125f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float* pq = pp + 8 * s2;
                const float* dq_ = ds + 8 * s2;
                const bf16x8 pb = cvt_bf16x8(pq[0], pq[1], pq[2], pq[3], pq[4], pq[5], pq[6], pq[7]);
                const bf16x8 db = cvt_bf16x8(dq_[0], dq_[1], dq_[2], dq_[3], dq_[4], dq_[5], dq_[6], dq_[7]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dvacc[k2][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[s2][dt], pb, dvacc[k2][dt], 0, 0, 0);
                    gacc[k2][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[s2][dt], db, gacc[k2][dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
                pk.y = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
                *reinterpret_cast<uint2*>(Ds + (dso[k2][g] ^ (qt << 6))) = pk;
            }
        }
    }

    // dK | dV partial slab of this query block: layout (nqb, B, TKtot, 2d)
    bf16_t* slab = dkv_part + (((size_t)qb * B + b) * TKtot + kc0) * (size_t)(2 * d) + (size_t)h * XA_DH;
#if XA_STAGED
    __syncthreads();                 // every dS^T tile is in Ds; nobody reads the Q / dout images any more: they become the row staging
    {
        static_assert(2 * XA_QB * 128 >= NW * 4096, "the Q and dout images must hold a 4 KB staging per wave");
        unsigned char* stg = Qs + wave * 4096;
#pragma unroll
        for (int k2 = 0; k2 < KT2; ++k2) {
            // rows of this wave's 32-key tile are consecutive keys: myrow[k2] = first key + r
            bf16_t* dkp = slab + (size_t)(myrow[k2] - r) * (2 * d);
            xa_store_rows(stg, gacc[k2], 1.f, dkp, 2 * d, lane);
            xa_store_rows(stg, dvacc[k2], 1.f, dkp + d, 2 * d, lane);
        }
    }
#else
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        bf16_t* dkp = slab + (size_t)myrow[k2] * (2 * d);
        bf16_t* dvp = dkp + d;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(gacc[k2][dt][4 * g], gacc[k2][dt][4 * g + 1]);
                pk.y = pack_bf16x2(gacc[k2][dt][4 * g + 2], gacc[k2][dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(dkp + dt * 32 + 8 * g + 4 * hh) = pk;
                pk.x = pack_bf16x2(dvacc[k2][dt][4 * g], dvacc[k2][dt][4 * g + 1]);
                pk.y = pack_bf16x2(dvacc[k2][dt][4 * g + 2], dvacc[k2][dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(dvp + dt * 32 + 8 * g + 4 * hh) = pk;
            }
    }
    __syncthreads();
#endif

    // dQ^T[dh][q] = K^T dS^T: the 8 (query tile, dh half) outputs are dealt over the waves
    constexpr int NOUT = (XA_QB / 32) * 2;
    for (int oi = wave; oi < NOUT; oi += NW) {
        const int qt = oi >> 1, dt = oi & 1;
        f32x16 dqa = {0};
        const int rl = 8 * hh + trq;                       // key row inside a 16-key step (second read: +4)
        const int gq = qt * 8 + 4 * trc + trp;             // 8-byte granule of the dS^T row
        const int do0 = xa_ds_off(rl, gq), do1 = xa_ds_off(rl + 4, gq);
        const int kpc = dt * 4 + 2 * trc + (trp >> 1);
        const int ko0 = xa_off(rl, kpc) + 8 * (trp & 1), ko1 = xa_off(rl + 4, kpc) + 8 * (trp & 1);
#pragma unroll
        for (int kb = 0; kb < TK; kb += 16) {
            const bf16x8 bfrag = xa_tr_frag(Ds + kb * DSROW + do0, Ds + kb * DSROW + do1);
            const bf16x8 afrag = xa_tr_frag(Ks + kb * 128 + ko0, Ks + kb * 128 + ko1);
            dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dqa, 0, 0, 0);
        }
        bf16_t* dqp = dq + ((size_t)b * Tq + (size_t)qb * XA_QB + qt * 32 + r) * ld_dq + (size_t)h * XA_DH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(dqa[4 * g], dqa[4 * g + 1]);
            pk.y = pack_bf16x2(dqa[4 * g + 2], dqa[4 * g + 3]);
            *reinterpret_cast<uint2*>(dqp + dt * 32 + 8 * g + 4 * hh) = pk;
        }
    }
}

// out = sum over slabs (bf16 in, fp32 accumulate, bf16 out), 8 elements per thread
__global__ __launch_bounds__(256) void sum_slabs_kernel(const bf16_t* __restrict__ part, int nslabs, size_t n8,
                                                        bf16_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < nslabs; ++s) {
            const uint4 t = reinterpret_cast<const uint4*>(part)[(size_t)s * n8 + i];
            const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += __uint_as_float(u[j] << 16);
                acc[2 * j + 1] += __uint_as_float(u[j] & 0xffff0000u);
            }
        }
        uint4 o;
        o.x = pack_bf16x2(acc[0], acc[1]);
        o.y = pack_bf16x2(acc[2], acc[3]);
        o.z = pack_bf16x2(acc[4], acc[5]);
        o.w = pack_bf16x2(acc[6], acc[7]);
        reinterpret_cast<uint4*>(out)[i] = o;
    }
}

static RttsLdsState g_xa_lds[4][2];
extern "C" int rtts_sum_slabs(const void* part, int nslabs, int64_t n, void* out, void* stream);

static int xa_check(const char* fn, int B, int H, int Tq, int Tk, int dh, int64_t ld_q, int64_t ld_kv) {
    RTTS_REQUIRE(dh == XA_DH, "%s: dh=%d unsupported (this build: 64)", fn, dh);
    RTTS_REQUIRE(Tk >= 128 && Tk % 128 == 0 && Tk <= 2048, "%s: T_k=%d unsupported (a multiple of 128 up to 2048)", fn, Tk);
    RTTS_REQUIRE(Tq > 0 && Tq % XA_QB == 0, "%s: T_q=%d must be a multiple of 128", fn, Tq);
    RTTS_REQUIRE(B > 0 && H > 0, "%s: bad B/H", fn);
    RTTS_REQUIRE(ld_q >= (int64_t)H * dh && ld_q % 8 == 0 && ld_kv >= (int64_t)2 * H * dh && ld_kv % 8 == 0, "%s: bad row strides", fn);
    return 0;
}

// keys held on chip at a time: 256 when T_k is a multiple of it, else 128
static inline int xa_chunk(int Tk) { return Tk % 256 == 0 ? 256 : 128; }
extern "C" int rtts_xattn_key_chunks(int Tk) { return (Tk >= 128 && Tk % 128 == 0) ? Tk / xa_chunk(Tk) : -1; }

extern "C" int rtts_xattn_fwd(const void* q, int64_t ld_q, const void* kv, int64_t ld_kv, const uint8_t* kvalid, int B, int H, int Tq,
                              int Tk, int dh, void* o, int64_t ld_o, float* lse, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                              void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(q && kv && o && lse && drop_p >= 0.f && drop_p < 1.f, "rtts_xattn_fwd: bad arguments");
    if (xa_check("rtts_xattn_fwd", B, H, Tq, Tk, dh, ld_q, ld_kv)) return -1;
    RTTS_REQUIRE(ld_o >= (int64_t)H * dh && ld_o % 8 == 0, "rtts_xattn_fwd: bad ld_o");
    RTTS_REQUIRE((((uintptr_t)q | (uintptr_t)kv | (uintptr_t)o) & 15) == 0, "rtts_xattn_fwd: buffers must be 16-byte aligned");
    const dim3 grid(B * H * (Tq / XA_QB));
    const int chunk = xa_chunk(Tk);
    const size_t lds = 2 * (size_t)chunk * 128 + (size_t)chunk * 4;
#define GO(TK_)                                                                                                           \
    do {                                                                                                                  \
        auto kern = Tk == TK_ ? xattn_fwd_kernel<TK_, false> : xattn_fwd_kernel<TK_, true>;                               \
        RTTS_ENSURE_LDS("rtts_xattn_fwd", kern, lds, g_xa_lds[Tk == TK_ ? 0 : 3][TK_ == 256]);                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, ld_q, (const bf16_t*)kv, ld_kv, \
                           kvalid, H, Tq, Tk, (bf16_t*)o, ld_o, lse, seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p)); \
    } while (0)
    if (chunk == 256) GO(256); else GO(128);
#undef GO
    RTTS_LAUNCH_CHECK("rtts_xattn_fwd");
    return 0;
}

extern "C" int rtts_xattn_bwd(const void* q, int64_t ld_q, const void* kv, int64_t ld_kv, const uint8_t* kvalid, const void* dout,
                              int64_t ld_dout, const float* lse, const float* delta, int B, int H, int Tq, int Tk, int dh, void* dq,
                              int64_t ld_dq, void* dkv_part, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* dq_chunks,
                              void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(q && kv && dout && lse && delta && dq && dkv_part && drop_p >= 0.f && drop_p < 1.f, "rtts_xattn_bwd: bad arguments");
    if (xa_check("rtts_xattn_bwd", B, H, Tq, Tk, dh, ld_q, ld_kv)) return -1;
    RTTS_REQUIRE(ld_dout >= (int64_t)H * dh && ld_dout % 8 == 0 && ld_dq >= (int64_t)H * dh && ld_dq % 8 == 0,
                 "rtts_xattn_bwd: bad strides");
    RTTS_REQUIRE((((uintptr_t)q | (uintptr_t)kv | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dkv_part) & 15) == 0,
                 "rtts_xattn_bwd: buffers must be 16-byte aligned");
    const int chunk = xa_chunk(Tk), nchunks = Tk / chunk;
    // more than one key chunk: each chunk's workgroups write their share of dQ to dq_chunks (nchunks, B*Tq, H*dh) bf16 and
    // the shares are summed into dq (compact rows) by the slab-sum kernel
    RTTS_REQUIRE(nchunks == 1 || (dq_chunks && ld_dq == (int64_t)H * dh && ((uintptr_t)dq_chunks & 15) == 0),
                 "rtts_xattn_bwd: T_k=%d is worked in %d key chunks: needs dq_chunks (%d x B*T_q x H*dh bf16) and ld_dq == H*dh", Tk, nchunks,
                 nchunks);
    const dim3 grid(B * H * (Tq / XA_QB), nchunks);
    const size_t dq_stride = nchunks > 1 ? (size_t)B * Tq * H * dh : 0;
    bf16_t* dq_dst = (bf16_t*)(nchunks > 1 ? dq_chunks : dq);
    const size_t lds = (size_t)chunk * 128 + 2 * (size_t)XA_QB * 128 + (size_t)chunk * (XA_QB * 2) + XA_QB * 8;
#define GO(TK_)                                                                                                           \
    do {                                                                                                                  \
        auto kern = drop_p > 0.f ? xattn_bwd_kernel<TK_, true> : xattn_bwd_kernel<TK_, false>;                            \
        RTTS_ENSURE_LDS("rtts_xattn_bwd", kern, lds, g_xa_lds[1 + (drop_p > 0.f)][TK_ == 256]);                           \
        hipLaunchKernelGGL(kern, grid, dim3(TK_ * 2 / XA_KT2), lds, (hipStream_t)stream, (const bf16_t*)q, ld_q, (const bf16_t*)kv, ld_kv, \
                           kvalid, (const bf16_t*)dout, ld_dout, lse, delta, H, Tq, Tk, dq_dst, ld_dq, dq_stride, (bf16_t*)dkv_part, B, \
                           seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p));                               \
    } while (0)
    if (chunk == 256) GO(256); else GO(128);
#undef GO
    RTTS_LAUNCH_CHECK("rtts_xattn_bwd");
    if (nchunks > 1) return rtts_sum_slabs(dq_chunks, nchunks, (int64_t)dq_stride, dq, stream);
    return 0;
}

extern "C" int rtts_sum_slabs(const void* part, int nslabs, int64_t n, void* out, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(part && out && nslabs > 0 && n > 0 && n % 8 == 0, "rtts_sum_slabs: n must be a positive multiple of 8");
    size_t blocks = ((size_t)n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)part, nslabs,
                       (size_t)n / 8, (bf16_t*)out);
    RTTS_LAUNCH_CHECK("rtts_sum_slabs");
    return 0;
}
