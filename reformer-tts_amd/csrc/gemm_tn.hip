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
#include <stdlib.h>

#define GT_BM 64       // contraction rows per stage
#ifndef GT_PAD
#define GT_PAD 64      // bytes of padding per LDS row: a row shift of 16 banks, so the 4 rows x 16 dwords of a 32-lane
#endif                 // transposed read (ds_read_b64_tr_b16 is served in two 32-lane groups over 64 banks) hit 64 distinct
                       // banks; 32 B of pad left them 2-way conflicted (130.6 -> 127.7 us per decoder-layer group)

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

// WN x WK waves, each owning a register tile of TN x TK MFMA tiles (32 x 32 each): the workgroup tile is
// (32 TN WN) x (32 TK WK).  Every MFMA needs 2 KB of operand fragments from LDS; a TN x TK register tile reuses each
// fragment, so the LDS bytes per MFMA are (TN + TK) / (TN TK) KB against an LDS rate of 128 B/clk/CU and an MFMA
// rate of one per 8 clk per CU:
//   <2,2,2,2>: 128 x 128, 256 threads, 2 workgroups per CU -- any shape that tiles by 128; 1 KB per MFMA = LDS-bound
//              (measured ~600 TFLOP/s, a 4 x 4-wave 256-tile variant with the same 2 x 2 register tile just as slow)
//   <4,2,2,4>: 256 x 256, 512 threads, 1 workgroup per CU, 128 accumulator registers -- 0.75 KB per MFMA
template <int WN, int WK, int TN, int TK>
__global__ __launch_bounds__(64 * WN * WK) void gemm_tn_kernel(const GtGroup grp) {
    constexpr int BN = 32 * TN * WN, BK = 32 * TK * WK, NTHR = 64 * WN * WK;
    constexpr int ROWA = BN * 2 + GT_PAD, ROWB = BK * 2 + GT_PAD;       // LDS row strides (bytes)
    constexpr int PA = BN / 8, PB = BK / 8;                            // 16-byte pieces per staged row
    constexpr int ITA = GT_BM * PA / NTHR, ITB = GT_BM * PB / NTHR;    // staging loads per thread and operand
    static_assert(GT_BM * PA % NTHR == 0 && GT_BM * PB % NTHR == 0, "staging must tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                          // [2][64][ROWA]
    unsigned char* Bs = smem + 2 * GT_BM * ROWA;       // [2][64][ROWB]

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

    const int tiles_k = K / BK;
    const int tile = blk / split, sp = blk % split;
    const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
    // the token range is cut into `split` runs of whole stages, the first (stages % split) one stage longer: any split
    // factor works, not only divisors of the stage count (M = 12544 halo rows = 196 stages)
    const int st_all = M / GT_BM, st_base = st_all / split, st_rem = st_all % split;
    const int nstage = st_base + (sp < st_rem ? 1 : 0);
    const int m_begin = (sp * st_base + min(sp, st_rem)) * GT_BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave % WK;
    const int hh = lane >> 5;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;

    // staging map per operand: thread -> (row = it * (NTHR / P) + tid / P, piece = tid % P)
    const int arow = tid / PA, apiece = tid % PA, brow = tid / PB, bpiece = tid % PB;
    static_assert(ITA <= 4 && ITB <= 4, "staging registers");
    // named scalars, not arrays: arrays written under `if (s + 1 < nstage)` and read after the MFMAs end up in scratch
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GT_LD(reg, base, ld, step, it, IT) \
    if constexpr ((it) < (IT)) reg = *reinterpret_cast<const uint4*>((base) + (size_t)((it) * (step)) * (ld))
#define GT_ST(reg, base, bstep, it, IT) \
    if constexpr ((it) < (IT)) *reinterpret_cast<uint4*>((base) + (it) * (bstep)) = reg
#define GT_LOAD_STAGE(stage)                                                              \
    do {                                                                                  \
        const size_t m_ = (size_t)m_begin + (size_t)(stage) * GT_BM;                      \
        const bf16_t* pa_ = a + (m_ + arow) * lda + n0 + apiece * 8;                      \
        const bf16_t* pb_ = b + (m_ + brow) * ldb + k0 + bpiece * 8;                      \
        GT_LD(ra0, pa_, lda, NTHR / PA, 0, ITA); GT_LD(ra1, pa_, lda, NTHR / PA, 1, ITA); \
        GT_LD(ra2, pa_, lda, NTHR / PA, 2, ITA); GT_LD(ra3, pa_, lda, NTHR / PA, 3, ITA); \
        GT_LD(rb0, pb_, ldb, NTHR / PB, 0, ITB); GT_LD(rb1, pb_, ldb, NTHR / PB, 1, ITB); \
        GT_LD(rb2, pb_, ldb, NTHR / PB, 2, ITB); GT_LD(rb3, pb_, ldb, NTHR / PB, 3, ITB); \
    } while (0)
#define GT_STORE_STAGE(buf)                                                               \
    do {                                                                                  \
        unsigned char* pa_ = As + ((buf) * GT_BM + arow) * ROWA + apiece * 16;            \
        unsigned char* pb_ = Bs + ((buf) * GT_BM + brow) * ROWB + bpiece * 16;            \
        GT_ST(ra0, pa_, (NTHR / PA) * ROWA, 0, ITA); GT_ST(ra1, pa_, (NTHR / PA) * ROWA, 1, ITA); \
        GT_ST(ra2, pa_, (NTHR / PA) * ROWA, 2, ITA); GT_ST(ra3, pa_, (NTHR / PA) * ROWA, 3, ITA); \
        GT_ST(rb0, pb_, (NTHR / PB) * ROWB, 0, ITB); GT_ST(rb1, pb_, (NTHR / PB) * ROWB, 1, ITB); \
        GT_ST(rb2, pb_, (NTHR / PB) * ROWB, 2, ITB); GT_ST(rb3, pb_, (NTHR / PB) * ROWB, 3, ITB); \
    } while (0)

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = (f32x16){0};

    GT_LOAD_STAGE(0);
    GT_STORE_STAGE(0);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) GT_LOAD_STAGE(s + 1);      // in flight during the MFMAs below
        const unsigned char* Ab = As + buf * GT_BM * ROWA;
        const unsigned char* Bb = Bs + buf * GT_BM * ROWB;
        // fragments of step ks+1 are requested before the MFMAs of step ks are issued: without this the wave sits
        // through a full LDS round trip between every two MFMA bursts
#define GT_FRAGS(AF, BF, ks_)                                                                   \
    do {                                                                                        \
        const int mr_ = (ks_) * 16 + 8 * hh + trq;                                              \
        _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                        \
            const int ca_ = (wn * 32 * TN + t * 32 + 16 * trc + 4 * trp) * 2;                   \
            AF[t] = gt_tr_frag(Ab + mr_ * ROWA + ca_, Ab + (mr_ + 4) * ROWA + ca_);             \
        }                                                                                       \
        _Pragma("unroll") for (int t = 0; t < TK; ++t) {                                        \
            const int cb_ = (wk * 32 * TK + t * 32 + 16 * trc + 4 * trp) * 2;                   \
            BF[t] = gt_tr_frag(Bb + mr_ * ROWB + cb_, Bb + (mr_ + 4) * ROWB + cb_);             \
        }                                                                                       \
    } while (0)
#define GT_MFMAS(AF, BF)                                                                        \
    _Pragma("unroll") for (int i = 0; i < TN; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < TK; ++j)                                          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i], BF[j], acc[i][j], 0, 0, 0)
        bf16x8 af0[TN], bf0[TK], af1[TN], bf1[TK];
        // (sched_barrier pins the order: the machine scheduler otherwise sinks every fragment load back to its use)
        GT_FRAGS(af0, bf0, 0);
        GT_FRAGS(af1, bf1, 1);
        __builtin_amdgcn_sched_barrier(0);
        GT_MFMAS(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        GT_FRAGS(af0, bf0, 2);
        __builtin_amdgcn_sched_barrier(0);
        GT_MFMAS(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        GT_FRAGS(af1, bf1, 3);
        __builtin_amdgcn_sched_barrier(0);
        GT_MFMAS(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        GT_MFMAS(af1, bf1);
        static_assert(GT_BM == 64, "four k-steps per stage");
        if (s + 1 < nstage) GT_STORE_STAGE(buf ^ 1);   // other buffer: its readers passed the previous barrier
        __syncthreads();
    }
#undef GT_LOAD_STAGE
#undef GT_STORE_STAGE
#undef GT_LD
#undef GT_ST
#undef GT_FRAGS
#undef GT_MFMAS

    // epilogue: C tile rows n = (i&3) + 8*(i>>2) + 4*hh, cols k = lane&31
    float* dst = out + (split > 1 ? (size_t)sp * slab_stride : 0);
    const bool add = (split == 1) && accumulate;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int nb = n0 + wn * 32 * TN + i * 32 + 4 * hh;
            const int kc = k0 + wk * 32 * TK + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* p = dst + (size_t)(nb + (e & 3) + 8 * (e >> 2)) * ldo + kc;
                *p = add ? *p + acc[i][j][e] : acc[i][j][e];
            }
        }
}

// ---- ring variant of the 256 x 256 kernel: operands by LDS-DMA into a 4-deep ring of 32-row stages -----------------
// The register double buffer above has ONE 64-row stage (64 KB) in flight against a load latency that, with every CU
// streaming, is longer than the stage's MFMA time, so each stage ends in a stall.  Here global_load_lds_dwordx4 writes the
// stages straight into LDS (no staging registers, no ds_write pass); three stages (96 KB) are in flight while one is
// consumed, a wave waits with a COUNTED s_waitcnt vmcnt(N) for its own pieces of the oldest stage and a raw s_barrier (no
// vmcnt(0) drain) publishes it.  The images are unpadded (a DMA instruction writes 1 KB = 2 rows contiguously), the 64-byte
// chunks of a row XOR-swizzled with (row & 3) on the SOURCE side so that the transposed fragment reads (4 consecutive rows
// x 64 B) stay bank-conflict free.
#define GT_RS 32       // rows per ring stage
#ifndef GT_NST
#define GT_NST 4       // ring depth
#endif

template <int WN, int WK, int TN, int TK>
__global__ __launch_bounds__(64 * WN * WK) void gemm_tn_ring_kernel(const GtGroup grp) {
    constexpr int BN = 32 * TN * WN, BK = 32 * TK * WK, NTHR = 64 * WN * WK, NWAVE = WN * WK;
    static_assert(BN == 256 && BK == 256 && NWAVE == 8, "ring variant: 256 x 256 tile, 8 waves");
    constexpr int ROWB = 512;                                  // bytes per LDS row of either operand
    constexpr int STAGE = GT_RS * ROWB;                        // 16 KB per operand and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                                  // [GT_NST][32][512]
    unsigned char* Bs = smem + GT_NST * STAGE;

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
    const int tiles_k = K / BK;
    const int tile = blk / split, sp = blk % split;
    const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
    const int st_all = M / GT_RS, st_base = st_all / split, st_rem = st_all % split;      // (see gemm_tn_kernel)
    const int nstage = st_base + (sp < st_rem ? 1 : 0);
    const int m_begin = (sp * st_base + min(sp, st_rem)) * GT_RS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WK, wk = wave % WK;
    const int hh = lane >> 5;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;

    // DMA map: a wave-instruction fills 2 rows; wave w owns rows 4w .. 4w+3 of a stage (2 instructions per operand).
    // Lane l lands on physical 16-byte piece l & 31 of row (l >> 5): it fetches the logical chunk ((l & 31) >> 2) ^ (row & 3).
    const bf16_t* srcA[2];
    const bf16_t* srcB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * wave + 2 * i + (lane >> 5);
        const int pp = lane & 31;
        const int col = ((((pp >> 2) ^ (row & 3)) << 6) + ((pp & 3) << 4)) >> 1;     // bf16 elements
        srcA[i] = a + ((size_t)m_begin + row) * lda + n0 + col;
        srcB[i] = b + ((size_t)m_begin + row) * ldb + k0 + col;
    }
#define GT_ISSUE(stage_)                                                                                              \
    do {                                                                                                              \
        const int buf_ = (stage_) % GT_NST;                                                                           \
        const size_t mo_ = (size_t)(stage_) * GT_RS;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                               \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + mo_ * lda),    \
                                             (RTTS_LDS void*)(As + buf_ * STAGE + (4 * wave + 2 * i) * ROWB), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + mo_ * ldb),    \
                                             (RTTS_LDS void*)(Bs + buf_ * STAGE + (4 * wave + 2 * i) * ROWB), 16, 0, 0); \
        }                                                                                                             \
    } while (0)

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = (f32x16){0};

    // fragment offsets inside a stage: row 16*ks + 8*hh + trq (second read: +4), byte column (tile column + 16*trc + 4*trp) * 2;
    // (row & 3) = trq for both reads (16*ks + 8*hh and +4 do not touch the low two bits)
    int offA[TN], offB[TK];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int cb = (wn * 32 * TN + t * 32 + 16 * trc + 4 * trp) * 2;
        offA[t] = (8 * hh + trq) * ROWB + ((((cb >> 6) ^ trq) << 6) | (cb & 63));
    }
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const int cb = (wk * 32 * TK + t * 32 + 16 * trc + 4 * trp) * 2;
        offB[t] = (8 * hh + trq) * ROWB + ((((cb >> 6) ^ trq) << 6) | (cb & 63));
    }

    constexpr int AHEAD = GT_NST - 1;                // stages in flight beyond the one being consumed
    static_assert(AHEAD >= 2 && AHEAD <= 4, "ring depth 3..5");
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < nstage) GT_ISSUE(i);
    for (int s = 0; s < nstage; ++s) {
        // this wave's pieces of stage s have landed once at most the later stages' instructions (4 each) are outstanding
        const int later = min(AHEAD - 1, nstage - 1 - s);
        if (later >= 3) __builtin_amdgcn_s_waitcnt(0x0F7C);              // vmcnt(12)
        else if (later == 2) __builtin_amdgcn_s_waitcnt(0x0F78);         // vmcnt(8)
        else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0F74);         // vmcnt(4)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        asm volatile("s_barrier" ::: "memory");      // everybody's pieces of stage s are in LDS; everybody is done reading stage s-1
        if (s + AHEAD < nstage) GT_ISSUE(s + AHEAD); // into the buffer stage s-1 occupied
        const unsigned char* Ab = As + (s % GT_NST) * STAGE;
        const unsigned char* Bb = Bs + (s % GT_NST) * STAGE;
        // The fragment reads are issued as inline assembly: the compiler cannot prove that an LDS read does not alias the
        // LDS-DMA writes still in flight and would drain them all (s_waitcnt vmcnt(0)) in front of the first read, which
        // is exactly the pipelining this kernel exists for.  Which stage is complete is known here (counted wait + barrier
        // above), so the reads wait only on lgkmcnt; the wait statement takes every fragment as an in/out operand, which
        // keeps the MFMAs behind it.
        const uint32_t abase = (uint32_t)(uintptr_t)(RTTS_LDS unsigned char*)Ab;
        const uint32_t bbase = (uint32_t)(uintptr_t)(RTTS_LDS unsigned char*)Bb;
        short4v al[2][TN], ah[2][TN], bl[2][TK], bh[2][TK];
#define GT_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const uint32_t ad = abase + offA[t];
            GT_TR(al[0][t], ad, 0);
            GT_TR(ah[0][t], ad, 4 * ROWB);
            GT_TR(al[1][t], ad, 16 * ROWB);
            GT_TR(ah[1][t], ad, 20 * ROWB);
        }
#pragma unroll
        for (int t = 0; t < TK; ++t) {
            const uint32_t ad = bbase + offB[t];
            GT_TR(bl[0][t], ad, 0);
            GT_TR(bh[0][t], ad, 4 * ROWB);
            GT_TR(bl[1][t], ad, 16 * ROWB);
            GT_TR(bh[1][t], ad, 20 * ROWB);
        }
#undef GT_TR
        static_assert(TN == 2 && TK == 4, "operand list of the wait below");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(al[0][0]), "+v"(ah[0][0]), "+v"(al[1][0]), "+v"(ah[1][0]), "+v"(al[0][1]), "+v"(ah[0][1]), "+v"(al[1][1]),
                       "+v"(ah[1][1]), "+v"(bl[0][0]), "+v"(bh[0][0]), "+v"(bl[1][0]), "+v"(bh[1][0]), "+v"(bl[0][1]), "+v"(bh[0][1]),
                       "+v"(bl[1][1]), "+v"(bh[1][1]), "+v"(bl[0][2]), "+v"(bh[0][2]), "+v"(bl[1][2]), "+v"(bh[1][2]), "+v"(bl[0][3]),
                       "+v"(bh[0][3]), "+v"(bl[1][3]), "+v"(bh[1][3]));
        bf16x8 af0[TN], bf0[TK], af1[TN], bf1[TK];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const gt_short8 f0 = {al[0][t][0], al[0][t][1], al[0][t][2], al[0][t][3], ah[0][t][0], ah[0][t][1], ah[0][t][2], ah[0][t][3]};
            const gt_short8 f1 = {al[1][t][0], al[1][t][1], al[1][t][2], al[1][t][3], ah[1][t][0], ah[1][t][1], ah[1][t][2], ah[1][t][3]};
            af0[t] = __builtin_bit_cast(bf16x8, f0);
            af1[t] = __builtin_bit_cast(bf16x8, f1);
        }
#pragma unroll
        for (int t = 0; t < TK; ++t) {
            const gt_short8 f0 = {bl[0][t][0], bl[0][t][1], bl[0][t][2], bl[0][t][3], bh[0][t][0], bh[0][t][1], bh[0][t][2], bh[0][t][3]};
            const gt_short8 f1 = {bl[1][t][0], bl[1][t][1], bl[1][t][2], bl[1][t][3], bh[1][t][0], bh[1][t][1], bh[1][t][2], bh[1][t][3]};
            bf0[t] = __builtin_bit_cast(bf16x8, f0);
            bf1[t] = __builtin_bit_cast(bf16x8, f1);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0[i], bf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
    }
#undef GT_ISSUE

    // epilogue: C tile rows n = (i&3) + 8*(i>>2) + 4*hh, cols k = lane&31
    float* dst = out + (split > 1 ? (size_t)sp * slab_stride : 0);
    const bool add = (split == 1) && accumulate;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int nb = n0 + wn * 32 * TN + i * 32 + 4 * hh;
            const int kc = k0 + wk * 32 * TK + j * 32 + (lane & 31);
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

template <int WN, int WK, int TN, int TK>
static int gt_launch(const GtGroup& grp, int wg, hipStream_t s) {
    constexpr int BN = 32 * TN * WN, BK = 32 * TK * WK;
    const size_t lds = 2 * GT_BM * (BN * 2 + GT_PAD) + 2 * GT_BM * (BK * 2 + GT_PAD);
    static RttsLdsState st = {};
    RTTS_ENSURE_LDS("rtts_gemm_tn", (gemm_tn_kernel<WN, WK, TN, TK>), lds, st);
    hipLaunchKernelGGL((gemm_tn_kernel<WN, WK, TN, TK>), dim3(wg), dim3(64 * WN * WK), lds, s, grp);
    return 0;
}

extern "C" int rtts_gemm_tn_grouped(const rtts_gemm_tn_problem* problems, int n, float* slab_ws, int64_t slab_ws_floats, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(problems && n > 0 && n <= RTTS_GEMM_TN_MAX_GROUP, "rtts_gemm_tn_grouped: 1..%d problems", RTTS_GEMM_TN_MAX_GROUP);
    GtGroup grp;
    grp.n = n;
    bool big = getenv("RTTS_GEMM_TN_SMALL_TILES") == nullptr;     // 256 x 256 tiles when every problem tiles by 256
    long long flops = 0;
    for (int i = 0; i < n; ++i) {
        const rtts_gemm_tn_problem& q = problems[i];
        RTTS_REQUIRE(q.a && q.b && q.c, "rtts_gemm_tn: null pointer");
        RTTS_REQUIRE(q.N > 0 && q.K > 0 && q.M > 0 && q.N % 128 == 0 && q.K % 128 == 0 && q.M % GT_BM == 0,
                     "rtts_gemm_tn: need N %% 128 == 0, K %% 128 == 0, M %% 64 == 0 (got M=%d N=%d K=%d)", q.M, q.N, q.K);
        RTTS_REQUIRE(q.lda >= q.N && q.ldb >= q.K && q.ldc >= q.K && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0,
                     "rtts_gemm_tn: bad leading dimensions");
        RTTS_REQUIRE((((uintptr_t)q.a | (uintptr_t)q.b | (uintptr_t)q.c) & 15) == 0, "rtts_gemm_tn: buffers must be 16-byte aligned");
        big = big && q.N % 256 == 0 && q.K % 256 == 0;
        flops += 2LL * q.M * q.N * q.K;
    }
    // small groups cannot fill 256 CUs with 256 x 256 tiles without splitting the token range very finely
    // (measured: one 25.8 GFLOP problem 45 us with 128-tiles / 50 us with 256-tiles; the 87 GFLOP group of a decoder
    // layer 143 us / 127 us)
    static const long long big_min = [] { const char* e = getenv("RTTS_GEMM_TN_BIG_MIN_GFLOP"); return (e ? atoll(e) : 40LL) * 1000000000LL; }();
    if (flops < big_min) big = false;
    const int bt = big ? 256 : 128;
    int total_tiles = 0;
    for (int i = 0; i < n; ++i) total_tiles += (problems[i].N / bt) * (problems[i].K / bt);
    // one split target for the group: the workgroups of all problems together fill the chip
    // (2 per CU for the 128-tile kernel, 1 per CU for the 256-tile kernel), each with >= 4 stages
    const int wg_target = big ? 320 : 640;
    int want = 1;
    while (total_tiles * want * 2 <= wg_target) want *= 2;
    int wg = 0, rb = 0;
    size_t slab_used = 0;
    bool any_split = false;
    for (int i = 0; i < n; ++i) {
        const rtts_gemm_tn_problem& q = problems[i];
        GtProb& P = grp.p[i];
        const int tiles = (q.N / bt) * (q.K / bt), stages = q.M / GT_BM;
        const size_t slab = (size_t)q.N * q.K;
        int split = 1;
        while (split < want && split * 2 <= stages / 4) split *= 2;
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
    static const bool ring = getenv("RTTS_GEMM_TN_NO_RING") == nullptr;
    if (big && ring) {
        const size_t lds = 2 * GT_NST * GT_RS * 512;
        static RttsLdsState st = {};
        RTTS_ENSURE_LDS("rtts_gemm_tn", (gemm_tn_ring_kernel<4, 2, 2, 4>), lds, st);
        hipLaunchKernelGGL((gemm_tn_ring_kernel<4, 2, 2, 4>), dim3(wg), dim3(512), lds, s, grp);
    } else if (big) {
        if (gt_launch<4, 2, 2, 4>(grp, wg, s)) return -1;
    } else if (gt_launch<2, 2, 2, 2>(grp, wg, s)) return -1;
    if (any_split) hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, grp);
    RTTS_LAUNCH_CHECK("rtts_gemm_tn");
    return 0;
}

extern "C" int rtts_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int M, int N, int K, float* c, int64_t ldc,
                            int accumulate, float* slab_ws, int64_t slab_ws_floats, void* stream) {
    RTTS_ENTER(stream);
    rtts_gemm_tn_problem q;
    q.a = a; q.lda = lda; q.b = b; q.ldb = ldb; q.c = c; q.ldc = ldc;
    q.M = M; q.N = N; q.K = K; q.accumulate = accumulate;
    return rtts_gemm_tn_grouped(&q, 1, slab_ws, slab_ws_floats, stream);
}
