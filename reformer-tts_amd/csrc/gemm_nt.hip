// Forward / input-gradient GEMM of the Linear layers of the reversible stacks:
//     C[M][N] (bf16) = epilogue( A[M][K] (bf16) x W )        fp32 accumulation on v_mfma_f32_16x16x32_bf16
//   W_KN = 0:  W stored [N][K]  (y = x W^T, the forward of nn.Linear:   toqk/tov/to_out, in_proj/out_proj, FeedForward net.0/net.3;
//              /root/reference/reformer_tts/model/modules.py:195-207, reformer.py:161-217)
//   W_KN = 1:  W stored [K][N]  (dx = dy W, the input gradient of the same layers: the autograd backward of the lines above)
// Epilogues (fused into the store of the accumulators, nothing is re-read):
//   0 plain   1 + bias   2 relu(+ bias)   3 ReLU gate: C = acc * (gate > 0) with the column sums of C over the workgroup's
//   rows written as fp32 partial rows (the bias gradient of the first FeedForward layer; summed by rtts_colsum_final_grouped)
//
// Shape of the problem on MI355X: M = B*T = 12288 tokens, N and K only 512..2048 -- a few hundred output tiles and 8..32
// K-steps, so the kernel is sized for launch ramp and operand ingest, not for a long steady state:
//   * tile BM x BN = 192 x 128 (256 x 128 when 192 does not divide M): (M/192) x (N/128) = 256, 512, 1024 workgroups for
//     N = 512, 1024, 2048 = whole waves of one 8-wave workgroup per CU; small problems (the encoder, M = 3072) use 96 x 64;
//   * both operands arrive by LDS-DMA (global_load_lds_dwordx4) into a ring of 64-deep K stages as deep as LDS allows
//     (4 x 40 KB for the 192 x 128 tile): NST-1 stages in flight while one is consumed, a counted s_waitcnt vmcnt(N) +
//     ONE raw s_barrier per stage; the loop is software-pipelined over half stages so that every fragment read from LDS
//     is issued under the MFMAs of the previous half stage;
//   * LDS images are unpadded 128-byte rows ([row][64 k]) whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7 on the
//     SOURCE side of the DMA, so every ds_read_b128 fragment read (16 rows x 4 chunks) touches 16 distinct 16-byte slots;
//     the [K][N] weight image keeps k rows and is read with ds_read_b64_tr_b16 (32-byte chunks XOR-swizzled by the k row);
//   * the product is computed transposed (D^T = W-fragment x A-fragment) so that a lane ends up with 4 CONSECUTIVE output
//     columns of one row: the epilogue stores 8 bytes per lane straight from the accumulators, bias / gate are 16 / 8-byte loads;
//   * XCD-aware tile order: the N/BN tiles that share an A row panel run on one XCD back to back (one fetch of the panel
//     from HBM / Infinity Cache per XCD, the weight stays resident in every L2).
// Three launch forms of the same body (template MODE):
//   0  one problem, one tile per workgroup (the form every small problem keeps);
//   1  GROUPED: one grid over the tiles of up to GN_MAX_GROUP independent problems of one layout and epilogue (the cross-attention's
//      q and k|v projections; its input-gradient pair dxn / dkeys): a workgroup looks its problem up by tile range -- one launch
//      ramp, one tail and one kernel boundary instead of one per problem;
//   2  PERSISTENT: problems of several tiles per CU (N >= 1024 at M = 12288) run on a grid of resident workgroups that walk the
//      tiles; the LDS ring keeps turning across a tile boundary -- the first NST stages of the next tile are in flight while the
//      current tile's accumulators are stored -- so the 1.8 us launch-to-first-stage latency is paid once per workgroup, not per tile.
// More epilogues: 4 with `accum` (fp32 C += result: the keys' gradient accumulates over the decoder layers without a separate add
// launch) and 5 (input gradient of to_out / out_proj + delta = rowsum(out * dout) per (token, head) for the attention backward: the
// separate rtts_lsh_bwd_delta launch and its re-read of dout are gone).
#include "rtts_common.h"
#include <stdlib.h>
#include <atomic>

typedef __attribute__((ext_vector_type(4))) int gn_v4i;
typedef __attribute__((ext_vector_type(2))) int gn_v2i;


#include <type_traits>
template <int I, int N, typename F>
__device__ __forceinline__ void gn_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        gn_static_for<I + 1, N>(f);
    }
}
// LDS reads as inline assembly with immediate offsets (an "n" operand must be a template constant, not a loop variable)
template <int OFF>
__device__ __forceinline__ void gn_rd128(gn_v4i& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void gn_rdtr(gn_v2i& dst, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void gn_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void gn_wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

#define GN_BK 64

struct GnArgs {
    const bf16_t* a;
    const bf16_t* w;
    bf16_t* c;
    const float* bias;
    const bf16_t* gate;
    float* colsum;       // [(M / BM) * WM][N] partial rows (epilogue 3), may be null
    unsigned long long* bits;   // 1-bit ReLU gate, one word per lane and tile: written by epilogue 2, read by epilogue 3 (instead of `gate`); may be null
    int64_t lda, ldw, ldc, ldg;
    int M, N, K, epi;
    // implicit-GEMM Conv1d(k = 5): conv_cpt = 64-deep channel blocks per tap (0 = plain GEMM).  K stage s is tap s / conv_cpt,
    // channel block s % conv_cpt: the A rows are shifted by conv_sign * (tap - 2) (the caller's rows carry a zero halo), and a
    // [K][N] weight (input gradient) is addressed as row = channel, column offset tap * conv_wtap.
    int conv_cpt, conv_sign;
    int64_t conv_wtap;
    // epilogue 5: aux = the attention output `out` (M, N) bf16 (row stride ldaux), aux_out = delta f32 (B*H, T) with rows
    // m = b * T + t and heads of 64 columns; epilogue 4: accum != 0 -> C (fp32) += result
    const bf16_t* aux;
    int64_t ldaux;
    float* aux_out;
    int T, H, accum;
    int store_mode;      // bf16 outputs: 0 = 8-byte stores, 1 = 16-byte stores (lane-pair exchange), 2 = 16-byte write-through stores
    // epilogue 7 (a convolution in front of a BatchNorm): fp32 store as epilogue 4 AND the per-channel moments of the output over the
    // rows that carry data -- halo rows m = b * bn_P + bn_H + t, t < bn_L, m < bn_rows -- as partial rows in `colsum`:
    // row (m0 / BM) * WM + wm holds [sum y | sum y^2] (2 N floats): what the BatchNorm statistics kernel re-read y for
    int bn_P, bn_H, bn_L, bn_rows;
};

#define GN_MAX_GROUP 4
struct GnGroup {
    GnArgs p[GN_MAX_GROUP];
    int tile_end[GN_MAX_GROUP];        // MODE 1: cumulative tile counts of the problems
    int n;
    int spread;                        // MODE 1: every tile count is a multiple of 8 -> each XCD gets an eighth of EVERY problem
};

// swizzle of the [K][N] weight image: XOR applied to the index of a 32-byte chunk (16 columns) of k row `k`
template <int RP>
__device__ __forceinline__ int gn_kn_swz(int k) {
    if constexpr (RP == 128) return ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
    else return (k & 3) | (((k >> 3) & 1) << 2);
}

// the 2-deep ring of the 192 x 128 tile exists to put TWO 8-wave workgroups on a CU: 4 waves per SIMD, i.e. at most 128 registers
template <int BM, int BN, int WM, int WN, bool W_KN, int EPI, int NST, int MODE>
__global__ __launch_bounds__(64 * WM * WN, (BM == 192 && NST == 2) ? 4 : 1) void gemm_nt_kernel(const GnGroup G) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;          // 16 x 16 MFMA tiles per wave
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int PA = BM / 8 / NW, PW = BN / 8 / NW, PER = PA + PW;   // 1-KB DMA pieces per wave and stage
    constexpr int RP = 2 * BN;                                    // row pitch of the [K][N] image (bytes)
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA pieces must divide over the waves");
    static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "wave tiles");
    static_assert(!W_KN || RP == 128 || RP == 256 || RP == 512, "[K][N] image: BN in {64, 128, 256}");
    static_assert(!W_KN || ((BN / WN) % 64 == 0) || TN <= 2, "[K][N] image: wave column base");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef GN_STAMPS
    // diagnostic build (scripts/gemm_nt_stamps.py): wave 0 stamps the phase boundaries with the shader clock and the 100 MHz
    // real-time clock into a buffer of its own (passed in the colsum slot of a plain-epilogue call); never in the product build
    unsigned long long st_c[5], st_r[5];
#define GN_STAMP(i_)                                                                                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c[i_]), "=s"(st_r[i_])::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define GN_STAMP(i_)
#endif
    GN_STAMP(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t logical = xcd_remap(blockIdx.x, gridDim.x);
    // which problem, which tile: MODE 1 looks the problem up by tile range (wave-uniform: scalar compares on kernel arguments)
    int pi = 0;
    uint32_t tile = logical;
    if constexpr (MODE == 1) {
        if (G.spread) {
            // logical ids [x q, (x + 1) q) share an XCD (xcd_remap), q = total / 8: give every XCD an eighth of EVERY problem --
            // problems of different depth (K = 512 beside K = 1024) then load the eight XCDs alike.  Inside its slice a problem's
            // tiles stay consecutive (the N tiles of an A row panel back to back on one L2).
            const uint32_t q = gridDim.x >> 3, x = logical / q, rr = logical - x * q;
#pragma unroll
            for (int i = 0; i + 1 < GN_MAX_GROUP; ++i)
                if (i + 1 < G.n && rr >= ((uint32_t)G.tile_end[i] >> 3)) pi = i + 1;
            pi = __builtin_amdgcn_readfirstlane(pi);
            const uint32_t lo = pi ? ((uint32_t)G.tile_end[pi - 1] >> 3) : 0u, cnt = ((uint32_t)G.tile_end[pi] >> 3) - lo;
            tile = x * cnt + (rr - lo);
        } else {
#pragma unroll
            for (int i = 0; i + 1 < GN_MAX_GROUP; ++i)
                if (i + 1 < G.n && logical >= (uint32_t)G.tile_end[i]) pi = i + 1;
            pi = __builtin_amdgcn_readfirstlane(pi);
            tile = logical - (pi ? (uint32_t)G.tile_end[pi - 1] : 0u);
        }
    }
    const GnArgs& P = G.p[pi];
    const int ntn = P.N / BN;
    int m0 = (int)(tile / ntn) * BM, n0 = (int)(tile % ntn) * BN;
    const int nk = P.K / GN_BK;
    const int64_t lda = P.lda, ldw = P.ldw;
    // MODE 2: this workgroup works tiles tile, tile + gridDim.x, ... of the one problem
    const uint32_t ntiles = MODE == 2 ? (uint32_t)(P.M / BM) * (uint32_t)ntn : 0u;
    (void)ntiles;

    // ---- DMA sources: ONE per-lane pointer per operand (the wave's piece 0 of stage 0).  Piece t of a wave is 8 NW t rows
    // further on ([row][k] images; (1024 / RP) NW t k rows for the [K][N] weight image) -- a wave-uniform offset -- and needs
    // the same per-lane swizzle: the row bits the swizzles read ((row >> 1) & 7 resp. gn_kn_swz) do not change with t for an
    // even wave count (checked below), so the pointers of the other pieces are scalar adds at issue time, not registers.
    const bf16_t* srcA0;
    const bf16_t* srcW0;
    {
        const int row = 8 * wave + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        srcA0 = P.a + (size_t)(m0 + row) * lda + lc * 8;
        if constexpr (!W_KN) {
            srcW0 = P.w + (size_t)(n0 + row) * ldw + lc * 8;
        } else {
            constexpr int UPR = RP / 16;                          // 16-byte units per k row
            const int krow = (1024 / RP) * wave + lane / UPR, u = lane % UPR;
            const int lchunk = (u >> 1) ^ gn_kn_swz<RP>(krow);
            srcW0 = P.w + (size_t)krow * ldw + n0 + lchunk * 16 + (u & 1) * 8;
        }
    }
    static_assert(NW % 2 == 0 && (!W_KN || ((1024 / RP) * NW) % 16 == 0), "per-piece swizzle must not depend on the piece index");
    const size_t pstepA = (size_t)8 * NW * lda, pstepW = W_KN ? (size_t)(1024 / RP) * NW * ldw : (size_t)8 * NW * ldw;
    const size_t wstep = W_KN ? (size_t)GN_BK * ldw : (size_t)GN_BK;
    const int conv_cpt = P.conv_cpt;
    // MODE 2: element offsets from this tile's operand rows to the next tile's (wave-uniform); stages >= nk name the next tile
    int64_t nxtA = 0, nxtW = 0;
    bool has_next = false, first_tile = true;
    (void)nxtA; (void)nxtW; (void)has_next; (void)first_tile;
    // element offsets of K stage `st` into the A rows / the weight (wave-uniform scalar arithmetic)
    auto a_off = [&](int st) -> int64_t {
        if (conv_cpt == 0) return (int64_t)st * GN_BK;
        const int tap = st / conv_cpt, cb = st - tap * conv_cpt;
        return (int64_t)(P.conv_sign * (tap - 2)) * lda + (int64_t)cb * GN_BK;
    };
    auto w_off = [&](int st) -> int64_t {
        if (!W_KN || conv_cpt == 0) return (int64_t)st * (int64_t)wstep;
        const int tap = st / conv_cpt, cb = st - tap * conv_cpt;
        return (int64_t)cb * GN_BK * ldw + (int64_t)tap * P.conv_wtap;
    };
#define GN_ISSUE(stage_, buf_)                                                                                          \
    do {                                                                                                                \
        unsigned char* sb_ = smem + (buf_) * STAGE;                                                                     \
        const int64_t ka_ = a_off(stage_), kw_ = w_off(stage_);                                                         \
        _Pragma("unroll") for (int t = 0; t < PA; ++t)                                                                  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA0 + t * pstepA + ka_), \
                                             (RTTS_LDS void*)(sb_ + (wave + NW * t) * 1024), 16, 0, 0);                 \
        _Pragma("unroll") for (int t = 0; t < PW; ++t)                                                                  \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW0 + t * pstepW + kw_), \
                                             (RTTS_LDS void*)(sb_ + A_BYTES + (wave + NW * t) * 1024), 16, 0, 0);       \
    } while (0)

    // ---- fragment read offsets inside a stage
    const int r = lane & 15, g = lane >> 4;
    // [row][64 k] images: 16 rows x 4 chunks per fragment; k32-step 1 flips bit 2 of the chunk index (address ^ 64)
    const uint32_t offA0 = (uint32_t)(wm * (BM / WM) * 128 + r * 128 + ((g ^ ((r >> 1) & 7)) << 4));
    const uint32_t offA1 = offA0 ^ 64u;
    uint32_t offW[2];
    if constexpr (!W_KN) {
        offW[0] = (uint32_t)(A_BYTES + wn * (BN / WN) * 128 + r * 128 + ((g ^ ((r >> 1) & 7)) << 4));
        offW[1] = offW[0] ^ 64u;
    } else {
        // transposed read: lane 4q+p of 16-lane group g supplies k row 8g + q (second read: + 4), columns 4p .. 4p+3 of tile j.
        // The physical 32-byte chunk of tile j is (jj0 + j) ^ fl with jj0 = the wave's first tile, a multiple of TN (a power of
        // two), so (jj0 + j) ^ fl = (jj0 ^ fl) ^ j: ONE lane-dependent base, tile j at base ^ (j << 5) -- bits 5.. of the rest
        // of the base (A image, k row * RP, 8 p) are zero below the chunk field.
        static_assert((TN & (TN - 1)) == 0 && (A_BYTES % 256) == 0, "[K][N] image: tile count per wave must be a power of two");
        const int q = r >> 2, p = r & 3;
        const int fl = gn_kn_swz<RP>(8 * g + q);                   // does not depend on the k32-step or the half
        const int jj0 = wn * TN;
        offW[0] = (uint32_t)(A_BYTES + (8 * g + q) * RP + ((jj0 ^ fl) << 5) + 8 * p);
        offW[1] = 0;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop, software-pipelined over half stages (k32-steps):
    //   F0 = fragments of (stage s, step 0) are in registers at the top of iteration s;
    //   read F1 = (s, 1)  | MFMA(F0) | F1 back | stage s+1 landed (counted vmcnt) + ONE barrier: every wave is also done
    //   reading stage s | DMA of stage s+NST into the buffer of stage s | read F0 = (s+1, 0) | MFMA(F1)
    // so every LDS fragment read runs under 12..16 MFMAs of the same wave, and NST-1 stages are in flight during them.
    gn_v4i a0[TM], a1[TM], w0[TN], w1[TN];
    gn_v2i wl0[TN], wh0[TN], wl1[TN], wh1[TN];
    (void)w0; (void)w1; (void)wl0; (void)wh0; (void)wl1; (void)wh1;
    constexpr int RD_PER_KS = (W_KN ? 2 * TN : TN) + TM;
#define GN_READ(AF, WF, WL, WH, sb_, KS)                                                                                   \
    do {                                                                                                                   \
        if constexpr (!W_KN) {                                                                                             \
            const uint32_t aw_ = (sb_) + offW[KS];                                                                         \
            gn_static_for<0, TN>([&](auto jc) { constexpr int j = decltype(jc)::value; gn_rd128<j * 2048>(WF[j], aw_); }); \
        } else {                                                                                                           \
            gn_static_for<0, TN>([&](auto jc) {                                                                            \
                constexpr int j = decltype(jc)::value;                                                                     \
                const uint32_t aw_ = (sb_) + (offW[0] ^ (uint32_t)(j << 5));                                               \
                gn_rdtr<(32 * KS) * RP>(WL[j], aw_);                                                                       \
                gn_rdtr<(32 * KS + 4) * RP>(WH[j], aw_);                                                                   \
            });                                                                                                            \
        }                                                                                                                  \
        const uint32_t aa_ = (sb_) + (KS ? offA1 : offA0);                                                                 \
        gn_static_for<0, TM>([&](auto ic) { constexpr int i = decltype(ic)::value; gn_rd128<i * 2048>(AF[i], aa_); });     \
    } while (0)
    // One burst = the TM x TN MFMAs of a half stage.  The DMA pieces of the stage being prefetched are issued INSIDE the bursts,
    // one piece every GAP MFMAs (pieces [Q0, Q1) of stage ST_ into buffer BUF_ when COND_): a global_load_lds issued into a
    // busy texture-address queue blocks its wave for 60..200 cycles (MI355X_MICROARCH.md, LDS-DMA piece issue cost), and
    // with all eight waves issuing their pieces back to back behind the barrier the matrix pipes idled ~475 cycles per stage
    // (scripts/gemm_nt_stamps.py ablations, profiles/r02_gemm_nt_ablation.log); spread out, a blocked wave's SIMD partner keeps
    // the pipe busy.
    constexpr int SPB = (PER + 1) / 2;                     // pieces issued per burst
    constexpr int GAP = (TM * TN) / SPB > 0 ? (TM * TN) / SPB : 1;
    auto issue_piece = [&](auto qc, int stage_, int bufi_) {
        constexpr int q = decltype(qc)::value;
        unsigned char* sb_ = smem + bufi_ * STAGE;
        int64_t ka_, kw_;
        if constexpr (MODE == 2) {             // a stage index beyond this tile's last stage: the next tile's stage (stage_ - nk)
            const bool nx_ = stage_ >= nk;
            const int st_ = nx_ ? stage_ - nk : stage_;
            ka_ = a_off(st_) + (nx_ ? nxtA : 0);
            kw_ = w_off(st_) + (nx_ ? nxtW : 0);
        } else {
            ka_ = a_off(stage_);
            kw_ = w_off(stage_);
        }
        if constexpr (q < PA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA0 + q * pstepA + ka_),
                                             (RTTS_LDS void*)(sb_ + (wave + NW * q) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW0 + (q - PA) * pstepW + kw_),
                                             (RTTS_LDS void*)(sb_ + A_BYTES + (wave + NW * (q - PA)) * 1024), 16, 0, 0);
    };
#define GN_MFMA(AF, WF, WL, WH, COND_, ST_, BUF_, Q0, Q1)                                                                  \
    do {                                                                                                                   \
        bf16x8 wfr_[TN], afr_[TM];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                   \
            if constexpr (!W_KN) wfr_[j] = __builtin_bit_cast(bf16x8, WF[j]);                                              \
            else {                                                                                                         \
                const gn_v4i both_ = {WL[j][0], WL[j][1], WH[j][0], WH[j][1]};                                             \
                wfr_[j] = __builtin_bit_cast(bf16x8, both_);                                                               \
            }                                                                                                              \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) afr_[i] = __builtin_bit_cast(bf16x8, AF[i]);                        \
        const bool cond_ = (COND_);                                                                                        \
        /* D^T tile: rows = output columns n (weight fragment as the A operand), columns = output rows m */               \
        gn_static_for<0, TM * TN>([&](auto xc) {                                                                           \
            constexpr int x = decltype(xc)::value, i = x / TN, j = x % TN;                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr_[j], afr_[i], acc[i][j], 0, 0, 0);                     \
            if constexpr (x % GAP == GAP - 1 && (Q0) + x / GAP < (Q1)) {                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
                if (cond_) issue_piece(std::integral_constant<int, (Q0) + x / GAP>{}, (ST_), (BUF_));                      \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
        });                                                                                                                \
    } while (0)
    // this wave's pieces of a stage have landed once at most `later` younger stages (PER instructions each) are outstanding
#define GN_WAIT_STAGE(later_)                                                                                              \
    do {                                                                                                                   \
        const int l_ = (later_);                                                                                           \
        if (l_ <= 0) gn_wait_vm<0>();                                                                                      \
        else if (l_ == 1) gn_wait_vm<PER>();                                                                               \
        else if (l_ == 2) gn_wait_vm<2 * PER>();                                                                           \
        else gn_wait_vm<3 * PER>();                                                                                        \
    } while (0)
    static_assert(NST >= 2 && NST <= 4 && 3 * PER <= 63, "ring depth");

#pragma unroll
    for (int i = 0; i < NST; ++i)
        if (i < nk) GN_ISSUE(i, i);
    GN_WAIT_STAGE(min(NST - 1, nk - 1));
    asm volatile("s_barrier" ::: "memory");
    GN_STAMP(1);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(RTTS_LDS unsigned char*)smem;
    GN_READ(a0, w0, wl0, wh0, smem_base, 0);
    int buf = 0;
#ifdef GN_ABL_NOLDS          // timing-only builds (scripts/gemm_nt_stamps.py): results are wrong by construction
#undef GN_READ
#define GN_READ(AF, WF, WL, WH, sb_, KS) do { } while (0)
    gn_static_for<0, TM>([&](auto ic) { constexpr int i = decltype(ic)::value; a0[i] = a1[i] = (gn_v4i){lane, i, 3, 4}; });
    gn_static_for<0, TN>([&](auto jc) { constexpr int j = decltype(jc)::value; w0[j] = w1[j] = (gn_v4i){j, lane, 5, 6};
                                        wl0[j] = wl1[j] = wh0[j] = wh1[j] = (gn_v2i){lane, j}; });
#endif
#ifdef GN_ABL_NOMFMA
#undef GN_MFMA
#define GN_MFMA(AF, WF, WL, WH, COND_, ST_, BUF_, Q0, Q1)                                         \
    do {                                                                                          \
        if (COND_) gn_static_for<(Q0), (Q1)>([&](auto qc) { issue_piece(qc, (ST_), (BUF_)); });   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(AF[i]));             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                          \
            if constexpr (!W_KN) asm volatile("" ::"v"(WF[j]));                                   \
            else asm volatile("" ::"v"(WL[j]), "v"(WH[j]));                                       \
        }                                                                                         \
    } while (0)
#endif
    int pbuf = 0;                                    // buffer of stage s-1: the second half of stage s-1+NST's pieces goes there
    static_assert(MODE != 2 || (EPI != 4 && true), "persistent form: bf16 epilogues");
  for (;;) {                                         // MODE 2: one pass per tile of this workgroup; else a single pass
    if constexpr (MODE == 2) {
        const uint32_t nt_ = tile + gridDim.x;
        has_next = nt_ < ntiles;
        if (has_next) {
            const int m0n = (int)(nt_ / ntn) * BM, n0n = (int)(nt_ % ntn) * BN;
            nxtA = (int64_t)(m0n - m0) * lda;
            nxtW = W_KN ? (int64_t)(n0n - n0) : (int64_t)(n0n - n0) * ldw;
        }
    }
    for (int s = 0; s < nk; ++s) {
        const uint32_t sb = smem_base + buf * STAGE;
        int nb = buf + 1 == NST ? 0 : buf + 1;
        GN_READ(a1, w1, wl1, wh1, sb, 1);
        // F0 is back (LDS returns in order; only F1's reads may be outstanding).  The counter holds 15 at most: with more
        // reads per half stage (the 256-wide tile's transposed weight fragments) "at most 15 outstanding" already implies it
        gn_wait_lgkm<(RD_PER_KS < 15 ? RD_PER_KS : 15)>();
        __builtin_amdgcn_sched_barrier(0);
        // MODE 2: the ring does not stop at a tile boundary -- stage indices >= nk are the next tile's first stages (issue_piece),
        // the second half of the stage whose first half the previous tile's last step issued goes out at s = 0
#ifndef GN_ABL_NODMA
        GN_MFMA(a0, w0, wl0, wh0, MODE == 2 ? ((s >= 1 || !first_tile) && (s - 1 + NST < nk || has_next)) : (s >= 1 && s - 1 + NST < nk),
                s - 1 + NST, pbuf, SPB, PER);
#else
        GN_MFMA(a0, w0, wl0, wh0, false, 0, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        gn_wait_lgkm<0>();                           // F1 is back: this wave has no read of stage s left
        if (s + 1 < nk || (MODE == 2 && has_next)) {
            GN_WAIT_STAGE((MODE == 2 && has_next) ? NST - 2 : min(NST - 2, nk - 2 - s));
            asm volatile("s_barrier" ::: "memory");  // stage s+1 complete in LDS; everybody is done reading stage s
            GN_READ(a0, w0, wl0, wh0, smem_base + nb * STAGE, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef GN_ABL_NODMA
        GN_MFMA(a1, w1, wl1, wh1, s + NST < nk || (MODE == 2 && has_next), s + NST, buf, 0, SPB);    // (implies: behind the barrier)
#else
        GN_MFMA(a1, w1, wl1, wh1, false, 0, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
        pbuf = buf;
        buf = nb;
    }
    GN_STAMP(2);

    // ---- epilogue: lane holds C[m][n .. n+3], m = m0 + wave rows + 16 i + (lane & 15), n = n0 + wave cols + 16 j + 4 (lane >> 4)
    constexpr int epi = EPI;
    const int mrow = m0 + wm * (BM / WM) + r;
    const int ncol = n0 + wn * (BN / WN) + 4 * g;
    f32x4 cs[TN], cs2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs[j] = cs2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    (void)cs2;
    bool vrow[TM];                                 // epilogue 7: does row tile i of this lane carry data (not a halo / padding row)?
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        vrow[i] = false;
        if constexpr (EPI == 7) {
            const int m = mrow + 16 * i;
            const int rr = m % P.bn_P - P.bn_H;
            vrow[i] = m < P.bn_rows && (unsigned)rr < (unsigned)P.bn_L;
        }
    }
    (void)vrow;
    float dl[TM];                                  // epilogue 5: this lane's share of delta of row tile i (its 4 x TN columns)
#pragma unroll
    for (int i = 0; i < TM; ++i) dl[i] = 0.f;
    (void)dl;
    static_assert(EPI != 5 || (W_KN && BN / WN == 64), "epilogue 5: an input gradient whose waves each cover one 64-wide head");
    // 1-bit gate: bit ((i * TN + j) * 4 + e) of this lane's word <=> output (i, j, e) of this lane is a positive bf16.  The
    // forward (epilogue 2) and the input gradient (epilogue 3) of one (M, N) get the same tile shape, hence the same lane ->
    // element map; 16 x TM x TN bits fit one word for every tile but 256 x 256 (rtts_gemm_nt_gate_words() = 0 there).
    constexpr bool kBitsFit = TM * TN * 4 <= 64;
    unsigned long long gbits = 0;
    const size_t widx = ((size_t)(m0 / BM) * (P.N / BN) + (size_t)(n0 / BN)) * (64 * WM * WN) + tid;
    bool use_bits = false;
    if constexpr (kBitsFit && (epi == 2 || epi == 3)) use_bits = P.bits != nullptr;
    if constexpr (kBitsFit && epi == 3) if (use_bits) gbits = P.bits[widx];
    // store form (wave-uniform): 0 = 8 bytes per lane straight from the accumulators, 1 = 16 bytes after a lane-pair exchange,
    // 2 = the same as write-through (sc1) stores -- the output leaves the XCD's L2 while the kernel still runs instead of being
    // flushed as dirty lines at its end (no consumer sits on this L2 alone: the next kernel's workgroups are on all eight)
    const int wide = (TN % 2 == 0) ? P.store_mode : 0;
    static_assert(TN % 2 == 0, "column tiles are worked in pairs");
    // one (row tile i, column tile j) of this lane: 4 consecutive columns of one row through the epilogue's arithmetic.
    // -> true: `o` holds the four bf16 to store; false: the fp32 forms have stored already
    auto one = [&](auto ic_, auto jc_, const f32x4 bv, uint2& o) -> bool {
        constexpr int i = decltype(ic_)::value, j = decltype(jc_)::value;
        const int n = ncol + 16 * j;
        const size_t m = (size_t)(mrow + 16 * i);
        f32x4 v = acc[i][j] + bv;
        if constexpr (epi == 2) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        } else if constexpr (epi == 3) {
            if (kBitsFit && use_bits) {
                const unsigned nib = (unsigned)(gbits >> ((i * TN + j) * 4)) & 15u;
                v[0] = (nib & 1u) ? v[0] : 0.f;
                v[1] = (nib & 2u) ? v[1] : 0.f;
                v[2] = (nib & 4u) ? v[2] : 0.f;
                v[3] = (nib & 8u) ? v[3] : 0.f;
            } else {
                const uint2 hv = *reinterpret_cast<const uint2*>(P.gate + m * P.ldg + n);
                // bf16 h > 0  <=>  sign bit clear and magnitude non-zero
                v[0] = ((hv.x & 0x8000u) == 0 && (hv.x & 0x7FFFu) != 0) ? v[0] : 0.f;
                v[1] = ((hv.x & 0x80000000u) == 0 && (hv.x & 0x7FFF0000u) != 0) ? v[1] : 0.f;
                v[2] = ((hv.y & 0x8000u) == 0 && (hv.y & 0x7FFFu) != 0) ? v[2] : 0.f;
                v[3] = ((hv.y & 0x80000000u) == 0 && (hv.y & 0x7FFF0000u) != 0) ? v[3] : 0.f;
            }
            cs[j] += v;
        }
        // epilogue 4: unrounded fp32 result (rows of ldc floats): in front of a BatchNorm / the loss.  Epilogue 6 is the GROUP's
        // epilogue: each problem says at run time (a wave-uniform branch) whether it stores fp32 (its own epilogue 4) or bf16
        // (0, or 1 when it has a bias)
        if (epi == 4 || epi == 7 || (epi == 6 && P.epi == 4)) {
            f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(P.c) + m * P.ldc + n);
            if (P.accum) v += *dst;        // C += result (the keys' gradient over the decoder layers)
            *dst = v;
            if constexpr (epi == 7) {
                if (vrow[i]) {
                    cs[j] += v;
                    cs2[j] += v * v;
                }
            }
            return false;
        }
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        if constexpr (epi == 5) {          // delta += out * dout on the ROUNDED dout: what the attention backward will read
            const uint2 ov = *reinterpret_cast<const uint2*>(P.aux + m * P.ldaux + n);
            dl[i] += __uint_as_float(ov.x << 16) * __uint_as_float(o.x << 16) + __uint_as_float(ov.x & 0xFFFF0000u) * __uint_as_float(o.x & 0xFFFF0000u) +
                     __uint_as_float(ov.y << 16) * __uint_as_float(o.y << 16) + __uint_as_float(ov.y & 0xFFFF0000u) * __uint_as_float(o.y & 0xFFFF0000u);
        }
        if constexpr (kBitsFit && epi == 2) {      // the values are >= 0: positive <=> the rounded bf16 is not zero
            const unsigned nib = ((o.x & 0xFFFFu) ? 1u : 0u) | ((o.x >> 16) ? 2u : 0u) | ((o.y & 0xFFFFu) ? 4u : 0u) | ((o.y >> 16) ? 8u : 0u);
            gbits |= (unsigned long long)nib << ((i * TN + j) * 4);
        }
        return true;
    };
    gn_static_for<0, TN / 2>([&](auto jpc_) {
        constexpr int jl = 2 * decltype(jpc_)::value, jh = jl + 1;
        const int nl = ncol + 16 * jl, nh = nl + 16;
        f32x4 bvl = {0.f, 0.f, 0.f, 0.f}, bvh = {0.f, 0.f, 0.f, 0.f};
        if constexpr (epi == 1 || epi == 2) {
            bvl = *reinterpret_cast<const f32x4*>(P.bias + nl);
            bvh = *reinterpret_cast<const f32x4*>(P.bias + nh);
        }
        if constexpr (epi == 4 || epi == 6 || epi == 7) if (P.bias != nullptr) {
            bvl = *reinterpret_cast<const f32x4*>(P.bias + nl);
            bvh = *reinterpret_cast<const f32x4*>(P.bias + nh);
        }
        gn_static_for<0, TM>([&](auto ic_) {
            constexpr int i = decltype(ic_)::value;
            const size_t m = (size_t)(mrow + 16 * i);
            uint2 ol = {0u, 0u}, oh = {0u, 0u};
            const bool sl = one(ic_, std::integral_constant<int, jl>{}, bvl, ol);
            const bool sh = one(ic_, std::integral_constant<int, jh>{}, bvh, oh);
            if (!(sl && sh)) return;               // fp32 forms: stored inside
            if (wide) {
                // 16-byte stores: the lane pair (l, l ^ 16) -- column groups g and g ^ 1 of the same row -- trades halves of the tile
                // pair (jl, jh): the even group ends up with 8 consecutive columns of tile jl, the odd group with 8 of tile jh.
                // One store instruction then writes 16 rows x 64 contiguous bytes instead of 16 x 32, and half as many are issued.
                const bool odd = (g & 1) != 0;
                const uint2 send = odd ? ol : oh;
                uint2 recv;
                recv.x = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send.x, 0x401F);      // lane ^ 16 (bit mode: and 0x1F, xor 0x10)
                recv.y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)send.y, 0x401F);
                gn_v4i wv;
                wv[0] = (int)(odd ? recv.x : ol.x);
                wv[1] = (int)(odd ? recv.y : ol.y);
                wv[2] = (int)(odd ? oh.x : recv.x);
                wv[3] = (int)(odd ? oh.y : recv.y);
                bf16_t* dst = P.c + m * P.ldc + (odd ? nh - 4 : nl);
                // write-through; the s_nop pads the store-data hazard hipcc does not pad for an asm statement (cdna_hip_programming.md 5.7)
                if (wide == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(wv) : "memory");
                else *reinterpret_cast<gn_v4i*>(dst) = wv;
            } else {
                *reinterpret_cast<uint2*>(P.c + m * P.ldc + nl) = ol;
                *reinterpret_cast<uint2*>(P.c + m * P.ldc + nh) = oh;
            }
        });
    });
    if constexpr (kBitsFit && epi == 2) if (use_bits) P.bits[widx] = gbits;
    if constexpr (epi == 5) {
        // the four lane groups g hold different columns of the same 16 rows: sum over g (lane ^ 16 by ds_swizzle, lane ^ 32 by
        // v_permlane32_swap), then the lanes of group 0 write delta[(b * H + head) * T + t] of their row m = b * T + t
        const int head = (n0 + wn * (BN / WN)) >> 6;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float x = dl[i];
            x += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));      // bit mode: and 0x1F, xor 0x10
            x = rtts_xhalf_sum(x);
            if (g == 0) {
                const int m = mrow + 16 * i;
                const int b_ = m / P.T;
                P.aux_out[((size_t)b_ * P.H + head) * P.T + (m - b_ * P.T)] = x;
            }
        }
    }
    if constexpr (epi == 3) if (P.colsum != nullptr) {
        // sum over the wave's rows: the 16 lanes of a group hold 16 different rows of the same 4 columns
        float* dst = P.colsum + ((size_t)(m0 / BM) * WM + wm) * P.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = cs[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));     // lane ^ 1
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));     // lane ^ 2
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));    // row_half_mirror
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));    // row_mirror
                v[e] = x;
            }
            if (r == 0) *reinterpret_cast<f32x4*>(dst + ncol + 16 * j) = v;
        }
    }
    if constexpr (epi == 7) {
        // the same reduction over the wave's rows, for both moments: row (m0 / BM) * WM + wm of the partial buffer = [sum y | sum y^2]
        float* dst = P.colsum + ((size_t)(m0 / BM) * WM + wm) * (2 * (size_t)P.N);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 v = pl ? cs2[j] : cs[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e];
                    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));     // lane ^ 1
                    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));     // lane ^ 2
                    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));    // row_half_mirror
                    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));    // row_mirror
                    v[e] = x;
                }
                if (r == 0) *reinterpret_cast<f32x4*>(dst + pl * P.N + ncol + 16 * j) = v;
            }
    }
    if constexpr (MODE != 2) break;
    else {
        if (!has_next) break;
        // next tile: the ring already holds (or has in flight) its first stages, F0 of its stage 0 is in a0 / w0
        tile += gridDim.x;
        m0 = (int)(tile / ntn) * BM;
        n0 = (int)(tile % ntn) * BN;
        srcA0 += nxtA;
        srcW0 += nxtW;
        first_tile = false;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
#undef GN_READ
#undef GN_MFMA
#undef GN_WAIT_STAGE
#undef GN_ISSUE
#ifdef GN_STAMPS
    if constexpr (EPI == 0) {
        GN_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GN_STAMP(4);
        if (P.colsum != nullptr && tid == 0) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(P.colsum) + (size_t)blockIdx.x * 12;
            for (int i = 0; i < 5; ++i) { d[i] = st_c[i]; d[5 + i] = st_r[i]; }
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d[10] = xcc; d[11] = logical;
        }
    }
#endif
}

// ring depth: as deep as 160 KB of LDS allow, at most 4 stages
template <int BM, int BN>
constexpr int gn_nst() { return (4 * (BM + BN) * 128 <= 160 * 1024) ? 4 : ((3 * (BM + BN) * 128 <= 160 * 1024) ? 3 : 2); }

// TEST / A-B ONLY (rtts_debug_set_gemm_mode): how problems of several tiles per CU are launched.  0 = the library's pick,
// 1 = never persistent (round 3's forms), 2 = persistent on a 2-deep ring, two workgroups per CU, 3 = persistent on the
// deepest ring, one workgroup per CU.  Process-wide, atomic; no environment is read on the launch path.
static std::atomic<int> g_gn_mode{0};
static std::atomic<int> g_gn_store{-1};
extern "C" int rtts_debug_set_gemm_mode(int mode) {
    // 0..3: launch form (see above); 10 + s: store form s of the bf16 epilogues (0 = 8-byte, 1 = 16-byte, 2 = 16-byte write-through),
    // 9 = the library's pick of the store form again
    if (mode >= 9 && mode <= 12) { g_gn_store.store(mode - 10, std::memory_order_relaxed); return 0; }
    if (mode < 0 || mode > 3) { rtts_set_error("rtts_debug_set_gemm_mode: 0..3 or 9..12 (got %d)", mode); return -1; }
    g_gn_mode.store(mode, std::memory_order_relaxed);
    return 0;
}
// MEASURED (profiles/r04_gemm_nt_store_forms_ab.log, one box, interleaved, us per launch inside a replayed graph):
//   M = 12288, N x K = 512 x 512: 12.9 (8-byte) / 11.7 (16-byte) / 10.3 (16-byte write-through);  1024 x 512: 20.7 / 18.3 / 15.6;
//   2048 x 512: 41.7 / 36.4 / 33.4;  512 x 2048: 34.2 / 33.9 / 31.9;  M = 3072, 512 x 512: 6.3 / 6.1 / 5.8.
// With plain stores the whole output (12.6 .. 50 MB) sits in the XCDs' L2 as dirty lines when the kernel ends and is written back
// at the kernel boundary (B / 6 TB/s on top of the boundary itself: MI355X_MICROARCH.md, price list, "boundary"); written through,
// it leaves while the MFMAs still run -- the fabric is idle under an ingest-bound GEMM -- and nothing is left to flush.  No consumer
// loses an L2 hit: the next kernel's workgroups sit on all eight XCDs and read through their own L2 either way.
#ifndef GN_DEFAULT_STORE
#define GN_DEFAULT_STORE 2
#endif
// store form of a bf16 output: the 16-byte forms need 16-byte aligned rows
static int gn_store_mode(const void* c, int64_t ldc) {
    const int want = g_gn_store.load(std::memory_order_relaxed);
    const int m = want >= 0 ? want : GN_DEFAULT_STORE;
    return ((((uintptr_t)c & 15) == 0 && ldc % 8 == 0) ? m : 0);
}

template <int BM, int BN, int WM, int WN, bool W_KN, int EPI, int NST, int MODE>
static int gn_launch3(const GnGroup& G, int grid, hipStream_t s) {
    constexpr size_t lds = (size_t)NST * (BM + BN) * 128;
    static RttsLdsState attr;                    // per device: the dynamic-LDS limit is an attribute of the loaded function
    RTTS_ENSURE_LDS("rtts_gemm_nt", (gemm_nt_kernel<BM, BN, WM, WN, W_KN, EPI, NST, MODE>), lds, attr);
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, W_KN, EPI, NST, MODE>), dim3(grid), dim3(64 * WM * WN), lds, s, G);
    return 0;
}

template <int BM, int BN, int WM, int WN, bool W_KN, int EPI>
static int gn_launch2(const GnGroup& G, hipStream_t s) {
    constexpr int NST = gn_nst<BM, BN>();
    const GnArgs& P = G.p[0];
    const int tiles = (P.M / BM) * (P.N / BN);
    if constexpr (BM == 192 && BN == 128 && EPI != 4 && EPI != 5 && EPI != 7) {
        // Several tiles per CU (N >= 1024 at M = 12288: 512 / 1024 tiles).  Round 3 launched them all, two workgroups per CU on a
        // 2-deep ring, so that one workgroup's prologue / epilogue fills with the other's MFMAs (N = 2048, K = 512: 37.4 vs 45.5 us
        // on the deep ring; profiles/r02_gemm_nt_probe.log).  Round 4: a grid of RESIDENT workgroups walks the tiles (MODE 2)
        // and the ring keeps turning across the tile boundary -- measured forms in profiles/r04_gemm_nt_persistent_ab.log.
        const int mode = g_gn_mode.load(std::memory_order_relaxed);
        const bool multi = tiles >= 512 && tiles % 256 == 0 && P.K / GN_BK >= NST && P.conv_cpt == 0;
        // MEASURED (profiles/r04_gemm_nt_persistent_ab.log, one box, interleaved): the persistent forms LOSE -- N = 2048, K = 512:
        // 41.0 us one tile per workgroup, 44.9 persistent on the 2-deep ring, 49.8 on the deep ring; N = 1024: 20.3 / 21.3 / 24.2.
        // Two co-resident workgroups already hide each other's prologue and epilogue, the dispatcher re-fills a CU slot the moment a
        // workgroup retires (its stores drain while the successor's first stage loads), and a resident workgroup has to wait for its
        // own epilogue stores before the counted vmcnt of its next tile's second stage.  The library's pick stays round 3's form;
        // the persistent forms remain selectable (bit-identical results: tests/test_gemm_hip.py) for A/B runs on other shapes.
        if (multi && mode >= 2) {
            if (mode == 3) return gn_launch3<BM, BN, WM, WN, W_KN, EPI, NST, 2>(G, 256, s);
            return gn_launch3<BM, BN, WM, WN, W_KN, EPI, 2, 2>(G, 512, s);
        }
        if (tiles >= 512) return gn_launch3<BM, BN, WM, WN, W_KN, EPI, 2, 0>(G, tiles, s);
    }
    return gn_launch3<BM, BN, WM, WN, W_KN, EPI, NST, 0>(G, tiles, s);
}

template <int BM, int BN, int WM, int WN>
static int gn_launch(const GnGroup& G, int w_kn, hipStream_t s) {
    const GnArgs& P = G.p[0];
    if (w_kn) {                       // input gradients: plain / fp32 store or the ReLU gate of the FeedForward hidden layer
        if (P.epi == 3) return gn_launch2<BM, BN, WM, WN, true, 3>(G, s);
        if (P.epi == 0) return gn_launch2<BM, BN, WM, WN, true, 0>(G, s);
        if (P.epi == 4) return gn_launch2<BM, BN, WM, WN, true, 4>(G, s);
        if constexpr (BN / WN == 64) if (P.epi == 5) return gn_launch2<BM, BN, WM, WN, true, 5>(G, s);
        rtts_set_error("rtts_gemm_nt: a [K][N] weight (input gradient) takes epilogue 0, 3, 4 or 5 (5: tiles whose waves span one head), got %d", P.epi);
        return -1;
    }
    switch (P.epi) {
        case 0: return gn_launch2<BM, BN, WM, WN, false, 0>(G, s);
        case 1: return gn_launch2<BM, BN, WM, WN, false, 1>(G, s);
        case 2: return gn_launch2<BM, BN, WM, WN, false, 2>(G, s);
        case 3: return gn_launch2<BM, BN, WM, WN, false, 3>(G, s);
        case 4: return gn_launch2<BM, BN, WM, WN, false, 4>(G, s);
        case 7: return gn_launch2<BM, BN, WM, WN, false, 7>(G, s);
        default: rtts_set_error("rtts_gemm_nt: an [N][K] weight takes epilogue 0..4 or 7, got %d", P.epi); return -1;
    }
}

// grouped launch (MODE 1): every problem on one tile shape and one weight layout; epilogue 6 = each problem's own choice of
// {0, 1 (bias), 4 (fp32 store, + accumulate)} at run time
template <int BM, int BN, int WM, int WN>
static int gn_launch_group(const GnGroup& G, int total_tiles, int w_kn, hipStream_t s) {
    constexpr int NST = gn_nst<BM, BN>();
    // up to two workgroups per CU on the 2-deep ring take a grid that is not a whole number of one-per-CU waves (384 tiles of the
    // cross-attention's q + k|v projections) in one go; small grids keep the deep ring
    constexpr int NST2 = BM == 192 ? 2 : NST;
    const bool two = BM == 192 && total_tiles > 256;
    if (w_kn) {
        if (two) return gn_launch3<BM, BN, WM, WN, true, 6, NST2, 1>(G, total_tiles, s);
        return gn_launch3<BM, BN, WM, WN, true, 6, NST, 1>(G, total_tiles, s);
    }
    if (two) return gn_launch3<BM, BN, WM, WN, false, 6, NST2, 1>(G, total_tiles, s);
    return gn_launch3<BM, BN, WM, WN, false, 6, NST, 1>(G, total_tiles, s);
}

// Tile choice.  The kernel is bound by operand ingest (bytes per workgroup and K stage ~ BM + BN), so the largest tile wins
// as long as its grid fills the chip in WHOLE waves of one workgroup per CU: 256 x 256 when its workgroup count is a multiple
// of 256 (M = 16384, N >= 1024: 46 vs 54 us at N = 2048; at M = 12288 its 192 / 384 workgroups leave a quarter of the chip
// idle or a half-empty second wave and lose to 192 x 128: 23.4 vs 20.5 us, 43.1 vs 40.3 us; profiles/r02_gemm_nt_probe_256.log),
// else 192 x 128 (M = 12288, N = 512: exactly one workgroup per CU), 256 x 128, and the 4-wave tiles for small problems.
#define GN_NCAND 5
static const int gn_cand[GN_NCAND][3] = {{256, 256, 4}, {192, 128, 4}, {256, 128, 4}, {96, 64, 2}, {128, 64, 2}};   // BM, BN, WM
static int gn_pick(int M, int N) {
    static const int no256 = [] { const char* e = getenv("RTTS_GEMM_NT_NO256"); return e ? atoi(e) : 0; }();      // A/B runs
    int best = -1;
    for (int i = (no256 ? 1 : 0); i < GN_NCAND; ++i) {
        if (M % gn_cand[i][0] || N % gn_cand[i][1]) continue;
        const int wgs = (M / gn_cand[i][0]) * (N / gn_cand[i][1]);
        if (i == 0 && wgs % 256 != 0) continue;   // 256 x 256: whole waves only
        if (best < 0) best = i;                   // fallback: the first that tiles
        if (wgs >= 192) return i;
    }
    // nothing reaches 192 workgroups: take the smallest tile that fits (most workgroups)
    for (int i = GN_NCAND - 1; i >= GN_NCAND - 2; --i)
        if (M % gn_cand[i][0] == 0 && N % gn_cand[i][1] == 0) return i;
    return best;
}

extern "C" int rtts_gemm_nt_partial_rows(int M, int N) {
    const int i = gn_pick(M, N);
    return i < 0 ? -1 : (M / gn_cand[i][0]) * gn_cand[i][2];
}

static int gn_run(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, int M, int N, int K, void* c, int64_t ldc,
                  const float* bias, int epilogue, const void* gate, int64_t ldg, float* colsum_partial, int conv_cpt, int conv_sign,
                  int64_t conv_wtap, void* stream, unsigned long long* gate_bits = nullptr, const int* bn_geom = nullptr) {
    RTTS_REQUIRE(a && w && c, "rtts_gemm_nt: null pointer");
    RTTS_REQUIRE(M > 0 && N > 0 && K > 0 && K % GN_BK == 0, "rtts_gemm_nt: K must be a positive multiple of 64 (got M=%d N=%d K=%d)", M, N, K);
    RTTS_REQUIRE((epilogue >= 0 && epilogue <= 4) || (epilogue == 7 && bn_geom && colsum_partial && !w_is_kn),
                 "rtts_gemm_nt: epilogue 0..4 (5 and fp32 accumulation: rtts_gemm_nt_grouped; 7: rtts_conv1d_k5_moments)");
    RTTS_REQUIRE(!(epilogue == 1 || epilogue == 2) || bias, "rtts_gemm_nt: epilogue %d needs a bias", epilogue);
    RTTS_REQUIRE(epilogue != 3 || gate_bits || (gate && ldg >= N && ldg % 4 == 0),
                 "rtts_gemm_nt: epilogue 3 needs a gate (M, N) with ldg %% 4 == 0, or the forward's gate words");
    RTTS_REQUIRE(lda % 8 == 0 && (conv_cpt || lda >= K) && ldc >= N && ldc % 4 == 0, "rtts_gemm_nt: bad leading dimensions (lda=%lld ldc=%lld)",
                 (long long)lda, (long long)ldc);
    RTTS_REQUIRE(ldw % 8 == 0 && (conv_cpt || ldw >= (w_is_kn ? N : K)), "rtts_gemm_nt: bad ldw=%lld", (long long)ldw);
    RTTS_REQUIRE((((uintptr_t)a | (uintptr_t)w) & 15) == 0 && (((uintptr_t)c | (uintptr_t)gate) & ((epilogue == 4 || epilogue == 7) ? 15 : 7)) == 0 &&
                 (((uintptr_t)bias | (uintptr_t)colsum_partial) & 15) == 0, "rtts_gemm_nt: misaligned buffer");
    const int pick = gn_pick(M, N);
    RTTS_REQUIRE(pick >= 0, "rtts_gemm_nt: M x N = %d x %d tiles by none of 256x256, 192x128, 256x128, 96x64, 128x64", M, N);
    GnGroup G = {};
    G.n = 1;
    GnArgs& P = G.p[0];
    P.a = (const bf16_t*)a; P.w = (const bf16_t*)w; P.c = (bf16_t*)c; P.bias = bias; P.gate = (const bf16_t*)gate;
    P.colsum = colsum_partial; P.lda = lda; P.ldw = ldw; P.ldc = ldc; P.ldg = ldg; P.M = M; P.N = N; P.K = K; P.epi = epilogue;
    P.conv_cpt = conv_cpt; P.conv_sign = conv_sign; P.conv_wtap = conv_wtap;
    P.bits = gate_bits;
    P.store_mode = (epilogue == 4 || epilogue == 7) ? 0 : gn_store_mode(c, ldc);
    if (bn_geom) { P.bn_P = bn_geom[0]; P.bn_H = bn_geom[1]; P.bn_L = bn_geom[2]; P.bn_rows = bn_geom[3]; }
    RTTS_REQUIRE(!gate_bits || (pick != 0 && (epilogue == 2 || epilogue == 3) && ((uintptr_t)gate_bits & 7) == 0),
                 "rtts_gemm_nt_gated: gate words go with epilogue 2 (written) or 3 (read) and a tile shape that has them "
                 "(rtts_gemm_nt_gate_words(M, N) > 0)");
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
#ifdef GN_EXPERIMENT_ROWTILE
    // timing experiment only (scripts/build_ab.sh ... -DGN_EXPERIMENT_ROWTILE): a tile that spans the WHOLE row of an N = 512 product
    // (what a LayerNorm in the epilogue would need), one wave per 64 columns
    if (g_gn_mode.load(std::memory_order_relaxed) == 1 && N == 512 && !w_is_kn && epilogue == 0 && M % 64 == 0) {
        rc = gn_launch3<64, 512, 1, 8, false, 0, 2, 0>(G, M / 64, s);
        if (rc) return rc;
        RTTS_LAUNCH_CHECK("rtts_gemm_nt");
        return 0;
    }
#endif
    switch (pick) {
        case 0: rc = gn_launch<256, 256, 4, 2>(G, w_is_kn, s); break;
        case 1: rc = gn_launch<192, 128, 4, 2>(G, w_is_kn, s); break;
        case 2: rc = gn_launch<256, 128, 4, 2>(G, w_is_kn, s); break;
        case 3: rc = gn_launch<96, 64, 2, 2>(G, w_is_kn, s); break;
        default: rc = gn_launch<128, 64, 2, 2>(G, w_is_kn, s); break;
    }
    if (rc) return rc;
    RTTS_LAUNCH_CHECK("rtts_gemm_nt");
    return 0;
}

// ---- rtts_gemm_nt_grouped: n independent problems in one launch (n = 1: one problem with the extended epilogues)
static int gn_fill(GnArgs& P, const rtts_gemm_nt_problem& q, int w_is_kn, int i) {
    RTTS_REQUIRE(q.a && q.w && q.c, "rtts_gemm_nt_grouped: problem %d: null pointer", i);
    RTTS_REQUIRE(q.M > 0 && q.N > 0 && q.K > 0 && q.K % GN_BK == 0, "rtts_gemm_nt_grouped: problem %d: K must be a positive multiple of 64 (M=%d N=%d K=%d)",
                 i, q.M, q.N, q.K);
    RTTS_REQUIRE(q.lda % 8 == 0 && q.lda >= q.K && q.ldc >= q.N && q.ldc % 4 == 0 && q.ldw % 8 == 0 && q.ldw >= (w_is_kn ? q.N : q.K),
                 "rtts_gemm_nt_grouped: problem %d: bad leading dimensions", i);
    RTTS_REQUIRE(q.epilogue == 0 || q.epilogue == 1 || q.epilogue == 4 || q.epilogue == 5, "rtts_gemm_nt_grouped: problem %d: epilogue 0, 1, 4 or 5", i);
    RTTS_REQUIRE(q.epilogue != 1 || q.bias, "rtts_gemm_nt_grouped: problem %d: epilogue 1 needs a bias", i);
    RTTS_REQUIRE(!q.accumulate || q.epilogue == 4, "rtts_gemm_nt_grouped: problem %d: accumulate goes with the fp32 epilogue (4)", i);
    RTTS_REQUIRE((((uintptr_t)q.a | (uintptr_t)q.w | (uintptr_t)q.bias) & 15) == 0 && ((uintptr_t)q.c & (q.epilogue == 4 ? 15 : 7)) == 0,
                 "rtts_gemm_nt_grouped: problem %d: misaligned buffer", i);
    if (q.epilogue == 5) {
        RTTS_REQUIRE(w_is_kn && q.aux && q.aux_out && q.ld_aux >= q.N && q.ld_aux % 4 == 0 && ((uintptr_t)q.aux & 7) == 0 && q.T > 0 && q.M % q.T == 0 &&
                     q.H * 64 == q.N, "rtts_gemm_nt_grouped: problem %d: epilogue 5 (input gradient + delta) takes a [K][N] weight, aux = out (M, N) bf16, "
                     "aux_out = delta (B*H, T) f32, M = B*T, N = 64 H", i);
    }
    P = GnArgs{};
    P.a = (const bf16_t*)q.a; P.w = (const bf16_t*)q.w; P.c = (bf16_t*)q.c; P.bias = q.bias; P.lda = q.lda; P.ldw = q.ldw; P.ldc = q.ldc;
    P.M = q.M; P.N = q.N; P.K = q.K; P.epi = q.epilogue; P.aux = (const bf16_t*)q.aux; P.ldaux = q.ld_aux; P.aux_out = q.aux_out;
    P.T = q.T; P.H = q.H; P.accum = q.accumulate;
    P.store_mode = q.epilogue == 4 ? 0 : gn_store_mode(q.c, q.ldc);
    return 0;
}

extern "C" int rtts_gemm_nt_grouped(const rtts_gemm_nt_problem* problems, int n, int w_is_kn, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(problems && n >= 1 && n <= GN_MAX_GROUP, "rtts_gemm_nt_grouped: 1..%d problems", GN_MAX_GROUP);
    GnGroup G = {};
    G.n = n;
    for (int i = 0; i < n; ++i) {
        const int rc = gn_fill(G.p[i], problems[i], w_is_kn, i);
        if (rc) return rc;
        RTTS_REQUIRE(n == 1 || problems[i].epilogue != 5, "rtts_gemm_nt_grouped: epilogue 5 is a single-problem form (problem %d)", i);
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    if (n == 1) {
        const GnArgs& P = G.p[0];
        if (P.epi == 5) {
            // waves that span exactly one 64-wide head: 192 x 128 when its grid fills the chip, else 128 x 64 as four row waves
            if (P.M % 192 == 0 && P.N % 128 == 0 && (P.M / 192) * (P.N / 128) >= 192) rc = gn_launch<192, 128, 4, 2>(G, w_is_kn, s);
            else if (P.M % 128 == 0 && P.N % 64 == 0) rc = gn_launch<128, 64, 4, 1>(G, w_is_kn, s);
            else if (P.M % 192 == 0 && P.N % 128 == 0) rc = gn_launch<192, 128, 4, 2>(G, w_is_kn, s);
            else { rtts_set_error("rtts_gemm_nt_grouped: epilogue 5 needs M x N = %d x %d to tile by 192 x 128 or 128 x 64", P.M, P.N); return -1; }
        } else {
            const int pick = gn_pick(P.M, P.N);
            RTTS_REQUIRE(pick >= 0, "rtts_gemm_nt_grouped: M x N = %d x %d tiles by none of 256x256, 192x128, 256x128, 96x64, 128x64", P.M, P.N);
            switch (pick) {
                case 0: rc = gn_launch<256, 256, 4, 2>(G, w_is_kn, s); break;
                case 1: rc = gn_launch<192, 128, 4, 2>(G, w_is_kn, s); break;
                case 2: rc = gn_launch<256, 128, 4, 2>(G, w_is_kn, s); break;
                case 3: rc = gn_launch<96, 64, 2, 2>(G, w_is_kn, s); break;
                default: rc = gn_launch<128, 64, 2, 2>(G, w_is_kn, s); break;
            }
        }
    } else {
        // one tile shape for the whole group: 192 x 128 when every problem tiles by it and the grid is worth it, else 96 x 64 / 128 x 64
        auto tiles_by = [&](int bm, int bn) {
            int t = 0;
            for (int i = 0; i < n; ++i) {
                if (G.p[i].M % bm || G.p[i].N % bn) return -1;
                t += (G.p[i].M / bm) * (G.p[i].N / bn);
            }
            return t;
        };
        int bm = 192, bn = 128, total = tiles_by(192, 128);
        if (total < 192) {
            const int t96 = tiles_by(96, 64), t128 = tiles_by(128, 64);
            if (t96 > 0) { bm = 96; bn = 64; total = t96; }
            else if (t128 > 0) { bm = 128; bn = 64; total = t128; }
        }
        RTTS_REQUIRE(total > 0, "rtts_gemm_nt_grouped: the problems share none of the tile shapes 192x128, 96x64, 128x64");
        int acc = 0;
        G.spread = 1;
        for (int i = 0; i < n; ++i) {
            const int t = (G.p[i].M / bm) * (G.p[i].N / bn);
            if (t % 8) G.spread = 0;
            acc += t;
            G.tile_end[i] = acc;
        }
        if (bm == 192) rc = gn_launch_group<192, 128, 4, 2>(G, total, w_is_kn, s);
        else if (bm == 96) rc = gn_launch_group<96, 64, 2, 2>(G, total, w_is_kn, s);
        else rc = gn_launch_group<128, 64, 2, 2>(G, total, w_is_kn, s);
    }
    if (rc) return rc;
    RTTS_LAUNCH_CHECK("rtts_gemm_nt_grouped");
    return 0;
}

extern "C" int rtts_gemm_nt(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, int M, int N, int K, void* c,
                            int64_t ldc, const float* bias, int epilogue, const void* gate, int64_t ldg, float* colsum_partial,
                            void* stream) {
    RTTS_ENTER(stream);
    return gn_run(a, lda, w, ldw, w_is_kn, M, N, K, c, ldc, bias, epilogue, gate, ldg, colsum_partial, 0, 0, 0, stream);
}

extern "C" int64_t rtts_gemm_nt_gate_words(int M, int N) {
    const int i = gn_pick(M, N);
    if (i <= 0) return 0;          // no tile, or 256 x 256 (its lanes hold more outputs than a word has bits)
    return (int64_t)(M / gn_cand[i][0]) * (N / gn_cand[i][1]) * 64 * gn_cand[i][2] * 2;
}

extern "C" int rtts_gemm_nt_gated(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, int M, int N, int K, void* c,
                                  int64_t ldc, const float* bias, int epilogue, uint64_t* gate_words, float* colsum_partial, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(gate_words && (epilogue == 2 || epilogue == 3), "rtts_gemm_nt_gated: epilogue 2 (bias + ReLU, writes the words) or 3 (reads them)");
    return gn_run(a, lda, w, ldw, w_is_kn, M, N, K, c, ldc, bias, epilogue, nullptr, 0, colsum_partial, 0, 0, 0, stream,
                  (unsigned long long*)gate_words);
}

extern "C" int rtts_conv1d_k5(const void* x, int64_t ldx, const void* wp, int64_t ldw, int transposed, int M, int C_out, int C_in,
                              void* y, int64_t ldy, const float* bias, int out_f32, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(C_in > 0 && C_in % GN_BK == 0 && C_out > 0, "rtts_conv1d_k5: channel counts must be multiples of 64 (got %d -> %d)", C_in, C_out);
    RTTS_REQUIRE(!bias || out_f32, "rtts_conv1d_k5: a bias rides in the fp32 epilogue only");
    // forward:    y[m][co] = sum_{tap, ci} x[m + tap - 2][ci] * wp[co][tap * C_in + ci]              (wp (C_out, 5 C_in), NT)
    // transposed: y[m][co] = sum_{tap, ci} x[m - tap + 2][ci] * wp[ci][tap * C_out + co]             (wp (C_in, 5 C_out), [K][N])
    return gn_run(x, ldx, wp, ldw, transposed ? 1 : 0, M, C_out, 5 * C_in, y, ldy, bias, out_f32 ? 4 : 0, nullptr, 0, nullptr,
                  C_in / GN_BK, transposed ? -1 : 1, transposed ? (int64_t)C_out : 0, stream);
}

// The forward convolution in front of a BatchNorm: y (fp32, unrounded) as rtts_conv1d_k5(out_f32 = 1, no bias) AND, from the same
// accumulators, the per-channel sums of y and y^2 over the rows that carry data (halo layout: B sequences of L rows, `halo` zero
// rows on either side of each) as rtts_gemm_nt_partial_rows(M, C_out) partial rows of [sum y | sum y^2] (2 C_out floats each):
// what rtts_bn_stats re-read all of y for.  rtts_bn_stats_from_partials finishes them.
extern "C" int rtts_conv1d_k5_moments(const void* x, int64_t ldx, const void* wp, int64_t ldw, int M, int C_out, int C_in, void* y, int64_t ldy,
                                      int B, int L, int halo, float* partial, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(C_in > 0 && C_in % GN_BK == 0 && C_out > 0, "rtts_conv1d_k5_moments: channel counts must be multiples of 64 (got %d -> %d)", C_in, C_out);
    RTTS_REQUIRE(partial && B > 0 && L > 0 && halo >= 0 && (long long)B * (L + 2 * halo) <= M, "rtts_conv1d_k5_moments: bad geometry");
    const int geom[4] = {L + 2 * halo, halo, L, B * (L + 2 * halo)};
    return gn_run(x, ldx, wp, ldw, 0, M, C_out, 5 * C_in, y, ldy, nullptr, 7, nullptr, 0, partial, C_in / GN_BK, 1, 0, stream, nullptr, geom);
}
