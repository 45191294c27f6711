// LSH chunked attention backward: one workgroup per (batch*head, sorted chunk).
//
// Backward of SURVEY.md Appendix B steps 4-11 (the reference gets it from autograd over
// reformer_pytorch's eager graph, reached from reformer_tts/model/reversible.py:69-85).
// The n_hashes rounds form ONE softmax over the multiset of (round, key) pairs:
//     out[q] = sum_{r,k} exp(s_rqk - LSE[q]) v[k],   LSE = logsumexp over rounds and keys,
// so with P' = exp(s - LSE) and delta[q] = out[q].dout[q]:
//     dV[k] += P'^T dout      dS = P' * (dout V^T - delta)   (0 where the logit was replaced
//     by the self constant)   dS' = dS * kscale   dQ = dS' K     G' = dS'^T Q,   dK = G' - k^ (k^.G')
// where kscale[k] = dh^-1/2 / |k| is the key normalisation folded into the logits.
//
// Layout: KEY ON THE LANE.  Wave w owns keys [32w, 32w+32) of the chunk's 2*BS keys for all BS
// queries: S and dP come out of the MFMA as [query rows in registers][key on lane], which IS the
// B operand of the dV^T and G^T products (contraction over the register/row index), so dV and dK
// of a key are complete inside one wave, in registers.  Only dS crosses LDS, once, as dS^T, for
// dQ^T = K^T dS'^T (wave w finishes (query tile w/2, dh half w%2)).  Nothing is accumulated across workgroups:
// each workgroup writes its rows of two dqk slots / two dv slots at UNSORTED positions (slot 0: the own chunk's
// rows, query role + key role added on chip; slot 1: key role of the looked-back chunk's rows), every row of every
// slot exactly once, and rtts_lsh_bwd_reduce sums slots and rounds -- deterministic, no atomics.
#include "rtts_common.h"
#include <float.h>

#define AB_DH 64
#define AB_ROWB 128   // row staging: unpadded 128-byte rows, swizzled (ab_stg_w / ab_stg_r)
#ifndef AB_SPLIT_EPI
#define AB_SPLIT_EPI 1   // walking kernel: a step's row stores run beside the next step's main loop (0: round 3's form, for A/B runs)
#endif
#ifndef AB_DQ_UNROLL
#define AB_DQ_UNROLL 16
#endif
#ifndef AB_UNROLL
#define AB_UNROLL 4   // query-tile loop fully unrolled: the tail of tile i (dV, G MFMAs, dS stores) overlaps the head of tile i+1; scripts/phase_probe.py --unroll N measures others
#endif

typedef __attribute__((ext_vector_type(8))) short short8v;

// Phase probe (scripts/phase_probe.py builds a private copy of this file with -DAB_PHASE_TIMING): wave 0 of every
// workgroup records the shader clock at the phase boundaries.  Never defined in the product build.
#ifdef AB_PHASE_TIMING
__device__ unsigned long long g_ab_phase[32 * 8192];
#define AB_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_ab_phase[blockIdx.x * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int rtts_debug_ab_phases(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ab_phase), sizeof(g_ab_phase)); }
#define AB_WSTAMP(i, w) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) g_ab_phase[blockIdx.x * 32 + (i) + (w)] = __builtin_readcyclecounter(); } while (0)
#else
#define AB_STAMP(i) do { } while (0)
#define AB_WSTAMP(i, w) do { } while (0)
#endif

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p1);
    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// ---- LDS images ---------------------------------------------------------------------------------------------------
// K/Q and dout images: [row][128 B], no padding; the eight 16-byte pieces of a row are XOR-swizzled with
//     sw(row) = row bits (2,1) | (row bit 3 ^ row bit 1) << 2
// so that BOTH access shapes are bank-conflict free (64 banks x 4 B): ds_read_b128 of one piece from 16 rows (the MFMA
// A/B fragments) and ds_read_b64_tr_b16 of 4 consecutive rows x 64 B (the transposed fragments).  With 144-byte padded
// rows the transposed reads were 2-way conflicted (rows j and j+2 overlap in 8 banks).
__device__ __forceinline__ int ab_sw(int row) { return ((row >> 1) & 3) | ((((row >> 3) ^ (row >> 1)) & 1) << 2); }
__device__ __forceinline__ int ab_off(int row, int piece) { return row * 128 + ((piece ^ ab_sw(row)) << 4); }
// dS'^T image: [key][BS queries] bf16, no padding, 8-byte granules (4 queries) XOR-swizzled by the key so that the
// ds_write_b64 of 16 consecutive keys (banks mod 32) and the transposed read of 4 consecutive keys x 64 B (banks mod 64)
// are both conflict free (the padded image was 4-way conflicted on the read side: the dQ phase ran at LDS speed / 4).
template <int BS>
__device__ __forceinline__ int ab_ds_off(int key, int gran) {
    const int k0 = key & 1, k1 = (key >> 1) & 1, k2 = (key >> 2) & 1, k3 = (key >> 3) & 1;
    if (BS == 128) return key * 256 + ((gran ^ ((k1 << 4) | (k0 << 3) | (k1 << 2) | (k2 << 1) | k3)) << 3);
    return key * 128 + ((gran ^ ((k1 << 3) | (k0 << 2) | (k2 << 1) | k3)) << 3);
}

// Row staging of the epilogues: [32 keys][128 B].  The accumulators hold a key per lane, so lane (r, hh) writes the 8 bytes
// (16-byte piece p, half hh) of row r with ds_write_b64 (16 consecutive lanes = 16 rows per LDS cycle, banks mod 32: the
// 16 rows must land on the 16 distinct 8-byte slots of a 128-byte window) and the rows leave as 16-byte pieces read with
// ds_read_b128 (lane = (row & 7, piece), groups of 16 lanes = 4 rows x 4 pieces, banks mod 64: rows of equal parity must
// hold different pieces).  piece ^ (row & 7) serves both; the half is swapped in rows with bit 3 set, which the reader
// undoes for free: bit 3 of its row is the compile-time index of the read.  (The padded [32][144 B] staging of rounds 1-3 was
// 2-way conflicted on both sides: 10.9 % of this kernel's LDS cycles; tests/test_lds_swizzle_model.py holds the model.)
__device__ __forceinline__ int ab_stg_w(int r, int piece, int hh) { return r * 128 + ((piece ^ (r & 7)) << 4) + ((hh ^ ((r >> 3) & 1)) << 3); }
__device__ __forceinline__ int ab_stg_r(int i, int srow, int spiece) { return (i * 8 + srow) * 128 + ((spiece ^ srow) << 4); }
__device__ __forceinline__ uint4 ab_stg_fix(int i, const uint4 v) { return (i & 1) ? uint4{v.z, v.w, v.x, v.y} : v; }

// AB_KT2 = 32-key tiles owned by one wave.  1: 8 waves per 128-key chunk pair, two waves per SIMD (<= 256 registers each).
// 2: 4 waves, ONE wave per SIMD with the whole 512-entry register file: a query tile's fragments, per-query words and
// transposed fragments are read once for two key tiles, and the two tiles' MFMA bursts and softmax arithmetic are
// independent instruction streams the scheduler can interleave inside one wave.
#ifndef AB_KT2
#define AB_KT2 1
#endif

// DROP: dropout on the attention probabilities (lsh_attn_fwd.hip): with keep = 0 | 1/(1-p) from the same counter hash of the
// pair index, dV += (keep P')^T dout and dS = P' (keep dP - delta); delta = out . dout already contains the mask.
// |half a key row|^2 from its four fragments: v_dot2c_f32_bf16 on the packed pairs (16 instructions in four chains of four;
// unpack + fma was 64 in one chain of 32 dependent fmas) -- the same arithmetic as the forward kernel's norms.
typedef __attribute__((ext_vector_type(2))) __bf16 ab_bf2;
__device__ __forceinline__ float ab_frags_sumsq(const bf16x8 (&f)[4]) {
    float part[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 u = __builtin_bit_cast(uint4, f[ks]);
        float a = 0.f;
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ab_bf2, u.x), __builtin_bit_cast(ab_bf2, u.x), a, false);
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ab_bf2, u.y), __builtin_bit_cast(ab_bf2, u.y), a, false);
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ab_bf2, u.z), __builtin_bit_cast(ab_bf2, u.z), a, false);
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ab_bf2, u.w), __builtin_bit_cast(ab_bf2, u.w), a, false);
        part[ks] = a;
    }
    return (part[0] + part[1]) + (part[2] + part[3]);
}

struct AbDrop {
    uint32_t seed;
    const uint32_t* seed_dev;
    uint32_t thresh;
    float scale;
};

template <int BS, bool CAUSAL, bool MASKED, bool DROP>
__global__ __launch_bounds__(BS * 4 / AB_KT2, AB_KT2 == 2 ? 1 : 2) void lsh_attn_bwd_kernel(
    const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v, int64_t ld, const int32_t* __restrict__ st,
    const uint8_t* __restrict__ mask, const bf16_t* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse_tot,
    const float* __restrict__ delta, int H, int T, int n_hashes, bf16_t* __restrict__ dqk_part, bf16_t* __restrict__ dv_part,
    size_t slot_stride, AbDrop dr) {
    constexpr int KT2 = AB_KT2;
    constexpr int NK = 2 * BS;
    constexpr int NQT = BS / 32;
    constexpr int NW = NK / (32 * KT2);
    constexpr int NTHR = 64 * NW;
    constexpr int DSROW = BS * 2;         // bytes per row of the dS^T image [key][query]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // per-row words first: their addresses then fit the 16-bit offset field of the DS instructions
    // (the key-side scale and effective position of a wave's keys live in its registers; word arrays 0 and 2 of
    // the region are unused)
    int* kpos = reinterpret_cast<int*>(smem) + NK;             // original position of every row (self test, row stores)
    float* qlse = reinterpret_cast<float*>(kpos + 2 * NK);     // lse_tot * log2(e)
    float* qdel = qlse + BS;                                   // MINUS delta: the dP accumulator starts from it
    int* qpe_s = reinterpret_cast<int*>(qdel + BS);            // query-side effective position (-1: an invalid query)
    unsigned char* Ks = reinterpret_cast<unsigned char*>(qpe_s + BS);   // [NK][128]  qk rows (own chunk first), swizzled
    unsigned char* Os = Ks + NK * 128;                         // [BS][128]  dout rows of the queries, swizzled
    unsigned char* Ds = Os + BS * 128;                         // [NK][DSROW] dS'^T, swizzled
    unsigned char* Stg = Ds + NK * DSROW;                      // [NK / 32][32][144] per-key-tile staging of the row stores

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / C, c = wi % C;
    const int b = bh / H, h = bh % H;
    const int cprev = (c == 0) ? C - 1 : c - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int r = lane & 31, hh = lane >> 5;

    AB_STAMP(0);
    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* dobase = dout + (size_t)b * T * ld_do + (size_t)h * AB_DH;

    // ---- gather K rows (all 2*BS) and dout rows (own chunk) into LDS -----------------------
    // Two dependent global round trips, not three: everything that is indexed by the token position (rows AND the
    // per-token words mask / lse / delta) is requested as soon as the positions are known.
    constexpr int ITERS = NK * 8 / NTHR;   // 4 (KT2 = 1) or 8
    int trow[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = (it * NTHR + tid) >> 3;
        const int slot = (row < BS) ? c * BS + row : cprev * BS + (row - BS);
        trow[it] = st_row[slot];
    }
    // this wave owns the KT2 key tiles wave*KT2 + k2, one key per lane (both halves)
    int myrow[KT2], mypos[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        myrow[k2] = (wave * KT2 + k2) * 32 + r;
        mypos[k2] = st_row[(myrow[k2] < BS) ? c * BS + myrow[k2] : cprev * BS + (myrow[k2] - BS)];
    }
#ifdef AB_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the positions are here
    AB_STAMP(10);
#endif
    // Rows go global -> LDS by DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write pass).  One wave-instruction
    // fills 8 consecutive 128-byte rows of the image in lane order, so the swizzle goes on the SOURCE side: the lane
    // that lands on physical piece (lane & 7) of row (lane >> 3) fetches logical piece (lane & 7) ^ sw(row).
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int rowb = it * (NTHR / 8) + wave * 8;            // wave-uniform: first row of this instruction
        const int lp = (lane & 7) ^ ab_sw(rowb + (lane >> 3));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (size_t)trow[it] * ld + lp * 8),
                                         (RTTS_LDS void*)(Ks + rowb * 128), 16, 0, 0);
        if (it < ITERS / 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dobase + (size_t)trow[it] * ld_do + lp * 8),
                                             (RTTS_LDS void*)(Os + rowb * 128), 16, 0, 0);
    }
    int rvalid[ITERS / 2];
    float rlse[ITERS / 2], rdel[ITERS / 2];
#pragma unroll
    for (int it = 0; it < ITERS / 2; ++it) {
        rvalid[it] = MASKED ? (int)mask[(size_t)b * T + trow[it]] : 1;
        rlse[it] = lse_tot[(size_t)bh * T + trow[it]];
        rdel[it] = delta[(size_t)bh * T + trow[it]];
    }
    // V fragments of this wave's keys go straight to registers (no other wave needs them)
    int myvalid[KT2];
    bf16x8 vf[KT2][4];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        myvalid[k2] = MASKED ? (int)mask[(size_t)b * T + mypos[k2]] : 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[k2][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)mypos[k2] * ld + ks * 16 + 8 * hh);
    }
#ifdef AB_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0x0F70);
    AB_STAMP(11);
#endif
    // query-side words: lane `piece` of a row's eight lanes stores word `piece` (3 lse*log2e, 4 -delta, 5 effective
    // position, 1 position); the key-side words of a wave's own keys never leave its registers
    const int cbase = (tid & 7) < 3 ? (tid & 7) * (NK * 4) : 3 * NK * 4 + ((tid & 7) - 3) * (BS * 4);
#pragma unroll
    for (int it = 0; it < ITERS / 2; ++it) {
        const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
        const int eff = CAUSAL ? trow[it] : 0;
        int w = trow[it];
        w = piece == 3 ? __float_as_int(rlse[it] * 1.4426950408889634f) : w;
        w = piece == 4 ? __float_as_int(-rdel[it]) : w;
        // an invalid query sees nothing but itself: its effective position is below every key's
        w = piece == 5 ? (rvalid[it] ? eff : -1) : w;
        if (piece == 1 || (piece >= 3 && piece < 6)) *reinterpret_cast<int*>(smem + cbase + row * 4) = w;
    }
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)   // looked-back rows: only this wave reads them back (row stores)
        if (hh == 0 && myrow[k2] >= BS) kpos[myrow[k2]] = mypos[k2];
    AB_STAMP(1);
    __syncthreads();
    AB_STAMP(2);

    // ---- this wave's key-side constants -----------------------------------------------------
    // byte offsets of the A/B fragment pieces (ks*2+hh) of row r inside any 32-row block of a swizzled image
    int fro[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fro[ks] = ab_off(r, ks * 2 + hh);
    bf16x8 kf[KT2][4];
    float ksc[KT2];
    int kpk[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[k2][ks] = *reinterpret_cast<const bf16x8*>(Ks + (wave * KT2 + k2) * (32 * 128) + fro[ks]);
        float ss = ab_frags_sumsq(kf[k2]);
        ss = rtts_xhalf_sum(ss);
        ksc[k2] = 0.125f * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // dh^-1/2 / max(|k|, 1e-12)
        kpk[k2] = myvalid[k2] ? (CAUSAL ? mypos[k2] : 0) : 0x40000000;
    }

    f32x16 dvacc[KT2][2], gacc[KT2][2];   // [key tile][dh tile]: rows = dh, lane = key
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            dvacc[k2][d] = (f32x16){0};
            gacc[k2][d] = (f32x16){0};
        }
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
    // transposed fragments: lane -> (row 4*hh + trq (+8 for the second read), 8-byte granule dt*8 + 4*trc + trp);
    // sw(row + 8) = sw(row) ^ 4 and dt toggles the same piece bit, so the second read of tile dt sits at tro[dt ^ 1] + 8 rows
    int tro[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) tro[dt] = ab_off(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);
    // dS'^T store offsets of this lane's key rows at query tile 0; tile qt is granule + 8*qt = byte offset ^ (qt << 6)
    // (the row base is a multiple of 128 resp. 256 bytes, so bits 6.. of the offset belong to the granule index alone)
    int dso[KT2][4];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)
#pragma unroll
        for (int g = 0; g < 4; ++g) dso[k2][g] = ab_ds_off<BS>(myrow[k2], 2 * g + hh);
    const bool own_tile = wave * KT2 < BS / 32;  // wave-uniform (SGPR); the KT2 tiles of a wave sit on the same side
    const bool wrap = (cprev / nb) != (c / nb);
#if defined(AB_STAGGER) && AB_STAGGER > 0
    // The two waves of a SIMD run the same program from the same barrier: their MFMA bursts and their softmax arithmetic
    // coincide, so the matrix pipe idles while both do vector work.  Holding the second-dispatched half back by a fraction
    // of a tile puts one wave's vector phase under the other's MFMA burst (MI355X_MICROARCH.md, two waves per SIMD, item 9).
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= NTHR / 2) __builtin_amdgcn_s_sleep(AB_STAGGER);
#endif

#pragma unroll AB_UNROLL
    for (int qt = 0; qt < NQT; ++qt) {
        // dP starts at -delta[q] (read first: the second MFMA below waits for it): the accumulator then holds dP - delta,
        // one subtraction per logit less on the VALU
        f32x16 pinit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 d4 = *reinterpret_cast<const float4*>(qdel + qt * 32 + 8 * g + 4 * hh);
            pinit[4 * g] = d4.x;
            pinit[4 * g + 1] = d4.y;
            pinit[4 * g + 2] = d4.z;
            pinit[4 * g + 3] = d4.w;
        }
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Ks + qt * (32 * 128) + fro[ks]);
            dof[ks] = *reinterpret_cast<const bf16x8*>(Os + qt * (32 * 128) + fro[ks]);
        }
        // q-side row constants of this tile, issued now so that their LDS latency hides behind the MFMAs below
        float4 l4[4];
        int4 e4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = qt * 32 + 8 * g + 4 * hh;
            l4[g] = *reinterpret_cast<const float4*>(qlse + q0);
            e4[g] = *reinterpret_cast<const int4*>(qpe_s + q0);
        }
        // A fragments of the transposed products (element j <-> query 16*s2 + 8*(j>>2) + 4*hh + (j&3))
        bf16x8 qtf[2][2], dotf[2][2];   // [s2][dh tile]
        if (KT2 == 2) {                 // read once for both key tiles (at KT2 = 1 they are read late: registers)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int blk = (qt * 32 + 16 * s2) * 128;
                    qtf[s2][dt] = tr_frag(Ks + blk + tro[dt], Ks + blk + 8 * 128 + tro[dt ^ 1]);
                    dotf[s2][dt] = tr_frag(Os + blk + tro[dt], Os + blk + 8 * 128 + tro[dt ^ 1]);
                }
        }
#pragma unroll
        for (int k2 = 0; k2 < KT2; ++k2) {
            // Can a key of this tile BE one of this tile's queries (the self logit)?  Own keys: only on the diagonal
            // tile.  Looked-back keys: only when the previous chunk belongs to another hash round (the chunk ring wraps
            // over rounds, so the same token can then sit in both chunks).  Wave-uniform: the common path skips the test.
            const bool chk_self = own_tile ? (wave * KT2 + k2 == qt) : wrap;
            f32x16 sacc = {0}, pacc = pinit;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[k2][ks], sacc, 0, 0, 0);    // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[k2][ks], pacc, 0, 0, 0);   // dP[q][key] - delta[q]
            }
            // P' = exp2(s*ksc*log2e - lse*log2e); dS' = P' (dP - delta) ksc  (0 at the self logit: it was a constant).
            // ksc multiplies dS' once here: G' = dS'^T Q then gives dK = G' - k^ (k^ . G'), and dQ^T = K^T dS'^T.
            uint32_t kbits = 0u;          // DROP: bit i = pair i of this tile is kept
            if constexpr (DROP) {
                // with dropout the softmax's backward sees keep * dP: pacc = dP - delta and pinit = -delta, so the masked
                // value is keep * (pacc - pinit) + pinit; the arithmetic below then runs unchanged, and P' gets its keep-scale
                // (for dV) after it
                const uint32_t seed = dr.seed + (dr.seed_dev ? dr.seed_dev[0] : 0u);
                const uint32_t pair0 = ((uint32_t)wi * BS + (uint32_t)(qt * 32 + 4 * hh)) * (uint32_t)NK + (uint32_t)myrow[k2];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool kept = rtts_drop_hash(seed, pair0 + (uint32_t)(8 * (i >> 2) + (i & 3)) * (uint32_t)NK) >= dr.thresh;
                    kbits |= kept ? (1u << i) : 0u;
                    pacc[i] = kept ? __builtin_fmaf(dr.scale, pacc[i] - pinit[i], pinit[i]) : pinit[i];
                }
            }
            float pp[16], ds[16];
            const float ksc2 = ksc[k2] * 1.4426950408889634f;
            if (chk_self) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w};
                    const int4 p4 = *reinterpret_cast<const int4*>(kpos + qt * 32 + 8 * g + 4 * hh);   // rare path: read here
                    const int pv[4] = {p4.x, p4.y, p4.z, p4.w}, ev[4] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * g + j;
                        const bool self = pv[j] == mypos[k2];
                        const bool dead = kpk[k2] > ev[j];
                        float x = sacc[i] * ksc2;
                        x = self ? (-5e4f * 1.4426950408889634f) : x;
                        float p = __builtin_amdgcn_exp2f(x - lv[j]);
                        p = (dead && !self) ? 0.f : p;
                        pp[i] = p;
                        ds[i] = self ? 0.f : p * pacc[i] * ksc[k2];
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w};
                    const int ev[4] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * g + j;
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], ksc2, -lv[j]));
                        p = (kpk[k2] > ev[j]) ? 0.f : p;
                        pp[i] = p;
                        ds[i] = p * pacc[i] * ksc[k2];
                    }
                }
            }
            if constexpr (DROP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pp[i] = ((kbits >> i) & 1u) ? pp[i] * dr.scale : 0.f;
            }
            if (KT2 == 1) {   // read only now, so that their registers are free during the softmax arithmetic above
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int blk = (qt * 32 + 16 * s2) * 128;
                        qtf[s2][dt] = tr_frag(Ks + blk + tro[dt], Ks + blk + 8 * 128 + tro[dt ^ 1]);
                        dotf[s2][dt] = tr_frag(Os + blk + tro[dt], Os + blk + 8 * 128 + tro[dt ^ 1]);
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
            // dS'^T[key][q] (bf16), 4 consecutive queries per 8-byte store
            const int dsq = qt << 6;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
                pk.y = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
                *reinterpret_cast<uint2*>(Ds + (dso[k2][g] ^ dsq)) = pk;
            }
        }
    }

    // ---- row stores.  The accumulators hold a key per lane and dh down the registers; each 32-key tile goes through
    //      a private [32][144 B] LDS staging so that a row leaves as eight 16-byte pieces (full 128-byte lines) instead
    //      of sixteen scattered 8-byte stores.  dV is final here: it leaves NOW, before the barrier, so that its HBM
    //      writes run under the dQ phase of the slower waves.
    AB_STAMP(3);
    AB_WSTAMP(16, wave);
    const int round = c / nb, round_prev = cprev / nb;
    const size_t obase = ((size_t)bh * n_hashes + (own_tile ? round : round_prev)) * T;
    const int srow = lane >> 3, spiece = lane & 7;
    int rpos[KT2][4];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        unsigned char* stg = Stg + (wave * KT2 + k2) * (32 * AB_ROWB);
#pragma unroll
        for (int i = 0; i < 4; ++i) rpos[k2][i] = kpos[(wave * KT2 + k2) * 32 + i * 8 + srow];
        bf16_t* dvdst = dv_part + (own_tile ? 0 : slot_stride);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(dvacc[k2][dt][4 * g], dvacc[k2][dt][4 * g + 1]);
                pk.y = pack_bf16x2(dvacc[k2][dt][4 * g + 2], dvacc[k2][dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(stg + ab_stg_w(r, dt * 4 + g, hh)) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        uint4 rowv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rowv[i] = ab_stg_fix(i, *reinterpret_cast<const uint4*>(stg + ab_stg_r(i, srow, spiece)));
#pragma unroll
        for (int i = 0; i < 4; ++i) rtts_store16_out(dvdst + (obase + rpos[k2][i]) * AB_DH + spiece * 8, rowv[i]);
    }

    AB_STAMP(4);
    AB_WSTAMP(24, wave);
    __syncthreads();   // every dS'^T tile is in Ds; nobody reads Os as dout any more

    AB_STAMP(5);
    // ---- dQ^T[dh][q] = K^T dS'^T over all 2*BS keys: the 2*NQT (query tile, dh half) outputs are dealt over the waves
    //      (KT2 = 1: one each; KT2 = 2: both dh halves of query tile `wave`, sharing the dS'^T fragments) and parked
    //      (bf16) in the dout image's rows: a chunk row is both a query and an own key, so its query-role and
    //      key-role gradients are added before they leave the chip
    {
        const int qt = (KT2 == 2) ? wave : (wave >> 1);
        f32x16 dq[KT2];
#pragma unroll
        for (int di = 0; di < KT2; ++di) dq[di] = (f32x16){0};
        const int rl = 8 * hh + trq;                       // key row inside a 16-key step (second read: +4)
        int ko0[KT2], ko1[KT2];
#pragma unroll
        for (int di = 0; di < KT2; ++di) {
            const int dt = (KT2 == 2) ? di : (wave & 1);
            const int kpc = dt * 4 + 2 * trc + (trp >> 1);     // 16-byte piece of the K row
            ko0[di] = ab_off(rl, kpc) + 8 * (trp & 1);
            ko1[di] = ab_off(rl + 4, kpc) + 8 * (trp & 1);
        }
        const int gq = qt * 8 + 4 * trc + trp;             // 8-byte granule of the dS^T row
        const int do0 = ab_ds_off<BS>(rl, gq), do1 = ab_ds_off<BS>(rl + 4, gq);
#pragma unroll AB_DQ_UNROLL
        for (int kb = 0; kb < NK; kb += 16) {
            const bf16x8 bfrag = tr_frag(Ds + kb * DSROW + do0, Ds + kb * DSROW + do1);
#pragma unroll
            for (int di = 0; di < KT2; ++di) {
                const bf16x8 afrag = tr_frag(Ks + kb * 128 + ko0[di], Ks + kb * 128 + ko1[di]);
                dq[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dq[di], 0, 0, 0);
            }
        }
#pragma unroll
        for (int di = 0; di < KT2; ++di) {
            const int dt = (KT2 == 2) ? di : (wave & 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(dq[di][4 * g], dq[di][4 * g + 1]);
                pk.y = pack_bf16x2(dq[di][4 * g + 2], dq[di][4 * g + 3]);
                *reinterpret_cast<uint2*>(Os + ab_off(qt * 32 + r, dt * 4 + g) + 8 * hh) = pk;
            }
        }
    }
    AB_STAMP(6);
    __syncthreads();   // dQ parked
    AB_STAMP(7);

    // ---- dK (+ dQ on own rows) of this wave's keys
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        unsigned char* stg = Stg + (wave * KT2 + k2) * (32 * AB_ROWB);
        bf16_t* dkdst = dqk_part + (own_tile ? 0 : slot_stride);
        // dK = G - k^ (k^ . G) with k^ = k / |k| = k * (8 ksc): on the raw bf16 row that is G - k * ((k . G) * (8 ksc)^2);
        // this lane holds 32 of the 64 dh, the partner half the other 32
        float kraw[2][16];
        float dot = 0.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 kk = *reinterpret_cast<const uint2*>(Ks + ab_off(myrow[k2], dt * 4 + g) + 8 * hh);
                kraw[dt][4 * g] = __uint_as_float(kk.x << 16);
                kraw[dt][4 * g + 1] = __uint_as_float(kk.x & 0xffff0000u);
                kraw[dt][4 * g + 2] = __uint_as_float(kk.y << 16);
                kraw[dt][4 * g + 3] = __uint_as_float(kk.y & 0xffff0000u);
#pragma unroll
                for (int j = 0; j < 4; ++j) dot = __builtin_fmaf(kraw[dt][4 * g + j], gacc[k2][dt][4 * g + j], dot);
            }
        const float ncoef = -rtts_xhalf_sum(dot) * (ksc[k2] * 8.f) * (ksc[k2] * 8.f);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dk[j] = __builtin_fmaf(kraw[dt][4 * g + j], ncoef, gacc[k2][dt][4 * g + j]);
                if (own_tile) {
                    const uint2 dqv = *reinterpret_cast<const uint2*>(Os + ab_off(myrow[k2], dt * 4 + g) + 8 * hh);
                    dk[0] += __uint_as_float(dqv.x << 16);
                    dk[1] += __uint_as_float(dqv.x & 0xffff0000u);
                    dk[2] += __uint_as_float(dqv.y << 16);
                    dk[3] += __uint_as_float(dqv.y & 0xffff0000u);
                }
                uint2 pk;
                pk.x = pack_bf16x2(dk[0], dk[1]);
                pk.y = pack_bf16x2(dk[2], dk[3]);
                *reinterpret_cast<uint2*>(stg + ab_stg_w(r, dt * 4 + g, hh)) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        uint4 rowv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rowv[i] = ab_stg_fix(i, *reinterpret_cast<const uint4*>(stg + ab_stg_r(i, srow, spiece)));
#pragma unroll
        for (int i = 0; i < 4; ++i) rtts_store16_out(dkdst + (obase + rpos[k2][i]) * AB_DH + spiece * 8, rowv[i]);
    }
    AB_STAMP(8);
#ifdef AB_PHASE_TIMING
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the row stores have been acknowledged
    AB_STAMP(9);
#endif
}



// LDS-DMA issued from inline assembly: the compiler's wait-count pass treats a __builtin_amdgcn_global_load_lds as a store
// to LDS that MAY alias any later ds_read (all images live in one dynamic array at run-time offsets), and makes the next
// LDS read of the main loop wait for the DMA to complete -- ~850 cycles per instruction (measured: the prefetch of 32 KB
// cost the main loop 3.4 k cycles).  The prefetched slot is read by nobody before the barrier at the top of the next step,
// which is preceded by an explicit s_waitcnt vmcnt(0), so the ordering the compiler tried to enforce is not needed.
// (Its own vmcnt(N) waits stay correct with these loads in the queue: returns are in order, extra entries only lengthen a wait.)
__device__ __forceinline__ void ab_dma16(const void* gptr, RTTS_LDS void* lds_dst_wave_uniform) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_wave_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(m0v) : "m0");
}

// ======================================================================================================================
// WALKING form: one workgroup works R consecutive chunks of one (batch, head) ring.
//
// The kernel above spends 6 k of its 22.5 k cycles per chunk in the gather (positions -> rows -> LDS image: two dependent
// round trips with the MFMA pipe idle) and holds one workgroup per CU, so nothing runs under it.  Consecutive chunks share
// rows: the own chunk of step j is the looked-back chunk of step j + 1.  Here the K rows live in a ring of three chunk slots
// (looked-back, own, next), the dout rows and the query words in two, the sort positions in four; while chunk j is worked,
// the rows of chunk j + 1 arrive by LDS-DMA (no registers held), the positions of chunk j + 2 and the query words of chunk
// j + 1 by plain loads that are stored to LDS late in the step.  After the first chunk of a run a step starts with its
// operands already on chip, and only half the K rows are fetched at all.  Same arithmetic, same outputs, same partial-row
// layout as the kernel above (the row staging reuses the dS'^T image, so dV leaves after the dQ product, not before it).
template <int BS, bool CAUSAL, bool MASKED, bool DROP>
__global__ __launch_bounds__(BS * 4, 2) void lsh_attn_bwd_walk_kernel(
    const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v, int64_t ld, const int32_t* __restrict__ st,
    const uint8_t* __restrict__ mask, const bf16_t* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse_tot,
    const float* __restrict__ delta, int H, int T, int n_hashes, bf16_t* __restrict__ dqk_part, bf16_t* __restrict__ dv_part,
    size_t slot_stride, int R, uint8_t* __restrict__ row_flags, AbDrop dr) {
    constexpr int NK = 2 * BS;
    constexpr int NQT = BS / 32;
    constexpr int NW = NK / 32;
    constexpr int NTHR = 64 * NW;
    constexpr int DSROW = BS * 2;
    constexpr int KSLOT = BS * 128;        // bytes of one chunk's row image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* kposS = reinterpret_cast<int*>(smem);                       // [4][BS] sort positions of the chunks in flight
    float* qlseS = reinterpret_cast<float*>(kposS + 4 * BS);         // [2][BS] lse_tot * log2(e)
    float* qdelS = qlseS + 2 * BS;                                   // [2][BS] MINUS delta
    int* qpeS = reinterpret_cast<int*>(qdelS + 2 * BS);              // [2][BS] query-side effective position
    unsigned char* KsR = reinterpret_cast<unsigned char*>(qpeS + 2 * BS);   // [3][BS][128] qk rows, swizzled
    unsigned char* OsR = KsR + 3 * KSLOT;                            // [2][BS][128] dout rows, later the parked dQ
    unsigned char* Ds = OsR + 2 * KSLOT;                             // [NK][DSROW] dS'^T; after the dQ product: row staging
    unsigned char* Stg = Ds;

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const int runs = C / R;
    const uint32_t run = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = run / runs, c0 = (run % runs) * R;
    const int b = bh / H, h = bh % H;
    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* dobase = dout + (size_t)b * T * ld_do + (size_t)h * AB_DH;

    // ---- prologue: the full gather of the first step (looked-back chunk -> slot 0, own chunk -> slot 1)
    {
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int cp0 = (c0 == 0) ? C - 1 : c0 - 1;
        constexpr int ITERS = NK * 8 / NTHR;   // 4
        int trow[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3;
            trow[it] = st_row[(row < BS) ? c0 * BS + row : cp0 * BS + (row - BS)];
        }
        int pnext = 0;
        if (R > 1 && tid < BS) pnext = st_row[(c0 + 1) * BS + tid];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rowb = it * (NTHR / 8) + wave * 8;            // wave-uniform: first row of this instruction
            const int lp = (lane & 7) ^ ab_sw(rowb + (lane >> 3));
            unsigned char* kdst = (rowb < BS) ? KsR + KSLOT + rowb * 128 : KsR + (rowb - BS) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qbase + (size_t)trow[it] * ld + lp * 8),
                                             (RTTS_LDS void*)kdst, 16, 0, 0);
            if (it < ITERS / 2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dobase + (size_t)trow[it] * ld_do + lp * 8),
                                                 (RTTS_LDS void*)(OsR + rowb * 128), 16, 0, 0);
        }
        int rvalid[ITERS / 2];
        float rlse[ITERS / 2], rdel[ITERS / 2];
#pragma unroll
        for (int it = 0; it < ITERS / 2; ++it) {
            rvalid[it] = MASKED ? (int)mask[(size_t)b * T + trow[it]] : 1;
            rlse[it] = lse_tot[(size_t)bh * T + trow[it]];
            rdel[it] = delta[(size_t)bh * T + trow[it]];
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
            if (piece == 1) kposS[(row < BS) ? BS + row : row - BS] = trow[it];      // own chunk: slot 1, looked-back: slot 0
            if (it < ITERS / 2) {
                if (piece == 3) qlseS[row] = rlse[it] * 1.4426950408889634f;
                if (piece == 4) qdelS[row] = -rdel[it];
                // an invalid query sees nothing but itself: its effective position is below every key's
                if (piece == 5) qpeS[row] = rvalid[it] ? (CAUSAL ? trow[it] : 0) : -1;
            }
        }
        if (R > 1 && tid < BS) kposS[2 * BS + tid] = pnext;
    }
#if AB_SPLIT_EPI
    // the builtin, so that the compiler's wait-count pass knows the gather's LDS-DMA has landed: it treats a pending LDS-DMA as a
    // store that may alias any LDS read, and would wait with vmcnt(0) in front of the first LDS read of EVERY step
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
#endif

    // Per-lane state that crosses a step: the K and V fragments, scale and position of this wave's 32 keys.  The wave groups
    // swap roles every step -- the group that works the OWN keys of chunk j works the same keys as LOOKED-BACK keys of chunk
    // j + 1 and keeps them in registers; only the other group fetches (the V rows of its next keys are requested as soon as
    // its main loop is done with the old ones, the K fragments come from the prefetched image).
    // The same holds for the gradients: the dV / dK accumulators of a chunk's keys run through BOTH steps (own role, then
    // looked-back role), so every key row is written ONCE, complete, with the query-role gradient of the row folded in when it
    // is parked (sdq remembers k . dQ, which the projection of the key-normalisation gradient must not see).  Only the ends of a
    // run leave partial rows: the looked-back keys of its first step (slot 1; their own part was the previous run's last step)
    // and the own keys of its last step (slot 0, flagged: rtts_lsh_bwd_reduce adds slot 1 only to flagged rows).
    bf16x8 vf[4], kf[4];
    float ksc = 0.f, sdq = 0.f;
    int myvalid = 1, mypos = 0, kpk = 0;
    f32x16 dvacc[2], gacc[2];   // [dh tile]: rows = dh, lane = key
    // (A static s_setprio 1 for the second-dispatched half of the workgroup -- MI355X_MICROARCH.md, two waves per SIMD, item 4 --
    //  measured null here: 220.9 against 221.3 / 220.4 us, profiles/r03_lsh_attn_bwd_prio_ab.log.)
#pragma unroll 1
    for (int j = 0; j < R; ++j) {
        // nothing else but scalars is carried from one step to the next: the lane id goes through an opaque move so that no address
        // arithmetic of the body is hoisted out of the loop and held in registers across it
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int r = lane & 31, hh = lane >> 5;
        const int c = c0 + j;
        const int cprev = (c == 0) ? C - 1 : c - 1;
        const int ks_own = (j + 1) % 3, ks_lb = j % 3, kp_own = (j + 1) & 3, kp_lb = j & 3, par = j & 1;
        unsigned char* Kown = KsR + ks_own * KSLOT;
        unsigned char* Klb = KsR + ks_lb * KSLOT;
        unsigned char* Os = OsR + par * KSLOT;
        const float* qlse = qlseS + par * BS;
        const float* qdel = qdelS + par * BS;
        const int* qpe_s = qpeS + par * BS;
        const int* kq = kposS + kp_own * BS;                   // positions of the own chunk's rows (queries = own keys)
        const int* kl = kposS + kp_lb * BS;

#define AB_JSTAMP(i) do { if (j == 2) AB_STAMP(i); } while (0)
        AB_JSTAMP(0);
#if AB_SPLIT_EPI
        // Only the first step waits here (for the prologue's gather).  Later steps: every wave has waited for its share of the
        // prefetch in front of the barrier that ends the previous step's dQ phase, and the previous step's row stores are NOT
        // waited for -- the group that stores rows (four of the eight waves) arrives late at this step's main loop, the other
        // group is already in it: one wave of each group shares a SIMD, so the stores' staging / store issue of one wave runs
        // under the MFMAs of the other instead of in front of an idle matrix pipe.
        // (that first wait sits in front of the loop)
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of the prefetch (issued from assembly) has landed
        __syncthreads();   // this step's rows and words are on chip (each wave waited for its own DMA); the previous step is over
#endif
        AB_JSTAMP(1);

        // ---- prefetch of chunk j + 1 (rows by LDS-DMA, query words, positions of chunk j + 2): issued piecewise INSIDE the main
        //      loop below -- a CU takes in ~31 B/clk beside the MFMA work, so 32 KB of rows issued in one go would hold every wave
        //      at the vector-memory queue for ~3 k cycles (measured) before the step's arithmetic starts
        const bool more = j + 1 < R;
        const int* kn = kposS + ((j + 2) & 3) * BS;
        unsigned char* Knext = KsR + ((j + 2) % 3) * KSLOT;
        unsigned char* Onext = OsR + (par ^ 1) * KSLOT;
        float nlse = 0.f, ndel = 0.f;
        int nval = 1, npos = 0, p2 = 0;
        constexpr int PF_IT = BS * 8 / NTHR;                 // 2: DMA instructions per wave and operand
        constexpr int PF_PER_TILE = 2 * PF_IT / NQT;         // 1 (BS = 128) or 2 (BS = 64) per query tile
        static_assert(PF_PER_TILE * NQT == 2 * PF_IT, "prefetch pieces must tile the query loop");
        int pfpos[PF_IT];                                    // read now: no LDS round trip in front of a DMA issue inside the loop
#pragma unroll
        for (int it = 0; it < PF_IT; ++it) pfpos[it] = more ? kn[it * (NTHR / 8) + wave * 8 + (lane >> 3)] : 0;
        AB_JSTAMP(2);
        // ---- this wave's key tile
        const int grp = wave / (NW / 2), wt = wave % (NW / 2);   // wave-uniform
        const bool own_tile = grp == (j & 1);                    // the groups take turns: own keys in one step, looked-back in the next
        const int myrow = (own_tile ? 0 : BS) + wt * 32 + r;     // row of the dS'^T image: own keys first
        const unsigned char* Kt = (own_tile ? Kown : Klb) + wt * (32 * 128);
        int fro[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fro[ks] = ab_off(r, ks * 2 + hh);
        if (own_tile || j == 0) {          // fresh keys (the looked-back group of the first step has nothing to keep either)
            const int pos_now = own_tile ? kq[wt * 32 + r] : kl[wt * 32 + r];
            if (j == 0) {                  // later steps: V rows and validity were requested by the previous one
                myvalid = MASKED ? (int)mask[(size_t)b * T + pos_now] : 1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)pos_now * ld + ks * 16 + 8 * hh);
#if AB_SPLIT_EPI
                // used HERE (an empty asm that reads them), on the first step's path only: the compiler waits for the loads in front of
                // it; where the two paths into the code below meet it would otherwise wait with vmcnt(0) in every step -- behind the
                // previous step's row stores, which leave from asm statements it cannot count
                asm volatile("" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(myvalid));
#endif
            }
            mypos = pos_now;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(Kt + fro[ks]);
            float ss = ab_frags_sumsq(kf);
            ss = rtts_xhalf_sum(ss);
            ksc = 0.125f * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // dh^-1/2 / max(|k|, 1e-12)
            kpk = myvalid ? (CAUSAL ? mypos : 0) : 0x40000000;
        }

        if (own_tile || j == 0) {          // fresh keys start from zero; looked-back keys continue what their own step left
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                dvacc[d] = (f32x16){0};
                gacc[d] = (f32x16){0};
            }
            sdq = 0.f;
        }
        const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
        int tro[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) tro[dt] = ab_off(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);
        int dso[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) dso[g] = ab_ds_off<BS>(myrow, 2 * g + hh);
        const bool wrap = (cprev / nb) != (c / nb);
#ifdef AB_PHASE_TIMING
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): V fragments (and this wave's share of the prefetch) have arrived
        AB_JSTAMP(3);
#endif

#pragma unroll AB_UNROLL
        for (int qt = 0; qt < NQT; ++qt) {
            if (more) {
#pragma unroll
                for (int pf = 0; pf < PF_PER_TILE; ++pf) {
                    const int piece = qt * PF_PER_TILE + pf;         // 0 .. 2 * PF_IT - 1: (iteration, K | dout)
                    const int it = piece >> 1;
                    const int rowb = it * (NTHR / 8) + wave * 8;
                    const int pos = pfpos[it];
                    const int lp = (lane & 7) ^ ab_sw(rowb + (lane >> 3));
                    if ((piece & 1) == 0)
                        ab_dma16(qbase + (size_t)pos * ld + lp * 8, (RTTS_LDS void*)(Knext + rowb * 128));
                    else
                        ab_dma16(dobase + (size_t)pos * ld_do + lp * 8, (RTTS_LDS void*)(Onext + rowb * 128));
                }
                if (qt == 0 && tid < BS) {
                    npos = kn[tid];
                    nval = MASKED ? (int)mask[(size_t)b * T + npos] : 1;
                    nlse = lse_tot[(size_t)bh * T + npos];
                    ndel = delta[(size_t)bh * T + npos];
                    if (j + 2 < R) p2 = st_row[(c + 2) * BS + tid];
                }
            }
            f32x16 pinit;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 d4 = *reinterpret_cast<const float4*>(qdel + qt * 32 + 8 * g + 4 * hh);
                pinit[4 * g] = d4.x;
                pinit[4 * g + 1] = d4.y;
                pinit[4 * g + 2] = d4.z;
                pinit[4 * g + 3] = d4.w;
            }
            bf16x8 qf[4], dof[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[ks] = *reinterpret_cast<const bf16x8*>(Kown + qt * (32 * 128) + fro[ks]);
                dof[ks] = *reinterpret_cast<const bf16x8*>(Os + qt * (32 * 128) + fro[ks]);
            }
            float4 l4[4];
            int4 e4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qt * 32 + 8 * g + 4 * hh;
                l4[g] = *reinterpret_cast<const float4*>(qlse + q0);
                e4[g] = *reinterpret_cast<const int4*>(qpe_s + q0);
            }
            const bool chk_self = own_tile ? (wt == qt) : wrap;
            f32x16 sacc = {0}, pacc = pinit;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], sacc, 0, 0, 0);    // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[ks], pacc, 0, 0, 0);   // dP[q][key] - delta[q]
            }
            uint32_t kbits = 0u;          // DROP: bit i = pair i of this tile is kept
            if constexpr (DROP) {
                const uint32_t seed = dr.seed + (dr.seed_dev ? dr.seed_dev[0] : 0u);
                const uint32_t pair0 = (((uint32_t)bh * C + (uint32_t)c) * BS + (uint32_t)(qt * 32 + 4 * hh)) * (uint32_t)NK + (uint32_t)myrow;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool kept = rtts_drop_hash(seed, pair0 + (uint32_t)(8 * (i >> 2) + (i & 3)) * (uint32_t)NK) >= dr.thresh;
                    kbits |= kept ? (1u << i) : 0u;
                    pacc[i] = kept ? __builtin_fmaf(dr.scale, pacc[i] - pinit[i], pinit[i]) : pinit[i];
                }
            }
            float pp[16], ds[16];
            const float ksc2 = ksc * 1.4426950408889634f;
            if (chk_self) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w};
                    const int4 p4 = *reinterpret_cast<const int4*>(kq + qt * 32 + 8 * g + 4 * hh);
                    const int pv[4] = {p4.x, p4.y, p4.z, p4.w}, ev[4] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int i = 4 * g + jj;
                        const bool self = pv[jj] == mypos;
                        const bool dead = kpk > ev[jj];
                        float x = sacc[i] * ksc2;
                        x = self ? (-5e4f * 1.4426950408889634f) : x;
                        float p = __builtin_amdgcn_exp2f(x - lv[jj]);
                        p = (dead && !self) ? 0.f : p;
                        pp[i] = p;
                        ds[i] = self ? 0.f : p * pacc[i] * ksc;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w};
                    const int ev[4] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int i = 4 * g + jj;
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], ksc2, -lv[jj]));
                        p = (kpk > ev[jj]) ? 0.f : p;
                        pp[i] = p;
                        ds[i] = p * pacc[i] * ksc;
                    }
                }
            }
            if constexpr (DROP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pp[i] = ((kbits >> i) & 1u) ? pp[i] * dr.scale : 0.f;
            }
            bf16x8 qtf[2][2], dotf[2][2];   // [s2][dh tile]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int blk = (qt * 32 + 16 * s2) * 128;
                    qtf[s2][dt] = tr_frag(Kown + blk + tro[dt], Kown + blk + 8 * 128 + tro[dt ^ 1]);
                    dotf[s2][dt] = tr_frag(Os + blk + tro[dt], Os + blk + 8 * 128 + tro[dt ^ 1]);
                }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float* pq = pp + 8 * s2;
                const float* dq_ = ds + 8 * s2;
                const bf16x8 pb = cvt_bf16x8(pq[0], pq[1], pq[2], pq[3], pq[4], pq[5], pq[6], pq[7]);
                const bf16x8 db = cvt_bf16x8(dq_[0], dq_[1], dq_[2], dq_[3], dq_[4], dq_[5], dq_[6], dq_[7]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[s2][dt], pb, dvacc[dt], 0, 0, 0);
                    gacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[s2][dt], db, gacc[dt], 0, 0, 0);
                }
            }
            const int dsq = qt << 6;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
                pk.y = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
                *reinterpret_cast<uint2*>(Ds + (dso[g] ^ dsq)) = pk;
            }
        }

        AB_JSTAMP(4);
        __syncthreads();   // every dS'^T tile is in Ds; nobody reads Os as dout any more
        AB_JSTAMP(5);
        if (more && !own_tile) {        // this group works the own keys of chunk j + 1 next: its V rows and validity
            const int nmypos = kn[wt * 32 + r];
            myvalid = MASKED ? (int)mask[(size_t)b * T + nmypos] : 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)nmypos * ld + ks * 16 + 8 * hh);
        }

        // ---- dQ^T[dh][q] = K^T dS'^T over the own keys, then the looked-back keys; parked (bf16) in the dout image's rows
        {
            const int qt = wave >> 1, dt = wave & 1;
            f32x16 dq = {0};
            const int rl = 8 * hh + trq;
            const int kpc = dt * 4 + 2 * trc + (trp >> 1);
            const int ko0 = ab_off(rl, kpc) + 8 * (trp & 1), ko1 = ab_off(rl + 4, kpc) + 8 * (trp & 1);
            const int gq = qt * 8 + 4 * trc + trp;
            const int do0 = ab_ds_off<BS>(rl, gq), do1 = ab_ds_off<BS>(rl + 4, gq);
            if (qt < NQT) {        // BS = 64: NW = 4 waves, 2 * NQT = 4 outputs -- every wave has one; BS = 128: 8 and 8
#pragma unroll AB_DQ_UNROLL
                for (int kb = 0; kb < BS; kb += 16) {
                    const bf16x8 bfrag = tr_frag(Ds + kb * DSROW + do0, Ds + kb * DSROW + do1);
                    const bf16x8 afrag = tr_frag(Kown + kb * 128 + ko0, Kown + kb * 128 + ko1);
                    dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dq, 0, 0, 0);
                }
#pragma unroll AB_DQ_UNROLL
                for (int kb = 0; kb < BS; kb += 16) {
                    const bf16x8 bfrag = tr_frag(Ds + (BS + kb) * DSROW + do0, Ds + (BS + kb) * DSROW + do1);
                    const bf16x8 afrag = tr_frag(Klb + kb * 128 + ko0, Klb + kb * 128 + ko1);
                    dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dq, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_bf16x2(dq[4 * g], dq[4 * g + 1]);
                    pk.y = pack_bf16x2(dq[4 * g + 2], dq[4 * g + 3]);
                    *reinterpret_cast<uint2*>(Os + ab_off(qt * 32 + r, dt * 4 + g) + 8 * hh) = pk;
                }
            }
        }
        AB_JSTAMP(6);
#if AB_SPLIT_EPI
        // What the NEXT step's main loop overwrites is read out in front of the two barriers below, so that a wave may enter that
        // main loop while others still store rows: the next chunk's words are published, the raw key rows (slot j % 3 = the
        // slot the next step's prefetch lands in) are taken into registers, and each wave waits for ITS share of this step's
        // prefetch (and the V rows it has just requested).
        if (more && tid < BS) {
            qlseS[(par ^ 1) * BS + tid] = nlse * 1.4426950408889634f;
            qdelS[(par ^ 1) * BS + tid] = -ndel;
            qpeS[(par ^ 1) * BS + tid] = nval ? (CAUSAL ? npos : 0) : -1;
            if (j + 2 < R) kposS[((j + 3) & 3) * BS + tid] = p2;
        }
        uint2 kk2[2][4];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) kk2[dt][g] = *reinterpret_cast<const uint2*>(Kt + ab_off(r, dt * 4 + g) + 8 * hh);
        // (the empty asm reads the V rows requested above: the compiler's wait-count pass then KNOWS they have arrived -- otherwise it
        //  waits for them in the next main loop with vmcnt(0), i.e. behind this step's row stores, which it cannot count)
        asm volatile("" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(myvalid));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // dQ parked; the dS'^T image is free; the next step's rows and words are on chip
        AB_JSTAMP(7);
        uint2 dqp[2][4];          // (every wave reads: a conditional fill of the array sends it through scratch)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) dqp[dt][g] = *reinterpret_cast<const uint2*>(Os + ab_off(wt * 32 + r, dt * 4 + g) + 8 * hh);
        __syncthreads();   // the parked dQ rows have been read: the next step's prefetch may land in this dout image
#else
        __syncthreads();   // dQ parked; the dS'^T image is free: it becomes the row staging
        AB_JSTAMP(7);

        // the prefetched words of the next chunk have long arrived: publish them (read after the barrier at the top of the next step)
        if (more && tid < BS) {
            qlseS[(par ^ 1) * BS + tid] = nlse * 1.4426950408889634f;
            qdelS[(par ^ 1) * BS + tid] = -ndel;
            qpeS[(par ^ 1) * BS + tid] = nval ? (CAUSAL ? npos : 0) : -1;
            if (j + 2 < R) kposS[((j + 3) & 3) * BS + tid] = p2;
        }
#endif

        // ---- end of the step.  Own keys: the parked query-role gradient of the row joins the key-role accumulator (they are the
        //      same rows).  Looked-back keys (and the own keys of a run's last step): the row is complete -- dK = G - k^ (k^ . G) on
        //      the key part only, + dQ -- and leaves through a [32][144 B] staging as full 128-byte rows, with dV.
        float kraw[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#if AB_SPLIT_EPI
                const uint2 kk = kk2[dt][g];
#else
                const uint2 kk = *reinterpret_cast<const uint2*>(Kt + ab_off(r, dt * 4 + g) + 8 * hh);
#endif
                kraw[dt][4 * g] = __uint_as_float(kk.x << 16);
                kraw[dt][4 * g + 1] = __uint_as_float(kk.x & 0xffff0000u);
                kraw[dt][4 * g + 2] = __uint_as_float(kk.y << 16);
                kraw[dt][4 * g + 3] = __uint_as_float(kk.y & 0xffff0000u);
            }
        if (own_tile) {
            float sd4[4] = {0.f, 0.f, 0.f, 0.f};        // four chains of 8, not one of 32 dependent fmas
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#if AB_SPLIT_EPI
                    const uint2 dqv = dqp[dt][g];
#else
                    const uint2 dqv = *reinterpret_cast<const uint2*>(Os + ab_off(myrow, dt * 4 + g) + 8 * hh);
#endif
                    const float dqf[4] = {__uint_as_float(dqv.x << 16), __uint_as_float(dqv.x & 0xffff0000u), __uint_as_float(dqv.y << 16),
                                          __uint_as_float(dqv.y & 0xffff0000u)};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        gacc[dt][4 * g + jj] += dqf[jj];
                        sd4[jj] = __builtin_fmaf(kraw[dt][4 * g + jj], dqf[jj], sd4[jj]);
                    }
                }
            sdq = rtts_xhalf_sum((sd4[0] + sd4[1]) + (sd4[2] + sd4[3]));
        }
        if (!own_tile || j == R - 1) {
            const int slot = (!own_tile && j == 0) ? 1 : 0;
            const size_t obase = ((size_t)bh * n_hashes + (own_tile ? c / nb : cprev / nb)) * T;
            const int srow = lane >> 3, spiece = lane & 7;
#if AB_SPLIT_EPI
            // the staging (4 KB) must not lie where another wave's next main loop writes dS'^T: it takes the rows of the dS'^T
            // image this wave itself writes NEXT (32 keys x DSROW >= 4 KB: own keys after a looked-back step; the own keys of a
            // run's last step take their looked-back twin's rows)
            unsigned char* stg = Ds + ((own_tile ? BS : 0) + wt * 32) * DSROW;
            static_assert(32 * DSROW >= 32 * AB_ROWB, "the staging must fit a wave's rows of the dS'^T image");
#else
            unsigned char* stg = Stg + wave * (32 * AB_ROWB);
#endif
            int rpos[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) rpos[i] = own_tile ? kq[wt * 32 + i * 8 + srow] : kl[wt * 32 + i * 8 + srow];
            {
                bf16_t* dvdst = dv_part + slot * slot_stride;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 pk;
                        pk.x = pack_bf16x2(dvacc[dt][4 * g], dvacc[dt][4 * g + 1]);
                        pk.y = pack_bf16x2(dvacc[dt][4 * g + 2], dvacc[dt][4 * g + 3]);
                        *reinterpret_cast<uint2*>(stg + ab_stg_w(r, dt * 4 + g, hh)) = pk;
                    }
                __builtin_amdgcn_wave_barrier();
                uint4 rowv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rowv[i] = ab_stg_fix(i, *reinterpret_cast<const uint4*>(stg + ab_stg_r(i, srow, spiece)));
#pragma unroll
                for (int i = 0; i < 4; ++i) rtts_store16_out(dvdst + (obase + rpos[i]) * AB_DH + spiece * 8, rowv[i]);
                __builtin_amdgcn_wave_barrier();
            }
            {
                bf16_t* dkdst = dqk_part + slot * slot_stride;
                float dot4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dot4[e & 3] = __builtin_fmaf(kraw[dt][e], gacc[dt][e], dot4[e & 3]);
                const float dot = (dot4[0] + dot4[1]) + (dot4[2] + dot4[3]);
                // k . (G + dQ) - k . dQ = k . G; k^ = k / |k| = k * (8 ksc)
                const float ncoef = -(rtts_xhalf_sum(dot) - sdq) * (ksc * 8.f) * (ksc * 8.f);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float dk[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) dk[jj] = __builtin_fmaf(kraw[dt][4 * g + jj], ncoef, gacc[dt][4 * g + jj]);
                        uint2 pk;
                        pk.x = pack_bf16x2(dk[0], dk[1]);
                        pk.y = pack_bf16x2(dk[2], dk[3]);
                        *reinterpret_cast<uint2*>(stg + ab_stg_w(r, dt * 4 + g, hh)) = pk;
                    }
                __builtin_amdgcn_wave_barrier();
                uint4 rowv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rowv[i] = ab_stg_fix(i, *reinterpret_cast<const uint4*>(stg + ab_stg_r(i, srow, spiece)));
#pragma unroll
                for (int i = 0; i < 4; ++i) rtts_store16_out(dkdst + (obase + rpos[i]) * AB_DH + spiece * 8, rowv[i]);
            }
            // slot-0 rows say whether a slot-1 partner exists: only the own keys of a run's last step have one
            if (slot == 0 && spiece == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) row_flags[obase + rpos[i]] = own_tile ? 1 : 0;
            }
        }
        AB_JSTAMP(8);
#ifdef AB_PHASE_TIMING
        __builtin_amdgcn_s_waitcnt(0x0F70);
        AB_JSTAMP(9);
#endif
    }
#undef AB_JSTAMP
}

extern "C" int rtts_lsh_bwd_qk_slots(void) { return RTTS_LSH_BWD_QK_SLOTS; }

// chunks a workgroup walks: the longest run (8, 4) that divides the ring and still leaves >= 3 workgroups per CU; 0 = the
// one-chunk kernel (small problems).  Measured (kbench, backward + reduce, us): decoder shape (128-row buckets) 272 + 83 ->
// 231 + 40 at runs of 8; encoder shape (64-row buckets, T = 256) 50.8 + 16.0 -> 46.9 + 13.9 at runs of 4 (51.2 + 12.3 at 8:
// too few workgroups); T = 4096 with 64-row buckets 246 + 109 -> 206 + 72 at runs of 8.
// rtts_debug_set_walk() forces a run length for tests and A/B runs (0: the one-chunk kernel).
extern "C" int rtts_lsh_attn_bwd_run_length(int B, int H, int T, int n_hashes, int bucket_size) {
    if (B <= 0 || H <= 0 || n_hashes <= 0 || bucket_size <= 0 || T <= 0 || T % bucket_size) return -1;
    const int C = n_hashes * (T / bucket_size);
    const long long chunks = (long long)B * H * C;
    int R = 0;
    if (AB_KT2 == 1)
        for (int cand = 8; cand >= 4; cand >>= 1)
            if (C % cand == 0 && chunks / cand >= 768) { R = cand; break; }
    const int w = rtts_walk_override(1);          // tests / A-B runs only (rtts_debug_set_walk); -1 in every product call
    if (w >= 0 && AB_KT2 == 1) R = (w >= 1 && C % w == 0) ? w : 0;
    return R;
}

template <int BS>
static int launch_attn_bwd(const bf16_t* qk, const bf16_t* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                           const bf16_t* dout, int64_t ld_do, const float* lse_tot, const float* delta, int B, int H, int T,
                           int n_hashes, int causal, bf16_t* dqk_part, bf16_t* dv_part, uint8_t* row_flags, float drop_p,
                           uint32_t drop_seed, const uint32_t* seed_dev, hipStream_t stream) {
    constexpr int NK = 2 * BS;
    const size_t slot_stride = (size_t)B * H * n_hashes * T * AB_DH;
    const bool drop = drop_p > 0.f;
    const int vi = (drop ? 4 : 0) + (causal ? 2 : 0) + (mask ? 1 : 0);
    const AbDrop dr{drop_seed, seed_dev, rtts_drop_thresh(drop_p), 1.f / (1.f - drop_p)};
    const int C = n_hashes * (T / BS);
    const long long chunks = (long long)B * H * C;
    const int R = rtts_lsh_attn_bwd_run_length(B, H, T, n_hashes, BS);
    if (R >= 1 && AB_KT2 == 1) {
        RTTS_REQUIRE(row_flags, "rtts_lsh_attn_bwd: this shape is worked by the walking kernel (rtts_lsh_attn_bwd_run_length() = %d): "
                                "row_flags (B*H*n_hashes*T bytes) is required", R);
        const size_t ds_bytes = (size_t)NK * (BS * 2), stg_bytes = (size_t)(NK / 32) * 32 * AB_ROWB;
        const size_t wlds = (size_t)BS * 40 + 5 * (size_t)BS * 128 + (ds_bytes > stg_bytes ? ds_bytes : stg_bytes);
        static RttsLdsState wattr[8];
        const dim3 wgrid((unsigned)(chunks / R)), wblock(BS * 4);
#define AB_WGO(C_, M_, D_)                                                                                                 \
    do {                                                                                                                   \
        auto kern = lsh_attn_bwd_walk_kernel<BS, C_, M_, D_>;                                                              \
        RTTS_ENSURE_LDS("rtts_lsh_attn_bwd", kern, wlds, wattr[vi]);                                                       \
        hipLaunchKernelGGL(kern, wgrid, wblock, wlds, stream, qk, v, ld, st, mask, dout, ld_do, lse_tot, delta, H, T, n_hashes, \
                           dqk_part, dv_part, slot_stride, R, row_flags, dr);                                              \
    } while (0)
#define AB_WGO2(C_, M_) do { if (drop) AB_WGO(C_, M_, true); else AB_WGO(C_, M_, false); } while (0)
        if (causal) {
            if (mask) AB_WGO2(true, true); else AB_WGO2(true, false);
        } else {
            if (mask) AB_WGO2(false, true); else AB_WGO2(false, false);
        }
#undef AB_WGO2
#undef AB_WGO
        RTTS_LAUNCH_CHECK("rtts_lsh_attn_bwd");
        return 0;
    }
    const size_t lds = NK * 128 + BS * 128 + NK * (BS * 2) + (NK / 32) * 32 * AB_ROWB + NK * 12 + BS * 12;
    const dim3 grid(B * H * n_hashes * (T / BS)), block(BS * 4 / AB_KT2);
    static RttsLdsState attr[8];                 // per device: the dynamic-LDS limit is an attribute of the loaded function
#define AB_GO(C_, M_, D_)                                                                                                  \
    do {                                                                                                                   \
        auto kern = lsh_attn_bwd_kernel<BS, C_, M_, D_>;                                                                   \
        RTTS_ENSURE_LDS("rtts_lsh_attn_bwd", kern, lds, attr[vi]);                                                         \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, qk, v, ld, st, mask, dout, ld_do, lse_tot, delta, H, T, n_hashes, \
                           dqk_part, dv_part, slot_stride, dr);                                                            \
    } while (0)
#define AB_GO2(C_, M_) do { if (drop) AB_GO(C_, M_, true); else AB_GO(C_, M_, false); } while (0)
    if (causal) {
        if (mask) AB_GO2(true, true); else AB_GO2(true, false);
    } else {
        if (mask) AB_GO2(false, true); else AB_GO2(false, false);
    }
#undef AB_GO2
#undef AB_GO
    RTTS_LAUNCH_CHECK("rtts_lsh_attn_bwd");
    return 0;
}

extern "C" int rtts_lsh_attn_bwd(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                                 const void* dout, int64_t ld_dout, const float* lse_tot, const float* delta, int B, int H,
                                 int T, int dh, int n_hashes, int bucket_size, int causal, void* dqk_part, void* dv_part,
                                 uint8_t* row_flags, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(qk && v && st && dout && lse_tot && delta && dqk_part && dv_part, "rtts_lsh_attn_bwd: null pointer");
    RTTS_REQUIRE(dh == AB_DH, "rtts_lsh_attn_bwd: dh=%d unsupported (this build: 64)", dh);
    RTTS_REQUIRE(bucket_size == 64 || bucket_size == 128, "rtts_lsh_attn_bwd: bucket_size=%d unsupported (64 or 128)", bucket_size);
    RTTS_REQUIRE(T > 0 && T % (2 * bucket_size) == 0,
                 "rtts_lsh_attn_bwd: Sequence length (%d) needs to be divisible by target bucket size x 2 - %d", T, 2 * bucket_size);
    RTTS_REQUIRE(B > 0 && H > 0 && n_hashes > 0, "rtts_lsh_attn_bwd: bad B/H/n_hashes");
    RTTS_REQUIRE(ld >= (int64_t)H * dh && ld % 8 == 0 && ld_dout >= (int64_t)H * dh && ld_dout % 8 == 0,
                 "rtts_lsh_attn_bwd: row strides must be >= H*dh and multiples of 8");
    RTTS_REQUIRE((((uintptr_t)qk | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dqk_part | (uintptr_t)dv_part) & 15) == 0,
                 "rtts_lsh_attn_bwd: buffers must be 16-byte aligned");
    RTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "rtts_lsh_attn_bwd: drop_p must be in [0, 1)");
    RTTS_REQUIRE(drop_p == 0.f || (uint64_t)B * H * n_hashes * T * 2 * bucket_size < (1ull << 32),
                 "rtts_lsh_attn_bwd: dropout on more than 2^32 query-key pairs (the forward refuses the same shape)");
    hipStream_t s = (hipStream_t)stream;
    if (bucket_size == 64)
        return launch_attn_bwd<64>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta,
                                   B, H, T, n_hashes, causal, (bf16_t*)dqk_part, (bf16_t*)dv_part, row_flags, drop_p, drop_seed, seed_dev, s);
    return launch_attn_bwd<128>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta, B,
                                H, T, n_hashes, causal, (bf16_t*)dqk_part, (bf16_t*)dv_part, row_flags, drop_p, drop_seed, seed_dev, s);
}
