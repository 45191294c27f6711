// LSH bucket assignment + stable counting sort, one workgroup per (batch*head, hash round).
//
// Replaces hash_vectors / sort_key_val of the reference's LSH layer (reformer_pytorch 0.19.1,
// reached from reformer_tts/model/reformer.py:217).
//
// HBM-bound integer path: reads each qk row (128 B) once per round (rounds of one head run
// on the same XCD, so 7 of the 8 reads are L2 hits), writes 4 B (+8 B optional) per token.
// The projection is a k-ordered fp32 fmaf chain, so bucket ids are bit-identical to
// oracle/lsh_int.c; the sort is a stable counting sort (keys are unique => permutation unique).
//
// Hash: a lane owns a row (64 values in registers) and runs the fmaf chains of two buckets per instruction
// (v_pk_fma_f32: the two halves are independent IEEE fmas, so packing does not change a bit).  The rotation matrix is
// broadcast from LDS -- every lane of the 12 waves of a CU fetches the same 4*HALF bytes per row element, which makes
// LDS bandwidth the floor of this phase (measured by ablation on MI355X, decoder shape: empty kernel 2.0 us, hash phase
// 13.3 us = 3.3 us per 64-row tile, sort + stores 3.0 us).  Scalar loads of the matrix (constant address space,
// s_load + SGPR operands) were tried and are slower (20.3 us: the loads serialise on their latency).
// Sort: per 64-token group the lanes that share a bucket are found with log2(n_buckets) ballots (AND of
// ballot / ~ballot per bucket-id bit), a lane's stable rank is the popcount of that mask below it, and the
// per-(wave, bucket) running offsets live in LDS -- O(log n_buckets) wave ops per group instead of O(n_buckets).
#include "rtts_common.h"
#include <stdlib.h>

#define HS_DH 64
#ifndef HS_PF32
#define HS_PF32 1
#endif
#ifndef HS_W8_MIN_T
#define HS_W8_MIN_T 2048   // sequences from this length on get 8 waves per (head, round)
#endif
#define HS_ROWB 144   // LDS row stride in bytes for a staged 64 x 64 bf16 tile (conflict-free b128 reads)

template <int HALF>
__device__ __forceinline__ int hash_row(const float* q, const float* rot_lds, int half) {
    // rot_lds[f * HALF + i], columns i >= half are zero padding (n_buckets / 2 = half need not be a power of two: a
    // 768-frame mel at bucket size 128 has 6 buckets); returns argmax over [xR, -xR] restricted to the real columns,
    // the first maximum winning
    int idx = 0;
    float best;
    if constexpr (HALF >= 2) {
        f32x2 acc[HALF / 2];
#pragma unroll
        for (int i = 0; i < HALF / 2; ++i) acc[i] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int f = 0; f < HS_DH; ++f) {
            const f32x2 qq = {q[f], q[f]};
#pragma unroll
            for (int i = 0; i < HALF / 2; ++i)
                acc[i] = __builtin_elementwise_fma(qq, *reinterpret_cast<const f32x2*>(rot_lds + f * HALF + 2 * i), acc[i]);
        }
        best = acc[0][0];
#pragma unroll
        for (int i = 1; i < HALF; ++i)
            if (i < half && acc[i >> 1][i & 1] > best) { best = acc[i >> 1][i & 1]; idx = i; }
#pragma unroll
        for (int i = 0; i < HALF; ++i)
            if (i < half && -acc[i >> 1][i & 1] > best) { best = -acc[i >> 1][i & 1]; idx = half + i; }
    } else {
        float a = 0.f;
#pragma unroll
        for (int f = 0; f < HS_DH; ++f) a = __builtin_fmaf(q[f], rot_lds[f], a);
        best = a;
        if (-a > best) idx = 1;
    }
    return idx;
}

// lanes of the wave whose value v (0 <= v < 2^BITS, or -1 = inactive) equals this lane's
template <int BITS>
__device__ __forceinline__ unsigned long long match_lanes(int v) {
    unsigned long long m = __ballot(v >= 0);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
        const unsigned long long bm = __ballot((v >> bit) & 1);
        m &= ((v >> bit) & 1) ? bm : ~bm;
    }
    return m;
}

// histogram step of one wave: lane adds 1 to cnt[v] for every active lane (v >= 0).  Sixty-four lanes on a handful of buckets
// made the plain per-lane LDS add an N-way same-address serialisation (43 % of the sort kernel's LDS cycles were bank
// conflicts: profiles/r03g_pmc_*); the first lane of each group of equal values adds the group's size instead -- distinct
// addresses, one add each.
// With many buckets (32, 64: the T = 4096 configuration) equal values in one wave are rare, the per-lane add has nothing to
// serialise, and the BITS + 1 ballots would cost more than they save (measured: 71 against 62-67 us at 64 buckets): plain adds there.
template <int BITS>
__device__ __forceinline__ void count_lanes(int* cnt, int v, int lane) {
    if constexpr (BITS <= 4) {
        const unsigned long long m = match_lanes<BITS>(v);
        if (v >= 0 && (m & ((1ull << lane) - 1ull)) == 0) cnt[v] += __popcll(m);
    } else {
        if (v >= 0) atomicAdd(&cnt[v], 1);                     // integer LDS add: order-free, deterministic
    }
}

template <int HALF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lsh_hash_sort_kernel(
    const bf16_t* __restrict__ qk, int64_t ld, const float* __restrict__ rotations, int rot_rows,
    int H, int T, int n_hashes, int32_t* __restrict__ buckets, int32_t* __restrict__ st, int32_t* __restrict__ undo, int half) {
    // HALF = the compile-time column capacity (a power of two >= half), half = n_buckets / 2 of this call
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 2 * half;
    constexpr int NBC = 2 * HALF;
    constexpr int BITS = (NBC <= 2) ? 1 : (NBC <= 4) ? 2 : (NBC <= 8) ? 3 : (NBC <= 16) ? 4 : (NBC <= 32) ? 5 : 6;
    constexpr int NTHR = 64 * WAVES;
    // carve: rot [64*HALF] f32 | tile [WAVES][64 rows][144 B] | bkt [T] u16 | cntw [WAVES][64] i32 | tot [64] i32
    float* rot_lds = reinterpret_cast<float*>(smem);
    unsigned char* tile = smem + ((HS_DH * HALF * 4 + 15) & ~15);
    uint16_t* bkt = reinterpret_cast<uint16_t*>(tile + WAVES * 64 * HS_ROWB);
    int* cntw = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(bkt) + ((T * 2 + 15) & ~15));
    int* tot = cntw + WAVES * 64;

    // work item: round r of head bh; rounds of one bh are 8 ids apart => same XCD (L2 reuse of qk)
    const uint32_t nblk = gridDim.x;
    const uint32_t w = xcd_remap(blockIdx.x, nblk);
    const int bh = w / n_hashes, r = w % n_hashes;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const float* rot_src = rotations + (size_t)(rot_rows == 1 ? 0 : bh) * HS_DH * n_hashes * half;
    for (int i = tid; i < HS_DH * HALF; i += NTHR) {
        const int f = i / HALF, k = i % HALF;
        rot_lds[i] = k < half ? rot_src[((size_t)f * n_hashes + r) * half + k] : 0.f;
    }
    for (int i = tid; i < WAVES * 64; i += NTHR) cntw[i] = 0;
    __syncthreads();

    float rot_reg[HALF == 32 ? 32 : 1];      // MFMA path: R^T fragments, rot_reg[j] = R[2j + (lane >> 5)][lane & 31]
    if constexpr (HALF == 32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) rot_reg[j] = rot_lds[(2 * j + (lane >> 5)) * HALF + (lane & 31)];
    }
    // ---- hash: each wave stages 64 rows (coalesced 16-B pieces), then one lane hashes one row.
    //      Wave w hashes the token segment it will sort, so its bucket counts need no other wave's rows.
    const int seg = T / WAVES;              // multiple of 32 because T % 128 == 0 and WAVES in {4, 8}
    const int s0 = wave * seg;
    const bf16_t* base = qk + (size_t)b * T * ld + (size_t)h * HS_DH;
    unsigned char* wt = tile + wave * 64 * HS_ROWB;
    // Row tiles are fetched PF at a time (all their loads in flight together): with few buckets the fmaf chains are
    // short and a wave would otherwise sit through one full memory round trip per 64 rows.  PF = 1 where the hash is
    // long enough to hide the next tile's latency behind it and the registers are needed for the accumulators.
    constexpr int PF = (HALF <= 8) ? 4 : (HALF == 32) ? HS_PF32 : 1;   // MFMA path: deeper prefetch measured neutral (A/B, 1/2/4)
    for (int tb = 0; tb < seg; tb += 64 * PF) {
        uint4 pre[PF][8];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t0 = tb + 64 * u;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int row = p * 8 + (lane >> 3), piece = lane & 7;
                const int tr = min(t0 + row, seg - 1);      // past the end: re-read the last row (never stored)
                pre[u][p] = *reinterpret_cast<const uint4*>(base + (size_t)(s0 + tr) * ld + piece * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t0 = tb + 64 * u;
            const int rows = min(64, seg - t0);         // 64, 32, or <= 0 past the end of the segment
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int row = p * 8 + (lane >> 3), piece = lane & 7;
                if (row < rows) *reinterpret_cast<uint4*>(wt + row * HS_ROWB + piece * 16) = pre[u][p];
            }
            if (rows <= 0) continue;           // wave-uniform; (no break: the unrolled body keeps `pre` in registers)
            __builtin_amdgcn_wave_barrier();   // same wave wrote and reads: LDS ops of one wave execute in order
            if constexpr (HALF == 32) {
                // 33..64 buckets: the projections go to the matrix pipe.  v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered
                // f32 fmaf chain (one rounding per product, no wider accumulation), i.e. exactly the chain of
                // oracle/lsh_int.c, at the f32 vector rate but with the rotation matrix resident in 32 registers instead
                // of being broadcast from LDS for every row element.  D[bucket][token] = sum_k R[k][bucket] q[token][k]:
                // A = R^T (lane: bucket l & 31, k = 2j + (l >> 5)), B = q^T (lane: token l & 31, same k), so a lane ends
                // up with 16 of its token's 32 projections and the argmax stays in registers but for one half swap.
                const int hh = lane >> 5;
                int idx2[2];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    f32x16 acc = {0};
                    const unsigned char* rowp = wt + (32 * sub + (lane & 31)) * HS_ROWB;
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const uint4 val = *reinterpret_cast<const uint4*>(rowp + p * 16);
                        const uint32_t uu[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float bq = __uint_as_float(hh ? (uu[k] & 0xffff0000u) : (uu[k] << 16));   // q[token][2j + hh]
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rot_reg[p * 4 + k], bq, acc, 0, 0, 0);
                        }
                    }
                    // this lane's 16 buckets: 8 * (i >> 2) + 4 * hh + (i & 3), ascending in i; [xR, -xR], first maximum wins
                    float best = -__builtin_inff();
                    int bi = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int m = 8 * (i >> 2) + 4 * hh + (i & 3);
                        if (m < half && acc[i] > best) { best = acc[i]; bi = m; }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int m = 8 * (i >> 2) + 4 * hh + (i & 3);
                        if (m < half && -acc[i] > best) { best = -acc[i]; bi = half + m; }
                    }
                    // the partner half holds the other 16 buckets of the same token: larger value wins, the smaller index on a tie
                    const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                    const auto si = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
                    const float ob = __uint_as_float(hh ? sv[0] : sv[1]);
                    const int oi = (int)(hh ? si[0] : si[1]);
                    idx2[sub] = (ob > best || (ob == best && oi < bi)) ? oi : bi;
                }
                const int idx = hh ? idx2[1] : idx2[0];     // lane <-> row of the 64-row tile again
                if (lane < rows) {
                    bkt[s0 + t0 + lane] = (uint16_t)idx;
                    if (buckets) buckets[((size_t)bh * n_hashes + r) * T + s0 + t0 + lane] = idx + r * NB;
                }
                count_lanes<BITS>(cntw + wave * 64, lane < rows ? idx : -1, lane);      // this wave's own row: no other wave adds to it
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            float q[HS_DH];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint4 val = *reinterpret_cast<const uint4*>(wt + lane * HS_ROWB + p * 16);
                const uint32_t uu[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    q[p * 8 + 2 * k] = __uint_as_float(uu[k] << 16);
                    q[p * 8 + 2 * k + 1] = __uint_as_float(uu[k] & 0xffff0000u);
                }
            }
            const int idx = hash_row<HALF>(q, rot_lds, half);   // lanes >= rows hash stale LDS rows: results dropped below
            if (lane < rows) {
                bkt[s0 + t0 + lane] = (uint16_t)idx;
                if (buckets) buckets[((size_t)bh * n_hashes + r) * T + s0 + t0 + lane] = idx + r * NB;
            }
            count_lanes<BITS>(cntw + wave * 64, lane < rows ? idx : -1, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- stable counting sort: first sorted slot of (bucket k, wave w) = tokens in smaller buckets + tokens of
    //      bucket k in earlier waves' segments
    if (tid < NB) {
        int t = 0;
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) t += cntw[w2 * 64 + tid];
        tot[tid] = t;
    }
    __syncthreads();
    int basek = 0;
    if (lane < NB) {
        for (int k = 0; k < lane; ++k) basek += tot[k];
        for (int w2 = 0; w2 < wave; ++w2) basek += cntw[w2 * 64 + lane];
    }
    __syncthreads();                         // every wave has read the counts before they become running offsets
    int* run = cntw + wave * 64;             // this wave's running offset per bucket
    if (lane < NB) run[lane] = basek;
    __builtin_amdgcn_wave_barrier();
    int32_t* st_out = st + ((size_t)bh * n_hashes + r) * T;
    int32_t* undo_out = undo ? undo + ((size_t)bh * n_hashes + r) * T : nullptr;
    for (int t0 = 0; t0 < seg; t0 += 64) {
        const bool act = t0 + lane < seg;
        const int mb = act ? (int)bkt[s0 + t0 + lane] : -1;
        const unsigned long long m = match_lanes<BITS>(mb);              // lanes of this group in my bucket
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        int pos = 0;
        if (act) pos = run[mb] + __popcll(below);
        __builtin_amdgcn_wave_barrier();
        if (act && below == 0) run[mb] += __popcll(m);                   // the first lane of each bucket advances it
        __builtin_amdgcn_wave_barrier();
        if (act) {
            const int t = s0 + t0 + lane;
            st_out[pos] = t;
            if (undo_out) undo_out[t] = pos;
        }
    }
}

// ======================================================================================================================
// Two-launch form (default): HASH for all rounds at once, then SORT per (head, round).
//
// The one-launch kernel above reads and stages every qk row once per hash round (8 times: 768 workgroups, four dependent
// load -> hash trips each) and broadcasts the rotation matrix from LDS for every row element: 23-28 us at the decoder shape,
// 0.10 of the HBM roofline, 72 % of its wave cycles waiting.  Here
//   * hash kernel: a workgroup = 128 tokens of one (batch, head), 32 per wave; its rows are fetched and staged ONCE and projected
//     on the columns of ALL rounds by v_mfma_f32_32x32x2_f32 -- bit for bit the k-ordered fp32 fmaf chain of
//     oracle/lsh_int.c (one rounding per product, no wider accumulation; the 33..64-bucket path above already relies on it).
//     The A operand is R^T with the rounds side by side: pass p holds the columns 32p .. 32p+31 of the (round, column)
//     grid at capacity HALF per round (decoder: 8 rounds x 4 columns = ONE pass), 32 registers per lane loaded straight
//     from the (tiny, L2-resident) rotation tensor -- no LDS broadcast at all.  The q values of a lane's two tokens stay in
//     registers across passes.  Bucket ids go to the `st` array itself (as scratch) and to `buckets` when asked for.
//   * sort kernel: one workgroup per (head, round) reads its T bucket ids (4 KB) and runs the same stable counting sort.
// Traffic: 128 B per token and head read once + 2 x 4 B x rounds written + 4 B x rounds read.
template <int HALF>
__global__ __launch_bounds__(256) void lsh_hash_rounds_kernel(const bf16_t* __restrict__ qk, int64_t ld, const float* __restrict__ rotations,
                                                              int rot_rows, int H, int T, int n_hashes, int half,
                                                              int32_t* __restrict__ buckets, int32_t* __restrict__ ids) {
    // 4 waves x 32 tokens: a wave's chain (rows -> LDS -> 32 registers, rotation columns -> 32 registers, 32 dependent
    // f32 MFMAs per pass, argmax, stores) is short and ~100 registers, so four waves per SIMD overlap each other's latencies
    // (two 64-token waves per workgroup measured 11.7 us at the decoder shape)
    __shared__ __attribute__((aligned(16))) unsigned char tile[4 * 32 * HS_ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blocks_per_head = T / 128;
    const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = w / blocks_per_head, t0 = (w % blocks_per_head) * 128 + wave * 32;
    const int b = bh / H, h = bh % H;
    const int NB = 2 * half;
    const bf16_t* base = qk + ((size_t)b * T + t0) * ld + (size_t)h * HS_DH;
    unsigned char* wt = tile + wave * 32 * HS_ROWB;
    {   // stage this wave's 32 rows: coalesced 16-byte pieces, all four loads in flight
        uint4 pre[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) pre[p] = *reinterpret_cast<const uint4*>(base + (size_t)(p * 8 + (lane >> 3)) * ld + (lane & 7) * 8);
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<uint4*>(wt + (p * 8 + (lane >> 3)) * HS_ROWB + (lane & 7) * 16) = pre[p];
    }
    __builtin_amdgcn_wave_barrier();
    const int hh = lane >> 5, l31 = lane & 31;
    // B operand: q[token][2j + hh], token = lane & 31: 32 registers, kept for every pass
    float bq[1][32];
    {
        const unsigned char* rowp = wt + l31 * HS_ROWB;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint4 val = *reinterpret_cast<const uint4*>(rowp + p * 16);
            const uint32_t uu[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) bq[0][p * 4 + k] = __uint_as_float(hh ? (uu[k] & 0xffff0000u) : (uu[k] << 16));
        }
    }
    const float* rot_src = rotations + (size_t)(rot_rows == 1 ? 0 : bh) * HS_DH * n_hashes * half;
    const int passes = (n_hashes * HALF + 31) / 32;
    constexpr int RPP = HALF >= 32 ? 1 : 32 / HALF;        // rounds per pass
    for (int p = 0; p < passes; ++p) {
        // A operand: column m = lane & 31 of this pass = (round, column in round) at capacity HALF; zero outside the real grid
        const int cg = 32 * p + l31;
        const int ar = cg / HALF, aci = cg % HALF;
        const bool areal = ar < n_hashes && aci < half;
        float rot_reg[32];
#pragma unroll
        for (int j = 0; j < 32; ++j)
            rot_reg[j] = areal ? rot_src[((size_t)(2 * j + hh) * n_hashes + ar) * half + aci] : 0.f;
        {
            constexpr int sub = 0;
            f32x16 acc = {0};
#pragma unroll
            for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rot_reg[j], bq[sub][j], acc, 0, 0, 0);
            const int tok = t0 + l31;
            // acc[i] = projection on column m = 8 * (i >> 2) + 4 * hh + (i & 3) of this pass; argmax over [xR, -xR] of each round,
            // the first maximum winning (ascending column, +x before -x)
            if constexpr (HALF <= 4) {
                // a lane's group g (four consecutive columns) holds 4 / HALF complete rounds
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int jr = 0; jr < 4 / HALF; ++jr) {
                        const int rr = p * RPP + (8 * g + 4 * hh) / HALF + jr;
                        float best = -__builtin_inff();
                        int bi = 0;
#pragma unroll
                        for (int ci = 0; ci < HALF; ++ci) {
                            const float x = acc[4 * g + jr * HALF + ci];
                            if (ci < half && x > best) { best = x; bi = ci; }
                        }
#pragma unroll
                        for (int ci = 0; ci < HALF; ++ci) {
                            const float x = -acc[4 * g + jr * HALF + ci];
                            if (ci < half && x > best) { best = x; bi = half + ci; }
                        }
                        if (rr < n_hashes) {
                            ids[((size_t)bh * n_hashes + rr) * T + tok] = bi;
                            if (buckets) buckets[((size_t)bh * n_hashes + rr) * T + tok] = bi + rr * NB;
                        }
                    }
            } else {
                // HALF = 8, 16, 32: a round spans both lane halves (and HALF / 8 groups): lane-local first maximum, then the
                // partner half's -- larger value wins, the smaller index on a tie
                constexpr int GPR = HALF / 8;                  // groups per round
#pragma unroll
                for (int q = 0; q < 4 / GPR; ++q) {
                    const int rr = p * RPP + q;
                    float best = -__builtin_inff();
                    int bi = 0;
#pragma unroll
                    for (int gg = 0; gg < GPR; ++gg)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ci = 8 * gg + 4 * hh + j;
                            const float x = acc[4 * (q * GPR + gg) + j];
                            if (ci < half && x > best) { best = x; bi = ci; }
                        }
#pragma unroll
                    for (int gg = 0; gg < GPR; ++gg)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ci = 8 * gg + 4 * hh + j;
                            const float x = -acc[4 * (q * GPR + gg) + j];
                            if (ci < half && x > best) { best = x; bi = half + ci; }
                        }
                    const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
                    const auto si = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
                    const float ob = __uint_as_float(hh ? sv[0] : sv[1]);
                    const int oi = (int)(hh ? si[0] : si[1]);
                    const int idx = (ob > best || (ob == best && oi < bi)) ? oi : bi;
                    if (rr < n_hashes && hh == (q & 1)) {      // both halves know the result: they take turns storing
                        ids[((size_t)bh * n_hashes + rr) * T + tok] = idx;
                        if (buckets) buckets[((size_t)bh * n_hashes + rr) * T + tok] = idx + rr * NB;
                    }
                }
            }
        }
    }
}

// stable counting sort of one (head, round): ids (in `st`, written by the hash kernel) -> st (sorted slot -> token), undo
template <int BITS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lsh_sort_ids_kernel(int T, int NB, int32_t* __restrict__ st, int32_t* __restrict__ undo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* bkt = reinterpret_cast<uint16_t*>(smem);
    int* cntw = reinterpret_cast<int*>(smem + ((T * 2 + 15) & ~15));
    int* tot = cntw + WAVES * 64;
    const uint32_t w = xcd_remap(blockIdx.x, gridDim.x);       // (head, round) in the order of the arrays
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* st_out = st + (size_t)w * T;
    int32_t* undo_out = undo ? undo + (size_t)w * T : nullptr;
    const int seg = T / WAVES, s0 = wave * seg;
    for (int i = tid; i < WAVES * 64; i += 64 * WAVES) cntw[i] = 0;
    __syncthreads();
    for (int t0 = 0; t0 < seg; t0 += 64) {                     // wave w counts the segment it will place
        const bool act = t0 + lane < seg;
        const int id = act ? st_out[s0 + t0 + lane] : -1;
        if (act) bkt[s0 + t0 + lane] = (uint16_t)id;
        count_lanes<BITS>(cntw + wave * 64, id, lane);         // this wave's own row of counters
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                           // all ids are in LDS: st may be overwritten from here on
    if (tid < NB) {
        int t = 0;
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) t += cntw[w2 * 64 + tid];
        tot[tid] = t;
    }
    __syncthreads();
    int basek = 0;
    if (lane < NB) {
        for (int k = 0; k < lane; ++k) basek += tot[k];
        for (int w2 = 0; w2 < wave; ++w2) basek += cntw[w2 * 64 + lane];
    }
    __syncthreads();
    int* run = cntw + wave * 64;
    if (lane < NB) run[lane] = basek;
    __builtin_amdgcn_wave_barrier();
    for (int t0 = 0; t0 < seg; t0 += 64) {
        const bool act = t0 + lane < seg;
        const int mb = act ? (int)bkt[s0 + t0 + lane] : -1;
        const unsigned long long m = match_lanes<BITS>(mb);
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        int pos = 0;
        if (act) pos = run[mb] + __popcll(below);
        __builtin_amdgcn_wave_barrier();
        if (act && below == 0) run[mb] += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (act) {
            const int t = s0 + t0 + lane;
            st_out[pos] = t;
            if (undo_out) undo_out[t] = pos;
        }
    }
}

template <int HALF>
static int launch_hash_then_sort(const bf16_t* qk, int64_t ld, const float* rot, int rot_rows, int B, int H, int T, int n_hashes,
                                 int32_t* buckets, int32_t* st, int32_t* undo, int half, hipStream_t stream) {
    constexpr int NBC = 2 * HALF;
    constexpr int BITS = (NBC <= 2) ? 1 : (NBC <= 4) ? 2 : (NBC <= 8) ? 3 : (NBC <= 16) ? 4 : (NBC <= 32) ? 5 : 6;
    hipLaunchKernelGGL((lsh_hash_rounds_kernel<HALF>), dim3(B * H * (T / 128)), dim3(256), 0, stream, qk, ld, rot, rot_rows, H, T, n_hashes,
                       half, buckets, st);
    RTTS_LAUNCH_CHECK("rtts_lsh_hash_sort (hash)");
    const size_t lds = ((T * 2 + 15) & ~15) + (8 * 64 + 64) * 4;
    if (T >= HS_W8_MIN_T && T % 512 == 0)
        hipLaunchKernelGGL((lsh_sort_ids_kernel<BITS, 8>), dim3(B * H * n_hashes), dim3(512), lds, stream, T, 2 * half, st, undo);
    else
        hipLaunchKernelGGL((lsh_sort_ids_kernel<BITS, 4>), dim3(B * H * n_hashes), dim3(256), lds, stream, T, 2 * half, st, undo);
    RTTS_LAUNCH_CHECK("rtts_lsh_hash_sort (sort)");
    return 0;
}

template <int HALF, int WAVES>
static int launch_hash_sort_w(const bf16_t* qk, int64_t ld, const float* rot, int rot_rows, int B, int H, int T,
                              int n_hashes, int32_t* buckets, int32_t* st, int32_t* undo, int half, hipStream_t stream) {
    const size_t lds = ((HS_DH * HALF * 4 + 15) & ~15) + WAVES * 64 * HS_ROWB + ((T * 2 + 15) & ~15) + (WAVES * 64 + 64) * 4;
    static RttsLdsState lds_state = {};
    RTTS_ENSURE_LDS("rtts_lsh_hash_sort", (lsh_hash_sort_kernel<HALF, WAVES>), lds, lds_state);
    const dim3 grid(B * H * n_hashes);
    hipLaunchKernelGGL((lsh_hash_sort_kernel<HALF, WAVES>), grid, dim3(64 * WAVES), lds, stream, qk, ld, rot, rot_rows, H, T,
                       n_hashes, buckets, st, undo, half);
    RTTS_LAUNCH_CHECK("rtts_lsh_hash_sort");
    return 0;
}

// 4 waves per (head, round) for short sequences; 8 from T = 2048 on (one workgroup still owns a whole sort, and a
// long row has enough tokens to keep 8 waves = 2 per SIMD busy through the fmaf chains)
template <int HALF>
static int launch_hash_sort(const bf16_t* qk, int64_t ld, const float* rot, int rot_rows, int B, int H, int T,
                            int n_hashes, int32_t* buckets, int32_t* st, int32_t* undo, int half, hipStream_t stream) {
    if (T >= HS_W8_MIN_T && T % 256 == 0)
        return launch_hash_sort_w<HALF, 8>(qk, ld, rot, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, stream);
    return launch_hash_sort_w<HALF, 4>(qk, ld, rot, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, stream);
}

static int hs_use_v1() {
    static const int v1 = [] { const char* e = getenv("RTTS_HASH_SORT_V1"); return e ? atoi(e) : 0; }();      // A/B runs: the one-launch kernel
    return v1;
}

// how rtts_lsh_hash_sort works a sequence length: 2 = lsh_hash_rounds_kernel + lsh_sort_ids_kernel, 1 = lsh_hash_sort_kernel
extern "C" int rtts_lsh_hash_sort_launches(int T) { return (!hs_use_v1() && T >= 1024) ? 2 : 1; }

extern "C" int rtts_lsh_hash_sort(const void* qk, int64_t ld_qk, const float* rotations, int rot_rows, int B, int H,
                                  int T, int dh, int n_hashes, int bucket_size, int32_t* buckets, int32_t* st,
                                  int32_t* undo, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(qk && rotations && st, "rtts_lsh_hash_sort: null pointer");
    RTTS_REQUIRE(dh == HS_DH, "rtts_lsh_hash_sort: dh=%d unsupported (this build: 64)", dh);
    RTTS_REQUIRE(bucket_size > 0 && T > 0 && T % (2 * bucket_size) == 0,
                 "rtts_lsh_hash_sort: Sequence length (%d) needs to be divisible by target bucket size x 2 - %d", T,
                 2 * bucket_size);
    RTTS_REQUIRE(T % 128 == 0 && T <= 8192, "rtts_lsh_hash_sort: T=%d must be a multiple of 128 and <= 8192", T);
    RTTS_REQUIRE(B > 0 && H > 0 && n_hashes > 0 && n_hashes <= 16, "rtts_lsh_hash_sort: bad B/H/n_hashes");
    RTTS_REQUIRE(rot_rows == 1 || rot_rows == B * H, "rtts_lsh_hash_sort: rot_rows must be 1 or B*H");
    RTTS_REQUIRE(ld_qk >= (int64_t)H * dh && ld_qk % 8 == 0, "rtts_lsh_hash_sort: ld_qk must be >= H*dh and a multiple of 8");
    RTTS_REQUIRE(((uintptr_t)qk & 15) == 0, "rtts_lsh_hash_sort: qk must be 16-byte aligned");
    const int half = T / bucket_size / 2;
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* q = (const bf16_t*)qk;
    RTTS_REQUIRE(half >= 1 && half <= 32, "rtts_lsh_hash_sort: n_buckets=%d unsupported (2 .. 64)", 2 * half);
    // column capacity = next power of two; the extra columns are zero padding that the argmax ignores
    // short sequences keep the one-launch kernel: at T = 256 (encoder shape) the two launches measure 9.2 us against 6.7 us --
    // the second kernel boundary costs more than the staging it saves; from T = 1024 on: 14.8 against 18.4 us (decoder shape, the
    // same box: gpurun_out/r03_kbench_hash_v{1,2}.log), 61.7 against 66.1 us at T = 4096 with 64 buckets (f32-MFMA-bound there:
    // 4.3 GFLOP of exact fp32 projections at the 155 TF f32 matrix rate = 28 us)
    if (rtts_lsh_hash_sort_launches(T) == 2) {
        if (half <= 1) return launch_hash_then_sort<1>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
        if (half <= 2) return launch_hash_then_sort<2>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
        if (half <= 4) return launch_hash_then_sort<4>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
        if (half <= 8) return launch_hash_then_sort<8>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
        if (half <= 16) return launch_hash_then_sort<16>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
        return launch_hash_then_sort<32>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    }
    if (half <= 1) return launch_hash_sort<1>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    if (half <= 2) return launch_hash_sort<2>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    if (half <= 4) return launch_hash_sort<4>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    if (half <= 8) return launch_hash_sort<8>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    if (half <= 16) return launch_hash_sort<16>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
    return launch_hash_sort<32>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, half, s);
}
