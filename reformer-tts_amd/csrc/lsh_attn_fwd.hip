// LSH chunked attention forward: one workgroup per (batch*head, sorted chunk).
//
// Replaces the gather / look-one-back / dots / mask / softmax / PV / unsort block of the
// reference's LSH layer (reformer_pytorch 0.19.1 via reformer_tts/model/reformer.py:217;
// SURVEY.md Appendix B steps 4-10) without materialising dots, masks or sorted copies.
//
// Per workgroup (BS = bucket size, NK = 2*BS keys = own chunk + previous chunk):
//   1. gather the NK qk rows and NK v rows (128 B each, coalesced 16-B pieces) by `st`
//      straight into LDS; while a row passes through registers its L2 norm is reduced, so the
//      key normalisation becomes a per-key scale applied to the logits (never rounded to bf16);
//   2. two waves share queries [32t, 32t+32), one per half of the keys (own chunk / previous chunk), and
//      merge (m, l, O) through LDS at the end; each walks its keys in tiles of 32 with an online softmax
//      (base-2 domain): S^T tile = K Q^T with v_mfma_f32_32x32x16_bf16 -- a lane holds one query
//      column, so the row reduction stays in registers (+1 cross-half shuffle per tile);
//   3. masks in the reference's order (padding, causal, self = -5e4) folded into two compares:
//      `dead` = key's effective position > query's effective position, `self` = positions equal;
//   4. O^T += V^T P^T: the P^T accumulators are the B operand as they stand (rows of X are the
//      contraction index), V^T fragments come from the row-major V image by ds_read_b64_tr_b16;
//   5. rows of o and lse are written directly at their UNSORTED position (round, t).
// The kernel is built for occupancy: ~120 VGPRs, tile loop not unrolled, 2 workgroups (16 waves) per CU.  What bounds it is the
// VECTOR ALU and the transcendental unit, not latency (an earlier reading of the counters said "latency"
// and was wrong: SQ_ACTIVE_INST_VALU is in quad-cycles per wave, four waves share a SIMD, the vector ALU is ~90 % busy): per
// wave and chunk ~1030 vector instructions against 32 MFMAs, of which the tile loop is 4 x (111 vector + 16 v_exp_f32).  Hence
// the lazy running maximum (the rescale of O and l almost never runs), v_dot2c row norms, tree-shaped sums, and the walking form
// below (every K / V row gathered once).  Ablations: profiles/r03_lsh_attn_fwd_ablation.log, r03_lsh_attn_fwd_walk_ablation.log.
#include "rtts_common.h"
#include <float.h>

#define AF_DH 64
#define AF_ROWB 144   // LDS row stride (bytes): 128 B of bf16 + 16 B pad => conflict-free ds_read_b128
#define AF_LOG2E 1.4426950408889634f
#define AF_NEG (-1.0e30f)
#define AF_BIGPOS 0x40000000
#define AF_SLACK 8.0f   // see the tile loop: the running maximum may lag the true one by this many log2 units

typedef __attribute__((ext_vector_type(8))) short af_short8;
typedef __attribute__((ext_vector_type(2))) __bf16 af_bf2;

// Phase probe (scripts/phase_probe.py --fwd builds a private copy with -DAF_PHASE_TIMING): waves 0 and NQT (the two key
// halves of query tile 0) stamp the shader clock at the phase boundaries.  Never defined in the product build.
#ifdef AF_PHASE_TIMING
__device__ unsigned long long g_af_phase[16 * 8192];
#define AF_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == BS / 32)) \
        g_af_phase[blockIdx.x * 16 + ((threadIdx.x >> 6) ? 8 : 0) + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int rtts_debug_af_phases(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_af_phase), sizeof(g_af_phase)); }
// walking form: waves 0 (own keys) and NQT (looked-back keys) stamp step AF_WSTEP of every run; slots [0, 8) / [8, 16)
#ifndef AF_WSTEP
#define AF_WSTEP 1
#endif
#define AF_WSTAMP(i) do { if (j == AF_WSTEP && lane == 0 && blockIdx.x < 8192 && qt == 0) \
        g_af_phase[blockIdx.x * 16 + (kh ? 8 : 0) + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define AF_STAMP(i) do { } while (0)
#define AF_WSTAMP(i) do { } while (0)
#endif

// key-tile loop unroll: 2 for bucket size 64 (4-wave workgroups, registers to spare: -4 %), 1 for 128 (unrolling costs the
// second workgroup per CU: +35 %); scripts/ab_attn.py --fwd measures others
#ifndef AF_UNROLL
#define AF_UNROLL BS == 64 ? 2 : 1
#endif
// V image: [row][128 B] without padding, the eight 16-byte pieces of a row XOR-swizzled (same function as the backward's
// images, lsh_attn_bwd.hip) so that the transposed reads (4 consecutive rows x 64 B) are bank-conflict free; it is filled by
// LDS-DMA, the swizzle applied on the source side.
__device__ __forceinline__ int af_sw(int row) { return ((row >> 1) & 3) | ((((row >> 3) ^ (row >> 1)) & 1) << 2); }
__device__ __forceinline__ int af_voff(int row, int piece) { return row * 128 + ((piece ^ af_sw(row)) << 4); }

// Sum of a lane's 16 probabilities as a TREE (depth 4): the plain `l += p` loop is a chain of 16 dependent adds, and with four
// waves per SIMD the chain's latency, not its issue slots, is what the tile pays (ablation r03: dropping the sum bought 6 %
// of the kernel, dropping the 32 mask instructions of a tile 7 %).
__device__ __forceinline__ float af_sum16(const f32x16& a) {
    const float s0 = (a[0] + a[1]) + (a[2] + a[3]), s1 = (a[4] + a[5]) + (a[6] + a[7]);
    const float s2 = (a[8] + a[9]) + (a[10] + a[11]), s3 = (a[12] + a[13]) + (a[14] + a[15]);
    return (s0 + s1) + (s2 + s3);
}

// DROP: dropout on the attention probabilities (the reference layer's `dropout` knob, reformer_tts/model/config.py:27: the
// per-chunk softmax output is dropped before it multiplies the values; lse -- hence the round weights -- sees the undropped
// probabilities).  Counter-hash mask keyed by (seed, pair index): pair = ((head * chunks + chunk) * BS + query row) * 2BS +
// key row, the same index in the backward kernels, so no mask is stored and the reversible recompute redraws the same one.
// ---- the softmax arithmetic of one 32 x 32 tile, shared by both kernel forms (so that they agree bit for bit) -------------
// acc: this lane's query against the tile's 32 keys, raw q.k (rows = keys 8g + 4hh + j).  -> acc = 2^(x - m) with x = q.k * scale[key],
// masked; (m, l, O) updated.  AF_FOLD (default): scale and the reference ride in ONE fma, x' = fma(q.k, scale, -mref), so the tile
// costs 16 fma + the masks + the maxima + 16 exp + the sums -- the separate 16 multiplies and 16 subtractions of round 3 are gone
// (the ISA census had them at 32 of 113 vector instructions per tile).  mref = m, or 0 for a lane that has seen no live key but itself yet
// (m = AF_NEG, or the self constant, would swallow the logit in the fma); when the lazy reference moves (rarely), the 16 values are shifted once.
#ifndef AF_FOLD
#define AF_FOLD 1
#endif
#ifndef AF_TILE_SKIP
#define AF_TILE_SKIP 1
#endif
__device__ __forceinline__ void af_tile_probs(f32x16& acc, const float* __restrict__ ksc_t, const int* __restrict__ kpos_t,
                                              const int* __restrict__ kpe_t, int keyt, int hh, bool chk_self, int qpos, int qpe, float& m,
                                              float& l, f32x16 (&oacc)[2], bool live_all = false) {
    // "fresh": nothing live seen yet (m = AF_NEG), or only the query itself (m = -5e4 log2 e = -72134.75: a reference of that size
    // would cost the next real logit its low bits in the fma -- fp32 has 2^-7 of resolution there; seen as lse errors of 2.7e-3
    // on early causal positions before this line read `m < -1e4`).  Real logits are bounded by |q| dh^-1/2 log2 e.
    const bool fresh = m < -1.0e4f;
    const float mref = fresh ? 0.f : m;
    const float nref = -mref, selfv = (-5e4f * AF_LOG2E) - mref;
    float gmax[2] = {AF_NEG, AF_NEG};                    // two independent chains, joined below
    // can a key of this tile BE the query itself?  (wave-uniform: the common path skips the test)
    if (chk_self) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key0 = keyt + 8 * g + 4 * hh;
            const float4 sc = *reinterpret_cast<const float4*>(ksc_t + key0);
            const int4 kp = *reinterpret_cast<const int4*>(kpos_t + key0);
            const int4 ke = *reinterpret_cast<const int4*>(kpe_t + key0);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
            const int kpv[4] = {kp.x, kp.y, kp.z, kp.w};
            const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = __builtin_fmaf(acc[4 * g + j], scv[j], nref);
                x = (kev[j] > qpe) ? AF_NEG : x;
                x = (kpv[j] == qpos) ? selfv : x;
                acc[4 * g + j] = x;
                gmax[g & 1] = fmaxf(gmax[g & 1], x);
            }
        }
    } else if (live_all) {
        // every key of the tile is visible to every query of this wave (wave-uniform, from the tile's largest effective key
        // position): no compares, no selects -- 32 of the tile's ~100 vector instructions
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 sc = *reinterpret_cast<const float4*>(ksc_t + keyt + 8 * g + 4 * hh);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = __builtin_fmaf(acc[4 * g + j], scv[j], nref);
                acc[4 * g + j] = x;
                gmax[g & 1] = fmaxf(gmax[g & 1], x);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key0 = keyt + 8 * g + 4 * hh;
            const float4 sc = *reinterpret_cast<const float4*>(ksc_t + key0);
            const int4 ke = *reinterpret_cast<const int4*>(kpe_t + key0);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
            const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = __builtin_fmaf(acc[4 * g + j], scv[j], nref);
                x = (kev[j] > qpe) ? AF_NEG : x;
                acc[4 * g + j] = x;
                gmax[g & 1] = fmaxf(gmax[g & 1], x);
            }
        }
    }
    const float tmax = rtts_xhalf_max(fmaxf(gmax[0], gmax[1]));       // relative to mref
    // LAZY reference: m follows the running maximum only when a tile exceeds it by more than AF_SLACK (log2 units), so
    // probabilities stay below 2^AF_SLACK (exact in fp32, and bf16 keeps its relative precision) and the rescale of O
    // and l runs for the first live tile of a wave and then almost never (wave-uniform branch).  (m, l, O) stay a
    // consistent triple, so the merge and lse = m + log2 l need no change.  A fresh lane moves at its first live key.
    if (__any(tmax > (fresh ? 0.5f * AF_NEG : AF_SLACK))) {
        const float mnew = fmaxf(m, tmax + mref);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        const float d = mnew - mref;
        m = mnew;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            oacc[0][i] *= alpha;
            oacc[1][i] *= alpha;
            acc[i] -= d;
        }
        l *= alpha;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i]);
    l += af_sum16(acc);
}

template <int BS, bool CAUSAL, bool MASKED, bool DROP>
__global__ __launch_bounds__(BS * 4, 2) void lsh_attn_fwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v,
                                                                 int64_t ld, const int32_t* __restrict__ st,
                                                                 const uint8_t* __restrict__ mask, int H, int T, int n_hashes,
                                                                 bf16_t* __restrict__ o, float* __restrict__ lse, uint32_t drop_seed,
                                                                 const uint32_t* __restrict__ seed_dev, uint32_t drop_thresh,
                                                                 float drop_scale) {
    constexpr int NK = 2 * BS;
    constexpr int NKT = NK / 32;
    constexpr int NTHR = BS * 4;          // two waves per 32-query tile: each walks one half of the keys
    constexpr int NQT = BS / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the three per-key word arrays come FIRST: the tile loop reads them at (one base register + immediate), and a DS
    // immediate offset is 16 bits -- behind the 72 KB of images every one of those reads needed its own address add
    float* ksc = reinterpret_cast<float*>(smem);                 // dh^-1/2 / |k| * log2(e)
    int* kpos = reinterpret_cast<int*>(ksc + NK);                // original position (self test)
    int* kpe = kpos + NK;                                        // effective position for the `dead` compare
    unsigned char* Ks = smem + NK * 12;
    unsigned char* Vs = Ks + NK * AF_ROWB;                       // [NK][128] swizzled (the output staging reuses it, padded)

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / C, c = wi % C;
    const int b = bh / H, h = bh % H;
    const int cprev = (c == 0) ? C - 1 : c - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * AF_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * AF_DH;

    AF_STAMP(0);
    // ---- 1. gather -------------------------------------------------------------------
    constexpr int ITERS = NK * 8 / NTHR;   // = 4
    {
        int trow[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3;
            const int slot = (row < BS) ? c * BS + row : cprev * BS + (row - BS);
            trow[it] = st_row[slot];
        }
        uint4 kreg[ITERS];
        int rvalid[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int piece = tid & 7;
            kreg[it] = *reinterpret_cast<const uint4*>(qbase + (size_t)trow[it] * ld + piece * 8);
            // V rows: global -> LDS by DMA; one wave-instruction fills 8 consecutive rows in lane order, so the lane that
            // lands on physical piece (lane & 7) of row (lane >> 3) fetches logical piece (lane & 7) ^ sw(row)
            const int rowb = it * (NTHR / 8) + wave * 8;
            const int lp = (lane & 7) ^ af_sw(rowb + (lane >> 3));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (size_t)trow[it] * ld + lp * 8),
                                             (RTTS_LDS void*)(Vs + rowb * 128), 16, 0, 0);
            // everything indexed by the position is requested now: two dependent global round trips, not three
            rvalid[it] = MASKED ? (int)mask[(size_t)b * T + trow[it]] : 1;
        }
        const int cbase = (tid & 7) * (NK * 4);   // word array of this lane: 0 ksc, 1 kpos, 2 kpe
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
            *reinterpret_cast<uint4*>(Ks + row * AF_ROWB + piece * 16) = kreg[it];
            // |row piece|^2 straight from the packed pairs: v_dot2c_f32_bf16, 4 instructions per 16 bytes (unpack + fma: 16)
            float ss = 0.f;
            ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, kreg[it].x), __builtin_bit_cast(af_bf2, kreg[it].x), ss, false);
            ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, kreg[it].y), __builtin_bit_cast(af_bf2, kreg[it].y), ss, false);
            ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, kreg[it].z), __builtin_bit_cast(af_bf2, kreg[it].z), ss, false);
            ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, kreg[it].w), __builtin_bit_cast(af_bf2, kreg[it].w), ss, false);
            ss = rtts_sum8(ss);
            // the row's three words leave in one ds_write_b32: lane `piece` stores word `piece`
            int w = __float_as_int((0.125f * AF_LOG2E) * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)));   // 1 / max(|k|, 1e-12)
            w = piece == 1 ? trow[it] : w;
            // dead <=> kpe > qpe.  causal: positions; otherwise 0.  An invalid key is beyond every query.
            w = piece == 2 ? (rvalid[it] ? (CAUSAL ? trow[it] : 0) : AF_BIGPOS) : w;
            if (piece < 3) *reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ksc) + cbase + row * 4) = w;
        }
    }
    AF_STAMP(1);
    __syncthreads();
    AF_STAMP(2);

    // ---- 2. per-wave online softmax over key tiles ---------------------------------------
    const int r = lane & 31, hh = lane >> 5;
    const int qt = wave % NQT, kh = wave / NQT;
    const int qrow = qt * 32 + r;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(Ks + qrow * AF_ROWB + (ks * 16 + 8 * hh) * 2);
    const int qpos = kpos[qrow];
    // an invalid query sees nothing but itself: effective position below every key's
    const int qpe = (kpe[qrow] == AF_BIGPOS) ? -1 : (CAUSAL ? qpos : 0);
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;

    int tro[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) tro[dt] = af_voff(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);

    const bool wrap = (cprev / nb) != (c / nb);
    float m = AF_NEG, l = 0.f;
    f32x16 oacc[2] = {{0}, {0}};
#pragma unroll AF_UNROLL
    for (int kt = kh * (NKT / 2); kt < (kh + 1) * (NKT / 2); ++kt) {
        f32x16 acc = {0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + r) * AF_ROWB + (ks * 16 + 8 * hh) * 2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
        }
#if AF_FOLD && !defined(AF_ABLATE)
        af_tile_probs(acc, ksc, kpos, kpe, kt * 32, hh, (kt < NQT) ? (kt == qt) : wrap, qpos, qpe, m, l, oacc);
#else
        float tmax = AF_NEG;
        float gmax[2] = {AF_NEG, AF_NEG};                    // two independent chains, joined below
#if defined(AF_ABLATE) && AF_ABLATE == 3
        // timing experiment only: no scale, no masks, no max, no exp
        tmax = 0.f;
#else
        // can a key of this tile BE the query itself?  own keys: only the diagonal tile; looked-back keys: only when the
        // previous chunk belongs to another hash round.  Wave-uniform: the common path skips the test.
        const bool chk_self = (kt < NQT) ? (kt == qt) : wrap;
        if (chk_self) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kt * 32 + 8 * g + 4 * hh;
                const float4 sc = *reinterpret_cast<const float4*>(ksc + key0);
                const int4 kp = *reinterpret_cast<const int4*>(kpos + key0);
                const int4 ke = *reinterpret_cast<const int4*>(kpe + key0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
                const int kpv[4] = {kp.x, kp.y, kp.z, kp.w};
                const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[4 * g + j] * scv[j];
                    x = (kev[j] > qpe) ? AF_NEG : x;
                    x = (kpv[j] == qpos) ? (-5e4f * AF_LOG2E) : x;
                    acc[4 * g + j] = x;
                    gmax[g & 1] = fmaxf(gmax[g & 1], x);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = kt * 32 + 8 * g + 4 * hh;
                const float4 sc = *reinterpret_cast<const float4*>(ksc + key0);
                const int4 ke = *reinterpret_cast<const int4*>(kpe + key0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
                const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[4 * g + j] * scv[j];
#if !defined(AF_ABLATE) || AF_ABLATE != 2
                    x = (kev[j] > qpe) ? AF_NEG : x;
#endif
                    acc[4 * g + j] = x;
                    gmax[g & 1] = fmaxf(gmax[g & 1], x);
                }
            }
        }
        tmax = rtts_xhalf_max(fmaxf(gmax[0], gmax[1]));
#endif
        // LAZY reference: m follows the running maximum only when a tile exceeds it by more than AF_SLACK (log2 units), so
        // probabilities stay below 2^AF_SLACK (exact in fp32, and bf16 keeps its relative precision) and the rescale of O
        // and l -- 34 multiplies and an exp per tile -- runs for the first live tile of a wave and then almost never
        // (wave-uniform branch).  (m, l, O) stay a consistent triple, so the merge and lse = m + log2 l need no change.
        if (__any(tmax > m + AF_SLACK)) {
            const float mnew = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oacc[0][i] *= alpha;
                oacc[1][i] *= alpha;
            }
            l *= alpha;
        }
#if defined(AF_ABLATE) && AF_ABLATE == 3
#pragma unroll
        for (int i = 0; i < 16; ++i) l += acc[i];
#else
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i] - m);
        l += af_sum16(acc);
#endif
#endif
        if constexpr (DROP) {
            const uint32_t seed = drop_seed + (seed_dev ? seed_dev[0] : 0u);
            const uint32_t pair0 = ((uint32_t)wi * BS + (uint32_t)qrow) * (uint32_t)NK + (uint32_t)(kt * 32 + 4 * hh);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= rtts_drop_keep(seed, pair0 + 8 * (i >> 2) + (i & 3), drop_thresh, drop_scale);
        }
#if defined(AF_ABLATE) && AF_ABLATE == 4
        oacc[0][0] += acc[0] + acc[5] + acc[10] + acc[15];      // timing experiment only: no P V product
        if (false)
#endif
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int o8 = 8 * s2;
            const bf16x8 pf = cvt_bf16x8(acc[o8], acc[o8 + 1], acc[o8 + 2], acc[o8 + 3], acc[o8 + 4], acc[o8 + 5], acc[o8 + 6], acc[o8 + 7]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // rows 4*hh + trq (+8) of the 16-key block, 8-byte granule dt*8 + 4*trc + trp; sw(row + 8) = sw(row) ^ 4
                const int blk = (kt * 32 + 16 * s2) * 128;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)(Vs + blk + tro[dt]));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)(Vs + blk + 8 * 128 + tro[dt ^ 1]));
                const af_short8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf, oacc[dt], 0, 0, 0);
            }
        }
    }
    l = rtts_xhalf_sum(l);

    // ---- merge the two key halves of a query tile through LDS (aliases the K image) ----------
    float* part = reinterpret_cast<float*>(Ks) + (size_t)qt * 34 * 64;   // [34][64]: 32 x O, m, l per lane
    AF_STAMP(3);
    __syncthreads();                       // every wave is done reading K
    AF_STAMP(4);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            part[i * 64 + lane] = oacc[0][i];
            part[(16 + i) * 64 + lane] = oacc[1][i];
        }
        part[32 * 64 + lane] = m;
        part[33 * 64 + lane] = l;
    }
    __syncthreads();
    if (kh == 1) return;
    {
        const float m2 = part[32 * 64 + lane], l2 = part[33 * 64 + lane];
        const float mm = fmaxf(m, m2);
        float a1 = __builtin_amdgcn_exp2f(m - mm), a2 = __builtin_amdgcn_exp2f(m2 - mm);
        l = l * a1 + l2 * a2;
        m = mm;
        const float inv_l = 1.f / l;         // the normalisation rides in the merge coefficients
        a1 *= inv_l;
        a2 *= inv_l;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            oacc[0][i] = oacc[0][i] * a1 + part[i * 64 + lane] * a2;
            oacc[1][i] = oacc[1][i] * a1 + part[(16 + i) * 64 + lane] * a2;
        }
    }

    // ---- 5. write o, lse at the unsorted position ------------------------------------------
    // The accumulator holds a query per lane and dh down the registers; the tile goes through a [32][144 B] LDS
    // staging (the V image is dead by now) so that a row leaves as eight 16-byte pieces = one full 128-byte line.
    const int round = c / nb;
    const size_t obase = ((size_t)bh * n_hashes + round) * T;
    unsigned char* stg = Vs + qt * (32 * AF_ROWB);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(oacc[dt][4 * g], oacc[dt][4 * g + 1]);
            pk.y = pack_bf16x2(oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]);
            *reinterpret_cast<uint2*>(stg + r * AF_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2) = pk;
        }
    }
    if (hh == 0) lse[obase + qpos] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;   // v_log_f32 is log2
    __builtin_amdgcn_wave_barrier();
    const int srow = lane >> 3, spiece = lane & 7;
    uint4 rowv[4];
    int rpos[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rowv[i] = *reinterpret_cast<const uint4*>(stg + (i * 8 + srow) * AF_ROWB + spiece * 16);
        rpos[i] = kpos[qt * 32 + i * 8 + srow];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rtts_store16_out(o + (obase + rpos[i]) * AF_DH + spiece * 8, rowv[i]);
    AF_STAMP(5);
}


// ======================================================================================================================
// Walking form (round 3): one workgroup works a RUN of L consecutive chunks of one (batch, head) ring.
//
// The own chunk of step j is the looked-back chunk of step j+1, so its K / V rows, norms and positions stay in LDS: each row is
// gathered ONCE (the one-chunk kernel above fetches every row twice), and the fetch of the next chunk hides behind the merge
// and the row stores of the current one:
//   * LDS holds two chunk slots (K image [BS][144 B] + V image [BS][128 B] swizzled + norms); slot (j & 1) is the own chunk
//     of step j.  Positions and effective positions of ALL L+1 chunks of the run are fetched once, up front;
//   * the waves keep the split of the one-chunk kernel: for every 32-query tile one wave walks the own keys ("own waves"),
//     one the looked-back keys ("back waves"), merged through LDS.  After the tiles the looked-back slot is dead: it takes the
//     back waves' partial (m, l, O) -- exactly BS x 272 B -- and then the output staging;
//   * while the own waves merge, normalise, stage and store the rows, the back waves -- whose 32 accumulator registers are
//     free by then -- have the NEXT chunk's K and V rows in flight (global -> registers), and write them into the dead slot
//     as soon as the own waves are done with it.
// Per step: tiles -> barrier -> [partials] -> barrier -> [merge + stores | loads in flight] -> barrier -> [rows -> LDS] -> barrier.
// The pair index of the dropout mask, the masks and the arithmetic of a tile are those of the one-chunk kernel: the two forms
// agree bit for bit (tests/test_lsh_hip.py::test_walking_forward_matches_the_one_chunk_kernel).
template <int BS, bool DROP>
__device__ __forceinline__ void af_walk_tiles(const unsigned char* __restrict__ Kt, const unsigned char* __restrict__ Vt,
                                              const float* __restrict__ ksc_t, const int* __restrict__ kpos_t,
                                              const int* __restrict__ kpe_t, const int* __restrict__ kmin_t, const int* __restrict__ kmax_t,
                                              const bf16x8 (&qf)[4], int qpos, int qpe, int self_lo,
                                              int self_hi, int r, int hh, const int (&tro)[2], float& m, float& l, f32x16 (&oacc)[2],
                                              uint32_t pair_base, uint32_t seed, uint32_t drop_thresh, float drop_scale) {
    constexpr int NT = BS / 32;
#pragma unroll AF_UNROLL
    for (int t = 0; t < NT; ++t) {
        // AF_TILE_SKIP: positions inside a bucket are sorted, so whole 32 x 32 tiles are often uniformly dead (every key behind every
        // query of this wave: nothing to add -- the tile is skipped, MFMAs included) or uniformly live (no masks needed): 18 % and
        // 19 % of the tiles at random inputs, more on real text.  Both decisions are wave-uniform, from the tile's smallest / largest
        // effective key position (kmin_t / kmax_t, one word per tile, written once per run); a tile that may hold the query itself is
        // never skipped.  A skipped tile contributes exact zeros either way: results are bit-identical with and without.
        const bool self_tile = t >= self_lo && t < self_hi;
        bool live_all = false;
#if AF_TILE_SKIP
        if (!self_tile) {
            if (__all(kmin_t[t] > qpe)) continue;
            live_all = __all(kmax_t[t] <= qpe);
        }
#endif
        f32x16 acc = {0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kt + (t * 32 + r) * AF_ROWB + (ks * 16 + 8 * hh) * 2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
        }
#if AF_FOLD && !defined(AF_ABLATE)
        af_tile_probs(acc, ksc_t, kpos_t, kpe_t, t * 32, hh, self_tile, qpos, qpe, m, l, oacc, live_all);
#else
        float tmax = AF_NEG;
        float gmax[2] = {AF_NEG, AF_NEG};                    // two independent chains, joined below
        if (t >= self_lo && t < self_hi) {          // wave-uniform: a key of this tile can be the query itself
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = t * 32 + 8 * g + 4 * hh;
                const float4 sc = *reinterpret_cast<const float4*>(ksc_t + key0);
                const int4 kp = *reinterpret_cast<const int4*>(kpos_t + key0);
                const int4 ke = *reinterpret_cast<const int4*>(kpe_t + key0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
                const int kpv[4] = {kp.x, kp.y, kp.z, kp.w};
                const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[4 * g + j] * scv[j];
                    x = (kev[j] > qpe) ? AF_NEG : x;
                    x = (kpv[j] == qpos) ? (-5e4f * AF_LOG2E) : x;
                    acc[4 * g + j] = x;
                    gmax[g & 1] = fmaxf(gmax[g & 1], x);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key0 = t * 32 + 8 * g + 4 * hh;
                const float4 sc = *reinterpret_cast<const float4*>(ksc_t + key0);
                const int4 ke = *reinterpret_cast<const int4*>(kpe_t + key0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
                const int kev[4] = {ke.x, ke.y, ke.z, ke.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[4 * g + j] * scv[j];
#if !defined(AF_ABLATE) || AF_ABLATE != 2
                    x = (kev[j] > qpe) ? AF_NEG : x;
#endif
                    acc[4 * g + j] = x;
                    gmax[g & 1] = fmaxf(gmax[g & 1], x);
                }
            }
        }
        tmax = rtts_xhalf_max(fmaxf(gmax[0], gmax[1]));
        if (__any(tmax > m + AF_SLACK)) {           // lazy reference, as in the one-chunk kernel
            const float mnew = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - mnew);
            m = mnew;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                oacc[0][i] *= alpha;
                oacc[1][i] *= alpha;
            }
            l *= alpha;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#if defined(AF_ABLATE) && AF_ABLATE == 5
            acc[i] = acc[i] - m;                              // timing experiment only: no v_exp_f32
#else
            acc[i] = __builtin_amdgcn_exp2f(acc[i] - m);
#endif
        }
#if !defined(AF_ABLATE) || AF_ABLATE != 6
        l += af_sum16(acc);
#endif
#endif
        if constexpr (DROP) {
            const uint32_t pair0 = pair_base + (uint32_t)(t * 32 + 4 * hh);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= rtts_drop_keep(seed, pair0 + 8 * (i >> 2) + (i & 3), drop_thresh, drop_scale);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int o8 = 8 * s2;
            const bf16x8 pf = cvt_bf16x8(acc[o8], acc[o8 + 1], acc[o8 + 2], acc[o8 + 3], acc[o8 + 4], acc[o8 + 5], acc[o8 + 6], acc[o8 + 7]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int blk = (t * 32 + 16 * s2) * 128;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)(Vt + blk + tro[dt]));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)(Vt + blk + 8 * 128 + tro[dt ^ 1]));
                const af_short8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf, oacc[dt], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ float af_piece_sumsq(const uint4 q) {
    float ss = 0.f;
    ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, q.x), __builtin_bit_cast(af_bf2, q.x), ss, false);
    ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, q.y), __builtin_bit_cast(af_bf2, q.y), ss, false);
    ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, q.z), __builtin_bit_cast(af_bf2, q.z), ss, false);
    ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(af_bf2, q.w), __builtin_bit_cast(af_bf2, q.w), ss, false);
    return ss;
}

template <int BS, bool CAUSAL, bool MASKED, bool DROP>
__global__ __launch_bounds__(BS * 4, 2) void lsh_attn_fwd_walk_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v,
                                                                      int64_t ld, const int32_t* __restrict__ st,
                                                                      const uint8_t* __restrict__ mask, int H, int T, int n_hashes,
                                                                      bf16_t* __restrict__ o, float* __restrict__ lse,
                                                                      uint32_t drop_seed, const uint32_t* __restrict__ seed_dev,
                                                                      uint32_t drop_thresh, float drop_scale, int L) {
    constexpr int NK = 2 * BS;
    constexpr int NTHR = BS * 4;
    constexpr int NQT = BS / 32;
    const int NPOS = (L + 1) * BS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* allpos = reinterpret_cast<int*>(smem);                  // positions of the L+1 chunks of the run: [0] looked back at first
    int* allkpe = allpos + NPOS;                                 // effective positions (`dead` compare)
    float* ksc = reinterpret_cast<float*>(allkpe + NPOS);        // [2][BS]  dh^-1/2 / |k| * log2(e) of the rows in the two slots
    int* tkmin = reinterpret_cast<int*>(ksc + NK);               // smallest / largest effective position of every 32-key tile of the run
    int* tkmax = tkmin + 128;                                    // ((L + 1) * BS / 32 <= 68 entries each; 128 keeps the images 16-byte aligned)
    unsigned char* Ks = reinterpret_cast<unsigned char*>(tkmax + 128);   // [2][BS][144]
    unsigned char* Vs = Ks + NK * AF_ROWB;                       // [2][BS][128] swizzled

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const int runs = C / L;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / runs, c0 = (wi % runs) * L;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * AF_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * AF_DH;

    // ---- positions of the whole run ----------------------------------------------------------
    for (int i = tid; i < NPOS; i += NTHR) {
        int ch = c0 - 1 + i / BS;
        ch = ch < 0 ? ch + C : ch;                  // c0 + L - 1 < C: only the first looked-back chunk can wrap
        const int pos = st_row[ch * BS + (i % BS)];
        const int valid = MASKED ? (int)mask[(size_t)b * T + pos] : 1;
        allpos[i] = pos;
        allkpe[i] = valid ? (CAUSAL ? pos : 0) : AF_BIGPOS;      // dead <=> kpe > qpe; an invalid key is beyond every query
    }
    __syncthreads();
    // per-tile extremes of the effective key positions (read after the next barrier): wave w takes tiles w, w + waves, ...
    for (int tile = wave; tile < NPOS / 32; tile += NTHR / 64) {
        int lo = allkpe[tile * 32 + (lane & 31)], hi = lo;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if (lane == 0) { tkmin[tile] = lo; tkmax[tile] = hi; }
    }
    // ---- first step: both chunks (image rows [0, BS) = slot 0 = chunk c0, [BS, 2BS) = slot 1 = chunk c0 - 1) --------------
    {
        constexpr int ITERS = NK * 8 / NTHR;   // = 4
        uint4 kreg[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
            const int pos = allpos[row < BS ? BS + row : row - BS];
            kreg[it] = *reinterpret_cast<const uint4*>(qbase + (size_t)pos * ld + piece * 8);
            const int rowb = it * (NTHR / 8) + wave * 8;
            const int lp = (lane & 7) ^ af_sw(rowb + (lane >> 3));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (size_t)pos * ld + lp * 8),
                                             (RTTS_LDS void*)(Vs + rowb * 128), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
            *reinterpret_cast<uint4*>(Ks + row * AF_ROWB + piece * 16) = kreg[it];
            const float ss = rtts_sum8(af_piece_sumsq(kreg[it]));
            if (piece == 0) ksc[row] = (0.125f * AF_LOG2E) * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // 1 / max(|k|, 1e-12)
        }
    }
    __syncthreads();

    const int qt = wave % NQT, kh = wave / NQT;      // kh 0: own keys (and the merge / stores), 1: looked-back keys (and the fetch)
    const uint32_t seed = DROP ? drop_seed + (seed_dev ? seed_dev[0] : 0u) : 0u;
    int lane_o = lane;

#pragma unroll 1
    for (int j = 0; j < L; ++j) {
        // Nothing but scalars crosses the back edge: the lane id passes through an opaque move at the top of every step, so the
        // per-lane address arithmetic below is redone per step (a dozen instructions) instead of living in ~25 registers across
        // the whole loop, which put the tile loop over the 128 registers that two resident workgroups allow (80 B of scratch).
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o;
        const int tid = wave * 64 + lane;
        const int r = lane & 31, hh = lane >> 5;
        const int qrow = qt * 32 + r;
        const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
        int tro[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) tro[dt] = af_voff(4 * hh + trq, dt * 4 + 2 * trc + (trp >> 1)) + 8 * (trp & 1);
        AF_WSTAMP(0);
        const int so = j & 1, sp = so ^ 1;           // own slot, looked-back slot (dead after the tiles)
        const int c = c0 + j;
        const int cprev = (c == 0) ? C - 1 : c - 1;
        const int ms = so ^ kh;                      // slot of this wave's keys
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(Ks + (so * BS + qrow) * AF_ROWB + (ks * 16 + 8 * hh) * 2);
        const int qpos = allpos[(j + 1) * BS + qrow];
        // an invalid query sees nothing but itself: effective position below every key's
        const int qpe = (allkpe[(j + 1) * BS + qrow] == AF_BIGPOS) ? -1 : (CAUSAL ? qpos : 0);
        // can a key BE the query?  own keys: the diagonal tile; looked-back keys: only when that chunk belongs to another round
        const bool wrap = (cprev / nb) != (c / nb);
        const int self_lo = kh ? 0 : qt, self_hi = kh ? (wrap ? NQT : 0) : qt + 1;
        float m = AF_NEG, l = 0.f;
        f32x16 oacc[2] = {{0}, {0}};
        af_walk_tiles<BS, DROP>(Ks + ms * (BS * AF_ROWB), Vs + ms * (BS * 128), ksc + ms * BS, allpos + (j + 1 - kh) * BS,
                                allkpe + (j + 1 - kh) * BS, tkmin + (j + 1 - kh) * NQT, tkmax + (j + 1 - kh) * NQT, qf, qpos, qpe, self_lo,
                                self_hi, r, hh, tro, m, l, oacc,
                                ((uint32_t)(bh * C + c) * BS + (uint32_t)qrow) * (uint32_t)NK + (uint32_t)(kh * BS), seed,
                                drop_thresh, drop_scale);
        l = rtts_xhalf_sum(l);
        AF_WSTAMP(1);

        // the dead slot as the partial buffer: per query tile 18 rows of 64 words in its K image (O[0..15], m, l), 16 in its V image
        float* partA = reinterpret_cast<float*>(Ks + sp * (BS * AF_ROWB)) + qt * (18 * 64);
        float* partB = reinterpret_cast<float*>(Vs + sp * (BS * 128)) + qt * (16 * 64);
        __syncthreads();                             // (1) every wave is done with the K / V images of this step
        AF_WSTAMP(2);
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                partA[i * 64 + lane] = oacc[0][i];
                partB[i * 64 + lane] = oacc[1][i];
            }
            partA[16 * 64 + lane] = m;
            partA[17 * 64 + lane] = l;
            if (j + 1 == L) {
                __syncthreads();                     // (2)
                return;
            }
            // the next chunk's rows: global -> registers now (the accumulators are free), registers -> dead slot after (3)
            const int t2 = tid - NTHR / 2;           // 0 .. 2BS-1
            const int piece = t2 & 7, row0 = t2 >> 3;            // rows row0 + it * (NTHR / 16)
            const int* npos = allpos + (j + 2) * BS + row0;
            const bf16_t* kp = qbase + piece * 8;
            const bf16_t* vp = vbase + piece * 8;
#define AF_FETCH(IT)                                                                                   \
            const size_t off##IT = (size_t)npos[IT * (NTHR / 16)] * ld;                                \
            const uint4 kreg##IT = *reinterpret_cast<const uint4*>(kp + off##IT);                      \
            const uint4 vreg##IT = *reinterpret_cast<const uint4*>(vp + off##IT)
            AF_FETCH(0); AF_FETCH(1); AF_FETCH(2); AF_FETCH(3);
#undef AF_FETCH
            AF_WSTAMP(3);
            __syncthreads();                         // (2) partials visible
            AF_WSTAMP(4);
            __syncthreads();                         // (3) the own waves are done with the dead slot
            AF_WSTAMP(5);
#define AF_PLACE(IT)                                                                                   \
            do {                                                                                       \
                const int row = row0 + IT * (NTHR / 16);                                               \
                *reinterpret_cast<uint4*>(Ks + (sp * BS + row) * AF_ROWB + piece * 16) = kreg##IT;     \
                *reinterpret_cast<uint4*>(Vs + sp * (BS * 128) + af_voff(row, piece)) = vreg##IT;      \
                const float ss = rtts_sum8(af_piece_sumsq(kreg##IT));                                  \
                if (piece == 0) ksc[sp * BS + row] = (0.125f * AF_LOG2E) * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)); \
            } while (0)
            AF_PLACE(0); AF_PLACE(1); AF_PLACE(2); AF_PLACE(3);
#undef AF_PLACE
            AF_WSTAMP(6);
        } else {
            AF_WSTAMP(3);
            __syncthreads();                         // (2)
            AF_WSTAMP(4);
            {
                const float m2 = partA[16 * 64 + lane], l2 = partA[17 * 64 + lane];
                const float mm = fmaxf(m, m2);
                float a1 = __builtin_amdgcn_exp2f(m - mm), a2 = __builtin_amdgcn_exp2f(m2 - mm);
                l = l * a1 + l2 * a2;
                m = mm;
                const float inv_l = 1.f / l;
                a1 *= inv_l;
                a2 *= inv_l;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    oacc[0][i] = oacc[0][i] * a1 + partA[i * 64 + lane] * a2;
                    oacc[1][i] = oacc[1][i] * a1 + partB[i * 64 + lane] * a2;
                }
            }
            // rows of o and lse at their UNSORTED position; the tile goes through a [32][144 B] staging in this wave's own
            // 4608 bytes of the dead slot's K image (its partial rows are in registers by now: LDS operations of a wave are in order)
            const size_t obase = ((size_t)bh * n_hashes + c / nb) * T;
            unsigned char* stg = Ks + sp * (BS * AF_ROWB) + qt * (32 * AF_ROWB);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_bf16x2(oacc[dt][4 * g], oacc[dt][4 * g + 1]);
                    pk.y = pack_bf16x2(oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]);
                    *reinterpret_cast<uint2*>(stg + r * AF_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2) = pk;
                }
            }
            if (hh == 0) lse[obase + qpos] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;   // v_log_f32 is log2
            __builtin_amdgcn_wave_barrier();
            const int srow = lane >> 3, spiece = lane & 7;
            uint4 rowv[4];
            int rpos[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rowv[i] = *reinterpret_cast<const uint4*>(stg + (i * 8 + srow) * AF_ROWB + spiece * 16);
                rpos[i] = allpos[(j + 1) * BS + qt * 32 + i * 8 + srow];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) rtts_store16_out(o + (obase + rpos[i]) * AF_DH + spiece * 8, rowv[i]);
            AF_WSTAMP(5);
            if (j + 1 == L) return;
            __syncthreads();                         // (3) staging read: the dead slot may take the next chunk
            AF_WSTAMP(6);
        }
        __syncthreads();                             // (4) next chunk in place
        AF_WSTAMP(7);
    }
}

static RttsLdsState g_fwd_lds[2][8];
static RttsLdsState g_fwd_walk_lds[2][8];

// Chunks a workgroup of the walking form works in a row: the longest run (8, 4, 2) that divides the ring and still leaves three
// rounds of workgroups for the chip (256 CUs x 2 resident workgroups of 128-row buckets, x 4 of 64-row buckets); 0 = the
// one-chunk kernel (small grids: the encoder's T = 256).  rtts_debug_set_walk() forces a run length (tests, A/B runs).
extern "C" int rtts_lsh_attn_fwd_run_length(int B, int H, int T, int n_hashes, int bucket_size) {
    if (B <= 0 || H <= 0 || n_hashes <= 0 || (bucket_size != 64 && bucket_size != 128) || T <= 0 || T % bucket_size) return -1;
    const int C = n_hashes * (T / bucket_size);
    const long long chunks = (long long)B * H * C;
    const long long resident = 256ll * (bucket_size == 128 ? 2 : 4);
    int R = 0;
    for (int cand = 8; cand >= 2; cand >>= 1)
        if (C % cand == 0 && chunks / cand >= 3 * resident) { R = cand; break; }
    const int w = rtts_walk_override(0);          // tests / A-B runs only (rtts_debug_set_walk); -1 in every product call
    if (w >= 0) R = (w >= 1 && w <= 16 && C % w == 0) ? w : 0;
    return R;
}

template <int BS>
static int launch_attn_fwd(const bf16_t* qk, const bf16_t* v, int64_t ld, const int32_t* st, const uint8_t* mask, int B, int H,
                           int T, int n_hashes, int causal, bf16_t* o, float* lse, float drop_p, uint32_t seed,
                           const uint32_t* seed_dev, hipStream_t stream) {
    constexpr int NK = 2 * BS;
    const int L = rtts_lsh_attn_fwd_run_length(B, H, T, n_hashes, BS);
    const int C = n_hashes * (T / BS);
    const size_t lds = L > 0 ? (size_t)(2 * (L + 1) * BS + NK + 256) * 4 + (size_t)NK * (AF_ROWB + 128) : 2 * NK * AF_ROWB + NK * 12;
    const dim3 grid(L > 0 ? B * H * (C / L) : B * H * C), block(BS * 4);
    const bool drop = drop_p > 0.f;
    const int vi = (drop ? 4 : 0) + (causal ? 2 : 0) + (mask ? 1 : 0);
    const uint32_t th = rtts_drop_thresh(drop_p);
    const float sc = 1.f / (1.f - drop_p);
#define AF_GO(C_, M_, D_)                                                                                                  \
    do {                                                                                                                   \
        if (L > 0) {                                                                                                       \
            auto kern = lsh_attn_fwd_walk_kernel<BS, C_, M_, D_>;                                                          \
            RTTS_ENSURE_LDS("rtts_lsh_attn_fwd", kern, lds, g_fwd_walk_lds[BS == 128][vi]);                                \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, qk, v, ld, st, mask, H, T, n_hashes, o, lse, seed, seed_dev, th, sc, L); \
        } else {                                                                                                           \
            auto kern = lsh_attn_fwd_kernel<BS, C_, M_, D_>;                                                               \
            RTTS_ENSURE_LDS("rtts_lsh_attn_fwd", kern, lds, g_fwd_lds[BS == 128][vi]);                                     \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, qk, v, ld, st, mask, H, T, n_hashes, o, lse, seed, seed_dev, th, sc); \
        }                                                                                                                  \
    } while (0)
#define AF_GO2(C_, M_) do { if (drop) AF_GO(C_, M_, true); else AF_GO(C_, M_, false); } while (0)
    if (causal) {
        if (mask) AF_GO2(true, true); else AF_GO2(true, false);
    } else {
        if (mask) AF_GO2(false, true); else AF_GO2(false, false);
    }
#undef AF_GO2
#undef AF_GO
    RTTS_LAUNCH_CHECK("rtts_lsh_attn_fwd");
    return 0;
}

extern "C" int rtts_lsh_attn_fwd(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask, int B,
                                 int H, int T, int dh, int n_hashes, int bucket_size, int causal, void* o, float* lse,
                                 float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream) {
    RTTS_ENTER(stream);
    RTTS_REQUIRE(qk && v && st && o && lse, "rtts_lsh_attn_fwd: null pointer");
    RTTS_REQUIRE(dh == AF_DH, "rtts_lsh_attn_fwd: dh=%d unsupported (this build: 64)", dh);
    RTTS_REQUIRE(bucket_size == 64 || bucket_size == 128, "rtts_lsh_attn_fwd: bucket_size=%d unsupported (64 or 128)", bucket_size);
    RTTS_REQUIRE(T > 0 && T % (2 * bucket_size) == 0,
                 "rtts_lsh_attn_fwd: Sequence length (%d) needs to be divisible by target bucket size x 2 - %d", T, 2 * bucket_size);
    RTTS_REQUIRE(B > 0 && H > 0 && n_hashes > 0, "rtts_lsh_attn_fwd: bad B/H/n_hashes");
    RTTS_REQUIRE(ld >= (int64_t)H * dh && ld % 8 == 0, "rtts_lsh_attn_fwd: ld must be >= H*dh and a multiple of 8");
    RTTS_REQUIRE((((uintptr_t)qk | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "rtts_lsh_attn_fwd: qk, v, o must be 16-byte aligned");
    RTTS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "rtts_lsh_attn_fwd: drop_p must be in [0, 1)");
    // the dropout mask is a hash of the 32-bit pair index: beyond 2^32 pairs masks would repeat (forward and backward wrap alike,
    // so gradients would still match; refused all the same).  Without dropout nothing indexes pairs: no limit.
    RTTS_REQUIRE(drop_p == 0.f || (uint64_t)B * H * n_hashes * T * 2 * bucket_size < (1ull << 32),
                 "rtts_lsh_attn_fwd: dropout on more than 2^32 query-key pairs");
    hipStream_t s = (hipStream_t)stream;
    if (bucket_size == 64)
        return launch_attn_fwd<64>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, B, H, T, n_hashes, causal, (bf16_t*)o, lse, drop_p,
                                   drop_seed, seed_dev, s);
    return launch_attn_fwd<128>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, B, H, T, n_hashes, causal, (bf16_t*)o, lse, drop_p,
                                drop_seed, seed_dev, s);
}
