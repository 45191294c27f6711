// LSH chunked attention backward, "ring" form: one PERSISTENT workgroup per (batch*head, hash round)
// walks the round's chunks in sorted order.
//
// Same mathematics as lsh_attn_bwd.hip (see its header: one softmax over all rounds, P' = exp(s - LSE),
// dS' = P'(dP - delta) kscale, dQ = dS' K, G' = dS'^T Q, dK = G' - k^(k^.G'), dV = P'^T dout; backward of
// SURVEY.md Appendix B steps 4-11, which the reference obtains from autograd through reformer_pytorch via
// reformer_tts/model/reversible.py:69-85).  What changes is the data movement:
//
//   * chunk c attends to keys of chunks c and c-1, so consecutive chunks share half of their keys.  The
//     workgroup keeps a 2-slot ring of K images in LDS and two groups of waves that own the key tiles of the
//     two resident chunks; when it advances to chunk c+1 only that chunk is gathered (each K/V/dout row is
//     read ONCE per round instead of twice) and the group that held chunk c-1 takes chunk c+1 -- the group
//     that holds chunk c simply keeps its registers (K/V fragments, dK/dV accumulators);
//   * dK and dV of a chunk are therefore COMPLETE in registers after two consecutive iterations (own-chunk
//     queries, then the next chunk's queries) and are written once: 3 gradient slots per (round, token)
//     (dq, dk, dv) instead of 5 partial ones.  Only the first chunk of a round needs the last chunk of the
//     previous round (another workgroup's keys): its contribution goes to small "halo" buffers
//     (BS rows per round) that rtts_lsh_bwd_reduce_ring folds in through the inverse permutation `undo`;
//   * the next chunk's rows are fetched into registers while the current chunk computes.
// Nothing is accumulated across workgroups; no atomics; deterministic.
#include "rtts_common.h"
#include <float.h>

#define RB_DH 64
#define RB_ROWB 144
#define RB_LOG2E 1.4426950408889634f
#define RB_BIG 0x40000000

typedef __attribute__((ext_vector_type(8))) short rb_short8;

__device__ __forceinline__ bf16x8 rb_tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p1);
    const rb_short8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

template <int BS, bool CAUSAL, bool MASKED>
__global__ __launch_bounds__(BS * 4, 2) void lsh_attn_bwd_ring_kernel(
    const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v, int64_t ld, const int32_t* __restrict__ st,
    const uint8_t* __restrict__ mask, const bf16_t* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse_tot,
    const float* __restrict__ delta, int H, int T, int n_hashes, bf16_t* __restrict__ dq_part, bf16_t* __restrict__ dk_part,
    bf16_t* __restrict__ dv_part, bf16_t* __restrict__ halo_dk, bf16_t* __restrict__ halo_dv) {
    constexpr int NK = 2 * BS;
    constexpr int NQT = BS / 32;
    constexpr int NTHR = BS * 4;
    constexpr int NG = BS / 32;              // waves per group = 32-key tiles per chunk
    constexpr int DSROW = BS * 2 + 16;       // bytes per row of the dS'^T image [key][query]
    constexpr int PIECES = BS * 8 / NTHR;    // 16-B pieces per thread per staged image (= 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Kring = smem;                                  // [2][BS][144]
    unsigned char* Os = Kring + 2 * BS * RB_ROWB;                 // [BS][144]   dout rows of the query chunk
    unsigned char* Ds = Os + BS * RB_ROWB;                        // [NK][DSROW] rows 0..BS-1: own keys, BS..: previous chunk's
    float* kscale = reinterpret_cast<float*>(Ds + NK * DSROW);    // [2][BS]
    int* kpos = reinterpret_cast<int*>(kscale + 2 * BS);          // [2][BS]
    int* kpe = kpos + 2 * BS;                                     // [2][BS]
    float* qlse = reinterpret_cast<float*>(kpe + 2 * BS);         // [BS] lse_tot * log2(e)
    float* qdel = qlse + BS;                                      // [BS]

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / n_hashes, round = wi % n_hashes;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int grp = wave / NG, wtile = wave % NG;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;

    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * RB_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * RB_DH;
    const bf16_t* dobase = dout + (size_t)b * T * ld_do + (size_t)h * RB_DH;
    const size_t rt_base = ((size_t)bh * n_hashes + round) * T;   // row base of this (head, round) in the (.., T, 64) slots

    // ---- staging helpers: thread -> (row, piece) of a BS-row image -------------------------------------------
    int srow[PIECES];
#pragma unroll
    for (int it = 0; it < PIECES; ++it) srow[it] = (it * NTHR + tid) >> 3;
    const int spiece = tid & 7;
    int stok[PIECES];
    uint4 kst[PIECES], ost[PIECES];

    int snext[PIECES];   // token indices of the chunk after the one in flight (index loads run one iteration ahead)
#define RB_LOAD_IDX(chunk)                                                                                    \
    do {                                                                                                      \
        _Pragma("unroll") for (int it = 0; it < PIECES; ++it) snext[it] = st_row[(size_t)(chunk) * BS + srow[it]]; \
    } while (0)
#define RB_LOAD_CHUNK(with_do)                                                                                \
    do {                                                                                                      \
        _Pragma("unroll") for (int it = 0; it < PIECES; ++it) {                                               \
            stok[it] = snext[it];                                                                             \
            kst[it] = *reinterpret_cast<const uint4*>(qbase + (size_t)stok[it] * ld + spiece * 8);            \
            if (with_do) ost[it] = *reinterpret_cast<const uint4*>(dobase + (size_t)stok[it] * ld_do + spiece * 8); \
        }                                                                                                     \
    } while (0)

#define RB_STORE_CHUNK(slot, with_do)                                                                         \
    do {                                                                                                      \
        _Pragma("unroll") for (int it = 0; it < PIECES; ++it) {                                               \
            const int row_ = srow[it];                                                                        \
            *reinterpret_cast<uint4*>(Kring + ((slot) * BS + row_) * RB_ROWB + spiece * 16) = kst[it];        \
            if (with_do) *reinterpret_cast<uint4*>(Os + row_ * RB_ROWB + spiece * 16) = ost[it];              \
            const uint32_t u_[4] = {kst[it].x, kst[it].y, kst[it].z, kst[it].w};                              \
            float ss_ = 0.f;                                                                                  \
            _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                                \
                const float a_ = __uint_as_float(u_[k_] << 16), b_ = __uint_as_float(u_[k_] & 0xffff0000u);   \
                ss_ = __builtin_fmaf(a_, a_, ss_);                                                            \
                ss_ = __builtin_fmaf(b_, b_, ss_);                                                            \
            }                                                                                                 \
            ss_ += __shfl_xor(ss_, 1);                                                                        \
            ss_ += __shfl_xor(ss_, 2);                                                                        \
            ss_ += __shfl_xor(ss_, 4);                                                                        \
            if (spiece == 0) {                                                                                \
                const int tok_ = stok[it];                                                                    \
                const int valid_ = MASKED ? (int)mask[(size_t)b * T + tok_] : 1;                              \
                kscale[(slot) * BS + row_] = 0.125f * __builtin_amdgcn_rsqf(fmaxf(ss_, 1e-24f));              \
                kpos[(slot) * BS + row_] = tok_;                                                              \
                kpe[(slot) * BS + row_] = valid_ ? (CAUSAL ? tok_ : 0) : RB_BIG;                              \
                if (with_do) {                                                                                \
                    qlse[row_] = lse_tot[(size_t)bh * T + tok_] * RB_LOG2E;                                   \
                    qdel[row_] = delta[(size_t)bh * T + tok_];                                                \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)

    // ---- per-wave key state -----------------------------------------------------------------------------------
    bf16x8 kf[4], vf[4];
    float ksc = 0.f;
    int kpk = 0, mypos = 0;
    f32x16 dvacc[2], gacc[2];   // [dh tile]: rows = dh, lane = key

#define RB_LOAD_KEYS(slot)                                                                                     \
    do {                                                                                                      \
        const int krow_ = (slot) * BS + wtile * 32 + r;                                                       \
        mypos = kpos[krow_];                                                                                  \
        ksc = kscale[krow_];                                                                                  \
        kpk = kpe[krow_];                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                    \
            kf[ks] = *reinterpret_cast<const bf16x8*>(Kring + krow_ * RB_ROWB + (ks * 16 + 8 * hh) * 2);      \
            vf[ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)mypos * ld + ks * 16 + 8 * hh);         \
        }                                                                                                     \
        dvacc[0] = dvacc[1] = gacc[0] = gacc[1] = (f32x16){0};                                                \
    } while (0)

    // dK = G' - k^ (k^ . G'), dV: write this wave's 32 keys (rows of `dkdst`, `dvdst`)
#define RB_WRITE_KEYS(slot, dkdst, dvdst)                                                                      \
    do {                                                                                                      \
        const int krow_ = (slot) * BS + wtile * 32 + r;                                                       \
        float kv_[2][16];                                                                                     \
        float dot_ = 0.f;                                                                                     \
        const float inv_norm_ = ksc * 8.f;                                                                    \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) _Pragma("unroll") for (int g = 0; g < 4; ++g) {      \
            const uint2 kk_ = *reinterpret_cast<const uint2*>(Kring + krow_ * RB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2); \
            kv_[dt][4 * g] = __uint_as_float(kk_.x << 16) * inv_norm_;                                        \
            kv_[dt][4 * g + 1] = __uint_as_float(kk_.x & 0xffff0000u) * inv_norm_;                            \
            kv_[dt][4 * g + 2] = __uint_as_float(kk_.y << 16) * inv_norm_;                                    \
            kv_[dt][4 * g + 3] = __uint_as_float(kk_.y & 0xffff0000u) * inv_norm_;                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) dot_ = __builtin_fmaf(kv_[dt][4 * g + j], gacc[dt][4 * g + j], dot_); \
        }                                                                                                     \
        dot_ += __shfl_xor(dot_, 32);                                                                         \
        _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) _Pragma("unroll") for (int g = 0; g < 4; ++g) {      \
            uint2 pk_;                                                                                        \
            pk_.x = pack_bf16x2(gacc[dt][4 * g] - kv_[dt][4 * g] * dot_, gacc[dt][4 * g + 1] - kv_[dt][4 * g + 1] * dot_); \
            pk_.y = pack_bf16x2(gacc[dt][4 * g + 2] - kv_[dt][4 * g + 2] * dot_, gacc[dt][4 * g + 3] - kv_[dt][4 * g + 3] * dot_); \
            *reinterpret_cast<uint2*>((dkdst) + dt * 32 + 8 * g + 4 * hh) = pk_;                              \
            pk_.x = pack_bf16x2(dvacc[dt][4 * g], dvacc[dt][4 * g + 1]);                                      \
            pk_.y = pack_bf16x2(dvacc[dt][4 * g + 2], dvacc[dt][4 * g + 3]);                                  \
            *reinterpret_cast<uint2*>((dvdst) + dt * 32 + 8 * g + 4 * hh) = pk_;                              \
        }                                                                                                     \
    } while (0)

    // ---- prologue: halo chunk (previous round's last chunk) into slot 1 / group 1, first chunk into slot 0 ----------
    const int c_first = round * nb;
    const int c_halo = (c_first == 0) ? C - 1 : c_first - 1;
    RB_LOAD_IDX(c_halo);
    RB_LOAD_CHUNK(false);
    RB_LOAD_IDX(c_first);
    RB_STORE_CHUNK(1, false);
    RB_LOAD_CHUNK(true);
    if (nb > 1) RB_LOAD_IDX(c_first + 1);
    RB_STORE_CHUNK(0, true);
    __syncthreads();
    if (grp == 1) RB_LOAD_KEYS(1);

#pragma unroll 1
    for (int i = 0; i < nb; ++i) {
        const int p = i & 1;                 // slot / group of the query (= own) chunk
        const bool own = grp == p;           // wave-uniform
        if (own) RB_LOAD_KEYS(p);
        if (i + 1 < nb) {
            RB_LOAD_CHUNK(true);                                 // rows of chunk i+1: in flight during the compute below
            if (i + 2 < nb) RB_LOAD_IDX(c_first + i + 2);        // and the indices of the one after
        }
        const unsigned char* Qs = Kring + p * BS * RB_ROWB;     // query rows = own chunk's K image
        const int* qpos_a = kpos + p * BS;
        const int* qpe_a = kpe + p * BS;
        const int dsrow = (own ? 0 : BS) + wtile * 32 + r;

#pragma unroll 1
        for (int qt = 0; qt < NQT; ++qt) {
            bf16x8 qf[4], dof[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[ks] = *reinterpret_cast<const bf16x8*>(Qs + (qt * 32 + r) * RB_ROWB + (ks * 16 + 8 * hh) * 2);
                dof[ks] = *reinterpret_cast<const bf16x8*>(Os + (qt * 32 + r) * RB_ROWB + (ks * 16 + 8 * hh) * 2);
            }
            bf16x8 qtf[2][2], dotf[2][2];   // [s2][dh tile]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int qb = qt * 32 + 16 * s2 + 4 * hh + trq;
                    const int col = (dt * 32 + 16 * trc + 4 * trp) * 2;
                    qtf[s2][dt] = rb_tr_frag(Qs + qb * RB_ROWB + col, Qs + (qb + 8) * RB_ROWB + col);
                    dotf[s2][dt] = rb_tr_frag(Os + qb * RB_ROWB + col, Os + (qb + 8) * RB_ROWB + col);
                }
            f32x16 sacc = {0}, pacc = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], sacc, 0, 0, 0);    // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[ks], pacc, 0, 0, 0);   // dP[q][key]
            }
            float pp[16], ds[16];
            const float ksc2 = ksc * RB_LOG2E;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qt * 32 + 8 * g + 4 * hh;
                const float4 l4 = *reinterpret_cast<const float4*>(qlse + q0);
                const float4 d4 = *reinterpret_cast<const float4*>(qdel + q0);
                const int4 p4 = *reinterpret_cast<const int4*>(qpos_a + q0);
                const int4 e4 = *reinterpret_cast<const int4*>(qpe_a + q0);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
                const int pv[4] = {p4.x, p4.y, p4.z, p4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * g + j;
                    const bool self = pv[j] == mypos;
                    const int qpe = (ev[j] == RB_BIG) ? -1 : ev[j];     // an invalid query sees only itself
                    const bool dead = kpk > qpe;
                    float x = sacc[e] * ksc2;
                    x = self ? (-5e4f * RB_LOG2E) : x;
                    float pr = __builtin_amdgcn_exp2f(x - lv[j]);
                    pr = (dead && !self) ? 0.f : pr;
                    pp[e] = pr;
                    ds[e] = self ? 0.f : pr * (pacc[e] - dl[j]) * ksc;
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
                    dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[s2][dt], pb, dvacc[dt], 0, 0, 0);
                    gacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[s2][dt], db, gacc[dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
                pk.y = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
                *reinterpret_cast<uint2*>(Ds + dsrow * DSROW + (qt * 32 + 8 * g + 4 * hh) * 2) = pk;
            }
        }
        __syncthreads();

        // ---- the previous chunk's keys are complete now (halo keys at i == 0: this round's part only) ----------
        if (!own) {
            if (i == 0) {
                const size_t hrow = ((size_t)bh * n_hashes + round) * BS + wtile * 32 + r;
                RB_WRITE_KEYS(1 - p, halo_dk + hrow * RB_DH, halo_dv + hrow * RB_DH);
            } else {
                // the previous chunk belongs to this round: row (round, mypos)
                RB_WRITE_KEYS(1 - p, dk_part + (rt_base + mypos) * RB_DH, dv_part + (rt_base + mypos) * RB_DH);
            }
        }
        // ---- dQ^T[dh][q] = K^T dS'^T over own + previous keys: wave w finishes (query tile w/2, dh half w%2) ----
        {
            const int qt = wave >> 1, dt = wave & 1;
            f32x16 dq = {0};
            const int col = (dt * 32 + 16 * trc + 4 * trp) * 2;
            const int qcol = (qt * 32 + 16 * trc + 4 * trp) * 2;
#pragma unroll 4
            for (int kb = 0; kb < NK; kb += 16) {
                const int keyr = kb + 8 * hh + trq;
                const int slot = (kb < BS) ? p : 1 - p;
                const int krow = slot * BS + (keyr & (BS - 1));
                const bf16x8 bfrag = rb_tr_frag(Ds + keyr * DSROW + qcol, Ds + (keyr + 4) * DSROW + qcol);
                const bf16x8 afrag = rb_tr_frag(Kring + krow * RB_ROWB + col, Kring + (krow + 4) * RB_ROWB + col);
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dq, 0, 0, 0);
            }
            const int qpos = qpos_a[qt * 32 + r];
            bf16_t* dqp = dq_part + (rt_base + qpos) * RB_DH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(dq[4 * g], dq[4 * g + 1]);
                pk.y = pack_bf16x2(dq[4 * g + 2], dq[4 * g + 3]);
                *reinterpret_cast<uint2*>(dqp + dt * 32 + 8 * g + 4 * hh) = pk;
            }
        }
        __syncthreads();                     // everyone is done with slot 1-p, Os and Ds
        if (i + 1 < nb) {
            RB_STORE_CHUNK(1 - p, true);
            __syncthreads();
        }
    }
    // ---- the round's last chunk: own-query contributions only (the next round adds its part through the halo) -------
    {
        const int p = (nb - 1) & 1;
        if (grp == p) RB_WRITE_KEYS(p, dk_part + (rt_base + mypos) * RB_DH, dv_part + (rt_base + mypos) * RB_DH);
    }
}

// dqk[b,t,h] = sum_r dq + dk (+ halo_dk if t sits in the last chunk of round r), dv likewise; 8 lanes per token.head
template <int DUMMY>
__global__ __launch_bounds__(256) void lsh_bwd_reduce_ring_kernel(const bf16_t* __restrict__ dq_part, const bf16_t* __restrict__ dk_part,
                                                                   const bf16_t* __restrict__ dv_part, const bf16_t* __restrict__ halo_dk,
                                                                   const bf16_t* __restrict__ halo_dv, const int32_t* __restrict__ undo,
                                                                   int H, int T, int n_hashes, int BS, size_t rows,
                                                                   bf16_t* __restrict__ dqk, bf16_t* __restrict__ dv, int64_t ld_d) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> 3;
    const int piece = gid & 7;
    if (row >= rows) return;
    const int bh = row / T, t = row % T;
    const int b = bh / H, h = bh % H;
    const int last0 = T - BS;                // first sorted slot of a round's last chunk
    float aq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, av[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add8 = [](float* acc, const bf16_t* p) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __uint_as_float(w[k] << 16);
            acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
    };
    for (int r = 0; r < n_hashes; ++r) {
        const size_t off = (((size_t)bh * n_hashes + r) * T + t) * RB_DH + piece * 8;
        add8(aq, dq_part + off);
        add8(aq, dk_part + off);
        add8(av, dv_part + off);
        const int pos = undo[((size_t)bh * n_hashes + r) * T + t];
        if (pos >= last0) {
            const int rn = (r + 1 == n_hashes) ? 0 : r + 1;
            const size_t hoff = (((size_t)bh * n_hashes + rn) * BS + (pos - last0)) * RB_DH + piece * 8;
            add8(aq, halo_dk + hoff);
            add8(av, halo_dv + hoff);
        }
    }
    uint4 oq, ov;
    oq.x = pack_bf16x2(aq[0], aq[1]); oq.y = pack_bf16x2(aq[2], aq[3]); oq.z = pack_bf16x2(aq[4], aq[5]); oq.w = pack_bf16x2(aq[6], aq[7]);
    ov.x = pack_bf16x2(av[0], av[1]); ov.y = pack_bf16x2(av[2], av[3]); ov.z = pack_bf16x2(av[4], av[5]); ov.w = pack_bf16x2(av[6], av[7]);
    const size_t oo = ((size_t)b * T + t) * ld_d + h * RB_DH + piece * 8;
    *reinterpret_cast<uint4*>(dqk + oo) = oq;
    *reinterpret_cast<uint4*>(dv + oo) = ov;
}

static bool g_rb_attr[2][4];

template <int BS>
static int launch_ring(const bf16_t* qk, const bf16_t* v, int64_t ld, const int32_t* st, const uint8_t* mask, const bf16_t* dout,
                       int64_t ld_do, const float* lse_tot, const float* delta, int B, int H, int T, int n_hashes, int causal,
                       bf16_t* dq_part, bf16_t* dk_part, bf16_t* dv_part, bf16_t* halo_dk, bf16_t* halo_dv, hipStream_t stream) {
    constexpr int NK = 2 * BS;
    const size_t lds = 2 * BS * RB_ROWB + BS * RB_ROWB + NK * (BS * 2 + 16) + 2 * BS * 12 + BS * 8;
    const dim3 grid(B * H * n_hashes), block(BS * 4);
    const int vi = (causal ? 2 : 0) + (mask ? 1 : 0);
#define RB_GO(C_, M_)                                                                                                      \
    do {                                                                                                                   \
        auto kern = lsh_attn_bwd_ring_kernel<BS, C_, M_>;                                                                  \
        if (!g_rb_attr[BS == 128][vi]) {                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            g_rb_attr[BS == 128][vi] = true;                                                                               \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, qk, v, ld, st, mask, dout, ld_do, lse_tot, delta, H, T, n_hashes, \
                           dq_part, dk_part, dv_part, halo_dk, halo_dv);                                                   \
    } while (0)
    if (causal) {
        if (mask) RB_GO(true, true); else RB_GO(true, false);
    } else {
        if (mask) RB_GO(false, true); else RB_GO(false, false);
    }
#undef RB_GO
    RTTS_LAUNCH_CHECK("rtts_lsh_attn_bwd_ring");
    return 0;
}

extern "C" int rtts_lsh_attn_bwd_ring(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                                      const void* dout, int64_t ld_dout, const float* lse_tot, const float* delta, int B, int H, int T,
                                      int dh, int n_hashes, int bucket_size, int causal, void* dq_part, void* dk_part, void* dv_part,
                                      void* halo_dk, void* halo_dv, void* stream) {
    RTTS_REQUIRE(qk && v && st && dout && lse_tot && delta && dq_part && dk_part && dv_part && halo_dk && halo_dv,
                 "rtts_lsh_attn_bwd_ring: null pointer");
    RTTS_REQUIRE(dh == RB_DH, "rtts_lsh_attn_bwd_ring: dh=%d unsupported (this build: 64)", dh);
    RTTS_REQUIRE(bucket_size == 64 || bucket_size == 128, "rtts_lsh_attn_bwd_ring: bucket_size=%d unsupported (64 or 128)", bucket_size);
    RTTS_REQUIRE(T > 0 && T % (2 * bucket_size) == 0,
                 "rtts_lsh_attn_bwd_ring: Sequence length (%d) needs to be divisible by target bucket size x 2 - %d", T, 2 * bucket_size);
    RTTS_REQUIRE(B > 0 && H > 0 && n_hashes > 0, "rtts_lsh_attn_bwd_ring: bad B/H/n_hashes");
    RTTS_REQUIRE(ld >= (int64_t)H * dh && ld % 8 == 0 && ld_dout >= (int64_t)H * dh && ld_dout % 8 == 0,
                 "rtts_lsh_attn_bwd_ring: row strides must be >= H*dh and multiples of 8");
    RTTS_REQUIRE((((uintptr_t)qk | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq_part | (uintptr_t)dk_part | (uintptr_t)dv_part |
                   (uintptr_t)halo_dk | (uintptr_t)halo_dv) & 15) == 0, "rtts_lsh_attn_bwd_ring: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (bucket_size == 64)
        return launch_ring<64>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta, B, H, T,
                               n_hashes, causal, (bf16_t*)dq_part, (bf16_t*)dk_part, (bf16_t*)dv_part, (bf16_t*)halo_dk, (bf16_t*)halo_dv, s);
    return launch_ring<128>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta, B, H, T,
                            n_hashes, causal, (bf16_t*)dq_part, (bf16_t*)dk_part, (bf16_t*)dv_part, (bf16_t*)halo_dk, (bf16_t*)halo_dv, s);
}

extern "C" int rtts_lsh_bwd_reduce_ring(const void* dq_part, const void* dk_part, const void* dv_part, const void* halo_dk,
                                        const void* halo_dv, const int32_t* undo, int B, int H, int T, int dh, int n_hashes,
                                        int bucket_size, void* dqk, void* dv, int64_t ld_d, void* stream) {
    RTTS_REQUIRE(dq_part && dk_part && dv_part && halo_dk && halo_dv && undo && dqk && dv, "rtts_lsh_bwd_reduce_ring: null pointer");
    RTTS_REQUIRE(dh == RB_DH && B > 0 && H > 0 && T > 0 && n_hashes > 0 && bucket_size > 0 && T % bucket_size == 0,
                 "rtts_lsh_bwd_reduce_ring: bad shape");
    RTTS_REQUIRE(ld_d >= (int64_t)H * dh && ld_d % 8 == 0, "rtts_lsh_bwd_reduce_ring: bad stride");
    const size_t rows = (size_t)B * H * T;
    const unsigned grid = (unsigned)((rows * 8 + 255) / 256);
    hipLaunchKernelGGL(lsh_bwd_reduce_ring_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dq_part,
                       (const bf16_t*)dk_part, (const bf16_t*)dv_part, (const bf16_t*)halo_dk, (const bf16_t*)halo_dv, undo, H, T, n_hashes,
                       bucket_size, rows, (bf16_t*)dqk, (bf16_t*)dv, ld_d);
    RTTS_LAUNCH_CHECK("rtts_lsh_bwd_reduce_ring");
    return 0;
}
