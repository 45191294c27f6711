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
#define AB_ROWB 144

typedef __attribute__((ext_vector_type(8))) short short8v;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTTS_LDS short4v*)p1);
    const short8v both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

template <int BS, bool CAUSAL, bool MASKED>
__global__ __launch_bounds__(BS * 4, 2) void lsh_attn_bwd_kernel(
    const bf16_t* __restrict__ qk, const bf16_t* __restrict__ v, int64_t ld, const int32_t* __restrict__ st,
    const uint8_t* __restrict__ mask, const bf16_t* __restrict__ dout, int64_t ld_do, const float* __restrict__ lse_tot,
    const float* __restrict__ delta, int H, int T, int n_hashes, bf16_t* __restrict__ dqk_part, bf16_t* __restrict__ dv_part,
    size_t slot_stride) {
    constexpr int NK = 2 * BS;
    constexpr int NQT = BS / 32;
    constexpr int NTHR = BS * 4;
    constexpr int KT2 = 1;                // 32-key tiles owned by one wave
    constexpr int DSROW = BS * 2 + 16;   // bytes per row of the dS^T image [key][query]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;                                  // [NK][144]  qk rows (own chunk first)
    unsigned char* Os = Ks + NK * AB_ROWB;                     // [BS][144]  dout rows of the queries
    unsigned char* Ds = Os + BS * AB_ROWB;                     // [NK][DSROW] dS'^T
    float* kscale = reinterpret_cast<float*>(Ds + NK * DSROW);
    int* kpos = reinterpret_cast<int*>(kscale + NK);
    int* kpe = kpos + NK;                                      // effective position: dead <=> kpe[key] > qpe[query]
    float* qlse = reinterpret_cast<float*>(kpe + NK);          // lse_tot * log2(e)
    float* qdel = qlse + BS;
    int* qpe_s = reinterpret_cast<int*>(qdel + BS);            // query-side effective position (-1: an invalid query)

    const int nb = T / BS;
    const int C = n_hashes * nb;
    const uint32_t wi = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = wi / C, c = wi % C;
    const int b = bh / H, h = bh % H;
    const int cprev = (c == 0) ? C - 1 : c - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    const int32_t* st_row = st + (size_t)bh * n_hashes * T;
    const bf16_t* qbase = qk + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* vbase = v + (size_t)b * T * ld + (size_t)h * AB_DH;
    const bf16_t* dobase = dout + (size_t)b * T * ld_do + (size_t)h * AB_DH;

    // ---- gather K rows (all 2*BS) and dout rows (own chunk) into LDS -----------------------
    constexpr int ITERS = NK * 8 / NTHR;   // 4
    int trow[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = (it * NTHR + tid) >> 3;
        const int slot = (row < BS) ? c * BS + row : cprev * BS + (row - BS);
        trow[it] = st_row[slot];
    }
    uint4 kreg[ITERS], oreg[ITERS / 2];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) kreg[it] = *reinterpret_cast<const uint4*>(qbase + (size_t)trow[it] * ld + (tid & 7) * 8);
#pragma unroll
    for (int it = 0; it < ITERS / 2; ++it)   // rows < BS are the first half of the iterations
        oreg[it] = *reinterpret_cast<const uint4*>(dobase + (size_t)trow[it] * ld_do + (tid & 7) * 8);
    // V fragments of this wave's keys go straight to registers (no other wave needs them)
    int myrow[KT2];
    bf16x8 vf[KT2][4];
    int mypos[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
        myrow[k2] = wave * (32 * KT2) + 32 * k2 + r;
        const int row = myrow[k2];
        const int slot = (row < BS) ? c * BS + row : cprev * BS + (row - BS);
        mypos[k2] = st_row[slot];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            vf[k2][ks] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)mypos[k2] * ld + ks * 16 + 8 * hh);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = (it * NTHR + tid) >> 3, piece = tid & 7;
        *reinterpret_cast<uint4*>(Ks + row * AB_ROWB + piece * 16) = kreg[it];
        if (it < ITERS / 2) *reinterpret_cast<uint4*>(Os + row * AB_ROWB + piece * 16) = oreg[it];
        const uint32_t u[4] = {kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w};
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = __uint_as_float(u[k] << 16), bq = __uint_as_float(u[k] & 0xffff0000u);
            ss = __builtin_fmaf(a, a, ss);
            ss = __builtin_fmaf(bq, bq, ss);
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        if (piece == 0) {
            kscale[row] = 0.125f * __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // dh^-1/2 / max(|k|, 1e-12)
            kpos[row] = trow[it];
            const int valid = MASKED ? (int)mask[(size_t)b * T + trow[it]] : 1;
            kpe[row] = valid ? (CAUSAL ? trow[it] : 0) : 0x40000000;
            if (row < BS) {
                qlse[row] = lse_tot[(size_t)bh * T + trow[it]] * 1.4426950408889634f;
                qdel[row] = delta[(size_t)bh * T + trow[it]];
                // an invalid query sees nothing but itself: its effective position is below every key's
                qpe_s[row] = valid ? (CAUSAL ? trow[it] : 0) : -1;
            }
        }
    }
    __syncthreads();

    // ---- this wave's key-side constants -----------------------------------------------------
    bf16x8 kf[KT2][4];
    float ksc[KT2];
    int kpk[KT2];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            kf[k2][ks] = *reinterpret_cast<const bf16x8*>(Ks + myrow[k2] * AB_ROWB + (ks * 16 + 8 * hh) * 2);
        ksc[k2] = kscale[myrow[k2]];
        kpk[k2] = kpe[myrow[k2]];
    }

    f32x16 dvacc[KT2][2], gacc[KT2][2];   // [key tile][dh tile]: rows = dh, lane = key
#pragma unroll
    for (int a = 0; a < KT2; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            dvacc[a][d] = (f32x16){0};
            gacc[a][d] = (f32x16){0};
        }
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
    const bool own_tile = myrow[0] < BS;            // wave-uniform
    const bool wrap = (cprev / nb) != (c / nb);

#pragma unroll 1
    for (int qt = 0; qt < NQT; ++qt) {
        bf16x8 qf[4], dof[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(Ks + (qt * 32 + r) * AB_ROWB + (ks * 16 + 8 * hh) * 2);
            dof[ks] = *reinterpret_cast<const bf16x8*>(Os + (qt * 32 + r) * AB_ROWB + (ks * 16 + 8 * hh) * 2);
        }
        // q-side row constants of this tile, issued first so that their LDS latency hides behind the MFMAs below
        float4 l4[4], d4[4];
        int4 e4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = qt * 32 + 8 * g + 4 * hh;
            l4[g] = *reinterpret_cast<const float4*>(qlse + q0);
            d4[g] = *reinterpret_cast<const float4*>(qdel + q0);
            e4[g] = *reinterpret_cast<const int4*>(qpe_s + q0);
        }
        // Can a key of this wave's tile BE one of this tile's queries (the self logit)?  Own keys: only on the diagonal
        // tile.  Looked-back keys: only when the previous chunk belongs to another hash round (the chunk ring wraps
        // over rounds, so the same token can then sit in both chunks).  Wave-uniform: the common path skips the test.
        const bool chk_self = own_tile ? (wave == qt) : wrap;
#pragma unroll
        for (int k2 = 0; k2 < KT2; ++k2) {
            f32x16 sacc = {0}, pacc = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[k2][ks], sacc, 0, 0, 0);    // S[q][key]
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vf[k2][ks], pacc, 0, 0, 0);   // dP[q][key]
            }
            // P' = exp2(s*ksc*log2e - lse*log2e); dS' = P' (dP - delta) ksc  (0 at the self logit: it was a constant).
            // ksc multiplies dS' once here: G' = dS'^T Q then gives dK = G' - k^ (k^ . G'), and dQ^T = K^T dS'^T.
            float pp[16], ds[16];
            const float ksc2 = ksc[k2] * 1.4426950408889634f;
            if (chk_self) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w}, dv_[4] = {d4[g].x, d4[g].y, d4[g].z, d4[g].w};
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
                        ds[i] = self ? 0.f : p * (pacc[i] - dv_[j]) * ksc[k2];
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float lv[4] = {l4[g].x, l4[g].y, l4[g].z, l4[g].w}, dv_[4] = {d4[g].x, d4[g].y, d4[g].z, d4[g].w};
                    const int ev[4] = {e4[g].x, e4[g].y, e4[g].z, e4[g].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 4 * g + j;
                        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[i], ksc2, -lv[j]));
                        p = (kpk[k2] > ev[j]) ? 0.f : p;
                        pp[i] = p;
                        ds[i] = p * (pacc[i] - dv_[j]) * ksc[k2];
                    }
                }
            }
            // A fragments of the transposed products (element j <-> query 16*s2 + 8*(j>>2) + 4*hh + (j&3)): read only now,
            // so that their registers are free during the softmax arithmetic above
            bf16x8 qtf[2][2], dotf[2][2];   // [s2][dh tile]
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int qb = qt * 32 + 16 * s2 + 4 * hh + trq;
                    const int col = (dt * 32 + 16 * trc + 4 * trp) * 2;
                    qtf[s2][dt] = tr_frag(Ks + qb * AB_ROWB + col, Ks + (qb + 8) * AB_ROWB + col);
                    dotf[s2][dt] = tr_frag(Os + qb * AB_ROWB + col, Os + (qb + 8) * AB_ROWB + col);
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
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(ds[4 * g], ds[4 * g + 1]);
                pk.y = pack_bf16x2(ds[4 * g + 2], ds[4 * g + 3]);
                *reinterpret_cast<uint2*>(Ds + myrow[k2] * DSROW + (qt * 32 + 8 * g + 4 * hh) * 2) = pk;
            }
        }
    }

    __syncthreads();   // every dS'^T tile is in Ds; nobody reads Os as dout any more

    // ---- dQ^T[dh][q] = K^T dS'^T over all 2*BS keys: wave w finishes (query tile w/2, dh half w%2) and parks it
    //      (bf16) in the dout image's rows: a chunk row is both a query and an own key, so its query-role and
    //      key-role gradients are added before they leave the chip
    {
        const int qt = wave >> 1, dt = wave & 1;
        f32x16 dq = {0};
        const int col = (dt * 32 + 16 * trc + 4 * trp) * 2;
        const int qcol = (qt * 32 + 16 * trc + 4 * trp) * 2;
#pragma unroll 4
        for (int kb = 0; kb < NK; kb += 16) {
            const int keyr = kb + 8 * hh + trq;
            const bf16x8 bfrag = tr_frag(Ds + keyr * DSROW + qcol, Ds + (keyr + 4) * DSROW + qcol);
            const bf16x8 afrag = tr_frag(Ks + keyr * AB_ROWB + col, Ks + (keyr + 4) * AB_ROWB + col);
            dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, dq, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack_bf16x2(dq[4 * g], dq[4 * g + 1]);
            pk.y = pack_bf16x2(dq[4 * g + 2], dq[4 * g + 3]);
            *reinterpret_cast<uint2*>(Os + (qt * 32 + r) * AB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2) = pk;
        }
    }
    __syncthreads();   // dQ parked; Ds is free: it becomes the per-wave staging of the row stores below

    // ---- key-side outputs: dK (+ dQ on own rows) and dV of this wave's 32 keys.  The accumulators hold a key per
    //      lane and dh down the registers; each tile goes through a [32][144 B] LDS staging so that a row leaves as
    //      eight 16-byte pieces (full 128-byte lines) instead of sixteen scattered 8-byte stores.
    const int round = c / nb, round_prev = cprev / nb;
    static_assert(KT2 == 1, "one key tile per wave");
    {
        const bool own = myrow[0] < BS;   // wave-uniform
        const size_t obase = ((size_t)bh * n_hashes + (own ? round : round_prev)) * T;
        bf16_t* dkdst = dqk_part + (own ? 0 : slot_stride);
        bf16_t* dvdst = dv_part + (own ? 0 : slot_stride);
        unsigned char* stg = Ds + wave * (32 * AB_ROWB);
        // k^ . G over the 64 dh (this lane holds 32 of them, the partner half the other 32)
        float kv_[2][16];
        float dot = 0.f;
        const float inv_norm = ksc[0] * 8.f;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 kk = *reinterpret_cast<const uint2*>(Ks + myrow[0] * AB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2);
                kv_[dt][4 * g] = __uint_as_float(kk.x << 16) * inv_norm;
                kv_[dt][4 * g + 1] = __uint_as_float(kk.x & 0xffff0000u) * inv_norm;
                kv_[dt][4 * g + 2] = __uint_as_float(kk.y << 16) * inv_norm;
                kv_[dt][4 * g + 3] = __uint_as_float(kk.y & 0xffff0000u) * inv_norm;
#pragma unroll
                for (int j = 0; j < 4; ++j) dot = __builtin_fmaf(kv_[dt][4 * g + j], gacc[0][dt][4 * g + j], dot);
            }
        dot += __shfl_xor(dot, 32);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float dk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) dk[j] = gacc[0][dt][4 * g + j] - kv_[dt][4 * g + j] * dot;
                if (own) {
                    const uint2 dqv = *reinterpret_cast<const uint2*>(Os + myrow[0] * AB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2);
                    dk[0] += __uint_as_float(dqv.x << 16);
                    dk[1] += __uint_as_float(dqv.x & 0xffff0000u);
                    dk[2] += __uint_as_float(dqv.y << 16);
                    dk[3] += __uint_as_float(dqv.y & 0xffff0000u);
                }
                uint2 pk;
                pk.x = pack_bf16x2(dk[0], dk[1]);
                pk.y = pack_bf16x2(dk[2], dk[3]);
                *reinterpret_cast<uint2*>(stg + r * AB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        const int srow = lane >> 3, spiece = lane & 7;
        uint4 rowv[4];
        int rpos[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rowv[i] = *reinterpret_cast<const uint4*>(stg + (i * 8 + srow) * AB_ROWB + spiece * 16);
            rpos[i] = kpos[wave * 32 + i * 8 + srow];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(dkdst + (obase + rpos[i]) * AB_DH + spiece * 8) = rowv[i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 pk;
                pk.x = pack_bf16x2(dvacc[0][dt][4 * g], dvacc[0][dt][4 * g + 1]);
                pk.y = pack_bf16x2(dvacc[0][dt][4 * g + 2], dvacc[0][dt][4 * g + 3]);
                *reinterpret_cast<uint2*>(stg + r * AB_ROWB + (dt * 32 + 8 * g + 4 * hh) * 2) = pk;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) rowv[i] = *reinterpret_cast<const uint4*>(stg + (i * 8 + srow) * AB_ROWB + spiece * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(dvdst + (obase + rpos[i]) * AB_DH + spiece * 8) = rowv[i];
    }
}

static bool g_bwd_attr_set[2][4];

template <int BS>
static int launch_attn_bwd(const bf16_t* qk, const bf16_t* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                           const bf16_t* dout, int64_t ld_do, const float* lse_tot, const float* delta, int B, int H, int T,
                           int n_hashes, int causal, bf16_t* dqk_part, bf16_t* dv_part, hipStream_t stream) {
    constexpr int NK = 2 * BS;
    const size_t lds = NK * AB_ROWB + BS * AB_ROWB + NK * (BS * 2 + 16) + NK * 12 + BS * 12;
    const dim3 grid(B * H * n_hashes * (T / BS)), block(BS * 4);
    const size_t slot_stride = (size_t)B * H * n_hashes * T * AB_DH;
    const int vi = (causal ? 2 : 0) + (mask ? 1 : 0);
#define AB_GO(C_, M_)                                                                                                      \
    do {                                                                                                                   \
        auto kern = lsh_attn_bwd_kernel<BS, C_, M_>;                                                                       \
        if (!g_bwd_attr_set[BS == 128][vi]) {                                                                              \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            g_bwd_attr_set[BS == 128][vi] = true;                                                                          \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, qk, v, ld, st, mask, dout, ld_do, lse_tot, delta, H, T, n_hashes, \
                           dqk_part, dv_part, slot_stride);                                                                \
    } while (0)
    if (causal) {
        if (mask) AB_GO(true, true); else AB_GO(true, false);
    } else {
        if (mask) AB_GO(false, true); else AB_GO(false, false);
    }
#undef AB_GO
    RTTS_LAUNCH_CHECK("rtts_lsh_attn_bwd");
    return 0;
}

extern "C" int rtts_lsh_attn_bwd(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                                 const void* dout, int64_t ld_dout, const float* lse_tot, const float* delta, int B, int H,
                                 int T, int dh, int n_hashes, int bucket_size, int causal, void* dqk_part, void* dv_part,
                                 void* stream) {
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
    hipStream_t s = (hipStream_t)stream;
    if (bucket_size == 64)
        return launch_attn_bwd<64>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta,
                                   B, H, T, n_hashes, causal, (bf16_t*)dqk_part, (bf16_t*)dv_part, s);
    return launch_attn_bwd<128>((const bf16_t*)qk, (const bf16_t*)v, ld, st, mask, (const bf16_t*)dout, ld_dout, lse_tot, delta, B,
                                H, T, n_hashes, causal, (bf16_t*)dqk_part, (bf16_t*)dv_part, s);
}
