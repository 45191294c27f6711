// LSH bucket assignment + stable counting sort, one workgroup per (batch*head, hash round).
//
// Replaces hash_vectors / sort_key_val of the reference's LSH layer (reformer_pytorch 0.19.1,
// reached from reformer_tts/model/reformer.py:217).
//
// HBM-bound integer path: reads each qk row (128 B) once per round (rounds of one head run
// on the same XCD, so 7 of the 8 reads are L2 hits), writes 4 B (+8 B optional) per token.
// The projection is a k-ordered fp32 fmaf chain, so bucket ids are bit-identical to
// oracle/lsh_int.c; the sort is a stable counting sort (keys are unique => permutation unique).
#include "rtts_common.h"

#define HS_THREADS 256
#define HS_WAVES 4
#define HS_DH 64
#define HS_ROWB 144   // LDS row stride in bytes for a staged 64 x 64 bf16 tile (conflict-free b128 reads)

template <int HALF>
__device__ __forceinline__ int hash_row(const float* q, const float* rot_lds) {
    // rot_lds[f * HALF + i]; returns argmax over [xR, -xR] with the first maximum winning
    float acc[HALF];
#pragma unroll
    for (int i = 0; i < HALF; ++i) acc[i] = 0.f;
#pragma unroll
    for (int f = 0; f < HS_DH; ++f) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) acc[i] = __builtin_fmaf(q[f], rot_lds[f * HALF + i], acc[i]);
    }
    float best = acc[0];
    int idx = 0;
#pragma unroll
    for (int i = 1; i < HALF; ++i)
        if (acc[i] > best) { best = acc[i]; idx = i; }
#pragma unroll
    for (int i = 0; i < HALF; ++i)
        if (-acc[i] > best) { best = -acc[i]; idx = HALF + i; }
    return idx;
}

template <int HALF>
__global__ __launch_bounds__(HS_THREADS) void lsh_hash_sort_kernel(
    const bf16_t* __restrict__ qk, int64_t ld, const float* __restrict__ rotations, int rot_rows,
    int H, int T, int n_hashes, int32_t* __restrict__ buckets, int32_t* __restrict__ st, int32_t* __restrict__ undo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve: rot [64*HALF] f32 | tile [4 waves][64 rows][144 B] | bkt [T] u16 | cntw [4][64] i32 | tot [64] i32
    float* rot_lds = reinterpret_cast<float*>(smem);
    unsigned char* tile = smem + ((HS_DH * HALF * 4 + 15) & ~15);
    uint16_t* bkt = reinterpret_cast<uint16_t*>(tile + HS_WAVES * 64 * HS_ROWB);
    int* cntw = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(bkt) + ((T * 2 + 15) & ~15));
    int* tot = cntw + HS_WAVES * 64;

    constexpr int NB = 2 * HALF;
    // work item: round r of head bh; rounds of one bh are 8 ids apart => same XCD (L2 reuse of qk)
    const uint32_t nblk = gridDim.x;
    const uint32_t w = xcd_remap(blockIdx.x, nblk);
    const int bh = w / n_hashes, r = w % n_hashes;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    const float* rot_src = rotations + (size_t)(rot_rows == 1 ? 0 : bh) * HS_DH * n_hashes * HALF;
    for (int i = tid; i < HS_DH * HALF; i += HS_THREADS) {
        const int f = i / HALF, k = i % HALF;
        rot_lds[i] = rot_src[((size_t)f * n_hashes + r) * HALF + k];
    }
    __syncthreads();

    // ---- hash: each wave stages 64 rows (coalesced 16-B pieces), then one lane hashes one row
    const bf16_t* base = qk + (size_t)b * T * ld + (size_t)h * HS_DH;
    unsigned char* wt = tile + wave * 64 * HS_ROWB;
    for (int t0 = wave * 64; t0 < T; t0 += HS_WAVES * 64) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = p * 8 + (lane >> 3), piece = lane & 7;
            const uint4 val = *reinterpret_cast<const uint4*>(base + (size_t)(t0 + row) * ld + piece * 8);
            *reinterpret_cast<uint4*>(wt + row * HS_ROWB + piece * 16) = val;
        }
        __builtin_amdgcn_wave_barrier();   // same wave wrote and reads: LDS ops of one wave execute in order
        float q[HS_DH];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint4 val = *reinterpret_cast<const uint4*>(wt + lane * HS_ROWB + p * 16);
            const uint32_t u[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[p * 8 + 2 * k] = __uint_as_float(u[k] << 16);
                q[p * 8 + 2 * k + 1] = __uint_as_float(u[k] & 0xffff0000u);
            }
        }
        const int idx = hash_row<HALF>(q, rot_lds);
        bkt[t0 + lane] = (uint16_t)idx;
        if (buckets) buckets[((size_t)bh * n_hashes + r) * T + t0 + lane] = idx + r * NB;
    }
    __syncthreads();

    // ---- stable counting sort.  Wave w owns the contiguous token segment [w*T/4, (w+1)*T/4);
    // lane k of every wave keeps the counter of bucket k (NB <= 64).
    const int seg = T / HS_WAVES;           // multiple of 32 because T % 128 == 0
    const int s0 = wave * seg;
    int cnt = 0;
    for (int t0 = 0; t0 < seg; t0 += 64) {
        const bool act = t0 + lane < seg;
        const int mb = act ? (int)bkt[s0 + t0 + lane] : -1;
        for (int k = 0; k < NB; ++k) {
            const unsigned long long m = __ballot(mb == k);
            if (lane == k) cnt += __popcll(m);
        }
    }
    if (lane < NB) cntw[wave * 64 + lane] = cnt;
    __syncthreads();
    if (tid < NB) tot[tid] = cntw[tid] + cntw[64 + tid] + cntw[128 + tid] + cntw[192 + tid];
    __syncthreads();
    int basek = 0;   // first sorted slot of (bucket = lane, this wave's segment)
    if (lane < NB) {
        for (int k = 0; k < lane; ++k) basek += tot[k];
        for (int w2 = 0; w2 < wave; ++w2) basek += cntw[w2 * 64 + lane];
    }
    int32_t* st_out = st + ((size_t)bh * n_hashes + r) * T;
    int32_t* undo_out = undo ? undo + ((size_t)bh * n_hashes + r) * T : nullptr;
    for (int t0 = 0; t0 < seg; t0 += 64) {
        const bool act = t0 + lane < seg;
        const int mb = act ? (int)bkt[s0 + t0 + lane] : -1;
        int pos = 0;
        for (int k = 0; k < NB; ++k) {
            const unsigned long long m = __ballot(mb == k);
            const int bk = __shfl(basek, k);
            if (mb == k) pos = bk + __popcll(m & ((1ull << lane) - 1ull));
            if (lane == k) basek += __popcll(m);
        }
        if (act) {
            const int t = s0 + t0 + lane;
            st_out[pos] = t;
            if (undo_out) undo_out[t] = pos;
        }
    }
}

template <int HALF>
static int launch_hash_sort(const bf16_t* qk, int64_t ld, const float* rot, int rot_rows, int B, int H, int T,
                            int n_hashes, int32_t* buckets, int32_t* st, int32_t* undo, hipStream_t stream) {
    const size_t lds = ((HS_DH * HALF * 4 + 15) & ~15) + HS_WAVES * 64 * HS_ROWB + ((T * 2 + 15) & ~15) + (HS_WAVES * 64 + 64) * 4;
    const dim3 grid(B * H * n_hashes);
    hipLaunchKernelGGL(lsh_hash_sort_kernel<HALF>, grid, dim3(HS_THREADS), lds, stream, qk, ld, rot, rot_rows, H, T,
                       n_hashes, buckets, st, undo);
    RTTS_LAUNCH_CHECK("rtts_lsh_hash_sort");
    return 0;
}

extern "C" int rtts_lsh_hash_sort(const void* qk, int64_t ld_qk, const float* rotations, int rot_rows, int B, int H,
                                  int T, int dh, int n_hashes, int bucket_size, int32_t* buckets, int32_t* st,
                                  int32_t* undo, void* stream) {
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
    switch (half) {
        case 1: return launch_hash_sort<1>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        case 2: return launch_hash_sort<2>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        case 4: return launch_hash_sort<4>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        case 8: return launch_hash_sort<8>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        case 16: return launch_hash_sort<16>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        case 32: return launch_hash_sort<32>(q, ld_qk, rotations, rot_rows, B, H, T, n_hashes, buckets, st, undo, s);
        default: break;
    }
    rtts_set_error("rtts_lsh_hash_sort: n_buckets=%d unsupported (need a power of two in [2,64])", 2 * half);
    return -1;
}
