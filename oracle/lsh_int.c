/* C twin of the integer stages of LSH attention (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * Restates hash_vectors + the (bucket, position) sort of the reference's LSH layer
 * (reformer_pytorch 0.19.1 via /root/reference/reformer_tts/model/reformer.py:217;
 * SURVEY.md Appendix B steps 2-3) with the accumulation order fixed, so that the HIP kernel
 * can be compared BIT-EXACTLY: the projection is a k-ordered chain of fused multiply-adds
 * (fmaf), argmax takes the first maximum of [xR, -xR].
 *
 * Parity: unpinned against reformer_pytorch itself (absent); pinned against
 * oracle/lsh_ref.py (same algorithm in PyTorch) and HuggingFace's fixtures by tests/. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* qk: (BH, T, dh) f32; rot: (rot_rows, dh, R, half) f32; buckets: (BH, R, T) i32 incl. round offset */
void oracle_lsh_hash(const float* qk, const float* rot, int rot_rows, int BH, int T, int dh, int R, int half,
                     int32_t* buckets) {
    for (int bh = 0; bh < BH; ++bh) {
        const float* rb = rot + (size_t)(rot_rows == 1 ? 0 : bh) * dh * R * half;
        for (int r = 0; r < R; ++r)
            for (int t = 0; t < T; ++t) {
                const float* q = qk + ((size_t)bh * T + t) * dh;
                float best = 0.f;
                int idx = 0;
                for (int pass = 0; pass < 2; ++pass)
                    for (int i = 0; i < half; ++i) {
                        float acc = 0.f;
                        for (int f = 0; f < dh; ++f) acc = fmaf(q[f], rb[((size_t)f * R + r) * half + i], acc);
                        const float val = pass ? -acc : acc;
                        if ((pass == 0 && i == 0) || val > best) { best = val; idx = pass * half + i; }
                    }
                buckets[((size_t)bh * R + r) * T + t] = idx + r * 2 * half;
            }
    }
}

/* stable sort of each round by (bucket, t): st[slot] = t, undo[t] = slot (both per round) */
void oracle_lsh_sort(const int32_t* buckets, int BH, int T, int R, int n_buckets, int32_t* st, int32_t* undo) {
    int* cnt = (int*)malloc(sizeof(int) * (n_buckets + 1));
    for (size_t seg = 0; seg < (size_t)BH * R; ++seg) {
        const int r = (int)(seg % R);
        const int32_t* b = buckets + seg * T;
        for (int k = 0; k <= n_buckets; ++k) cnt[k] = 0;
        for (int t = 0; t < T; ++t) cnt[b[t] - r * n_buckets + 1]++;
        for (int k = 0; k < n_buckets; ++k) cnt[k + 1] += cnt[k];
        for (int t = 0; t < T; ++t) {
            const int pos = cnt[b[t] - r * n_buckets]++;
            st[seg * T + pos] = t;
            undo[seg * T + t] = pos;
        }
    }
    free(cnt);
}
