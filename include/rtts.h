/*
 * rtts.h -- C ABI of librtts_hip.so, the MI355X (gfx950) implementation of the
 * Reformer-TTS training hot path.
 *
 * The reference (kowaalczyk/reformer-tts) is pure Python: it has no FFI of its
 * own.  The seam this library plugs into is the `implementation` switch of
 * LSHSelfAttentionWrapper (reference reformer_tts/model/reformer.py:189-220)
 * and the reversible-block protocol (reformer_tts/model/reversible.py:46-98,
 * 134-203); each entry point below names the reference lines whose arithmetic
 * it replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the library never allocates, frees or synchronises: outputs and
 *     workspaces are caller-owned, launches go to the `stream` argument
 *     (a hipStream_t passed as void*), so calls are graph-capturable;
 *   - DEVICE of a call: the stream's.  A forward is called from the Python main
 *     thread, a backward from PyTorch's autograd worker thread for the device
 *     (nested inside the reversible function's backward), and HIP's current
 *     device is per thread -- so every launching entry point first binds the
 *     calling thread to the device its `stream` belongs to (hipStreamGetDevice +
 *     hipSetDevice when it differs): the explicit device ordinal SURVEY.md 8(b)
 *     asks for travels inside the stream handle and cannot disagree with it.
 *     A NULL stream (the legacy default stream) means the thread's current
 *     device.  Every pointer of a call must belong to that device; return -3 =
 *     the stream handle is not a stream of this process;
 *   - `bf16` buffers hold IEEE bfloat16 (uint16_t storage);
 *   - activations of width d = H*dh are row-major with an explicit row stride
 *     `ld` in ELEMENTS (so qk and v may be the two halves of one (B*T, 2d) GEMM
 *     output); head h of token (b,t) starts at element (b*T+t)*ld + h*dh;
 *   - return value 0 = launched, <0 = rejected before any launch
 *     (rtts_last_error() gives the reason, thread-local);
 *   - shapes supported by this build: dh == 64, bucket_size in {64,128},
 *     T % (2*bucket_size) == 0, T <= 8192, n_hashes <= 16.
 */
#ifndef RTTS_H
#define RTTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTTS_VERSION 1

int rtts_version(void);
const char* rtts_last_error(void);

/* ---- LSH attention: integer stages -------------------------------------------------
 * Replaces hash_vectors + sort of reformer_pytorch.LSHAttention as called from
 * reformer_tts/model/reformer.py:217.
 *   qk        bf16 (B,T,H*dh) with row stride ld_qk
 *   rotations f32  (rot_rows, dh, n_hashes, n_buckets/2), rot_rows = 1 (shared) or B*H
 *   buckets   i32  (B*H, n_hashes, T)  optional (NULL to skip): bucket id incl. round offset
 *   st        i32  (B*H, n_hashes, T)  sorted slot -> token position ("sticker % T")
 *   undo      i32  (B*H, n_hashes, T)  optional: token -> sorted slot within its round
 * Projection is a k-ordered fp32 fmaf chain (bit-reproducible, see oracle/lsh_int.c). */
int rtts_lsh_hash_sort(const void* qk, int64_t ld_qk, const float* rotations, int rot_rows,
                       int B, int H, int T, int dh, int n_hashes, int bucket_size,
                       int32_t* buckets, int32_t* st, int32_t* undo, void* stream);
/* how the call is worked: 2 = lsh_hash_rounds_kernel (the projections of ALL rounds through the f32 MFMA, every row staged
 * once; bucket ids pass through `st`) + lsh_sort_ids_kernel, from T = 1024 on; 1 = lsh_hash_sort_kernel (one launch, a
 * workgroup per (head, round)) for short sequences.  Same results bit for bit. */
int rtts_lsh_hash_sort_launches(int T);

/* ---- LSH attention: chunked attention forward (Appendix B steps 4-10) ----------------
 *   qk, v  bf16 rows as above (common stride ld)
 *   mask   u8 (B,T) 1 = valid token, or NULL
 *   o      bf16 (B*H, n_hashes, T, dh)  per-round output, already at UNSORTED positions
 *   lse    f32  (B*H, n_hashes, T)      per-round logsumexp of the masked logits */
int rtts_lsh_attn_fwd(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                      int B, int H, int T, int dh, int n_hashes, int bucket_size, int causal,
                      void* o, float* lse, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev, void* stream);
/* drop_p > 0: dropout on the attention probabilities (the layer's `dropout` knob, reference reformer_tts/model/config.py:27):
 * the chunk's softmax output is multiplied by keep = 0 | 1/(1-p) before it meets the values; lse is that of the undropped
 * probabilities.  keep is a counter hash of (drop_seed + *seed_dev, pair), pair = ((head * chunks + chunk) * bucket_size +
 * query row of the chunk) * 2*bucket_size + key row (own chunk's keys first): no mask is stored, rtts_lsh_attn_bwd called with
 * the same (drop_p, drop_seed, seed_dev) -- and the reversible recompute's forward -- redraw the same one.  seed_dev may be NULL. */
/* how rtts_lsh_attn_fwd works a shape: 0 = one workgroup per chunk (lsh_attn_fwd_kernel; small grids), R > 0 = workgroups
 * that walk runs of R consecutive chunks of a (batch, head) ring (lsh_attn_fwd_walk_kernel: every K / V row gathered once,
 * the next chunk's rows fetched while the current one is merged and stored; same results bit for bit).  -1: bad arguments.
 * rtts_debug_set_walk() below forces a run length for tests and A/B runs; the launch path reads no environment. */
int rtts_lsh_attn_fwd_run_length(int B, int H, int T, int n_hashes, int bucket_size);
/* TEST-ONLY, process-wide (two atomic words): run length of the walking forward / backward kernels.  -1 = the library's own
 * pick (the default: what every product call gets), 0 = the one-chunk kernel, n >= 1 = runs of n where n divides the ring
 * (else the one-chunk kernel).  Replaces the RTTS_LSH_{FWD,BWD}_WALK environment variables the launch path used to read on
 * every call; the Python binding reads those variables ONCE, when it loads the library, and calls this. */
int rtts_debug_set_walk(int fwd_run, int bwd_run);

/* ---- combine the hash rounds (step 11) and merge heads (first half of step 12) -------
 *   out     bf16 (B,T,H*dh) row stride ld_out
 *   lse_tot f32  (B*H, T)  logsumexp over rounds (kept for the backward) */
int rtts_lsh_combine_fwd(const void* o, const float* lse, int B, int H, int T, int dh, int n_hashes,
                         void* out, int64_t ld_out, float* lse_tot, void* stream);

/* ---- backward of steps 4-11 ------------------------------------------------------------
 * The rounds form ONE softmax over the multiset of (round, key) pairs, so the
 * backward needs only lse_tot and delta = rowsum(out * dout) per (token, head):
 *   rtts_lsh_bwd_delta : delta f32 (B*H,T) from out, dout (bf16, strides ld_out, ld_dout)
 *   rtts_lsh_attn_bwd  : per chunk, writes bf16 partial gradients at unsorted positions
 *        dqk_part : (rtts_lsh_bwd_qk_slots() = 2, B*H, n_hashes, T, dh)  slot 0 = the token as a row of its own
 *                   chunk (query role + key role, added on chip), slot 1 = key role as the looked-back chunk
 *        dv_part  : (2, B*H, n_hashes, T, dh)  slot 0 = own chunk, slot 1 = looked-back chunk
 *        every (slot, head, round, token) row is written exactly once: no zero-fill needed
 *        row_flags : u8 (B*H, n_hashes, T), required where rtts_lsh_attn_bwd_run_length() > 0 (else it may be NULL and is
 *                   not written): the walking kernel carries a chunk's dK/dV accumulators through the two steps that work
 *                   its keys and writes each key row ONCE into slot 0; only the ends of a run leave a slot-1 partner, and
 *                   row_flags[row] = 1 marks the slot-0 rows that have one (slot-1 rows without a flag are never written)
 *   rtts_lsh_bwd_reduce: dqk, dv bf16 (B,T,H*dh) stride ld_d = sum over slots and rounds; row_flags as written by the
 *        backward, or NULL = every row has both slots */
int rtts_lsh_bwd_qk_slots(void);
/* how rtts_lsh_attn_bwd works a shape: 0 = one workgroup per chunk (lsh_attn_bwd_kernel), R > 0 = workgroups that walk R
 * consecutive chunks of a ring with the next chunk's rows prefetched (lsh_attn_bwd_walk_kernel); same results bit for bit */
int rtts_lsh_attn_bwd_run_length(int B, int H, int T, int n_hashes, int bucket_size);
int rtts_lsh_bwd_delta(const void* out, int64_t ld_out, const void* dout, int64_t ld_dout,
                       int B, int H, int T, int dh, float* delta, void* stream);
int rtts_lsh_attn_bwd(const void* qk, const void* v, int64_t ld, const int32_t* st, const uint8_t* mask,
                      const void* dout, int64_t ld_dout, const float* lse_tot, const float* delta,
                      int B, int H, int T, int dh, int n_hashes, int bucket_size, int causal,
                      void* dqk_part, void* dv_part, uint8_t* row_flags, float drop_p, uint32_t drop_seed, const uint32_t* seed_dev,
                      void* stream);
int rtts_lsh_bwd_reduce(const void* dqk_part, const void* dv_part, int B, int H, int T, int dh, int n_hashes,
                        void* dqk, void* dv, int64_t ld_d, const uint8_t* row_flags, void* stream);

/* ---- optimizer step over the flat parameter buffer --------------------------------------
 * Replaces clip_grad_norm_ (pytorch-lightning gradient_clip_val, reference
 * reformer_tts/training/train.py:77-89) and transformers.optimization.AdamW.step as configured
 * at reformer_tts/training/wrappers.py:240-256,284-297 (beta 0.9/0.999, eps 1e-6, bias-corrected,
 * decoupled decay applied after the update to parameters whose decay_mask byte is 1).
 *   rtts_grad_clip_scale: scale_out[0] = grad_mult * min(1, max_norm / (grad_mult*|g| + 1e-6)),
 *        scale_out[1] = grad_mult*|g|;  partial_ws: >= 2048 floats.  grad_mult = 1/world_size
 *        after a sum all-reduce.  max_norm <= 0 disables clipping.
 *   rtts_adamw_step: grads are multiplied by scale[0] on the fly (scale may be NULL = 1);
 *        hyper (device) = {lr, lr*sqrt(1-beta2^t)/(1-beta1^t)} for the current step t, read by the kernel so that
 *        a captured hipGraph replays with the values the host wrote before the replay.  n % 4 == 0 (pad the flat
 *        buffers); bf16_mirror (may be NULL): the updated parameters rounded to bf16, written in the same pass. */
int rtts_grad_clip_scale(const float* grads, int64_t n, float grad_mult, float max_norm, float* partial_ws,
                         float* scale_out, void* stream);
int rtts_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                    int64_t n, const float* scale, const float* hyper, float beta1, float beta2, float eps,
                    float weight_decay, void* bf16_mirror, void* stream);

/* ---- row-wise fused kernels around the GEMMs of a reversible block --------------------------
 * Replace the ATen chains of WithNorm / FeedForward / residual adds (reference
 * reformer_tts/model/reformer.py:25-45, modules.py:195-207, reversible.py:56-98) and their
 * autograd backward.  Rows are contiguous (stride d); d in {128,256,384,512,768,1024,2048}.
 * Column sums (bias / LayerNorm-affine gradients) are ACCUMULATED into their outputs through
 * `partial_ws` (>= 2*256*d floats), deterministically.
 *   rtts_ln_fwd            xn(bf16) = LayerNorm(x; eps 1e-5) * gamma + beta; keeps mean, rstd (M)
 *   rtts_ln_bwd            dx_io += dLN(dxn); dgamma += ..., dbeta += ...
 *   rtts_cast_colsum       dyb(bf16) = dy(fp32); dbias += colsum(dy)   (dbias may be NULL); drop_p > 0: dy is first
 *                          multiplied by the keep-scale of the dropout that sat on the block's output
 *   rtts_colsum_bf16       dbias += colsum(dh); relu_gate: dh *= (h > 0) * gate_scale in place first (gate_scale = 1/(1-p)
 *                          when h went through dropout_p after the ReLU)
 *   rtts_residual_epilogue y = x + sign * dropout_p(g(bf16) + bias)    (bias may be NULL; drop_p = post_attn_dropout of
 *                          the LSH layer, reformer.py:198-200 knob list; mask = hash(seed + *seed_dev, element), the same
 *                          in the forward, the reconstruction and the backward)
 *   rtts_bias_act          h(bf16) = [relu](h + bias) in place
 *   rtts_cast_f32_bf16     flat cast (n % 4 == 0): the per-step bf16 mirror of all parameters */
int rtts_ln_fwd(const float* x, const float* gamma, const float* beta, void* xn, float* mean, float* rstd,
                int M, int d, void* stream);
int rtts_ln_bwd(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma,
                float* dx_io, float* dgamma, float* dbeta, float* partial_ws, int M, int d,
                void* dyb_next, float* partial_next, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* out-of-place form: dx_out = dx_in + dLN(dxn) (dx_in == dx_out is rtts_ln_bwd).  The stack executor uses it for the first update
 * of each of the two gradient streams of a reversible stack's backward (reference: reversible.py:114-129,155-158 start from two
 * chunks of one dy), which start as the caller's dout itself instead of a copy of it. */
int rtts_ln_bwd_to(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                   float* dx_out, float* dgamma, float* dbeta, float* partial_ws, int M, int d,
                   void* dyb_next, float* partial_next, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* the joining form: dx_out = dx_in + addend + dLN(dxn) (addend NULL: rtts_ln_bwd_to; must not alias dx_out).  The stack executor's LAST
 * LayerNorm backward: both streams of a reversible stack start as its input (reference reformer.py:85-93, x = cat([x, x])), so
 * d(input) = g1 + g2 -- the other stream joins in the pass that completes this one. */
int rtts_ln_bwd_join(const void* dxn, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dx_in,
                     const float* addend, float* dx_out, float* dgamma, float* dbeta, float* partial_ws, int M, int d,
                     void* dyb_next, float* partial_next, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* dyb_next (may be NULL): bf16 copy of the completed dx_io, times the keep-scale of (drop_p, seed) -- the next block's
 * rtts_cast_colsum folded in; partial_next then receives that copy's partial column sums (same layout as partial_ws). */
int rtts_cast_colsum(const float* dy, void* dyb, float* dbias, float* partial_ws, int M, int d, float drop_p, uint32_t seed,
                     const uint32_t* seed_dev, const float* scale_dev, void* stream);
/* scale_dev (may be NULL): one device float multiplied into dy first -- the upstream gradient of a scalar loss, so that the
 * root of a backward needs no separate scaling pass.  gated_out (may be NULL = in place): where the gated dh goes. */
int rtts_colsum_bf16(void* dh, const void* h, int64_t ld, float* dbias, float* partial_ws, int M, int d,
                     int relu_gate, float gate_scale, void* gated_out, void* stream);
/* Deferred finalisation of the column sums: rtts_ln_bwd (dgamma = dbeta = NULL), rtts_cast_colsum / rtts_colsum_bf16
 * (dbias = NULL) then only write their partial rows -- rtts_colsum_partial_rows(M) rows of d floats; rtts_ln_bwd writes two
 * such blocks, the second 256*d floats after the first -- and ONE grouped launch adds the sums of up to
 * RTTS_COLSUM_MAX_GROUP partial buffers into their outputs (same fixed summation order: deterministic). */
#define RTTS_COLSUM_MAX_GROUP 48
typedef struct {
    const float* partial;   /* (nrows, n) fp32, row stride ld */
    float* out;             /* (n): += column sums */
    int32_t nrows, n;
    int32_t ld, reserved;   /* ld = 0: n (a block of columns of a wider partial buffer: ld = its width) */
} rtts_colsum_job;
int rtts_colsum_partial_rows(int M);
int rtts_colsum_final_grouped(const rtts_colsum_job* jobs, int n, void* stream);
int rtts_residual_epilogue(const float* x, const void* g, const float* bias, float sign, float* y,
                           int64_t M, int d, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* The same epilogue fused with the NEXT block's LayerNorm: y = x + sign*(g + bias) (y = NULL: in place), then
 * xn(bf16) = LayerNorm(y)*gamma + beta with mean/rstd per row (reformer.py:25-33 applied to the stream a
 * reversible block has just updated or reconstructed, reversible.py:56-98). */
int rtts_residual_ln(float* x, const void* g, const float* bias, float sign, const float* gamma, const float* beta,
                     void* xn, float* mean, float* rstd, int M, int d, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                     float* y, void* stream);
/* out = a + b (the two streams of a reversible stack); out (fp32) and / or out_bf16 may be NULL; n % 4 == 0 */
int rtts_sum_streams(const float* a, const float* b, int64_t n, float* out, void* out_bf16, void* stream);
int rtts_bias_act(void* h, const float* bias, int64_t M, int d, int relu, void* stream);
int rtts_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);

/* ---- dense encoder-decoder attention (T_k a multiple of 128, 128 or 256 keys on chip at a time) ----
 * Replaces the attention core of nn.MultiheadAttention inside MultiheadAttentionWrapper
 * (reference reformer_tts/model/reformer.py:161-186); projections stay GEMMs outside.
 *   q   bf16 (B,Tq,H*dh) stride ld_q;  kv bf16 (B,Tk,2*H*dh) = [k | v] stride ld_kv
 *   kvalid u8 (B,Tk) 1 = attend (i.e. NOT key_padding_mask), or NULL
 *   o   bf16 (B,Tq,H*dh) stride ld_o;  lse f32 (B*H,Tq)
 *   backward: delta f32 (B*H,Tq) = rowsum(o*do) (rtts_lsh_bwd_delta), dq bf16 like q,
 *   dkv_part bf16 (Tq/128, B, Tk, 2*H*dh) partial slabs -> rtts_sum_slabs -> dkv (B,Tk,2*H*dh)
 *   T_k > 256 (or 384, 640, ...): the forward walks the keys in rtts_xattn_key_chunks(T_k) chunks with a running maximum and
 *   normaliser; the backward gives every chunk its own workgroups (P = exp(s - lse) needs only the forward's lse), which
 *   write their shares of dQ to dq_chunks (chunks, B*Tq, H*dh) bf16 -- summed into dq (ld_dq == H*dh) by the same call;
 *   one chunk: dq_chunks may be NULL */
int rtts_xattn_key_chunks(int Tk);
int rtts_xattn_fwd(const void* q, int64_t ld_q, const void* kv, int64_t ld_kv, const uint8_t* kvalid,
                   int B, int H, int Tq, int Tk, int dh, void* o, int64_t ld_o, float* lse,
                   float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
int rtts_xattn_bwd(const void* q, int64_t ld_q, const void* kv, int64_t ld_kv, const uint8_t* kvalid,
                   const void* dout, int64_t ld_dout, const float* lse, const float* delta,
                   int B, int H, int Tq, int Tk, int dh, void* dq, int64_t ld_dq, void* dkv_part,
                   float drop_p, uint32_t seed, const uint32_t* seed_dev, void* dq_chunks, void* stream);
/* drop_p = nn.MultiheadAttention's dropout on the normalised attention probabilities (config #4: 0.15); the keep
 * decision of (head, query, key) is hash(seed + *seed_dev, index), identical in the forward and the backward. */
int rtts_sum_slabs(const void* part, int nslabs, int64_t n, void* out, void* stream);

/* ---- convolutional edges (prenet / postnet) and the loss ------------------------------------------
 * Reference reformer_tts/model/modules.py:8-61 (EncoderPreNet), :103-169 (PostConvNet); loss.py:28-53.
 * HALO ROWS.  The activations of a convolution stack are channels-last rows (B, L + 2*halo, C), halo = 2, with `halo` zero
 * rows in front of and behind every sequence; a halo array may have a lead-in of `lead` rows before halo row 0 and any
 * number of rows behind row B*(L+2*halo) - 1 -- all zero.  Row m (counted from halo row 0) belongs to sequence m / (L+4)
 * at position m % (L+4) - 2.  A Conv1d(k=5, pad=2) is then the IMPLICIT GEMM rtts_conv1d_k5: tap k reads the same rows
 * shifted by k - 2, the halo supplies the zero padding, nothing like an im2col matrix is written; the transposed product
 * (input gradient) reads the output gradient shifted by 2 - k, and the weight gradient is five rtts_gemm_tn problems on
 * row-shifted views.  halo = 0 everywhere below means plain (B*L) rows.
 *   rtts_conv1d_k5      y (M, C_out) = conv(x) on halo rows: M rows starting at halo row 0 (any multiple of the GEMM's row
 *                       tile; x must be readable 2 rows before and after).  transposed = 0: wp (C_out, 5*C_in) bf16 from
 *                       rtts_conv_w_perm; transposed = 1: dx = conv^T(dy), wp (C_in, 5*C_out) is the SAME array seen as
 *                       [channel of dy][tap][channel of dx].  out_f32: unrounded fp32 result (+ optional bias), else bf16.
 *                       C_in % 64 == 0, (M, C_out) must tile like rtts_gemm_nt.
 *   rtts_to_halo        dst (bf16, `rows` rows of C channels, halo row 0 at row `lead`) = the first C_src channels of src
 *                       (plain (B*L, ld_src) rows, fp32 or bf16); zero in channels >= C_src and everywhere outside the valid
 *                       set (halo = 0: a cast of plain rows with the channels zero-padded to the GEMM's K granule)
 *   rtts_conv_w_perm    wp[co][k][ci] (bf16, ci < CP zero padded) = w[co][ci][k] (fp32 master layout of nn.Conv1d)
 *   rtts_conv_dw_unperm dw[co][ci][k] += dwp[co][k][ci]
 *   rtts_bn_stats       per-channel batch mean / rstd (eps 1e-5) over the B*L valid rows of y (fp32; halo or plain rows);
 *                       optional running-stat update (momentum 0.1, unbiased variance; mean_shift = the conv bias that was
 *                       left out of y)
 *   rtts_bn_act_fwd     z(bf16) = dropout_p(act(gamma*(y-mean)*rstd + beta)); act 1 = ReLU, 2 = tanh; z_halo = 1: z is a halo
 *                       array (z_rows rows, lead-in z_lead), zero outside the valid set; z_halo = 0: B*L plain rows
 *   rtts_bn_act_bwd     dy(bf16, y's layout: dy_rows rows, lead-in dy_lead, zero outside the valid set) = BatchNorm(train)
 *                       backward through act and dropout; dz in halo rows (dz_halo = 1, no lead-in) or plain rows;
 *                       dgamma, dbeta accumulate
 *   rtts_tts_loss       losses[4] = {total, raw, post, stop} and d_raw, d_post (rows, NM; row stride ld_grad >= NM,
 *                       the pad columns are written as zero), d_stop (rows):
 *                       masked MSE (kind 0) / L1 (kind 1) means over ALL elements + BCE-with-logits(pos_weight);
 *                       predictions / gradients have rows = batch * padded_len (the decoder's length, a multiple of
 *                       pad_base), targets batch * valid_len rows (reformer_tts.py:141-143 crops the predictions): rows
 *                       t >= valid_len get zero gradient and do not count.  res != NULL: the postnet prediction is
 *                       raw + res (reformer_tts.py:139-140) with res fp32 in halo rows (no lead-in, stride ld_res), `post`
 *                       is ignored and d_post is written in halo rows (dpost_rows rows, lead-in dpost_lead, zero outside
 *                       the valid set); res == NULL: halo = dpost_lead = 0, d_post in plain rows
 *   rtts_heads_grad     dheads (B*L, width) fp32 = d_raw + d_post (halo rows, lead-in dpost_lead) + dx0 (halo rows, no
 *                       lead-in; columns < n_mels), column n_mels = d_stop: the gradient of the [mel | stop] heads
 * partial_ws: >= (2*256 + 2)*C floats (bn) / 1536 floats (loss). */
int rtts_conv1d_k5(const void* x, int64_t ldx, const void* wp, int64_t ldw, int transposed, int M, int C_out, int C_in,
                   void* y, int64_t ldy, const float* bias, int out_f32, void* stream);
int rtts_to_halo(const void* src, int64_t ld_src, int64_t src_batch_stride, int C_src, int src_f32, int B, int L, int halo, int C,
                 void* dst, int lead, int64_t rows, void* stream);
/* src_batch_stride: elements between two samples of src (0 = L * ld_src; larger for a time-sliced view such as frames [0, L-1)) */
int rtts_conv_w_perm(const float* w, int Co, int Ci, int CP, void* wp, void* stream);
int rtts_conv_dw_unperm(const float* dwp, int Co, int Ci, int CP, float* dw, void* stream);
/* All convolution weights of a step re-laid-out in ONE launch (they only change in the optimizer step, and a launch costs
 * ~5 us whatever its size): jobs[i] is one rtts_conv_w_perm problem. */
#define RTTS_CONV_PERM_MAX_GROUP 8
typedef struct {
    const float* w;    /* (Co, Ci, 5) fp32 master */
    void* wp;          /* (>= Co, 5*CP) bf16 */
    int32_t Co, Ci, CP, reserved;
} rtts_conv_perm_job;
int rtts_conv_w_perm_grouped(const rtts_conv_perm_job* jobs, int n, void* stream);
/* The adjoint for the gradients, likewise grouped: jobs[i].w = dwp (Co_pad, 5*CP) fp32, jobs[i].wp = dw (Co, Ci, 5) fp32,
 * dw[co][ci][k] += dwp[co][k][ci]. */
int rtts_conv_dw_unperm_grouped(const rtts_conv_perm_job* jobs, int n, void* stream);
/* The convolution in front of a BatchNorm (reference modules.py:19-54,103-169: Conv1d -> BatchNorm1d): y (M, C_out) fp32 = the forward
 * of rtts_conv1d_k5 (no bias: BatchNorm cancels it) AND, from the same accumulators, the per-channel sums of y and y^2 over the rows that
 * carry data, as rtts_gemm_nt_partial_rows(M, C_out) partial rows of [sum y | sum y^2] (2 C_out floats each) in `partial`.
 * rtts_bn_stats_from_partials finishes them into mean / rstd / running statistics exactly as rtts_bn_stats does (which re-reads y). */
int rtts_conv1d_k5_moments(const void* x, int64_t ldx, const void* wp, int64_t ldw, int M, int C_out, int C_in, void* y, int64_t ldy,
                           int B, int L, int halo, float* partial, void* stream);
int rtts_bn_stats_from_partials(const float* partial, int nrows, int B, int L, int C, float* mean, float* rstd, float* run_mean,
                                float* run_var, const float* mean_shift, int64_t* num_batches, void* stream);
int rtts_bn_stats(const float* y, int B, int L, int halo, int C, float* mean, float* rstd, float* run_mean, float* run_var,
                  const float* mean_shift, int64_t* num_batches, float* partial_ws, void* stream);
int rtts_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                    float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C, void* z, int z_halo,
                    int z_lead, int64_t z_rows, void* stream);
int rtts_bn_act_bwd(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C,
                    void* dy, int dy_lead, int64_t dy_rows, float* dgamma, float* dbeta, float* partial_ws, void* stream);
/* Data-parallel BatchNorm (SyncBN; SURVEY.md 8(e), reference modules.py:29,127 normalise over the whole batch): the two
 * stages of rtts_bn_stats / rtts_bn_act_bwd as separate calls, so that the caller can all-reduce the 2*C floats between them.
 *   rtts_bn_moments        moments[0..C) = sum y, [C..2C) = sum y^2 over this rank's B*L valid rows
 *   rtts_bn_from_moments   mean / rstd (eps 1e-5; running statistics as rtts_bn_stats) from summed moments over `count` rows
 *   rtts_bn_act_bwd_sums   sums[0..C) = sum g, [C..2C) = sum g*yhat (g = dz through act and dropout); dgamma, dbeta accumulate
 *                          the LOCAL sums (the gradient all-reduce adds the ranks)
 *   rtts_bn_act_bwd_apply  dy from (all-reduced) sums over `count` rows
 * count = 0 in rtts_bn_from_moments / rtts_bn_act_bwd_apply: the row count is read from element [2*C] of moments / sums -- the
 * caller stores its own B*L there before the all-reduce, so ranks with different batch shapes need no second collective. */
int rtts_bn_moments(const float* y, int B, int L, int halo, int C, float* moments, float* partial_ws, void* stream);
int rtts_bn_from_moments(const float* moments, int64_t count, int C, float* mean, float* rstd, float* run_mean, float* run_var,
                         const float* mean_shift, int64_t* num_batches, void* stream);
int rtts_bn_act_bwd_sums(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C,
                         float* sums, float* dgamma, float* dbeta, float* partial_ws, void* stream);
int rtts_bn_act_bwd_apply(const float* y, const void* dz, int dz_halo, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, float drop_p, uint32_t seed, const uint32_t* seed_dev, int B, int L, int halo, int C,
                          const float* sums, int64_t count, void* dy, int dy_lead, int64_t dy_rows, void* stream);
int rtts_tts_loss(const float* raw, const float* post, int64_t ld_mel, const float* tgt, const float* mask, const float* stop,
                  int64_t ld_stop, const float* tstop, int rows, int NM, int kind, float pos_weight, float w_raw, float w_post,
                  float w_stop, float* d_raw, float* d_post, int64_t ld_grad, float* d_stop, float* losses, float* partial_ws,
                  int padded_len, int valid_len, const float* res, int64_t ld_res, int halo, int dpost_lead, int64_t dpost_rows,
                  int64_t tgt_batch_stride, const int32_t* valid_len_dev, int64_t mask_batch_stride, int64_t tstop_batch_stride,
                  void* stream);
/* tgt_batch_stride: elements between two samples of tgt (0 = valid_len * NM; larger for the view frames [1, L) of the batch).
 * valid_len_dev (may be NULL): the loss length as a device word, 1 <= *valid_len_dev <= valid_len -- a replayed hipGraph whose
 * batch buffers are padded to a fixed length serves batches of any length up to it; valid_len is then the time length of
 * the tgt / mask / tstop buffers only.  mask_batch_stride / tstop_batch_stride: elements between two samples of mask / tstop
 * (0 = valid_len * NM / valid_len).
 * rtts_heads_grad scale_dev (may be NULL): the upstream gradient of the total loss, multiplied into d_raw, d_post, d_stop. */
int rtts_heads_grad(const float* d_raw, const float* d_post, int dpost_lead, const float* dx0, int64_t ld_dx0, const float* d_stop,
                    int B, int L, int halo, int n_mels, int width, float* dheads, const float* scale_dev, void* stream);

/* Up to RTTS_SEGMENTS_MAX flat copies / additions in one launch (per-step refreshes of padded GEMM operands, additions of
 * padded gradient blocks into their parameters' gradients). */
#define RTTS_SEGMENTS_MAX 12
enum { RTTS_SEG_COPY_F32 = 0, RTTS_SEG_COPY_BF16 = 1, RTTS_SEG_ADD_F32 = 2, RTTS_SEG_CAST_F32_BF16 = 3 };
typedef struct {
    void* dst;
    const void* src;
    int64_t count;          /* elements */
    int32_t kind, reserved;
} rtts_segment;
int rtts_segments(const rtts_segment* jobs, int n, void* stream);

/* Scaled positional encoding (reference modules.py:172-192): out = y + alpha * dropout_p(table[t]), the mask shared over the
 * batch; dalpha += sum dy * dropout_p(table).  relu_drop: h = dropout_p(relu(h)) in place (decoder prenet, modules.py:82-100). */
int rtts_pe_add(const void* y, const float* table, const float* alpha, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                int T, int64_t M, int d, float* out, void* stream);
int rtts_pe_dalpha(const float* dy, const float* table, float drop_p, uint32_t seed, const uint32_t* seed_dev, int T, int64_t M,
                   int d, float* dalpha, float* partial_ws, void* stream);
int rtts_relu_drop(void* h, float drop_p, uint32_t seed, const uint32_t* seed_dev, int64_t n, void* stream);

/* What ReformerTTS.forward derives from a batch before its first layer (reference reformer_tts.py:119-125, wrappers.py:60), one
 * launch: pad_phonemes (B, Lp_pad) = phonemes right-padded with 0; phoneme_mask = pad_phonemes != 0 and its inverse (the
 * cross-attention's key_padding_mask); frame_mask (B, Lm_pad) = (mean over the n_mels columns of loss_mask != 0), 0 behind Lm.
 * Masks are one byte per element (torch.bool storage).  Strides in elements. */
int rtts_batch_masks(const int64_t* phonemes, int64_t ph_stride, int B, int Lp, int Lp_pad, const float* loss_mask, int64_t lm_bstride,
                     int64_t lm_rstride, int Lm, int Lm_pad, int n_mels, int64_t* pad_phonemes, uint8_t* phoneme_mask,
                     uint8_t* phoneme_pad_mask, uint8_t* frame_mask, void* stream);

/* nn.Embedding + the Dropout behind it (reference modules.py:17,22,56): out (rows, C) fp32 = dropout_p(E[ids]); backward:
 * dE[id] += sum of (dx * the same keep-scales) over the rows whose id matches; padding_idx skipped.  dE accumulates: it may be
 * the parameter's gradient itself. */
int rtts_embedding_fwd(const int64_t* ids, const float* E, int rows, int C, int n_embeddings, float drop_p, uint32_t seed,
                       const uint32_t* seed_dev, float* out, void* stream);
int rtts_embedding_bwd(const int64_t* ids, const float* dx, int rows, int C, int n_embeddings, int padding_idx, float* dE,
                       float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
/* the same with dx given as a (B, L, C) VIEW of a larger array (the convolution stack's gradient on its halo rows): batch stride and row
 * stride in floats, rows = B * L */
int rtts_embedding_bwd_strided(const int64_t* ids, const float* dx, int64_t batch_stride, int64_t row_stride, int L, int rows, int C,
                               int n_embeddings, int padding_idx, float* dE, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);

/* ---- weight-gradient GEMM, split over the token dimension ------------------------------------
 * c[N][K] (fp32, stride ldc) (+)= sum_m a[m][N] * b[m][K]   (a, b bf16 with strides lda, ldb)
 * = dW of a Linear layer y = x W^T (reference modules.py:195-207, reformer.py:161-217 via autograd).
 * N % 128 == 0, K % 128 == 0, M % 64 == 0.  slab_ws (>= N*K*16 floats for full split) holds the
 * per-split partial tiles; they are summed in a fixed order by a second small launch (deterministic).
 * (A single-launch variant in which the last workgroup of a tile folds the slabs was measured 6x slower:
 * the agent-scope release fence every workgroup needs writes back a whole XCD L2.)  accumulate=1: += . */
int rtts_gemm_tn(const void* a, int64_t lda, const void* b, int64_t ldb, int M, int N, int K, float* c, int64_t ldc,
                 int accumulate, float* slab_ws, int64_t slab_ws_floats, void* stream);

/* Up to RTTS_GEMM_TN_MAX_GROUP independent weight gradients in ONE pair of launches (the deferred gradients of a
 * reversible layer): the grid holds every problem's tiles, so the split factor -- and with it the slab traffic --
 * drops to what the whole group needs to fill the chip.  Same arithmetic per problem as rtts_gemm_tn. */
#define RTTS_GEMM_TN_MAX_GROUP 16
typedef struct {
    const void* a; int64_t lda;    /* dY (M x N) bf16 */
    const void* b; int64_t ldb;    /* X  (M x K) bf16 */
    float* c;      int64_t ldc;    /* dW (N x K) fp32 */
    int32_t M, N, K, accumulate;
} rtts_gemm_tn_problem;
int rtts_gemm_tn_grouped(const rtts_gemm_tn_problem* problems, int n, float* slab_ws, int64_t slab_ws_floats, void* stream);

/* ---- forward / input-gradient GEMM of the stacks' Linear layers ---------------------------------
 * c[M][N] (bf16, stride ldc) = epilogue( a[M][K] (bf16, stride lda) x W ), fp32 accumulation:
 *   w_is_kn = 0: W = w[N][K] (stride ldw)  -> y = x W^T, the forward of nn.Linear: toqk / tov / to_out of the LSH layer
 *                (reformer.py:198-217 via reformer_pytorch), in_proj / out_proj of nn.MultiheadAttention
 *                (reformer.py:161-186), FeedForward net.0 / net.3 (modules.py:195-207)
 *   w_is_kn = 1: W = w[K][N] (stride ldw)  -> dx = dy W, the input gradient autograd derives for the same lines
 * epilogue: 0 none; 1 + bias[N] (fp32); 2 relu(+ bias) (FeedForward's Linear -> ReLU, modules.py:200-201);
 *           3 ReLU gate: c = acc * (gate[M][N] > 0) (bf16, stride ldg) -- the backward of that ReLU -- and, if
 *             colsum_partial != NULL, fp32 partial column sums of c: rtts_gemm_nt_partial_rows(M, N) rows of N floats
 *             (the hidden bias gradient; sum them with rtts_colsum_final_grouped).
 * K % 64 == 0; (M, N) must tile by one of 192x128, 256x128, 96x64, 128x64; a, w 16-byte aligned, lda, ldw % 8 == 0.
 * Replaces the reference's ATen/cuBLAS calls for these layers; parity vs oracle/model_ref.py in tests/test_gemm_hip.py. */
int rtts_gemm_nt(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, int M, int N, int K, void* c,
                 int64_t ldc, const float* bias, int epilogue, const void* gate, int64_t ldg, float* colsum_partial,
                 void* stream);
int rtts_gemm_nt_partial_rows(int M, int N);
/* Grouped / extended form of rtts_gemm_nt (round 4).  n = 2..4 INDEPENDENT problems of one weight layout and one epilogue run as
 * ONE launch: one grid over all their tiles, a workgroup finds its problem by tile range -- the cross-attention's q and k|v
 * projections of one layer (reformer.py:161-186: two separate in_proj products inside nn.MultiheadAttention) and their
 * input-gradient pair (dxn = dq Wq beside dkeys += dkv Wkv) pay one launch ramp, tail and kernel boundary instead of two.
 * n = 1: one problem, with the epilogues rtts_gemm_nt has no arguments for:
 *   epilogue 4 + accumulate: c is fp32 (rows of ldc floats), c += result (+ bias) -- the gradient of the encoder keys summed over
 *                the decoder layers in the GEMM that produces each layer's share (no separate add launch);
 *   epilogue 5: [K][N] weights only -- the input gradient of to_out / out_proj (c = dout, bf16) AND delta[b*H + h][t] =
 *                sum over the 64 columns of head h of aux[m][.] * dout[m][.] (aux = the attention output, bf16 (M, N) stride
 *                ld_aux; rows m = b*T + t; delta f32 (B*H, T)), i.e. rtts_lsh_bwd_delta folded into the product that makes dout.
 * Epilogues 0, 1, 4 in a group (each problem its own); 0, 1, 4, 5 alone.  Shapes: as rtts_gemm_nt (a group needs one tile shape that tiles every
 * problem: 192x128, 96x64 or 128x64).  Tested in tests/test_gemm_hip.py against float64 on the same bf16 operands. */
typedef struct {
    const void* a; int64_t lda;
    const void* w; int64_t ldw;
    void* c; int64_t ldc;
    const float* bias;
    const void* aux; int64_t ld_aux;
    float* aux_out;
    int32_t M, N, K, epilogue;
    int32_t T, H, accumulate, reserved;
} rtts_gemm_nt_problem;
#define RTTS_GEMM_NT_MAX_GROUP 4
int rtts_gemm_nt_grouped(const rtts_gemm_nt_problem* problems, int n, int w_is_kn, void* stream);
/* TEST / A-B ONLY, process-wide: launch form of rtts_gemm_nt problems with several tiles per CU.  0 = the library's pick,
 * 1 = one tile per workgroup (round 3), 2 = persistent workgroups on a 2-deep ring (two per CU), 3 = persistent on the deepest
 * ring (one per CU).  10 / 11 / 12: store form of the bf16 epilogues (8-byte stores / 16-byte stores after a lane-pair exchange /
 * 16-byte write-through stores), 9: the library's pick again.  Results are identical bit for bit in every form. */
int rtts_debug_set_gemm_mode(int mode);
/* FeedForward pair with a 1-bit ReLU gate: the forward GEMM (epilogue 2: bias + ReLU) also writes one 64-bit word per
 * lane and tile -- bit b set <=> the b-th output of that lane is a positive bf16 -- and the input-gradient GEMM of the same
 * (M, N) (epilogue 3, [K][N] weight) reads the words instead of re-reading the (M, N) activation (50 MB at the baseline
 * shapes).  Both calls get the same tile shape for the same (M, N), hence the same lane -> element map; gate_words holds
 * rtts_gemm_nt_gate_words(M, N) words (0: the tile chosen for this shape has no word form -- use rtts_gemm_nt with the
 * activation as `gate`). */
int64_t rtts_gemm_nt_gate_words(int M, int N);
int rtts_gemm_nt_gated(const void* a, int64_t lda, const void* w, int64_t ldw, int w_is_kn, int M, int N, int K, void* c,
                       int64_t ldc, const float* bias, int epilogue, uint64_t* gate_words, float* colsum_partial, void* stream);

/* ---- on-box peak probes (bench.py; not on the training path) --------------------------------------
 * rtts_peak_copy: dst = src, float4 stream copy of `bytes` (multiple of 16) -> achieved HBM rate = 2*bytes / time.
 * rtts_peak_mfma: `workgroups` x 8 waves each issue iters x 8 back-to-back v_mfma_f32_16x16x32_bf16 on random register
 *                 operands -> dense bf16 matrix rate = workgroups*8*iters*8*16384 FLOP / time. */
int rtts_peak_copy(const void* src, void* dst, int64_t bytes, void* stream);
/* Probe, not on the training path: a ring all-reduce as the CUs see it -- `workgroups` resident workgroups (8-32) copy `bytes`
 * (+1 on every word) in 16 KB pieces with write-through stores and sleep `sleep` x ~0.5 us between pieces (pace of a link, not of
 * HBM).  scripts/comm_overlap_probe.py runs it beside the data-parallel chain of hipGraphs. */
int rtts_comm_probe(const void* src, void* dst, int64_t bytes, int workgroups, int sleep, void* stream);
/* TEST / MEASUREMENT ONLY: a one-lane kernel writes the device's 100 MHz wall clock into buf[slot] (uint64).  Capturable: markers inside
 * a replayed hipGraph give the timeline of an UNPROFILED step (scripts/replay_stamps.py). */
int rtts_debug_stamp(void* buf, int slot, void* stream);
int rtts_peak_mfma(float* sink, int workgroups, int iters, void* stream);

/* ---- SqueezeWave vocoder, inference -----------------------------------
 * Activations are channels-last rows; the 1x1 convolutions are GEMMs outside.  Replaces, per WN layer
 * (reference reformer_tts/squeeze_wave/modules.py):
 *   rtts_sw_depthwise_k3  BatchNorm1d(eval, folded into w/bias by the caller) + depthwise Conv1d(k=3, pad 1) (:88-122):
 *                         x fp32 (B,L,C), w (C,3), bias (C) -> y bf16
 *   rtts_sw_gate          fused_add_tanh_sigmoid_multiply (:10-24) of the pointwise output pw (B*L, 2C) and the layer's
 *                         slice [cond_offset, +2C) of the conditioning (B*Lm, ld_cond), nearest-upsampled by `upsample`
 *                         (nn.Upsample, :216-219) -> acts bf16 (B*L, C)
 *   rtts_sw_coupling_inv  a1 = (a1 - b) / exp(s) on channels [half, 2*half) of audio (rows, ld_audio), wn_out = [s | b] (:353-359)
 *   rtts_sw_coupling_inv1x1  the same coupling followed by the inverse invertible 1x1 convolution (:57-85, :360-361) in one
 *                         launch: out (rows, n) fp32 = [a0 | (a1 - b) / exp(s)] @ winv^T, winv (n, n) fp32 row-major = W^-1,
 *                         wn_out = [s | b] with row stride ld_wn (the WN block's end_conv GEMM output, consumed in place);
 *                         n even, <= 128; NOT in place; fp32 FMA arithmetic (the audio path never drops to bf16) */
int rtts_sw_depthwise_k3(const float* x, const float* w, const float* bias, int B, int L, int C, void* y,
                         const float* edge_lo, const float* edge_hi, void* stream);
/* edge_lo / edge_hi (C floats, may be NULL) are subtracted at l = 0 / l = L-1: the share of a folded BatchNorm constant
 * that the zero padding of the reference does not see. */
int rtts_sw_gate(const void* pw, const void* cond, int64_t ld_cond, int cond_offset, int upsample, int B, int L, int Lm, int C,
                 void* acts, void* stream);
int rtts_sw_coupling_inv(float* audio, int64_t ld_audio, const float* wn_out, int64_t rows, int half, void* stream);
int rtts_sw_coupling_inv1x1(const float* audio, int64_t ld_audio, const float* wn_out, int64_t ld_wn, const float* winv, int n,
                            int64_t rows, float* out, int64_t ld_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTTS_H */
