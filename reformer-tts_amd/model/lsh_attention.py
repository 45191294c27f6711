"""``implementation="hip"`` of the LSH self-attention layer the reference builds at
``/root/reference/reformer_tts/model/reformer.py:198-200`` (``reformer_pytorch.LSHSelfAttention``):
same constructor surface, same parameter names (``toqk``, ``tov``, ``to_out``), the arithmetic of
SURVEY.md Appendix B steps 2-11 in librtts_hip.so.

MI355X-first differences from the reference's mechanism (results are the same function):
  * the two projections run as ONE (M, d) x (d, 2d) GEMM whose halves are consumed in place
    (row stride 2d) by the kernels -- no head split / merge copies;
  * the forward keeps the sort permutation ``st`` (3 MB per decoder layer): the reversible
    recompute re-uses the buckets that the forward used, bit for bit, instead of hashing the
    reconstructed (rounding-perturbed) input again.

Recompute protocol at this seam (``forward(x, input_mask=...)`` is all the reference's wrapper calls,
``reformer.py:215-217``).  The reference re-runs the layer inside ``Deterministic.forward(set_rng=True)``
(``reversible.py:26-41``): the device RNG state recorded before the no_grad forward is restored under
``fork_rng`` and the layer is called again with gradients enabled.  So this layer
  * draws its rotations from the DEFAULT device generator, as ``reformer_pytorch`` does -- the replay
    reproduces them, and the generator is advanced by the same amount before ``post_attn_dropout``
    draws its mask, so the mask repeats too;
  * remembers, after a no_grad training call, the generator state it started from, the permutation and
    a per-row signature of its input; a grad-enabled training call that starts from THE SAME generator
    state re-uses that permutation where the signature agrees (a device-side select: no host sync) and
    hashes afresh otherwise.  No flag from the caller is needed; ``recompute=True`` (this package's own
    ``Deterministic``) is the same thing without the checks.
The explicit executor (``engine.LSHExec``) keeps its permutations in per-call slots and does not come
through ``forward``.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F

from .. import ops
from .._seeds import next_seed


class _LSHAttnFn(torch.autograd.Function):
    """qkv (B,T,2d) bf16 [qk | v], st (B*H,R,T) -> out (B,T,d) bf16 (heads merged)."""

    @staticmethod
    def forward(ctx, qkv, st, mask, heads, bucket_size, causal, drop=None):
        d = qkv.shape[-1] // 2
        qk, v = qkv[..., :d], qkv[..., d:]
        o, lse = ops.lsh_attn_fwd(qk, v, st, heads, bucket_size, causal, mask, drop)
        out, lse_tot = ops.lsh_combine_fwd(o, lse, qkv.shape[0], heads)
        ctx.save_for_backward(qkv, st, out, lse_tot, mask)
        ctx.cfg = (heads, bucket_size, causal, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, st, out, lse_tot, mask = ctx.saved_tensors
        heads, bucket_size, causal, drop = ctx.cfg
        d = qkv.shape[-1] // 2
        dqkv = torch.empty_like(qkv)
        if dout.dtype != torch.bfloat16 or dout.stride(2) != 1:
            dout = dout.to(torch.bfloat16).contiguous()
        ops.lsh_attn_bwd(qkv[..., :d], qkv[..., d:], st, out, dout, lse_tot, heads, bucket_size, causal, mask,
                         dqkv=(dqkv[..., :d], dqkv[..., d:]), drop=drop)
        return dqkv, None, None, None, None, None, None


def _gemm_tiles(m: int, n: int, k: int) -> bool:
    """Does rtts_gemm_nt take y (m, n) = x (m, k) W^T, its input gradient (m, k) = dy (m, n) W and rtts_gemm_tn the weight
    gradient?  (row tiles of 96 / 128 / 192 / 256, column tiles of 64 / 128, 64-deep K stages; dW: 128 | n, 128 | k, 64 | m)"""
    rows = any(m % r == 0 for r in (96, 128))
    return rows and n % 128 == 0 and k % 128 == 0 and m % 64 == 0


class _ProjectFn(torch.autograd.Function):
    """y (M, N) bf16 = x (M, K) bf16 @ cat(weights)^T [+ bias], every product on the in-tree MFMA kernels: forward and input
    gradient on ``rtts_gemm_nt`` (the weight read in place as [N][K] resp. [K][N]), weight gradient on ``rtts_gemm_tn``, bias
    gradient by ``rtts_colsum_bf16`` -- the projections of the layer a maintainer gets from INTEGRATION.md section 1
    (``/root/reference/reformer_tts/model/reformer.py:198-217``) run no library GEMM.  ``weights``: one fp32 parameter (N, K), or
    two that are stacked along N (toqk | tov: ONE (M, d) x (d, 2d) product)."""

    @staticmethod
    def forward(ctx, x, bias, *weights):
        from .. import engine
        wb = (weights[0] if len(weights) == 1 else torch.cat(weights, dim=0)).detach().to(torch.bfloat16)
        y = engine.gemm(x, wb, bias=None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(x, wb)
        ctx.split = [w.shape[0] for w in weights]
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib, engine
        x, wb = ctx.saved_tensors
        dy = dy if (dy.dtype == torch.bfloat16 and dy.is_contiguous()) else dy.to(torch.bfloat16).contiguous()
        dx = engine.gemm(dy, wb, kn=True) if ctx.needs_input_grad[0] else None
        dw = torch.empty(wb.shape, dtype=torch.float32, device=x.device)
        engine.wgrad(dw, dy, x, accumulate=False)
        db = None
        if ctx.has_bias:
            m, n = dy.shape
            db = torch.zeros(n, dtype=torch.float32, device=x.device)
            ws = torch.empty(256 * n, dtype=torch.float32, device=x.device)
            _lib.call("rtts_colsum_bf16", dy.data_ptr(), None, dy.stride(0), db.data_ptr(), ws.data_ptr(), m, n, 0, 1.0, None,
                      torch.cuda.current_stream().cuda_stream)
        return (dx, db, *torch.split(dw, ctx.split, dim=0))


def _project(x2, bias, *weights):
    """(M, K) bf16 -> (M, N) bf16 on the in-tree GEMMs where the shape tiles, else the library (said once)."""
    m, k = x2.shape
    n = sum(w.shape[0] for w in weights)
    if x2.is_cuda and _gemm_tiles(m, n, k):
        return _ProjectFn.apply(x2, bias, *weights)
    from .._lib import note_general_path
    note_general_path("LSH attention projections", f"({m} x {k}) -> {n} does not tile (rows % 96 or 128, widths % 128): library GEMM")
    w = (weights[0] if len(weights) == 1 else torch.cat(weights, dim=0)).to(torch.bfloat16)
    return F.linear(x2, w, None if bias is None else bias.to(torch.bfloat16))


class LSHSelfAttention(nn.Module):
    def __init__(self, dim, heads=8, bucket_size=64, n_hashes=8, causal=False, add_local_attn_hash=False,
                 attn_chunks=1, random_rotations_per_head=False, attend_across_buckets=True,
                 allow_duplicate_attention=True, num_mem_kv=0, one_value_head=False, use_full_attn=False,
                 full_attn_thres=None, return_attn=False, post_attn_dropout=0.0, dropout=0.0, seed=0):
        super().__init__()
        if dim % heads != 0:
            raise AssertionError("dimensions must be divisible by number of heads")
        unsupported = dict(add_local_attn_hash=add_local_attn_hash, attend_across_buckets=not attend_across_buckets,
                           allow_duplicate_attention=not allow_duplicate_attention, num_mem_kv=num_mem_kv,
                           one_value_head=one_value_head, use_full_attn=use_full_attn, return_attn=return_attn)
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"implementation='hip' supports the default value of {bad} only")
        if dim // heads != 64:
            raise NotImplementedError(f"implementation='hip' is built for dim/heads == 64 (got {dim // heads})")
        self.dim, self.heads, self.bucket_size, self.n_hashes, self.causal = dim, heads, bucket_size, n_hashes, causal
        self.random_rotations_per_head = random_rotations_per_head
        self.full_attn_thres = bucket_size if full_attn_thres is None else full_attn_thres
        self.attn_chunks = attn_chunks  # memory-only knob in the reference; no numeric effect
        self.toqk = nn.Linear(dim, dim, bias=False)
        self.tov = nn.Linear(dim, dim, bias=False)
        self.to_out = nn.Linear(dim, dim)
        self.post_attn_dropout = nn.Dropout(post_attn_dropout)
        self.dropout = float(dropout)        # on the attention probabilities (config.py:27): a counter-hash mask inside the kernels
        self._attn_drop = None               # (p, seed) of the last no_grad training call: the recompute redraws the same mask
        self.seed = seed
        self._gen: Optional[torch.Generator] = None
        self._saved: Optional[tuple] = None   # (generator state, st, input signature) of the last no_grad training call
        self.forced_rotations: Optional[torch.Tensor] = None  # tests: use these instead of sampling
        self.last_st: Optional[torch.Tensor] = None

    rotation_pool = None       # (flat fp32 normal samples, [next offset]) while a graph-mode training forward runs, else None

    def _rotations(self, x, n_buckets, default_generator: bool = False):
        if self.forced_rotations is not None and hasattr(self.forced_rotations, "__next__"):
            return next(self.forced_rotations).to(device=x.device, dtype=torch.float32).contiguous()   # tests: one per call
        if isinstance(self.forced_rotations, dict):            # tests: one tensor per bucket count (batches of several padded lengths)
            r = self.forced_rotations[n_buckets]
            if r.device != x.device or r.dtype != torch.float32 or not r.is_contiguous():     # moved once (never inside a capture)
                r = self.forced_rotations[n_buckets] = r.to(device=x.device, dtype=torch.float32).contiguous()
            return r
        if self.forced_rotations is not None:
            if self.forced_rotations.device != x.device or self.forced_rotations.dtype != torch.float32:
                self.forced_rotations = self.forced_rotations.to(device=x.device, dtype=torch.float32).contiguous()
            return self.forced_rotations
        rows = x.shape[0] * self.heads if self.random_rotations_per_head else 1
        shape = (rows, self.dim // self.heads, self.n_hashes, n_buckets // 2)
        if default_generator or getattr(self, "use_default_generator", False):   # graph-safe, and what Deterministic replays
            pool = LSHSelfAttention.rotation_pool             # one randn per training step for all layers (trainer.forward_loss)
            n = rows * shape[1] * shape[2] * shape[3]
            if pool is not None and pool[0].device == x.device and pool[1][0] + n <= pool[0].numel():
                off = pool[1][0]
                pool[1][0] = off + n
                return pool[0][off:off + n].view(shape)
            return torch.randn(shape, device=x.device, dtype=torch.float32)
        if self._gen is None or self._gen.device != x.device:
            self._gen = torch.Generator(device=x.device)
            self._gen.manual_seed(0x5EED + self.seed)
        return torch.randn(shape, device=x.device, dtype=torch.float32, generator=self._gen)

    @staticmethod
    def _generator_state(device):
        """(seed, offset) of the default generator of ``device`` -- what ``Deterministic`` records and restores
        (``reversible.py:21-24,36-40``); None while a hipGraph is being captured (the offset is then a graph quantity)."""
        if device.type != "cuda":
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        g = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        return (g.initial_seed(), g.get_offset())

    @staticmethod
    def _signature(x):
        """Per-row L1 norms of the layer input: equal to ~1e-6 between a forward and its reversible recompute (the
        reconstructed stream differs by fp32 rounding), O(1) apart for any other input."""
        return x.detach().float().abs().sum(-1)

    def forward(self, x, input_mask=None, recompute: bool = False, **_):
        b, t, e = x.shape
        if t <= self.full_attn_thres:
            raise NotImplementedError("full-attention shortcut (T <= full_attn_thres) is outside the HIP path")
        if t % (self.bucket_size * 2) != 0:
            raise AssertionError(f"Sequence length ({t}) needs to be divisible by target bucket size  x 2 - {self.bucket_size * 2}")
        # (B,T,2d) = [qk | v]: ONE product over the stacked weights, on rtts_gemm_nt (forward, input gradient) / rtts_gemm_tn
        qkv = _project(x.to(torch.bfloat16).reshape(b * t, e), None, self.toqk.weight, self.tov.weight).view(b, t, 2 * e)
        state = self._generator_state(x.device)
        saved, st = self._saved, None
        grad_on = torch.is_grad_enabled()        # read here: the block below runs under no_grad
        with torch.no_grad():
            # always drawn, also when the permutation is re-used: the generator must end up where the forward left it
            rot = self._rotations(x, t // self.bucket_size, default_generator=True)
            replay = (self.training and grad_on and saved is not None and saved[1].shape[0] == b * self.heads
                      and saved[1].shape[2] == t and (recompute or (state is not None and saved[0] == state)))
            if replay and recompute:
                st = saved[1]
            else:
                st, _, _ = ops.lsh_hash_sort(qkv[..., :e], rot, self.heads, self.bucket_size)
                if replay:
                    sig = self._signature(x)
                    same = (sig - saved[2]).abs().max() <= 1e-3 * saved[2].abs().max()
                    st = torch.where(same, saved[1], st)
            drop = None
            if self.training and self.dropout > 0.0:
                # the mask is a function of (seed, pair): a replayed call takes the seed of the call it replays
                drop = self._attn_drop if (replay and self._attn_drop is not None) else (self.dropout, next_seed())
            if replay:
                self._saved = None
            elif self.training and not grad_on:
                self._saved = (state, st, self._signature(x))      # reversible forward: kept for the recompute
                self._attn_drop = drop
        self.last_st = st
        mask = None if input_mask is None else input_mask.to(torch.uint8)
        out = _LSHAttnFn.apply(qkv, st, mask, self.heads, self.bucket_size, self.causal, drop)
        y = _project(out.reshape(b * t, e), self.to_out.bias, self.to_out.weight).view(b, t, e)     # bias in the GEMM's fp32 epilogue
        return self.post_attn_dropout(y.float())
