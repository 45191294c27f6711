"""Explicit (autograd-free) executor of a reversible stack on the GPU.

The reference runs every block under nested ``torch.autograd.backward`` calls
(``/root/reference/reformer_tts/model/reversible.py:62-98,148-170``) over eager ATen graphs; here
each block is a short, fixed sequence of launches of the hand-written kernels of librtts_hip.so
(the projections and feed-forward layers included: ``gemm`` -> csrc/gemm_nt.hip) -- for the forward, for the
reconstruction ``x = y - f(.)`` and for the backward, with parameter gradients accumulated
straight into the flat gradient buffer.  The two residual streams and their gradients are four
fp32 buffers updated IN PLACE; a swap exchanges two Python references.

Every block is the primitive   acc += fn(inp)   with
    forward :  acc += fn(inp)
    backward:  acc -= fn(inp)  (reconstruct), d_inp += J_fn(inp)^T d_acc, parameter grads += ...
The encoder's ReversibleBlock is two such steps (f then g), the decoder's layer three with swaps.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch

from . import _lib, ops
from ._seeds import next_seed, seed_base
from .model.lsh_attention import LSHSelfAttention


# Reversible recompute with a selective stash: the attention cores' outputs (bf16 (B,T,d) + one fp32 logsumexp per
# token.head, ~13 MB per decoder layer at the baseline shape) are kept from the forward, so the backward's recompute of
# f(x) = to_out(attention(LN x)) re-runs only LayerNorm and the projections, not the attention forward.  The same function
# (deterministic kernels), but NOT bitwise: the pure recompute sees the reconstructed stream, whose last bits differ from
# the forward's (tests/test_model_hip.py::test_attention_stash_matches_pure_recompute).  HBM capacity traded for time.
# False = the reference's pure recompute.
STASH_ATTENTION = True
# The same trade one step further: keep f(x) itself (bf16 (B*T, d), 12.6 MB per decoder block at the baseline shape,
# ~130 MB for the whole model) so that the reconstruction x = y - f(.) subtracts exactly what the forward added and
# the backward skips the output-projection / second FFN GEMM of the recompute (38.6 of 87 GFLOP per decoder layer).
# What is still recomputed: LayerNorm, the QKV / q, kv / first FFN projections (their outputs are backward operands).
STASH_BLOCK_OUTPUT = True
# Keep the projections of a sublayer's forward (LSH: qk|v; cross attention: q and k|v; feed-forward: the hidden activation) for its
# backward instead of recomputing them from the reconstructed input: ~0.3 GB at the baseline shapes against 288 GB of HBM,
# and the backward then has no forward GEMM left in it.  Off = the reference's memory behaviour (everything recomputed).
STASH_PROJECTIONS = True
# Keep the streams themselves: every sublayer writes its updated stream to a NEW buffer (same traffic as the in-place update)
# and leaves its LayerNorm input and output where they are, so the backward neither reconstructs a stream nor re-normalises it
# (the reference reconstructs x2 = y2 - g(y1), x1 = y1 - f(x2): reversible.py:69-98).  25 MB per sublayer and stream at the
# baseline shapes.  Needs the three switches above (a recomputing backward has to run on reconstructed streams).
STASH_STREAMS = True
_NO_G = object()          # "the block output is not needed" (no reconstruction): _internals then skips its GEMM


def _streams_kept() -> bool:
    return STASH_STREAMS and STASH_ATTENTION and STASH_BLOCK_OUTPUT and STASH_PROJECTIONS


# The four switches above as ONE named mode, from the reference's behaviour (everything recomputed from the reconstructed
# streams: /root/reference/reformer_tts/model/reversible.py:114-129, 69-98) to "the backward recomputes nothing", ordered by
# what they hold in HBM.  This is what ``TTSTrainingConfig.recompute`` names and what ``Trainer`` applies before every forward.
RECOMPUTE_MODES = ("full", "attention-stash", "output-stash", "projection-stash", "stash")


def set_recompute(mode: str) -> str:
    """Select what a stack's forward keeps for its backward; -> the mode that was active.  A forward and ITS backward agree by
    construction (the backward reads what the forward's slot holds), so the mode may change between steps."""
    global STASH_ATTENTION, STASH_BLOCK_OUTPUT, STASH_PROJECTIONS, STASH_STREAMS
    if mode not in RECOMPUTE_MODES:
        raise ValueError(f"recompute mode {mode!r}: expected one of {RECOMPUTE_MODES}")
    old = recompute_mode()
    rank = RECOMPUTE_MODES.index(mode)
    STASH_ATTENTION, STASH_BLOCK_OUTPUT, STASH_PROJECTIONS, STASH_STREAMS = rank >= 1, rank >= 2, rank >= 3, rank >= 4
    return old


def recompute_mode() -> str:
    """Name of the active mode (the four switches set by hand to a combination without a name: the nearest lower one)."""
    rank = 0
    for on in (STASH_ATTENTION, STASH_BLOCK_OUTPUT, STASH_PROJECTIONS, STASH_STREAMS):
        if not on:
            break
        rank += 1
    return RECOMPUTE_MODES[rank]


def stash_bytes(program, rows: int, d: int, mode: str, key_rows: int = 0) -> int:
    """Estimate of what the forward of one stack (``build_program`` list) holds in HBM for its backward in ``mode``, beyond
    the two final streams every mode keeps: ``rows`` = B x T_pad tokens of width ``d``, ``key_rows`` = B x T_text for the cross
    attention.  Counted per sublayer: attention-stash the attention output (bf16) + one fp32 logsumexp per token.head;
    output-stash + f(x) (bf16); projection-stash + qk|v, q, k|v, the feed-forward hidden activation and its 1-bit gate;
    stash + the sublayer's input stream (fp32), its LayerNorm output (bf16) and statistics."""
    if mode not in RECOMPUTE_MODES:
        raise ValueError(f"recompute mode {mode!r}: expected one of {RECOMPUTE_MODES}")
    rank = RECOMPUTE_MODES.index(mode)
    total = 0
    for step in program:
        for ex in step[1:3]:
            if ex is None:
                continue
            per = 0
            if isinstance(ex, (LSHExec, XAttnExec)):
                heads = ex.layer.heads if isinstance(ex, LSHExec) else ex.mha.num_heads
                if rank >= 1:
                    per += rows * d * 2 + rows * heads * 4
                if rank >= 3:
                    per += rows * 2 * d * 2 if isinstance(ex, LSHExec) else rows * d * 2 + key_rows * 2 * d * 2
            elif isinstance(ex, FFNExec):
                ff = ex.l1.weight.shape[0]
                if rank >= 3:
                    per += rows * ff * 2 + rows * ff // 8
            if rank in (2, 3):                      # with the streams kept nothing is reconstructed: f(x) is dropped again
                per += rows * d * 2
            if rank >= 4:
                per += rows * d * 4 + rows * d * 2 + rows * 8
            total += per
    return total
WEIGHT_EPOCH = [0]   # bumped by the trainer after every optimizer step (its kernels write parameters through raw pointers)


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _bf16(p: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a parameter: the trainer's per-step flat mirror when present."""
    m = getattr(p, "_bf16_mirror", None)
    return m if m is not None else p.detach().to(torch.bfloat16)


def bf16_twin(x: torch.Tensor) -> Optional[torch.Tensor]:
    """The bf16 copy a producer of this package left with its fp32 output (FusedStackFn: the sum of the two streams is
    written in both precisions by one launch), or None -- also when x was written since."""
    tw = getattr(x, "_rtts_bf16", None)
    if tw is None or tw[1] != x._version or tw[0].device != x.device:
        return None
    return tw[0]


def _adjacent(a: torch.Tensor, b: torch.Tensor) -> bool:
    """b starts where a ends, inside one storage (true for neighbours in the trainer's flat buffers)."""
    return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.is_contiguous() and b.is_contiguous()
            and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr())


def _grad(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class _WS:
    """Per-device scratch for the deterministic column sums."""
    _cache = {}

    @classmethod
    def partial(cls, device, d: int) -> torch.Tensor:
        key = (device, d, torch.cuda.current_stream(device).cuda_stream)      # two streams (encoder beside decoder) never share scratch
        if key not in cls._cache:
            cls._cache[key] = torch.empty(2 * 256 * d, dtype=torch.float32, device=device)
        return cls._cache[key]


def ln_fwd(x, norm):
    m, d = x.shape
    xn = torch.empty(m, d, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(m, dtype=torch.float32, device=x.device)
    rstd = torch.empty(m, dtype=torch.float32, device=x.device)
    _lib.call("rtts_ln_fwd", x.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), xn.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), m, d, _s())
    return xn, mean, rstd


# Column sums (bias and LayerNorm gradients) are leaves of the backward like the weight gradients: with DEFER_COLSUM the
# producing kernels only write their per-workgroup partial rows (into a buffer of their own, held until the flush) and ONE
# grouped launch per layer adds them into the gradients -- a decoder layer otherwise pays fifteen 5-us launches for it.
DEFER_COLSUM = True


class _Queue:
    """Deferred gradient work queued on ONE (device, stream): whoever flushes it -- autograd's worker thread for that device
    (where FusedStackFn.backward and the edge backwards run), the engine's end-of-backward callback, or the trainer's main
    thread after ``loss.backward()`` -- launches the entries on the stream they were queued on, so they are ordered after
    the kernels that produce their operands whichever thread drains them."""

    def __init__(self, device: int, stream: int):
        self.device, self.stream = device, stream
        self.colsums = []          # (partial buffer [kept alive], float offset of the block, rows, d, out tensor, ld)
        self.wgrads = []           # (grad_view, dy, x)
        self.conv_gemms = []       # edges.py: tap problems of the convolutions' weight gradients
        self.conv_items = []       # edges.py: (dwp, co, ci, cp, grad) re-layouts of finished convolution gradients
        self.flush_queued = False

    def __len__(self):
        return len(self.colsums) + len(self.wgrads) + len(self.conv_gemms) + len(self.conv_items)

    def clear(self):
        del self.colsums[:], self.wgrads[:], self.conv_gemms[:], self.conv_items[:]
        self.flush_queued = False


_QUEUES = {}                      # (device index, stream handle) -> _Queue
_QLOCK = __import__("threading").RLock()


def _queue() -> _Queue:
    key = (torch.cuda.current_device(), _s())
    q = _QUEUES.get(key)
    if q is None:
        with _QLOCK:
            q = _QUEUES.setdefault(key, _Queue(*key))
    return q


def _all_queues(keys=None):
    """Every queue, or -- ``keys`` = an iterable of (device index, stream handle) -- only those an owner names: a trainer
    flushes / discards / counts what ITS streams queued, not what another trainer or thread of the process has pending."""
    with _QLOCK:
        if keys is None:
            return list(_QUEUES.values())
        return [_QUEUES[k] for k in keys if k in _QUEUES]


def drop_stream_state(device_index: int, stream: int) -> None:
    """Forget the scratch and the (empty) queue of a stream that will not be used again (a warm-up stream): the 64 MB slab,
    the column-sum scratch and the queue object are keyed by stream handle and would otherwise live as long as the process."""
    with _QLOCK:
        q = _QUEUES.get((device_index, stream))
        if q is not None and not len(q):
            del _QUEUES[(device_index, stream)]
    for key in [k for k in _WS._cache if isinstance(k, tuple) and k[-1] == stream and
                (getattr(k[0], "index", None) == device_index or getattr(k[1], "index", None) == device_index)]:
        del _WS._cache[key]


def _partial_rows(m: int) -> int:
    return min((m + 3) // 4, 256)


def _queue_colsum(partial: torch.Tensor, offset_floats: int, rows: int, d: int, out: torch.Tensor, ld: int = 0):
    """``ld``: row stride of the partial buffer when ``d`` columns are a block of a wider one (0: d)."""
    _queue().colsums.append((partial, offset_floats, rows, d, out, ld))
    _queue_final_flush()


def flush_colsum(q: Optional[_Queue] = None):
    for q in ([q] if q is not None else _all_queues()):
        pending = q.colsums
        while pending:
            group = pending[:_lib.COLSUM_MAX_GROUP]
            del pending[:len(group)]
            arr = (_lib.ColsumJob * len(group))()
            for j, (partial, off, rows, d, out, ld) in zip(arr, group):
                j.partial, j.out, j.nrows, j.n, j.ld = partial.data_ptr() + 4 * off, out.data_ptr(), rows, d, ld
            with torch.cuda.device(q.device):
                _lib.call("rtts_colsum_final_grouped", arr, len(group), q.stream)


def ln_bwd(dxn, x, mean, rstd, norm, dx_io, next_cast=None, dx_in=None, join=None):
    """dx_io += dLN(dxn) (``dx_in`` given: dx_io = dx_in + dLN(dxn), out of place; ``join``: + that stream too -- the last update of a
    stack's backward, d(input) = g1 + g2); LayerNorm gradients queued (or added).  ``next_cast`` = (drop | None,): the completed dx_io is
    the next block's output gradient, so its bf16 copy (times that block's dropout keep-scale) and the partial column
    sums for that block's output bias are produced here -> (dyb, partial buffer, rows), else None."""
    m, d = x.shape
    dev = x.device
    nxt = None
    args_next = (None, None, 0.0, 0, None)
    if next_cast is not None:
        drop = next_cast[0]
        p, seed = drop if drop else (0.0, 0)
        dyb = torch.empty(m, d, dtype=torch.bfloat16, device=dev)
        pn = torch.empty(256 * d, dtype=torch.float32, device=dev)
        args_next = (dyb.data_ptr(), pn.data_ptr(), float(p), seed, seed_base(dev).data_ptr())
        nxt = (dyb, pn, _partial_rows(m))
    src = dx_io if dx_in is None else dx_in
    other = None if join is None else join.data_ptr()
    if DEFER_COLSUM:
        ws = torch.empty(2 * 256 * d, dtype=torch.float32, device=dev)
        _lib.call("rtts_ln_bwd_join", dxn.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), norm.weight.data_ptr(),
                  src.data_ptr(), other, dx_io.data_ptr(), None, None, ws.data_ptr(), m, d, *args_next, _s())
        rows = _partial_rows(m)
        _queue_colsum(ws, 0, rows, d, _grad(norm.weight))
        _queue_colsum(ws, 256 * d, rows, d, _grad(norm.bias))
        return nxt
    _lib.call("rtts_ln_bwd_join", dxn.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), norm.weight.data_ptr(),
              src.data_ptr(), other, dx_io.data_ptr(), _grad(norm.weight).data_ptr(), _grad(norm.bias).data_ptr(), _WS.partial(dev, d).data_ptr(),
              m, d, *args_next, _s())
    return nxt


def _out_grad(d_acc, dbias, drop, pre_cast):
    """bf16 copy of the block's output gradient (+ its bias gradient): taken from the previous LayerNorm backward when it
    produced one for this stream (``pre_cast``), else one rtts_cast_colsum launch."""
    if pre_cast is not None:
        dyb, pn, rows = pre_cast
        _queue_colsum(pn, 0, rows, d_acc.shape[1], dbias)
        return dyb
    return cast_colsum(d_acc, dbias, drop)


def cast_colsum(dy, dbias, drop=None, defer: bool = True, scale: Optional[torch.Tensor] = None):
    """dyb (bf16) = dy [* keep-scale of ``drop`` = (p, seed)] [* the device scalar ``scale``]; dbias += column sums of the same.
    ``dbias``: a tensor, None, or a list of (first column, columns, out) blocks -- the sums of a padded gradient go to several
    parameters' gradients (always deferred to the grouped final launch)."""
    m, d = dy.shape
    dyb = torch.empty(m, d, dtype=torch.bfloat16, device=dy.device)
    p, seed = drop if drop else (0.0, 0)
    sc = None if scale is None else scale.data_ptr()
    blocks = dbias if isinstance(dbias, (list, tuple)) else None
    if blocks is not None or (DEFER_COLSUM and defer and dbias is not None):
        ws = torch.empty(256 * d, dtype=torch.float32, device=dy.device)
        _lib.call("rtts_cast_colsum", dy.data_ptr(), dyb.data_ptr(), None, ws.data_ptr(), m, d, float(p), seed,
                  seed_base(dy.device).data_ptr(), sc, _s())
        if blocks is not None:
            for c0, n, out in blocks:
                _queue_colsum(ws, c0, _partial_rows(m), n, out, ld=d)
        else:
            _queue_colsum(ws, 0, _partial_rows(m), d, dbias)
        return dyb
    _lib.call("rtts_cast_colsum", dy.data_ptr(), dyb.data_ptr(), None if dbias is None else dbias.data_ptr(),
              _WS.partial(dy.device, d).data_ptr(), m, d, float(p), seed, seed_base(dy.device).data_ptr(), sc, _s())
    return dyb


def colsum_bf16(dh, dbias, h=None, gate_scale: float = 1.0, out: Optional[torch.Tensor] = None):
    """dbias += column sums of dh; with ``h``: dh * (h > 0) * gate_scale first (ReLU [+ dropout] backward), written to ``out``
    (same strides as dh) or back in place."""
    m, d = dh.shape
    o = None if out is None else out.data_ptr()
    if DEFER_COLSUM and dbias.is_contiguous():
        ws = torch.empty(256 * d, dtype=torch.float32, device=dh.device)
        _lib.call("rtts_colsum_bf16", dh.data_ptr(), None if h is None else h.data_ptr(), dh.stride(0), None, ws.data_ptr(), m, d,
                  int(h is not None), float(gate_scale), o, _s())
        _queue_colsum(ws, 0, _partial_rows(m), d, dbias)
        return
    _lib.call("rtts_colsum_bf16", dh.data_ptr(), None if h is None else h.data_ptr(), dh.stride(0), dbias.data_ptr(),
              _WS.partial(dh.device, d).data_ptr(), m, d, int(h is not None), float(gate_scale), o, _s())


def residual(acc, g, bias, sign: float, next_norm=None, drop=None, out=None):
    """out = acc + sign * dropout(g + bias) (``out`` None: in place; ``drop`` = (p, seed) or None).  With ``next_norm`` (the
    LayerNorm of the block that reads the result next) the row is normalised in the same launch: returns (xn, mean, rstd) of
    LayerNorm(result), else None."""
    m, d = acc.shape
    p, seed = drop if drop else (0.0, 0)
    sb = seed_base(acc.device).data_ptr()
    if next_norm is None:
        _lib.call("rtts_residual_epilogue", acc.data_ptr(), g.data_ptr(), None if bias is None else bias.data_ptr(), float(sign),
                  (acc if out is None else out).data_ptr(), m, d, float(p), seed, sb, _s())
        return None
    xn = torch.empty(m, d, dtype=torch.bfloat16, device=acc.device)
    mean = torch.empty(m, dtype=torch.float32, device=acc.device)
    rstd = torch.empty(m, dtype=torch.float32, device=acc.device)
    _lib.call("rtts_residual_ln", acc.data_ptr(), g.data_ptr(), None if bias is None else bias.data_ptr(), float(sign),
              next_norm.weight.data_ptr(), next_norm.bias.data_ptr(), xn.data_ptr(), mean.data_ptr(), rstd.data_ptr(), m, d,
              float(p), seed, sb, None if out is None else out.data_ptr(), _s())
    return xn, mean, rstd


def gate_words(m: int, n: int, device) -> Optional[torch.Tensor]:
    """Buffer for the 1-bit ReLU gate of an (m, n) feed-forward activation (one word per lane and tile of the GEMM that
    produces it), or None where the tile shape chosen for (m, n) has no word form."""
    if os.environ.get("RTTS_NO_GATE_WORDS"):          # A/B runs: gate by the activation
        return None
    words = _lib.load().rtts_gemm_nt_gate_words(m, n)
    return torch.empty(words, dtype=torch.int64, device=device) if words > 0 else None


def gemm(a: torch.Tensor, w: torch.Tensor, kn: bool = False, bias: Optional[torch.Tensor] = None, relu: bool = False,
         gate: Optional[torch.Tensor] = None, gate_bias_grad: Optional[torch.Tensor] = None, out_f32: bool = False,
         words: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C (M, N) bf16 = epilogue(a (M, K) @ W), csrc/gemm_nt.hip (hand-written MFMA kernel; no library GEMM on the stack path).
    ``kn=False``: w is (N, K) -- y = x W^T, the forward of nn.Linear; ``kn=True``: w is (K, N) -- dx = dy W, its input gradient.
    ``bias`` (fp32, N) [+ ``relu``] ride in the epilogue; ``gate`` (M, N) bf16: C = acc * (gate > 0), the backward of ReLU,
    with ``gate_bias_grad`` += column sums of C (queued with the other deferred column sums).  ``out_f32``: the unrounded fp32
    result (+ bias) instead of bf16.  ``words`` (``gate_words(m, n)``): with ``relu`` the epilogue also WRITES the sign pattern of
    its outputs there, one bit each; with ``gate=True`` the gate is READ from those words instead of an (M, N) activation."""
    m, k = a.shape
    n = w.shape[1] if kn else w.shape[0]
    if (w.shape[0] if kn else w.shape[1]) != k or a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError(f"gemm: operand shapes {tuple(a.shape)} x {tuple(w.shape)} (kn={kn}) do not match")
    c = torch.empty(m, n, dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    epi, cs = 0, None
    if out_f32:
        if gate is not None or relu:
            raise ValueError("gemm: out_f32 takes a plain or bias epilogue")
        epi = 4
    elif gate is not None:
        epi = 3
        if gate is True and words is None:
            raise ValueError("gemm: gate=True takes the forward's gate words")
        if gate_bias_grad is not None:
            rows = _lib.load().rtts_gemm_nt_partial_rows(m, n)
            if rows <= 0:
                raise _lib.RttsError(f"rtts_gemm_nt: no tile shape for M x N = {m} x {n}")
            cs = torch.empty(rows, n, dtype=torch.float32, device=a.device)
    elif bias is not None:
        epi = 2 if relu else 1
    ev = ops.TIMING.start(f"rtts_gemm_nt/{m}x{n}x{k}")
    if words is not None and epi in (2, 3):
        _lib.call("rtts_gemm_nt_gated", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), int(kn), m, n, k, c.data_ptr(), n,
                  None if bias is None else bias.data_ptr(), epi, words.data_ptr(), None if cs is None else cs.data_ptr(), _s())
    else:
        _lib.call("rtts_gemm_nt", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), int(kn), m, n, k, c.data_ptr(), n,
                  None if bias is None else bias.data_ptr(), epi, None if gate is None else gate.data_ptr(),
                  0 if gate is None else gate.stride(0), None if cs is None else cs.data_ptr(), _s())
    ops.TIMING.stop(ev, 2.0 * m * n * k)
    if cs is not None:
        _queue_colsum(cs, 0, cs.shape[0], n, gate_bias_grad)
    return c


def _nt_problem(e, a, w, kn, c, bias=None, epi=0, accumulate=False, aux=None, aux_out=None, t=0, h=0):
    m, k = a.shape
    n = w.shape[1] if kn else w.shape[0]
    if (w.shape[0] if kn else w.shape[1]) != k or a.stride(1) != 1 or w.stride(1) != 1 or c.shape != (m, n) or c.stride(1) != 1:
        raise ValueError(f"gemm: operand shapes {tuple(a.shape)} x {tuple(w.shape)} (kn={kn}) -> {tuple(c.shape)} do not match")
    e.a, e.lda, e.w, e.ldw, e.c, e.ldc = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), c.data_ptr(), c.stride(0)
    e.bias = None if bias is None else bias.data_ptr()
    e.aux, e.ld_aux = (None, 0) if aux is None else (aux.data_ptr(), aux.stride(0))
    e.aux_out = None if aux_out is None else aux_out.data_ptr()
    e.M, e.N, e.K, e.epilogue, e.T, e.H, e.accumulate, e.reserved = m, n, k, epi, t, h, int(accumulate), 0
    return 2.0 * m * n * k


def gemm_group(problems, kn: bool = False):
    """Up to ``_lib.GEMM_NT_MAX_GROUP`` INDEPENDENT products in ONE launch of csrc/gemm_nt.hip (``rtts_gemm_nt_grouped``: one
    grid over all their tiles).  ``problems``: dicts with ``a`` (M, K) bf16, ``w`` ((N, K), or (K, N) with ``kn``), optional
    ``bias`` (fp32, N), ``out_f32`` (unrounded fp32 result), ``into`` (an fp32 (M, N) tensor the result is ADDED to: the keys'
    gradient over the decoder layers).  -> list of outputs (bf16, fp32, or ``into`` itself)."""
    if not 1 <= len(problems) <= _lib.GEMM_NT_MAX_GROUP:
        raise ValueError(f"gemm_group: 1..{_lib.GEMM_NT_MAX_GROUP} problems")
    arr = (_lib.GemmNtProblem * len(problems))()
    outs, flop = [], 0.0
    for e, p in zip(arr, problems):
        a, w = p["a"], p["w"]
        m = a.shape[0]
        n = w.shape[1] if kn else w.shape[0]
        into = p.get("into")
        f32 = bool(p.get("out_f32")) or into is not None
        c = into if into is not None else torch.empty(m, n, dtype=torch.float32 if f32 else torch.bfloat16, device=a.device)
        if into is not None and (into.dtype != torch.float32 or into.shape != (m, n)):
            raise ValueError("gemm_group: `into` must be an fp32 (M, N) tensor")
        bias = p.get("bias")
        flop += _nt_problem(e, a, w, kn, c, bias, 4 if f32 else (1 if bias is not None else 0), accumulate=into is not None)
        outs.append(c)
    ev = ops.TIMING.start("rtts_gemm_nt/group" + "+".join(f"{e.M}x{e.N}x{e.K}" for e in arr))
    _lib.call("rtts_gemm_nt_grouped", arr, len(problems), int(kn), _s())
    ops.TIMING.stop(ev, flop)
    return outs


def gemm_dgrad_delta(dy: torch.Tensor, w: torch.Tensor, out: torch.Tensor, t: int, heads: int):
    """dout (M, N) bf16 = dy (M, K) @ w (K, N) -- the input gradient of to_out / out_proj -- AND delta (B*H, T) fp32 =
    rowsum over each 64-wide head of out * dout, in the same launch (epilogue 5): what ``rtts_lsh_bwd_delta`` computed in a
    launch of its own from a second read of dout.  ``out``: the attention output (M, N) bf16, rows m = b * T + t."""
    m = dy.shape[0]
    n = w.shape[1]
    if n != heads * 64 or m % t:
        raise ValueError("gemm_dgrad_delta: N must be 64 x heads and M a multiple of T")
    dout = torch.empty(m, n, dtype=torch.bfloat16, device=dy.device)
    delta = torch.empty((m // t) * heads, t, dtype=torch.float32, device=dy.device)
    arr = (_lib.GemmNtProblem * 1)()
    flop = _nt_problem(arr[0], dy, w, True, dout, None, 5, aux=out, aux_out=delta, t=t, h=heads)
    ev = ops.TIMING.start(f"rtts_gemm_nt/{m}x{n}x{dy.shape[1]}")
    _lib.call("rtts_gemm_nt_grouped", arr, 1, 1, _s())
    ops.TIMING.stop(ev, flop)
    return dout, delta


def dgrad_delta_ok(m: int, n: int) -> bool:
    """Does epilogue 5 tile (M, N)?  (192 x 128 or 128 x 64 tiles: waves that span one 64-wide head)"""
    return (m % 192 == 0 and n % 128 == 0) or (m % 128 == 0 and n % 64 == 0)


FUSE_DELTA = os.environ.get("RTTS_FUSE_DELTA", "1") != "0"          # A/B: the separate rtts_lsh_bwd_delta launch
GROUP_XATTN = os.environ.get("RTTS_GROUP_XATTN", "1") != "0"        # A/B: the q and k|v projections as two launches
GROUP_XATTN_BWD = os.environ.get("RTTS_GROUP_XATTN_BWD", "0") == "1"   # A/B: dxn and dkeys as one grouped launch (measured: no gain)

_SLAB_FLOATS = 16 * 1024 * 1024   # 64 MB: 16 splits of a 2048 x 512 gradient

# Weight gradients are leaves of the backward's dependency graph: nothing reads them before the block's all-reduce /
# the optimizer.  With DEFER_WGRAD they are queued (operands held) and launched together, up to 8 per grouped
# rtts_gemm_tn_grouped call: one grid of ~240 tiles per decoder layer instead of seven grids of 16..64 tiles, so the
# split factor and the fp32 slab traffic drop ~7x and launch ramps/tails are paid once per layer.
DEFER_WGRAD = True
# The queue is flushed once per layer's worth of problems, on one GPU too, where nobody waits for a block's gradients: holding
# the weight gradients of the WHOLE backward back for a few large mixed groups at its end (RTTS_WGRAD_FLUSH=end) measured
# 6.57 against 6.23 ms/step -- a layer's dy / x operands are still in the 256 MB Infinity Cache right after its backward and
# have left it by the end of the step.
WGRAD_FLUSH_PER_LAYER = os.environ.get("RTTS_WGRAD_FLUSH", "layer") == "layer"
WGRAD_MAX_PENDING = 64
# measurement only (scripts/replay_stamps.py): STAMPS = (uint64 device buffer, {name: slot}) makes stamp(name) launch a marker kernel
# on the current stream; None (always, in product runs): stamp() does nothing
STAMPS = None


def stamp(name: str) -> None:
    if STAMPS is None:
        return
    buf, names = STAMPS
    slot = names.setdefault(name, len(names))
    _lib.call("rtts_debug_stamp", buf.data_ptr(), slot, _s())


JOIN_STREAMS = os.environ.get("RTTS_JOIN_STREAMS", "1") != "0"      # A/B: d(input) = g1 + g2 as a separate (ATen) pass after the stack's backward
COPY_STREAMS = os.environ.get("RTTS_STREAM_COPIES", "0") == "1"     # A/B: copy x / dout into both streams instead of aliasing them
# (A second HIP stream for the weight gradients, forked/joined by events = parallel branches of the captured hipGraph,
#  was measured SLOWER on MI355X in round 1: 9.35 vs 8.94 ms/step; the cross-branch dependencies of the replayed graph
#  cost more than the overlapped tails recover.  Removed.)


# The overlapped one-process step ends with the ENCODER's backward alone on the chip: it cannot start before the decoder's lowest
# cross-attention has run backwards (the keys' gradient), it is a chain of small-grid kernels (3,072 rows), and the decoder branch
# beside it finishes 0.35-0.6 ms earlier (scripts/replay_stamps.py: markers inside the UNPROFILED replayed graph,
# profiles/r04_replay_stamps_*.log).  The encoder stack's weight gradients are not part of that chain -- nothing reads them before
# the optimizer -- so the trainer has them handed over: where the encoder stream's queue would be flushed, its weight-gradient
# entries are set aside together with an event recorded on that stream, and the MAIN stream launches them after the decoder's own
# work (Trainer.forward_backward_overlapped -> run_handed_over).  STEAL = None (always, outside that step).
STEAL = None          # dict(src=(device index, stream handle), src_stream=torch.cuda.Stream, pending=[(event, [(grad, dy, x)...])])


def run_handed_over(steal) -> int:
    """Launch, on the CURRENT stream, the weight gradients that were set aside from ``steal['src']``'s queue (each group behind the
    event that marks its operands' producers).  -> launches made."""
    main = torch.cuda.current_stream()
    n = 0
    for ev, entries in steal["pending"]:
        main.wait_event(ev)
        for gv, dy, x in entries:
            dy.record_stream(main)
            x.record_stream(main)
        pending = list(entries)
        while pending:
            group = pending[:_lib.GEMM_TN_MAX_GROUP]
            del pending[:len(group)]
            arr = (_lib.GemmTnProblem * len(group))()
            for e, (gv, dy, x) in zip(arr, group):
                e.a, e.lda, e.b, e.ldb, e.c, e.ldc = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gv.data_ptr(), gv.stride(0)
                e.M, e.N, e.K, e.accumulate = dy.shape[0], dy.shape[1], x.shape[1], 1
            ws = _slab_ws(group[0][1].device)
            _lib.call("rtts_gemm_tn_grouped", arr, len(group), ws.data_ptr(), ws.numel(), _s())
            n += 1
    del steal["pending"][:]
    return n


def _final_flush():
    flush_wgrad()


def _slab_ws(device, stream: Optional[int] = None):
    stream = torch.cuda.current_stream(device).cuda_stream if stream is None else stream
    key = ("slab", torch.device(device), stream)     # launches on one stream are ordered; two streams (two trainers, a side
    # stream) must not share partial-tile scratch
    if key not in _WS._cache:
        _WS._cache[key] = torch.empty(_SLAB_FLOATS, dtype=torch.float32, device=device)
    return _WS._cache[key]


def _queue_final_flush():
    q = _queue()
    if not q.flush_queued:
        # whatever is still queued when the running autograd pass ends is launched by the engine's callback
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_final_flush)
            q.flush_queued = True
        except RuntimeError:          # not inside a backward pass: the caller flushes
            pass


def flush_wgrad(colsums: bool = True, keys=None):
    """(``keys``: only the queues of these (device index, stream handle) pairs -- see ``_all_queues``.)
    Launch every queued weight gradient and (``colsums``) column-sum finalisation, grouped; release the held operands.
    ``colsums=False`` (the per-layer flush of the stack loop when nobody waits for a block's gradients): the small partial
    buffers stay queued for fewer, fuller launches -- the end-of-backward flush takes them.
    EVERY queue is drained (all devices, all streams, whichever thread filled them), each on its own stream: the trainer's
    flush after ``loss.backward()`` on the main thread sees what autograd's worker thread queued."""
    for q in _all_queues(keys):
        q.flush_queued = False         # the next deferral queues a fresh end-of-backward callback (extra ones are no-ops);
        #                                a backward that died half-way cannot leave the flag stuck
        if not len(q):
            continue
        with torch.cuda.device(q.device):
            if colsums or len(q.colsums) >= _lib.COLSUM_MAX_GROUP - 8:
                flush_colsum(q)
            elif q.colsums and (q.device, q.stream) == (torch.cuda.current_device(), _s()):
                _queue_final_flush()
            pending = q.wgrads
            steal = STEAL
            if steal is not None and steal.get("armed") and pending and (q.device, q.stream) == steal["src"]:
                ev = torch.cuda.Event()
                ev.record(steal["src_stream"])            # everything this stream has been given so far: the operands' producers
                steal["pending"].append((ev, list(pending)))
                del pending[:]
            while pending:
                group = pending[:_lib.GEMM_TN_MAX_GROUP]
                del pending[:len(group)]
                arr = (_lib.GemmTnProblem * len(group))()
                for e, (gv, dy, x) in zip(arr, group):
                    e.a, e.lda, e.b, e.ldb, e.c, e.ldc = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gv.data_ptr(), gv.stride(0)
                    e.M, e.N, e.K, e.accumulate = dy.shape[0], dy.shape[1], x.shape[1], 1
                ws = _slab_ws(group[0][1].device, q.stream)
                _lib.call("rtts_gemm_tn_grouped", arr, len(group), ws.data_ptr(), ws.numel(), q.stream)
            for hook in FLUSH_HOOKS:       # other deferred gradient work (edges.py: the convolutions' dW re-layout)
                hook(q)


def discard_pending(keys=None) -> int:
    """Drop everything queued (a backward that raised half-way must not leak its entries into the next step's gradients);
    ``keys``: only the caller's own queues.  -> number of entries dropped."""
    n = 0
    for q in _all_queues(keys):
        n += len(q)
        q.clear()
    return n


FLUSH_HOOKS = []


def pending_wgrads() -> int:
    """Weight gradients queued on the current (device, stream) -- what the stack loop's per-layer flush counts."""
    return len(_queue().wgrads)


def pending_all(keys=None) -> int:
    """Every deferred entry of every queue -- of the caller's own queues with ``keys`` (0 after a flush: Trainer.backward asserts it)."""
    return sum(len(q) for q in _all_queues(keys))


def wgrad(grad_view: torch.Tensor, dy: torch.Tensor, x: torch.Tensor, accumulate: bool = True):
    """grad_view (N,K) fp32 (+)= dy(M,N)^T @ x(M,K)   (bf16 operands, fp32 accumulation).
    Split-K kernel of csrc/gemm_tn.hip when the shape tiles (128 | N, 128 | K, 64 | M), else hipBLASLt.
    Accumulating calls are queued under DEFER_WGRAD (see flush_wgrad); dy and x must not be written afterwards."""
    m, n = dy.shape
    k = x.shape[1]
    if n % 128 == 0 and k % 128 == 0 and m % 64 == 0 and dy.stride(1) == 1 and x.stride(1) == 1 and grad_view.stride(1) == 1:
        if DEFER_WGRAD and accumulate:
            _queue().wgrads.append((grad_view, dy, x))
            _queue_final_flush()
            return
        ws = _slab_ws(dy.device)
        _lib.call("rtts_gemm_tn", dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), m, n, k, grad_view.data_ptr(),
                  grad_view.stride(0), int(accumulate), ws.data_ptr(), ws.numel(), _s())
    elif n % 128 == 0 and m % 64 == 0 and m >= 4096 and k < 128 and dy.stride(1) == 1:
        # a narrow input (the decoder prenet's 80 mel channels): the library's TN kernel for a (N x 80) output over 12288 rows
        # is 8 workgroups walking the whole token range (82 us); zero-padded to 128 columns the split-K kernel takes it
        xp = torch.zeros(m, 128, dtype=x.dtype, device=x.device)
        xp[:, :k].copy_(x)
        tmp = torch.empty(n, 128, dtype=torch.float32, device=x.device)
        ws = _slab_ws(dy.device)
        _lib.call("rtts_gemm_tn", dy.data_ptr(), dy.stride(0), xp.data_ptr(), 128, m, n, 128, tmp.data_ptr(), 128, 0, ws.data_ptr(),
                  ws.numel(), _s())
        if accumulate:
            grad_view.add_(tmp[:, :k])
        else:
            grad_view.copy_(tmp[:, :k])
    else:
        _lib.note_general_path("weight gradient", f"dW {n} x {k} over {m} rows does not tile (N, K % 128, M % 64): library GEMM")
        if accumulate:
            grad_view.add_(torch.mm(dy.t(), x, out_dtype=torch.float32))
        else:
            grad_view.copy_(torch.mm(dy.t(), x, out_dtype=torch.float32))


# ------------------------------------------------------------------------------------------ blocks
def _keep_streams(keep, slot, acc, inp, xn, mean, rstd, fresh=False):
    """``keep`` (the stack loop asks for it under STASH_STREAMS): -> the fresh buffer the updated stream goes to (also left in
    slot["acc_out"] for the loop to pick up), with the sublayer's LayerNorm input and output kept in the slot for its
    backward; else None: the stream is updated in place (an executor used on its own, or a recomputing mode)."""
    if not keep:
        if fresh:       # the stream still IS the stack's input (both streams start as x, no copy): its first update goes elsewhere
            out = slot["acc_out"] = torch.empty_like(acc)
            return out
        return None
    out = torch.empty_like(acc)
    slot.update(inp=inp, pre=(xn, mean, rstd), acc_out=out)
    if STASH_BLOCK_OUTPUT:
        slot["g"] = None                      # no reconstruction: f(x) is not read again
    return out


class LSHExec:
    """WithNorm(LayerNorm, LSHSelfAttentionWrapper): acc += to_out(LSH(LN(inp)))."""

    def __init__(self, withnorm):
        self.norm = withnorm.norm
        self.layer: LSHSelfAttention = withnorm.fn.layer
        # What a forward leaves for ITS backward lives in a per-call slot (a dict owned by FusedStackFn's ctx), never on the
        # executor: a second forward before the first backward (two losses, an eval forward in between, two models sharing
        # layers) must not hand the first backward the second forward's permutation / stash / dropout seed.  Keys:
        #   st (sort permutation), stash ((out, lse_tot) when STASH_ATTENTION), g (f(x) when STASH_BLOCK_OUTPUT), qkv (STASH_PROJECTIONS),
        #   drop ((p, seed) of the post-attention dropout)
        self._own_slot = {}   # direct use of one executor outside a stack (tests)

    @staticmethod
    def supported(withnorm) -> bool:
        # post_attn_dropout rides in the residual epilogue (counter-hash mask, reproduced by the backward); the projections are
        # rtts_gemm_nt launches: K = d and N = d, 2d must be multiples of its 64-wide stages / tiles
        return withnorm.fn.layer.dim % 64 == 0

    def _wqkv(self):
        lyr = self.layer
        a, b = _bf16(lyr.toqk.weight), _bf16(lyr.tov.weight)
        if _adjacent(a, b):                                        # neighbours in the flat mirror: zero-copy (2d, d) view
            return torch.as_strided(a, (2 * a.shape[0], a.shape[1]), (a.shape[1], 1))
        return torch.cat([a, b], dim=0)

    def _wqkv_grad(self):
        lyr = self.layer
        ga, gb = _grad(lyr.toqk.weight), _grad(lyr.tov.weight)
        if _adjacent(ga, gb):
            return torch.as_strided(ga, (2 * ga.shape[0], ga.shape[1]), (ga.shape[1], 1)), None
        return None, (ga, gb)

    def _internals(self, inp, b, t, mask, st, stash=None, g=None, pre=None, qkv=None, adrop=None):
        lyr = self.layer
        e = lyr.dim
        if t <= lyr.full_attn_thres:
            raise NotImplementedError("full-attention shortcut (T <= full_attn_thres) is outside the HIP path")
        xn, mean, rstd = pre if pre is not None else ln_fwd(inp, self.norm)
        wqkv = self._wqkv()
        if qkv is None:
            qkv = gemm(xn, wqkv).view(b, t, 2 * e)
        if st is None:
            rot = lyr._rotations(qkv, t // lyr.bucket_size)
            st, _, _ = ops.lsh_hash_sort(qkv[..., :e], rot, lyr.heads, lyr.bucket_size)
        lyr.last_st = st
        if stash is not None:
            out, lse_tot = stash
        else:
            o, lse = ops.lsh_attn_fwd(qkv[..., :e], qkv[..., e:], st, lyr.heads, lyr.bucket_size, lyr.causal, mask, adrop)
            out, lse_tot = ops.lsh_combine_fwd(o, lse, b, lyr.heads)
        if g is None:
            g = gemm(out.view(b * t, e), _bf16(lyr.to_out.weight))
        return xn, mean, rstd, wqkv, qkv, st, out, lse_tot, g

    def forward(self, acc, inp, b, t, mask=None, pre=None, next_norm=None, slot=None, keep_streams=False, fresh_acc=False, **_):
        slot = self._own_slot if slot is None else slot
        # dropout on the attention probabilities (the layer's `dropout` knob): a (p, seed) pair the recompute and the backward reuse
        pa = getattr(self.layer, "dropout", 0.0) if self.layer.training else 0.0
        adrop = (pa, next_seed()) if pa > 0.0 else None
        xn, mean, rstd, _, qkv, st, out, lse_tot, g = self._internals(inp, b, t, mask, None, pre=pre, adrop=adrop)
        p = self.layer.post_attn_dropout.p if self.layer.training else 0.0
        slot.clear()
        slot.update(st=st, stash=(out, lse_tot) if STASH_ATTENTION else None, g=g if STASH_BLOCK_OUTPUT else None,
                    qkv=qkv if STASH_PROJECTIONS else None, drop=(p, next_seed()) if p > 0.0 else None, adrop=adrop)
        return residual(acc, g, self.layer.to_out.bias, 1.0, next_norm, slot["drop"], out=_keep_streams(keep_streams, slot, acc, inp, xn, mean, rstd, fresh_acc))

    def backward(self, acc, inp, d_acc, d_inp, b, t, mask=None, pre=None, next_norm=None, pre_cast=None, next_cast=None, slot=None,
                 d_src=None, join=None, **_):
        slot = self._own_slot if slot is None else slot
        if "st" not in slot:
            raise RuntimeError("LSHExec.backward: no forward state for this call (backward run twice, or without its forward)")
        lyr = self.layer
        e = lyr.dim
        kept = "inp" in slot                       # STASH_STREAMS: the forward's own LayerNorm input / output, nothing reconstructed
        if kept:
            inp, pre = slot["inp"], slot["pre"]
        adrop = slot.get("adrop")
        xn, mean, rstd, wqkv, qkv, st, out, lse_tot, g = self._internals(inp, b, t, mask, slot["st"], slot["stash"],
                                                                         _NO_G if kept else slot["g"], pre, slot["qkv"], adrop)
        drop = slot["drop"]
        slot.clear()
        post = None if kept else residual(acc, g, lyr.to_out.bias, -1.0, next_norm, drop)   # reconstruct the stream (same dropout mask)
        dyb = _out_grad(d_acc, _grad(lyr.to_out.bias), drop, pre_cast)
        out2 = out.view(b * t, e)
        wgrad(_grad(lyr.to_out.weight), dyb, out2)
        delta = None
        if FUSE_DELTA and e == 64 * lyr.heads and dgrad_delta_ok(b * t, e):
            dout, delta = gemm_dgrad_delta(dyb, _bf16(lyr.to_out.weight), out2, t, lyr.heads)     # delta rides in the dgrad's epilogue
            dout = dout.view(b, t, e)
        else:
            dout = gemm(dyb, _bf16(lyr.to_out.weight), kn=True).view(b, t, e)
        dqkv = torch.empty_like(qkv)
        ops.lsh_attn_bwd(qkv[..., :e], qkv[..., e:], st, out, dout, lse_tot, lyr.heads, lyr.bucket_size, lyr.causal, mask,
                         dqkv=(dqkv[..., :e], dqkv[..., e:]), drop=adrop, delta=delta)
        dqkv2 = dqkv.view(b * t, 2 * e)
        gview, pair = self._wqkv_grad()
        if gview is not None:
            wgrad(gview, dqkv2, xn)
        else:
            _lib.note_general_path("weight gradient", "toqk / tov gradients are not neighbours in one buffer: library GEMM")
            full = torch.mm(dqkv2.t(), xn, out_dtype=torch.float32)
            pair[0].add_(full[:e])
            pair[1].add_(full[e:])
        dxn = gemm(dqkv2, wqkv, kn=True)
        return post, ln_bwd(dxn, inp, mean, rstd, self.norm, d_inp, next_cast, dx_in=d_src, join=join)


class FFNExec:
    """[Chunk(] WithNorm(LayerNorm, FeedForward) [)]: acc += W2 relu(W1 LN(inp) + b1) + b2."""

    def __init__(self, mod):
        wn = mod.fn if hasattr(mod, "chunks") else mod
        self.norm = wn.norm
        self.l1, self.l2 = wn.fn.net[0], wn.fn.net[3]
        self._own_slot = {}

    @staticmethod
    def supported(mod) -> bool:
        wn = mod.fn if hasattr(mod, "chunks") else mod
        lin1, lin2 = wn.fn.net[0], wn.fn.net[3]
        # both layers are rtts_gemm_nt launches: every width a multiple of its 64-wide stages / tiles
        return wn.fn.net[2].p == 0.0 and all(v % 64 == 0 for v in (lin1.in_features, lin1.out_features, lin2.out_features))

    def _internals(self, inp, g=None, pre=None, h=None):
        """h: (activation, gate words | None) kept from the forward, or None: computed here."""
        xn, mean, rstd = pre if pre is not None else ln_fwd(inp, self.norm)
        if h is None:   # bias + ReLU ride in the GEMM's epilogue (fp32 accumulate and fp32 bias, one rounding to bf16), which also
            #             leaves the 1-bit gate for the input gradient (the backward then does not re-read the activation for it)
            words = gate_words(xn.shape[0], self.l1.weight.shape[0], xn.device)
            h = (gemm(xn, _bf16(self.l1.weight), bias=self.l1.bias, relu=True, words=words), words)
        if g is None:
            g = gemm(h[0], _bf16(self.l2.weight))
        return xn, mean, rstd, h, g

    def forward(self, acc, inp, b, t, pre=None, next_norm=None, slot=None, keep_streams=False, fresh_acc=False, **_):
        slot = self._own_slot if slot is None else slot
        xn, mean, rstd, h, g = self._internals(inp, pre=pre)
        slot.clear()
        slot.update(g=g if STASH_BLOCK_OUTPUT else None, h=h if STASH_PROJECTIONS else None)
        return residual(acc, g, self.l2.bias, 1.0, next_norm, out=_keep_streams(keep_streams, slot, acc, inp, xn, mean, rstd, fresh_acc))

    def backward(self, acc, inp, d_acc, d_inp, b, t, pre=None, next_norm=None, pre_cast=None, next_cast=None, slot=None, d_src=None,
                 join=None, **_):
        slot = self._own_slot if slot is None else slot
        if "g" not in slot:
            raise RuntimeError("FFNExec.backward: no forward state for this call (backward run twice, or without its forward)")
        kept = "inp" in slot
        if kept:
            inp, pre = slot["inp"], slot["pre"]
        xn, mean, rstd, (h, words), g = self._internals(inp, _NO_G if kept else slot["g"], pre, slot["h"])
        slot.clear()
        post = None if kept else residual(acc, g, self.l2.bias, -1.0, next_norm)
        dyb = _out_grad(d_acc, _grad(self.l2.bias), None, pre_cast)
        wgrad(_grad(self.l2.weight), dyb, h)
        # the ReLU gate (1 bit per element, left by the forward's epilogue) and the partial column sums of db1 ride in the
        # epilogue of the dgrad GEMM
        dh = gemm(dyb, _bf16(self.l2.weight), kn=True, gate=h if words is None else True, gate_bias_grad=_grad(self.l1.bias), words=words)
        wgrad(_grad(self.l1.weight), dh, xn)
        dxn = gemm(dh, _bf16(self.l1.weight), kn=True)
        return post, ln_bwd(dxn, inp, mean, rstd, self.norm, d_inp, next_cast, dx_in=d_src, join=join)


class XAttnExec:
    """WithNorm(LayerNorm, MultiheadAttentionWrapper): acc += out_proj(MHA(q=LN(inp), k=v=keys))."""

    def __init__(self, withnorm):
        self.norm = withnorm.norm
        self.mha = withnorm.fn.layer
        self._own_slot = {}   # per-call keys: stash ((o, lse)), g (f(x)), pdrop ((p, seed) of the dropout on the attention
        #                       probabilities -- NOT an output dropout)

    @staticmethod
    def supported(withnorm) -> bool:
        m = withnorm.fn.layer      # dropout on the attention probabilities runs inside the kernels (counter-hash mask)
        return m.bias_k is None and not m.add_zero_attn and m._qkv_same_embed_dim and m.embed_dim % 64 == 0 and \
            m.embed_dim // m.num_heads == 64

    def _internals(self, inp, b, t, keys_bf16, kvalid, stash=None, g=None, pre=None, drop=None, proj=None):
        m = self.mha
        e, h = m.embed_dim, m.num_heads
        tk = keys_bf16.shape[0] // b
        w, bias = _bf16(m.in_proj_weight), m.in_proj_bias
        xn, mean, rstd = pre if pre is not None else ln_fwd(inp, self.norm)
        if proj is not None:
            q, kv = proj
        elif GROUP_XATTN:      # the two in_proj products of nn.MultiheadAttention (reformer.py:161-186) in one launch
            q, kv = gemm_group([dict(a=xn, w=w[:e], bias=bias[:e]), dict(a=keys_bf16, w=w[e:], bias=bias[e:])])
        else:
            q, kv = gemm(xn, w[:e], bias=bias[:e]), gemm(keys_bf16, w[e:], bias=bias[e:])
        if stash is not None:
            o, lse = stash
        else:
            o = torch.empty(b * t, e, dtype=torch.bfloat16, device=inp.device)
            lse = torch.empty(b * h, t, dtype=torch.float32, device=inp.device)
            p, seed = drop if drop else (0.0, 0)
            _lib.call("rtts_xattn_fwd", q.data_ptr(), e, kv.data_ptr(), 2 * e, None if kvalid is None else kvalid.data_ptr(), b, h, t,
                      tk, e // h, o.data_ptr(), e, lse.data_ptr(), float(p), seed, seed_base(inp.device).data_ptr(), _s())
        if g is None:
            g = gemm(o, _bf16(m.out_proj.weight))
        return xn, mean, rstd, w, q, kv, o, lse, g, tk

    def forward(self, acc, inp, b, t, keys_bf16=None, kvalid=None, pre=None, next_norm=None, slot=None, keep_streams=False,
                fresh_acc=False, **_):
        slot = self._own_slot if slot is None else slot
        p = self.mha.dropout if self.mha.training else 0.0
        pdrop = (p, next_seed()) if p > 0.0 else None
        xn, mean, rstd, _, q, kv, o, lse, g, _ = self._internals(inp, b, t, keys_bf16, kvalid, pre=pre, drop=pdrop)
        slot.clear()
        slot.update(stash=(o, lse) if STASH_ATTENTION else None, g=g if STASH_BLOCK_OUTPUT else None, pdrop=pdrop,
                    proj=(q, kv) if STASH_PROJECTIONS else None)
        return residual(acc, g, self.mha.out_proj.bias, 1.0, next_norm, out=_keep_streams(keep_streams, slot, acc, inp, xn, mean, rstd, fresh_acc))

    def backward(self, acc, inp, d_acc, d_inp, b, t, keys_bf16=None, kvalid=None, dkeys=None, pre=None, next_norm=None,
                 pre_cast=None, next_cast=None, slot=None, d_src=None, join=None, **_):
        slot = self._own_slot if slot is None else slot
        if "pdrop" not in slot:
            raise RuntimeError("XAttnExec.backward: no forward state for this call (backward run twice, or without its forward)")
        m = self.mha
        e, h = m.embed_dim, m.num_heads
        drop = slot["pdrop"]
        kept = "inp" in slot
        if kept:
            inp, pre = slot["inp"], slot["pre"]
        xn, mean, rstd, w, q, kv, o, lse, g, tk = self._internals(inp, b, t, keys_bf16, kvalid, slot["stash"],
                                                                  _NO_G if kept else slot["g"], pre, drop, slot["proj"])
        slot.clear()
        post = None if kept else residual(acc, g, m.out_proj.bias, -1.0, next_norm)
        dyb = _out_grad(d_acc, _grad(m.out_proj.bias), None, pre_cast)
        wgrad(_grad(m.out_proj.weight), dyb, o)
        dev = inp.device
        if FUSE_DELTA and e == 64 * h and dgrad_delta_ok(b * t, e):
            do, delta = gemm_dgrad_delta(dyb, _bf16(m.out_proj.weight), o, t, h)
        else:
            do = gemm(dyb, _bf16(m.out_proj.weight), kn=True)
            delta = torch.empty(b * h, t, dtype=torch.float32, device=dev)
            _lib.call("rtts_lsh_bwd_delta", o.data_ptr(), e, do.data_ptr(), e, b, h, t, e // h, delta.data_ptr(), _s())
        dq = torch.empty(b * t, e, dtype=torch.bfloat16, device=dev)
        nqb = t // 128
        part = torch.empty(nqb, b * tk, 2 * e, dtype=torch.bfloat16, device=dev)
        dp, dseed = drop if drop else (0.0, 0)
        nkc = _lib.load().rtts_xattn_key_chunks(tk)          # > 256 keys: worked in chunks, each with its share of dQ
        dq_chunks = torch.empty(nkc, b * t, e, dtype=torch.bfloat16, device=dev) if nkc > 1 else None
        _lib.call("rtts_xattn_bwd", q.data_ptr(), e, kv.data_ptr(), 2 * e, None if kvalid is None else kvalid.data_ptr(),
                  do.data_ptr(), e, lse.data_ptr(), delta.data_ptr(), b, h, t, tk, e // h, dq.data_ptr(), e, part.data_ptr(),
                  float(dp), dseed, seed_base(dev).data_ptr(), None if dq_chunks is None else dq_chunks.data_ptr(), _s())
        dkv = torch.empty(b * tk, 2 * e, dtype=torch.bfloat16, device=dev)
        _lib.call("rtts_sum_slabs", part.data_ptr(), nqb, dkv.numel(), dkv.data_ptr(), _s())
        gb, gw = _grad(m.in_proj_bias), _grad(m.in_proj_weight)
        colsum_bf16(dq, gb[:e])
        colsum_bf16(dkv, gb[e:])
        wgrad(gw[:e], dq, xn)
        wgrad(gw[e:], dkv, keys_bf16)
        # dkeys (fp32) += dkv W_kv in the product's own epilogue (no separate add launch).  Grouping it with dxn = dq W_q into one
        # launch measured no gain (28.7 against 27.7 us for the two launches + the add: a K = 1024 tile beside K = 512 tiles on two
        # workgroups per CU; profiles/r04_gemm_nt_persistent_ab.log), so the two products stay two launches
        if GROUP_XATTN_BWD:
            dxn, _ = gemm_group([dict(a=dq, w=w[:e]), dict(a=dkv, w=w[e:], into=dkeys)], kn=True)
        else:
            dxn = gemm(dq, w[:e], kn=True)
            gemm_group([dict(a=dkv, w=w[e:], into=dkeys)], kn=True)
        nxt = ln_bwd(dxn, inp, mean, rstd, self.norm, d_inp, next_cast, dx_in=d_src, join=join)
        return post, nxt


# ------------------------------------------------------------------------------------------ stacks
# LayerNorm chaining: every executor reads (through its LayerNorm) the stream the previous executor has just updated
# (forward) or reconstructed (backward), so that executor's residual epilogue also emits the next one's LayerNorm
# (rtts_residual_ln).  FUSE_RESIDUAL_LN = False launches the two kernels separately.
FUSE_RESIDUAL_LN = __import__("os").environ.get("RTTS_FUSE_RESIDUAL_LN", "1") != "0"


def _flat_calls(steps):
    """[(executor, step index, "f" | "g")] in forward order."""
    calls = []
    for i, (kind, f, g, _) in enumerate(steps):
        if kind == "half":
            calls.append((f, i, "f"))
        elif kind == "block":
            calls.append((f, i, "f"))
            calls.append((g, i, "g"))
    return calls


class _Chain:
    """Carries LayerNorm(acc) from the executor that wrote ``acc`` to the one that reads it next."""

    def __init__(self, calls, reverse: bool, slots=None):
        order = list(reversed(calls)) if reverse else calls
        self.next_norm, self.next_exec, self.next_slot = {}, {}, {}
        for (ex, i, w), (nxt, ni, nw) in zip(order, order[1:]):
            self.next_norm[(i, w)] = nxt.norm
            self.next_exec[(i, w)] = nxt
            self.next_slot[(i, w)] = slots[(ni, nw)] if slots is not None else {}
        self.pre, self.ptr = None, None
        self.cast, self.cast_ptr = None, None

    def args(self, i, which, inp):
        pre = self.pre if (self.pre is not None and self.ptr == inp.data_ptr()) else None
        return dict(pre=pre, next_norm=self.next_norm.get((i, which)) if FUSE_RESIDUAL_LN else None)

    def done(self, post, acc):
        self.pre, self.ptr = post, (acc.data_ptr() if post is not None else None)

    # backward only: the gradient stream a LayerNorm backward completes (d_inp) is the output gradient (d_acc) of the
    # executor that runs next in backward order; its bf16 cast + bias partial sums ride in that LayerNorm backward
    def grad_args(self, i, which, d_acc):
        pre_cast = self.cast if (self.cast is not None and self.cast_ptr == d_acc.data_ptr()) else None
        nxt = self.next_exec.get((i, which)) if FUSE_RESIDUAL_LN else None
        # "drop" = an executor's dropout on its OUTPUT (the LSH layers' post_attn_dropout); only that one masks d_acc
        return dict(pre_cast=pre_cast, next_cast=None if nxt is None else (self.next_slot[(i, which)].get("drop"),))

    def grad_done(self, nxt, d_inp):
        self.cast, self.cast_ptr = nxt, (d_inp.data_ptr() if nxt is not None else None)


def build_program(seq) -> Optional[List[tuple]]:
    """Translate a ReversibleSequence into a list of ("f"|"g"|"half"|"swap", executor) steps; None if
    some block needs the general (autograd) path (active dropout inside a block, exotic options)."""
    from .model.reformer import Chunk, LSHSelfAttentionWrapper, MultiheadAttentionWrapper, WithNorm
    from .model.reversible import ReversibleBlock, ReversibleHalfResidual, ReversibleSwap

    def make(net):
        inner = net.fn if isinstance(net, Chunk) else net
        if not isinstance(inner, WithNorm):
            return None
        if isinstance(inner.fn, LSHSelfAttentionWrapper):
            return LSHExec(inner) if LSHExec.supported(inner) else None
        if isinstance(inner.fn, MultiheadAttentionWrapper):
            return XAttnExec(inner) if XAttnExec.supported(inner) else None
        if hasattr(inner.fn, "net"):
            return FFNExec(net) if FFNExec.supported(net) else None
        return None

    prog = []
    for blk in seq.blocks:
        if isinstance(blk, ReversibleBlock):
            f, g = make(blk.f.net), make(blk.g.net)
            if f is None or g is None:
                return None
            prog.append(("block", f, g))
        elif isinstance(blk, ReversibleHalfResidual):
            f = make(blk.f.net)
            if f is None:
                return None
            prog.append(("half", f, None))
        elif isinstance(blk, ReversibleSwap):
            prog.append(("swap", None, None))
        else:
            return None
    return prog


def _mask_u8(m, cache):
    """bool (B,T) validity mask -> uint8, converted once per distinct tensor of a stack pass."""
    if m is None:
        return None
    if id(m) not in cache:
        # a bool tensor already is one byte of 0/1 per element: reinterpret instead of converting (no launch)
        cache[id(m)] = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8).contiguous()
    return cache[id(m)]


def _step_kwargs(kind, kwargs, extra, cache):
    """Reference kwargs routing (reformer.py:89-90,147-153) -> executor keyword arguments."""
    if kind == "block":
        return dict(mask=_mask_u8(kwargs.get("f_args", {}).get("input_mask"), cache)), {}
    out = {}
    if "input_mask" in kwargs and kwargs["input_mask"] is not None:
        out["mask"] = _mask_u8(kwargs["input_mask"], cache)
    if "key" in kwargs:
        out.update(extra)
    return out, None


class FusedStackFn(torch.autograd.Function):
    """out = sum of the two streams after the stack; keeps only the final streams."""

    @staticmethod
    def forward(ctx, x, context, seq, kwargs_list):
        b, t, d = x.shape
        prog = seq._program
        with torch.no_grad():
            kept = _streams_kept()
            if kept:
                # every sublayer writes its updated stream to a new buffer: both streams simply start as x itself
                s1 = x.detach().reshape(b * t, d)
                s1 = s2 = s1 if s1.dtype == torch.float32 and s1.is_contiguous() else s1.float().contiguous()
            else:
                # both streams start as x ITSELF (no copy): an executor's first update of a stream that is still x goes to a
                # fresh buffer (``fresh_acc``), later ones are in place -- the reference copies x twice (reformer.py:85,143)
                s1 = x.detach().reshape(b * t, d)
                s1 = s2 = s1 if s1.dtype == torch.float32 and s1.is_contiguous() else s1.float().contiguous()
            own1 = own2 = kept            # does the stream have a buffer of its own?  (kept: every update is out of place anyway)
            if not kept and COPY_STREAMS:          # A/B switch: round 2's form, two copies of x up front
                s1, s2, own1, own2 = s1.clone(), s1.clone(), True, True
            extra = {}
            if context is not None:
                kpm = next((k.get("key_padding_mask") for k in kwargs_list if "key" in k), None)
                kb = bf16_twin(context)                           # the encoder stack's own bf16 copy of its output
                if kb is None and getattr(context, "_rtts_ready", None) is not None:
                    torch.cuda.current_stream().wait_event(context._rtts_ready)      # no twin: the cast below reads the keys NOW
                kb = kb.view(-1, d) if kb is not None else context.detach().reshape(-1, d).to(torch.bfloat16)
                kv = None
                if kpm is not None:
                    kv = getattr(kpm, "_rtts_not", None)          # the validity mask the caller inverted into kpm: no second inversion
                    kv = ~kpm if kv is None or kv.shape != kpm.shape or kv.device != kpm.device else kv
                    kv = kv.contiguous().view(torch.uint8) if kv.dtype == torch.bool else kv.to(torch.uint8).contiguous()
                extra = dict(keys_bf16=kb, kvalid=kv)
            steps, mask_cache = [], {}
            for (kind, f, g), kwargs in zip(prog, kwargs_list):
                kw, kwg = _step_kwargs(kind, kwargs, extra, mask_cache)
                steps.append((kind, f, g, kw))
            chain = _Chain(_flat_calls(steps), reverse=False)
            slots = {(i, w): {} for (_, i, w) in _flat_calls(steps)}      # forward -> backward state of THIS call, one per executor
            keys_ready = getattr(context, "_rtts_ready", None) if context is not None else None
            for i, (kind, f, g, kw) in enumerate(steps):
                if kind == "swap":
                    s1, s2, own1, own2 = s2, s1, own2, own1
                elif kind == "half":
                    if keys_ready is not None and "keys_bf16" in kw:
                        # the encoder ran on a stream of its own beside the decoder's first blocks (Trainer, overlapped step):
                        # the first cross-attention is where the two meet
                        stamp("decoder: first cross-attention reached (before the wait for the encoder)")
                        torch.cuda.current_stream().wait_event(keys_ready)
                        keys_ready = None
                        stamp("decoder: first cross-attention starts (encoder output there)")
                    post = f.forward(s1, s2, b, t, slot=slots[(i, "f")], keep_streams=kept, fresh_acc=not own1, **kw,
                                     **chain.args(i, "f", s2))
                    s1, own1 = slots[(i, "f")].pop("acc_out", s1), True
                    chain.done(post, s1)
                else:
                    post = f.forward(s1, s2, b, t, slot=slots[(i, "f")], keep_streams=kept, fresh_acc=not own1, **kw,
                                     **chain.args(i, "f", s2))
                    s1, own1 = slots[(i, "f")].pop("acc_out", s1), True
                    chain.done(post, s1)
                    post = g.forward(s2, s1, b, t, slot=slots[(i, "g")], keep_streams=kept, fresh_acc=not own2, **chain.args(i, "g", s1))
                    s2, own2 = slots[(i, "g")].pop("acc_out", s2), True
                    chain.done(post, s2)
            if not (own1 and own2) and not kept:
                # a stack that never updated one of its streams: the backward reconstructs in place, so it must not be x
                s1, s2 = (s1 if own1 else s1.clone()), (s2 if own2 else s2.clone())
            if x.dtype == torch.float32 and (b * t * d) % 4 == 0:
                # the sum of the streams in fp32 (the autograd value) and in bf16 (what the heads / the cross attention's key
                # projection read), one launch
                out = torch.empty(b, t, d, dtype=torch.float32, device=x.device)
                twin = torch.empty(b * t, d, dtype=torch.bfloat16, device=x.device)
                _lib.call("rtts_sum_streams", s1.data_ptr(), s2.data_ptr(), b * t * d, out.data_ptr(), twin.data_ptr(), _s())
                out._rtts_bf16 = (twin, out._version)
            else:
                out = (s1 + s2).view(b, t, d)
        ctx.state = (s1, s2, steps, extra, b, t, d, context is not None, seq, slots)
        # with the streams kept, the first sublayer's saved input IS the caller's x (no copy): an in-place write to x between this
        # forward and its backward would silently change the gradients -- remember the version, check it in the backward
        ctx.x_guard = (x, x._version) if kept else None
        return out

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            gen = stack_backward_steps(ctx, dout)
            try:
                while True:
                    seq, done = next(gen)
                    if seq.block_done_hook is not None:
                        for j in done:
                            seq.block_done_hook(seq, j)
            except StopIteration as fin:
                dx, dkeys = fin.value
        return dx, dkeys, None, None


def stack_backward_steps(ctx, dout, complete_layers: Optional[bool] = None, notify_dkeys: bool = False, flush_at: int = 7):
    """The backward of one FusedStackFn.forward as a GENERATOR: runs the blocks in reverse and yields ``(seq, [block indices])``
    every time a layer's worth of weight gradients has been flushed -- from that point the listed blocks' slices of the flat
    gradient buffer are final -- and returns ``(dx, dkeys)`` through StopIteration.  FusedStackFn.backward drives it to the end
    and calls the block-done hooks at every stop; the data-parallel trainer's segmented capture (Trainer.capture) drives it
    itself, on the calling thread, and closes one hipGraph / opens the next at every stop, so that a decoder layer's
    gradient all-reduce can be issued while the next layer's backward replays.  Call under ``torch.no_grad()``.
    ``complete_layers``: also finalise the deferred column sums (bias / LayerNorm gradients) at every stop (default: only
    when a block-done hook is waiting for the gradients).  ``notify_dkeys``: additionally yield ``(seq, ("dkeys", dkeys))`` as soon
    as the gradient of the cross-attention keys is complete (after the LOWEST cross-attention block's backward): the encoder's
    backward can start there, beside the remaining decoder blocks.  ``flush_at``: weight gradients queued before a stop (7 = one
    decoder layer's worth -- LSH 2, cross-attention 3, feed-forward 2; 4 = one encoder block's)."""
    if ctx.state is None:
        raise RuntimeError("FusedStackFn.backward: this forward's state was already consumed (the streams are rebuilt in "
                           "place; a second backward through the same stack call is not possible)")
    s1, s2, steps, extra, b, t, d, has_ctx, seq, slots = ctx.state
    ctx.state = None
    guard = getattr(ctx, "x_guard", None)
    if guard is not None and guard[0]._version != guard[1]:
        raise RuntimeError("the input of a reversible stack was modified in place between its forward and its backward: with the streams "
                           "kept (engine.STASH_STREAMS) the first sublayer reads it again -- clone it before writing to it")
    # both gradient streams start as dout ITSELF (read only): the first LayerNorm backward that completes a stream writes it to a
    # fresh buffer (rtts_ln_bwd_to), later ones accumulate in place -- no (2, B*T, d) copy of dout
    g1 = dout.detach().reshape(b * t, d)
    g1 = g2 = g1 if g1.dtype == torch.float32 and g1.is_contiguous() else g1.float().contiguous()
    gown1 = gown2 = False
    if COPY_STREAMS:
        g1, g2, gown1, gown2 = g1.clone(), g1.clone(), True, True

    def target(g, owned):
        """-> (buffer the executor accumulates into, its source when that differs)"""
        return (g, None) if owned else (torch.empty_like(g), g)
    steal = STEAL if (STEAL is not None and (torch.cuda.current_device(), _s()) == STEAL["src"]) else None
    if steal is not None:
        flush_at = min(flush_at, 4)      # this stack's weight gradients are handed to another stream: one hand-over (one event) per block,
        #                                  so that the taker can start on the first blocks' while the chain is still in the last ones
        steal["armed"] = True            # only the STACK's: what the prenet queues behind it stays on this stream (its event would be the
        #                                  end of the chain, and the taker would wait for that)
    join_last = JOIN_STREAMS and steps[0][0] != "swap"

    def joined(i):
        """The stack's LAST LayerNorm backward (block 0's f) writes d(input) = g1 + g2 itself: the other stream joins in its pass."""
        return g1 if (i == 0 and join_last and g1 is not dst) else None
    dkeys = None
    if has_ctx:
        dkeys = torch.zeros(extra["keys_bf16"].shape, dtype=torch.float32, device=dout.device)
        extra = dict(extra, dkeys=dkeys)
    done = []
    chain = _Chain(_flat_calls(steps), reverse=True, slots=slots)
    key_blocks = [i for i, st in enumerate(steps) if st[0] == "half" and "keys_bf16" in st[3]]
    for i in range(len(steps) - 1, -1, -1):
        kind, f, g, kw = steps[i]
        if kind == "swap":
            s1, s2, g1, g2, gown1, gown2 = s2, s1, g2, g1, gown2, gown1
        elif kind == "half":
            if "keys_bf16" in kw:
                kw = dict(kw, dkeys=dkeys)
            dst, src = target(g2, gown2)
            post, nxt = f.backward(s1, s2, g1, dst, b, t, slot=slots[(i, "f")], d_src=src, join=joined(i), **kw, **chain.args(i, "f", s2),
                                   **chain.grad_args(i, "f", g1))
            g2, gown2 = dst, True
            chain.done(post, s1)
            chain.grad_done(nxt, g2)
            if notify_dkeys and key_blocks and i == key_blocks[0]:
                yield seq, ("dkeys", dkeys.view(b, -1, d))
        else:
            dst, src = target(g1, gown1)
            post, nxt = g.backward(s2, s1, g2, dst, b, t, slot=slots[(i, "g")], d_src=src, **chain.args(i, "g", s1),
                                   **chain.grad_args(i, "g", g2))
            g1, gown1 = dst, True
            chain.done(post, s2)
            chain.grad_done(nxt, g1)
            dst, src = target(g2, gown2)
            post, nxt = f.backward(s1, s2, g1, dst, b, t, slot=slots[(i, "f")], d_src=src, join=joined(i), **kw, **chain.args(i, "f", s2),
                                   **chain.grad_args(i, "f", g1))
            g2, gown2 = dst, True
            chain.done(post, s1)
            chain.grad_done(nxt, g2)
        done.append(i)
        hook = seq.block_done_hook
        waited_for = (hook is not None and getattr(hook, "active", lambda: True)()) if complete_layers is None else bool(complete_layers)
        if WGRAD_FLUSH_PER_LAYER or waited_for:
            if pending_wgrads() >= flush_at or i == 0:
                # one grouped launch per layer's worth of weight gradients; only then are the finished blocks'
                # gradient slices final, so their all-reduce hooks run here
                flush_wgrad(colsums=waited_for)
                yield seq, list(done)
                done.clear()
        elif pending_wgrads() >= WGRAD_MAX_PENDING:
            flush_wgrad()      # nobody waits for a block's gradients (one GPU): they go out in a few large groups at the
            #                    end of the backward; this only bounds the operands held
    if steal is not None:
        steal["armed"] = False
    dx = (g2 if join_last else g1 + g2).view(b, t, d)
    return dx, (None if dkeys is None else dkeys.view(b, -1, d))


class _ManualCtx:
    """Stands in for autograd's ctx when a stack's forward / backward are driven by hand (Trainer's segmented capture)."""
    state = None

