"""Tensor-level wrappers of the C ABI: shape/dtype/device checks on the host, raw
pointers and the current HIP stream handed to librtts_hip.so.  PyTorch is used for
device memory and streams only."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib


class _Timing:
    """Optional HIP-event timing of named launches on the stream they are launched on (bench.py's roofline legs).  Off by
    default; events are resolved in ``summary`` after a device sync.  A tag is enabled by name or by prefix
    (``enable("rtts_gemm_nt")`` records every ``rtts_gemm_nt/<shape>``)."""

    def __init__(self):
        self.tags = ()
        self.records = {}

    def enable(self, *tags: str):
        self.tags, self.records = tuple(tags), {}

    def disable(self):
        self.tags = ()

    def start(self, tag: str):
        if not self.tags or not any(tag == t or tag.startswith(t + "/") for t in self.tags):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return (tag, ev)

    def stop(self, started, work: float):
        if started is None:
            return
        tag, ev = started
        end = torch.cuda.Event(enable_timing=True)
        end.record(torch.cuda.current_stream())
        self.records.setdefault(tag, []).append((ev, end, work))

    def summary(self, tag: str):
        """-> (average launch ms, launches, algorithmic work per launch) over every record whose tag is ``tag`` or starts
        with ``tag + "/"``."""
        recs = [r for k, v in self.records.items() if k == tag or k.startswith(tag + "/") for r in v]
        if not recs:
            return 0.0, 0, 0.0
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b, _ in recs]
        return sum(ms) / len(ms), len(ms), sum(f for _, _, f in recs) / len(ms)


TIMING = _Timing()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check_rows(x: torch.Tensor, name: str) -> int:
    """(B, T, W) bf16 view whose last dim is contiguous and whose (B,T) dims collapse to rows."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 3):
        raise ValueError(f"{name}: expected a CUDA bfloat16 (B,T,W) tensor, got {x.dtype} {tuple(x.shape)} on {x.device}")
    if x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        raise ValueError(f"{name}: rows must be contiguous with a single row stride, got strides {x.stride()}")
    return x.stride(1)


def lsh_hash_sort(qk: torch.Tensor, rotations: torch.Tensor, heads: int, bucket_size: int,
                  want_buckets: bool = False, want_undo: bool = False):
    """qk (B,T,H*dh) bf16, rotations (1|B*H, dh, R, nb/2) f32 -> st (B*H,R,T) i32 [, buckets, undo]."""
    ld = _check_rows(qk, "qk")
    b, t, d = qk.shape
    dh = d // heads
    if t % (2 * bucket_size) != 0:
        raise AssertionError(f"Sequence length ({t}) needs to be divisible by target bucket size  x 2 - {bucket_size * 2}")
    if not (rotations.is_cuda and rotations.dtype == torch.float32 and rotations.is_contiguous() and rotations.dim() == 4):
        raise ValueError("rotations: expected a contiguous CUDA float32 (rows, dh, n_hashes, n_buckets/2) tensor")
    rows, rdh, n_hashes, half = rotations.shape
    if rdh != dh or 2 * half != t // bucket_size or rows not in (1, b * heads):
        raise ValueError(f"rotations shape {tuple(rotations.shape)} does not match dh={dh}, n_buckets={t // bucket_size}")
    st = torch.empty(b * heads, n_hashes, t, dtype=torch.int32, device=qk.device)
    buckets = torch.empty_like(st) if want_buckets else None
    undo = torch.empty_like(st) if want_undo else None
    ev = TIMING.start(f"rtts_lsh_hash_sort/nb{t // bucket_size}")
    _lib.call("rtts_lsh_hash_sort", qk.data_ptr(), ld, rotations.data_ptr(), rows, b, heads, t, dh, n_hashes, bucket_size,
              _ptr(buckets), st.data_ptr(), _ptr(undo), _stream())
    # SURVEY.md 8(d), per token and head: hash reads dh*2 B and writes n_hashes*4 B of bucket ids, the sort reads and writes
    # n_hashes*(4+4) B (permutation + inverse) -- the algorithm's bytes, whatever the fused kernel keeps on chip
    TIMING.stop(ev, float(b * heads * t) * (dh * 2 + n_hashes * 4 + n_hashes * 8))
    return st, buckets, undo


def _check_mask(mask: Optional[torch.Tensor], b: int, t: int, device) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if mask.shape != (b, t):
        raise ValueError(f"input_mask: expected shape {(b, t)}, got {tuple(mask.shape)}")
    return mask.to(device=device, dtype=torch.uint8).contiguous()


def _drop_args(drop, device):
    """(p, seed) | None -> (p, seed, device seed word) of a counter-hash dropout site."""
    if not drop or drop[0] <= 0.0:
        return 0.0, 0, None
    from ._seeds import seed_base
    return float(drop[0]), int(drop[1]) & 0xFFFFFFFF, seed_base(device).data_ptr()


def lsh_attn_fwd(qk, v, st, heads: int, bucket_size: int, causal: bool, mask=None, drop=None):
    """-> o (B*H,R,T,dh) bf16, lse (B*H,R,T) f32; rows already at unsorted positions.  ``drop`` = (p, seed): dropout on the
    attention probabilities (the layer's ``dropout`` knob); pass the same pair to ``lsh_attn_bwd``."""
    ld = _check_rows(qk, "qk")
    if _check_rows(v, "v") != ld or v.shape != qk.shape:
        raise ValueError("qk and v must share shape and row stride")
    b, t, d = qk.shape
    dh = d // heads
    n_hashes = st.shape[1]
    if st.shape != (b * heads, n_hashes, t) or st.dtype != torch.int32 or not st.is_contiguous():
        raise ValueError("st: expected contiguous int32 (B*H, n_hashes, T)")
    mask = _check_mask(mask, b, t, qk.device)
    o = torch.empty(b * heads, n_hashes, t, dh, dtype=torch.bfloat16, device=qk.device)
    lse = torch.empty(b * heads, n_hashes, t, dtype=torch.float32, device=qk.device)
    ev = TIMING.start(f"rtts_lsh_attn_fwd/bs{bucket_size}")
    _lib.call("rtts_lsh_attn_fwd", qk.data_ptr(), v.data_ptr(), ld, st.data_ptr(), _ptr(mask), b, heads, t, dh, n_hashes,
              bucket_size, int(causal), o.data_ptr(), lse.data_ptr(), *_drop_args(drop, qk.device), _stream())
    # two MFMA products (Q K^T and P V) of 2*bs*(2bs)*dh FLOP per chunk, n_hashes*T/bs chunks per head
    TIMING.stop(ev, 2.0 * 2.0 * bucket_size * (2 * bucket_size) * dh * (n_hashes * t // bucket_size) * b * heads)
    return o, lse


def lsh_combine_fwd(o, lse, batch: int, heads: int, out: Optional[torch.Tensor] = None):
    """-> out (B,T,H*dh) bf16 (merged heads), lse_tot (B*H,T) f32."""
    bh, n_hashes, t, dh = o.shape
    if out is None:
        out = torch.empty(batch, t, heads * dh, dtype=torch.bfloat16, device=o.device)
    ld_out = _check_rows(out, "out")
    lse_tot = torch.empty(bh, t, dtype=torch.float32, device=o.device)
    _lib.call("rtts_lsh_combine_fwd", o.data_ptr(), lse.data_ptr(), batch, heads, t, dh, n_hashes, out.data_ptr(), ld_out,
              lse_tot.data_ptr(), _stream())
    return out, lse_tot


def lsh_attn_bwd(qk, v, st, out, dout, lse_tot, heads: int, bucket_size: int, causal: bool, mask=None,
                 dqkv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, drop=None, delta: Optional[torch.Tensor] = None):
    """Backward of hash-sorted attention + round combine.  -> dqk, dv (B,T,H*dh) bf16.  ``delta`` (B*H, T) fp32 = rowsum(out *
    dout) per head when the caller already has it (the to_out input-gradient GEMM's epilogue 5), else computed here."""
    ld = _check_rows(qk, "qk")
    if _check_rows(v, "v") != ld:
        raise ValueError("qk and v must share a row stride")
    ld_out, ld_do = _check_rows(out, "out"), _check_rows(dout, "dout")
    b, t, d = qk.shape
    dh = d // heads
    n_hashes = st.shape[1]
    mask = _check_mask(mask, b, t, qk.device)
    dev = qk.device
    if delta is None:
        delta = torch.empty(b * heads, t, dtype=torch.float32, device=dev)
        _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), ld_out, dout.data_ptr(), ld_do, b, heads, t, dh, delta.data_ptr(), _stream())
    elif delta.shape != (b * heads, t) or delta.dtype != torch.float32 or not delta.is_contiguous():
        raise ValueError("delta: expected contiguous fp32 (B*H, T)")
    dqk_part = torch.empty(_lib.load().rtts_lsh_bwd_qk_slots(), b * heads, n_hashes, t, dh, dtype=torch.bfloat16, device=dev)
    dv_part = torch.empty(2, b * heads, n_hashes, t, dh, dtype=torch.bfloat16, device=dev)
    # the walking kernel writes most key rows once (slot 0 only) and flags the few that have a slot-1 partner
    walks = _lib.load().rtts_lsh_attn_bwd_run_length(b, heads, t, n_hashes, bucket_size) > 0
    flags = torch.empty(b * heads, n_hashes, t, dtype=torch.uint8, device=dev) if walks else None
    ev = TIMING.start(f"rtts_lsh_attn_bwd/bs{bucket_size}")
    _lib.call("rtts_lsh_attn_bwd", qk.data_ptr(), v.data_ptr(), ld, st.data_ptr(), _ptr(mask), dout.data_ptr(), ld_do,
              lse_tot.data_ptr(), delta.data_ptr(), b, heads, t, dh, n_hashes, bucket_size, int(causal), dqk_part.data_ptr(),
              dv_part.data_ptr(), _ptr(flags), *_drop_args(drop, dev), _stream())
    # five MFMA products of 2*bs*(2bs)*dh FLOP per chunk, n_hashes*T/bs chunks per head
    TIMING.stop(ev, 5.0 * 2.0 * bucket_size * (2 * bucket_size) * dh * (n_hashes * t // bucket_size) * b * heads)
    if dqkv is None:
        dqk = torch.empty(b, t, d, dtype=torch.bfloat16, device=dev)
        dv = torch.empty(b, t, d, dtype=torch.bfloat16, device=dev)
    else:
        dqk, dv = dqkv
    ld_d = _check_rows(dqk, "dqk")
    if _check_rows(dv, "dv") != ld_d:
        raise ValueError("dqk and dv must share a row stride")
    _lib.call("rtts_lsh_bwd_reduce", dqk_part.data_ptr(), dv_part.data_ptr(), b, heads, t, dh, n_hashes, dqk.data_ptr(),
              dv.data_ptr(), ld_d, _ptr(flags), _stream())
    return dqk, dv
