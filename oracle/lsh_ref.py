"""CPU restatement (fp32, eager PyTorch) of shared-QK LSH self-attention.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED for this
file: the reference obtains this arithmetic from ``reformer-pytorch==0.19.1``
(``/root/reference/requirements.txt:11``; constructed at
``reformer_tts/model/reformer.py:198-200``, called at ``:217``) and that package
is not present offline.  What is restated here is the published algorithm
(Kitaev et al. 2020, "Reformer", sections 2-3) with the constructor surface the
reference passes (``reformer_tts/model/config.py:11-27``); the step numbers in
the comments are those of SURVEY.md Appendix B.

Every function takes the random rotations as an explicit tensor so that the
HIP path, the C twin (``lsh_int.c``) and this file can be fed identical inputs.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn
import torch.nn.functional as F

MASK_VALUE = -torch.finfo(torch.float32).max  # step 8 (i), (ii)
SELF_VALUE = -5e4                             # step 8 (iii)


# --------------------------------------------------------------------------
# integer stages (steps 2-3)
# --------------------------------------------------------------------------
def hash_vectors(qk: torch.Tensor, rotations: torch.Tensor) -> torch.Tensor:
    """Step 2.  qk (BH,T,dh); rotations (1|BH, dh, R, nb/2) -> buckets (BH, R*T) int64.

    bucket = argmax over [xR, -xR]; round r is offset by r*nb so that rounds never
    share a bucket id.  (Same construction as HF modeling_reformer.py:727-768.)
    """
    bh, t, _ = qk.shape
    _, _, n_hashes, half = rotations.shape
    n_buckets = 2 * half
    rot = torch.einsum("btf,bfhi->bhti", qk, rotations.expand(bh, -1, -1, -1))
    both = torch.cat([rot, -rot], dim=-1)
    buckets = both.argmax(dim=-1)  # (BH, R, T)
    offs = (torch.arange(n_hashes, device=qk.device) * n_buckets).view(1, -1, 1)
    return (buckets + offs).reshape(bh, n_hashes * t)


def sort_buckets(buckets: torch.Tensor, seqlen: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Step 3.  Stable sort by (bucket, position).  Keys are unique, so the
    permutation is unique: ``sticker[j]`` = flat source index held by sorted slot
    j, ``undo_sort`` = its inverse."""
    n = buckets.shape[-1]
    ticker = torch.arange(n, device=buckets.device).unsqueeze(0).expand_as(buckets)
    keys = seqlen * buckets + (ticker % seqlen)
    _, sticker = keys.sort(dim=-1)
    undo_sort = torch.empty_like(sticker)
    undo_sort.scatter_(-1, sticker, ticker)
    return sticker, undo_sort


# --------------------------------------------------------------------------
# attention over sorted chunks (steps 4-11)
# --------------------------------------------------------------------------
def _with_previous_chunk(x: torch.Tensor) -> torch.Tensor:
    """Step 6: keys of chunk c are [chunk c, chunk c-1]; chunk 0 wraps to the last."""
    prev = torch.roll(x, shifts=1, dims=1)
    return torch.cat([x, prev], dim=2)


def lsh_attention_sorted(
    qk: torch.Tensor,
    v: torch.Tensor,
    sticker: torch.Tensor,
    undo_sort: torch.Tensor,
    bucket_size: int,
    n_hashes: int,
    causal: bool,
    input_mask: Optional[torch.Tensor] = None,
    return_parts: bool = False,
    keep: Optional[torch.Tensor] = None,
):
    """Steps 4-11 given the sort permutation.  qk, v: (BH,T,dh); mask (BH,T) bool.  ``keep`` (BH, chunks, bucket_size,
    2 * bucket_size): the dropout of step 9 as explicit keep-scales (0 or 1/(1-p)) on the chunk's probabilities, own chunk's
    keys first -- nn.Dropout's arithmetic with the mask given instead of drawn."""
    bh, t, dh = qk.shape
    n_chunks = n_hashes * (t // bucket_size)
    st = sticker % t                                            # step 4
    idx = st.unsqueeze(-1).expand(-1, -1, dh)
    sqk = qk.gather(1, idx)
    sv = v.gather(1, idx)
    bq_t = st.reshape(bh, n_chunks, bucket_size)
    bq = sqk.reshape(bh, n_chunks, bucket_size, dh)
    bv = sv.reshape(bh, n_chunks, bucket_size, dh)
    bk = F.normalize(bq, p=2, dim=-1)                           # step 5 (eps 1e-12)
    bk = _with_previous_chunk(bk)                               # step 6
    bv = _with_previous_chunk(bv)
    bkv_t = _with_previous_chunk(bq_t)
    dots = torch.einsum("bcie,bcje->bcij", bq, bk) * (dh ** -0.5)   # step 7
    if input_mask is not None:                                  # step 8 (i)
        mq = input_mask.gather(1, st).reshape(bh, n_chunks, bucket_size)
        mk = _with_previous_chunk(mq)
        dots = dots.masked_fill(~(mq[:, :, :, None] & mk[:, :, None, :]), MASK_VALUE)
    if causal:                                                  # step 8 (ii)
        dots = dots.masked_fill(bq_t[:, :, :, None] < bkv_t[:, :, None, :], MASK_VALUE)
    dots = dots.masked_fill(bq_t[:, :, :, None] == bkv_t[:, :, None, :], SELF_VALUE)  # (iii)
    lse = torch.logsumexp(dots, dim=-1, keepdim=True)           # step 9
    probs = torch.exp(dots - lse)
    if keep is not None:
        probs = probs * keep
    bo = torch.einsum("bcij,bcje->bcie", probs, bv)
    so = bo.reshape(bh, n_hashes * t, dh)
    slse = lse.reshape(bh, n_hashes * t)
    o = so.gather(1, undo_sort.unsqueeze(-1).expand(-1, -1, dh))    # step 10
    lse_u = slse.gather(1, undo_sort)
    o = o.reshape(bh, n_hashes, t, dh)
    lse_u = lse_u.reshape(bh, n_hashes, t, 1)
    w = torch.exp(lse_u - torch.logsumexp(lse_u, dim=1, keepdim=True))  # step 11
    out = (o * w).sum(dim=1)
    if return_parts:
        return out, o, lse_u.squeeze(-1)
    return out


def lsh_attention(
    qk: torch.Tensor,
    v: torch.Tensor,
    rotations: torch.Tensor,
    bucket_size: int,
    causal: bool,
    input_mask: Optional[torch.Tensor] = None,
):
    """Steps 2-11.  Returns (out, buckets, sticker, undo_sort)."""
    bh, t, _ = qk.shape
    assert t % (2 * bucket_size) == 0, (
        f"Sequence length ({t}) needs to be divisible by target bucket size x 2 - {2 * bucket_size}")
    n_hashes = rotations.shape[2]
    assert rotations.shape[3] * 2 == t // bucket_size
    with torch.no_grad():
        buckets = hash_vectors(qk.detach(), rotations)
        sticker, undo_sort = sort_buckets(buckets, t)
    out = lsh_attention_sorted(qk, v, sticker, undo_sort, bucket_size, n_hashes, causal, input_mask)
    return out, buckets, sticker, undo_sort


def lsh_attention_bruteforce(qk, v, sticker, bucket_size, n_hashes, causal, input_mask=None):
    """Independent float64 definition by explicit loops (tiny sizes only): for every
    round and token, enumerate the keys of its chunk and of the previous chunk,
    apply the three masks and softmax over the union of all rounds weighted as in
    step 11.  Used to check the vectorised restatement above against indexing slips.
    Returns (out, lse_tot).  Rows whose lse_tot is about -5e4 can only see themselves
    (first causal token, padded tokens): there fp32 rounds the logsumexp at a 4e-3
    grid, so fp32 implementations legitimately differ from this one by ~1e-2 relative."""
    bh, t, dh = qk.shape
    n_chunks = n_hashes * (t // bucket_size)
    q64, v64 = qk.double(), v.double()
    out = torch.zeros(bh, t, dh, dtype=torch.float64)
    lse_tot = torch.zeros(bh, t, dtype=torch.float64)
    for b in range(bh):
        st = (sticker[b] % t).tolist()
        chunks = [st[c * bucket_size:(c + 1) * bucket_size] for c in range(n_chunks)]
        kn = q64[b] / q64[b].norm(dim=-1, keepdim=True).clamp_min(1e-12)
        o_r = torch.zeros(n_hashes, t, dh, dtype=torch.float64)
        lse_r = torch.zeros(n_hashes, t, dtype=torch.float64)
        for c in range(n_chunks):
            r = c // (t // bucket_size)
            keys = chunks[c] + chunks[c - 1]
            for qi in chunks[c]:
                s = []
                for kj in keys:
                    val = float(q64[b, qi] @ kn[kj]) * dh ** -0.5
                    if input_mask is not None and not (bool(input_mask[b, qi]) and bool(input_mask[b, kj])):
                        val = MASK_VALUE
                    if causal and qi < kj:
                        val = MASK_VALUE
                    if qi == kj:
                        val = SELF_VALUE
                    s.append(val)
                s = torch.tensor(s, dtype=torch.float64)
                l = torch.logsumexp(s, 0)
                p = torch.exp(s - l)
                o_r[r, qi] = p @ v64[b, keys]
                lse_r[r, qi] = l
        w = torch.softmax(lse_r, dim=0)
        out[b] = (w.unsqueeze(-1) * o_r).sum(0)
        lse_tot[b] = torch.logsumexp(lse_r, dim=0)
    return out, lse_tot


# --------------------------------------------------------------------------
# module with the reformer_pytorch 0.19.1 constructor surface
# --------------------------------------------------------------------------
class LSHSelfAttention(nn.Module):
    """Stand-in with the parameter names (``toqk``, ``tov``, ``to_out``) and the
    keyword surface the reference passes at ``reformer_tts/model/reformer.py:200``.
    Knobs whose non-default value would change the arithmetic are rejected, not
    ignored."""

    def __init__(self, dim, heads=8, bucket_size=64, n_hashes=8, causal=False,
                 add_local_attn_hash=False, attn_chunks=1, random_rotations_per_head=False,
                 attend_across_buckets=True, allow_duplicate_attention=True, num_mem_kv=0,
                 one_value_head=False, use_full_attn=False, full_attn_thres=None,
                 return_attn=False, post_attn_dropout=0.0, dropout=0.0):
        super().__init__()
        assert dim % heads == 0
        if add_local_attn_hash or not attend_across_buckets or not allow_duplicate_attention \
                or num_mem_kv or one_value_head or use_full_attn or return_attn or dropout:
            raise NotImplementedError("oracle restates the default-knob LSH path only")
        self.dim, self.heads, self.bucket_size, self.n_hashes = dim, heads, bucket_size, n_hashes
        self.causal = causal
        self.random_rotations_per_head = random_rotations_per_head
        self.full_attn_thres = bucket_size if full_attn_thres is None else full_attn_thres
        self.toqk = nn.Linear(dim, dim, bias=False)
        self.tov = nn.Linear(dim, dim, bias=False)
        self.to_out = nn.Linear(dim, dim)
        self.post_attn_dropout = nn.Dropout(post_attn_dropout)
        self.rotation_log = None       # list -> every sampled rotation tensor is appended
        self.forced_rotations = None   # iterator of tensors -> used instead of sampling

    def forward(self, x, input_mask=None, **_):
        b, t, e = x.shape
        h, dh = self.heads, e // self.heads
        assert t > self.full_attn_thres, "full-attention shortcut is outside the restated path"
        qk = self.toqk(x).view(b, t, h, dh).transpose(1, 2).reshape(b * h, t, dh)     # step 1
        v = self.tov(x).view(b, t, h, dh).transpose(1, 2).reshape(b * h, t, dh)
        mask = None
        if input_mask is not None:
            mask = input_mask.bool().unsqueeze(1).expand(b, h, t).reshape(b * h, t)
        n_buckets = t // self.bucket_size
        if self.forced_rotations is not None:
            rotations = next(self.forced_rotations)
        else:
            lead = b * h if self.random_rotations_per_head else 1
            rotations = torch.randn(lead, dh, self.n_hashes, n_buckets // 2, dtype=x.dtype, device=x.device)
        if self.rotation_log is not None:
            self.rotation_log.append(rotations.detach().clone())
        out, *_ = lsh_attention(qk, v, rotations, self.bucket_size, self.causal, mask)
        out = out.view(b, h, t, dh).transpose(1, 2).reshape(b, t, e)                  # step 12
        return self.post_attn_dropout(self.to_out(out))
