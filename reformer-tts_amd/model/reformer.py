"""Encoder / decoder Reformer stacks with the reference's module tree
(``/root/reference/reformer_tts/model/reformer.py``), so state_dict names are unchanged."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn
import torch.nn.functional as F

from .lsh_attention import LSHSelfAttention
from .modules import FeedForward
from .reversible import ReversibleBlock, ReversibleHalfResidual, ReversibleSequence, ReversibleSwap


class WithNorm(nn.Module):
    """``reformer.py:25-33``."""

    def __init__(self, norm_class, dim, fn):
        super().__init__()
        self.norm = norm_class(dim)
        self.fn = fn
        self.accepts_recompute = getattr(fn, "accepts_recompute", False)

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)


class Chunk(nn.Module):
    """``reformer.py:36-45``.  The reference splits the sequence into ``chunks`` pieces to bound the
    (rows, hidden) intermediate and runs them in a Python loop; the pieces are independent rows of
    a position-wise function, so one call over all rows returns the same values (SURVEY.md Appendix
    A: max 5e-7 from GEMM blocking).  On a 288 GB part the intermediate (B*T x 2048 bf16 = 50 MB)
    is no concern, so the loop is not reproduced."""

    def __init__(self, chunks, fn, along_dim=-1):
        super().__init__()
        self.dim, self.chunks, self.fn = along_dim, chunks, fn

    def forward(self, x):
        return self.fn(x)


class LSHSelfAttentionWrapper(nn.Module):
    """``reformer.py:189-220``: picks the attention class by ``kwargs["implementation"]``.  This
    package adds ``"hip"``; the reference's two values name third-party eager implementations that
    are not part of it."""
    accepts_recompute = True

    def __init__(self, dim: int, causal: bool, **kwargs):
        super().__init__()
        assert "implementation" in kwargs
        assert kwargs["implementation"] in {"hip", "huggingface_transformers", "reformer_pytorch"}
        self.implementation = kwargs.pop("implementation")
        if self.implementation != "hip":
            raise NotImplementedError(
                f"implementation={self.implementation!r} is the reference's eager third-party path; "
                "this package provides implementation='hip'")
        self.layer = LSHSelfAttention(dim, causal=causal, **kwargs)

    def forward(self, x: torch.Tensor, input_mask: torch.Tensor = None, recompute: bool = False):
        return self.layer.forward(x, input_mask=input_mask, recompute=recompute)


class MultiheadAttentionWrapper(nn.Module):
    """``reformer.py:161-186``: ``nn.MultiheadAttention(dim, num_heads)`` parameters (names kept),
    query = decoder stream, key = value = encoder output, ``key_padding_mask`` True = ignore.
    In eval mode the head-averaged attention matrix is appended to ``attention_matrices_``."""

    def __init__(self, dim: int, attention_matrices: Optional[List[torch.Tensor]] = None, **kwargs):
        super().__init__()
        self.layer = nn.MultiheadAttention(dim, **kwargs)
        self.attention_matrices_ = attention_matrices

    def forward(self, query, **kwargs):
        assert 'key' in kwargs, "forward expects keyword argument 'key'"
        key, kpm = kwargs["key"], kwargs.get("key_padding_mask")
        lyr = self.layer
        b, tq, e = query.shape
        tk, h = key.shape[1], lyr.num_heads
        dh = e // h
        w, bias = lyr.in_proj_weight.to(torch.bfloat16), lyr.in_proj_bias.to(torch.bfloat16)
        q = F.linear(query.to(torch.bfloat16), w[:e], bias[:e]).view(b, tq, h, dh).transpose(1, 2)
        kv = F.linear(key.to(torch.bfloat16), w[e:], bias[e:]).view(b, tk, 2, h, dh)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        mask = None if kpm is None else (~kpm)[:, None, None, :]
        p_drop = lyr.dropout if self.training else 0.0
        if not self.training and self.attention_matrices_ is not None:
            s = (q.float() * dh ** -0.5) @ k.float().transpose(-1, -2)
            if mask is not None:
                s = s.masked_fill(~mask, float("-inf"))
            a = torch.softmax(s, dim=-1)
            self.attention_matrices_.append(a.mean(dim=1))
            o = (a.to(v.dtype) @ v)
        else:
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=p_drop)
        o = o.transpose(1, 2).reshape(b, tq, e)
        return F.linear(o, lyr.out_proj.weight.to(torch.bfloat16), lyr.out_proj.bias.to(torch.bfloat16)).float()


class ReformerEnc(nn.Module):
    """``reformer.py:51-93``."""

    def __init__(self, dim: int, depth: int, ff_chunks: int, attn_kwargs: Dict, ff_kwargs: Dict):
        super().__init__()
        self.dim, self.depth = dim, depth
        blocks = []
        for i in range(depth):
            self_attn = LSHSelfAttentionWrapper(dim, causal=False, **dict(attn_kwargs, seed=i))
            normed_ff = WithNorm(nn.LayerNorm, dim, FeedForward(dim, **ff_kwargs))
            if ff_chunks > 1:
                normed_ff = Chunk(ff_chunks, normed_ff, along_dim=-2)
            blocks.append(ReversibleBlock(f=WithNorm(nn.LayerNorm, dim, self_attn), g=normed_ff))
        self.layers = ReversibleSequence(nn.ModuleList(blocks))

    def forward(self, x, input_mask=None, kwargs_list=None):
        if kwargs_list is not None:
            assert len(kwargs_list) == self.depth, "list_kwargs should be the length of ReversibleSequence"
        else:
            kwargs_list = [dict() for _ in range(self.depth)]
        for kwargs in kwargs_list:
            kwargs["f_args"] = {"input_mask": input_mask}
        return self.layers.forward_sum(x, kwargs_list)             # cat([x, x]) ... sum of the halves (not the mean)


class ReformerDec(nn.Module):
    """``reformer.py:98-158``: per layer [LSH self-attn, swap, cross-attn, swap, FFN, swap]."""

    def __init__(self, dim: int, depth: int, ff_chunks: int, attn_kwargs: Dict, self_attn_kwargs: Dict, ff_kwargs: Dict):
        super().__init__()
        self.dim, self.depth = dim, depth
        self.attention_matrices_ = []
        blocks = []
        for i in range(depth):
            self_attn = LSHSelfAttentionWrapper(dim, causal=True, **dict(self_attn_kwargs, seed=100 + i))
            attn = MultiheadAttentionWrapper(dim, self.attention_matrices_, **attn_kwargs)
            normed_ff = WithNorm(nn.LayerNorm, dim, FeedForward(dim, **ff_kwargs))
            if ff_chunks > 1:
                normed_ff = Chunk(ff_chunks, normed_ff, along_dim=-2)
            blocks += [ReversibleHalfResidual(WithNorm(nn.LayerNorm, dim, self_attn)), ReversibleSwap(),
                       ReversibleHalfResidual(WithNorm(nn.LayerNorm, dim, attn)), ReversibleSwap(),
                       ReversibleHalfResidual(normed_ff), ReversibleSwap()]
        self.block_len = 6
        self.layers = ReversibleSequence(nn.ModuleList(blocks))

    def forward(self, x, keys, key_padding_mask=None, input_mask=None, kwargs_list=None):
        if kwargs_list is not None:
            assert len(kwargs_list) == self.block_len * self.depth, "list_kwargs should be the length of ReversibleSequence"
        else:
            kwargs_list = [dict() for _ in range(self.block_len * self.depth)]
        for kwargs in kwargs_list[2::6]:
            kwargs["key"] = keys
            kwargs["value"] = keys
            kwargs["key_padding_mask"] = key_padding_mask
        for kwargs in kwargs_list[::6]:
            kwargs["input_mask"] = input_mask
        self.attention_matrices_.clear()
        return self.layers.forward_sum(x, kwargs_list, context=keys), self.attention_matrices_
