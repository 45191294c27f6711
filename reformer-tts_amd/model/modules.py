"""Prenets, postnet, positional encoding and position-wise FFN with the reference's module and
parameter names (``/root/reference/reformer_tts/model/modules.py``).  On the GPU training path the
convolutions, BatchNorm and projections run on the kernels of librtts_hip.so (``edges.py``); the
eager bodies below are the general path (eval, exotic shapes) and say so when taken."""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import nn
import torch.nn.functional as F


class _GemmF32Out(torch.autograd.Function):
    """y(M,N) fp32 = a(M,K) bf16 @ w(N,K)^T bf16 with an fp32 result (hipBLASLt accumulates in fp32
    anyway; this keeps the result unrounded).  Backward: two bf16 GEMMs."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return torch.mm(a, w.t(), out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16)
        return dyb @ w, dyb.t() @ a


def conv1d_k5_rows(x, conv: nn.Conv1d):
    """Conv1d(kernel 5, padding 2) on channels-last rows: x (B, L, Cin) -> (B, L, Cout) as ONE GEMM over
    the (B*L, 5*Cin) window matrix (bf16 operands, fp32 accumulate).  Same arithmetic as
    ``nn.Conv1d`` on the transposed tensor; no MIOpen find/compile step, no (B,C,L) transposes."""
    b, l, cin = x.shape
    xp = F.pad(x.to(torch.bfloat16), (0, 0, 2, 2))
    cols = xp.unfold(1, 5, 1).reshape(b * l, cin * 5)                 # (ci, k) order == weight.view(Cout, Cin*5)
    w = conv.weight.to(torch.bfloat16).reshape(conv.out_channels, cin * 5)
    # fp32 result: the BatchNorm that follows removes the channel mean, which would otherwise leave bf16 rounding of the
    # MEAN behind
    y = _GemmF32Out.apply(cols, w) + conv.bias
    return y.view(b, l, conv.out_channels)


def batch_norm_rows(x, bn: nn.BatchNorm1d):
    """BatchNorm1d on channels-last rows (B, L, C): train mode uses the biased batch statistics over
    (B, L) and updates the running buffers (momentum 0.1, unbiased variance) like nn.BatchNorm1d."""
    xf = x.float()
    if bn.training:
        from .. import edges
        if edges.SYNC_BN is not None:
            raise NotImplementedError("sync_batchnorm is implemented by the fused edge executors (edges.ConvBNAct); this batch took "
                                      "the general path -- see the reformer_tts_amd log for the reason")
        var, mean = torch.var_mean(xf, dim=(0, 1), unbiased=False)
        with torch.no_grad():
            n = xf.shape[0] * xf.shape[1]
            bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
            bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
            bn.num_batches_tracked += 1
    else:
        mean, var = bn.running_mean, bn.running_var
    return (xf - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias


def _bf16_linear(x, lin: nn.Linear):
    b = None if lin.bias is None else lin.bias.to(torch.bfloat16)
    return F.linear(x.to(torch.bfloat16), lin.weight.to(torch.bfloat16), b)


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup + the Dropout behind it in one launch (counter-hash mask, reproduced by the backward); the backward
    is the deterministic per-id row sum of csrc/edges.hip, accumulated straight into the weight's gradient."""

    @staticmethod
    def forward(ctx, ids, weight, padding_idx, p):
        from .. import _lib
        from .._seeds import _seed_counter, seed_base
        n, c = weight.shape
        ids2 = ids.reshape(-1).contiguous()
        out = torch.empty(*ids.shape, c, dtype=torch.float32, device=weight.device)
        seed = next(_seed_counter) * 2654435761 % (1 << 32)
        w = weight.detach()
        w = w if w.dtype == torch.float32 and w.is_contiguous() else w.float().contiguous()
        _lib.call("rtts_embedding_fwd", ids2.data_ptr(), w.data_ptr(), ids2.numel(), c, n, float(p), seed,
                  seed_base(weight.device).data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        ctx.state = (ids2, weight, padding_idx, float(p), seed)
        return out

    @staticmethod
    def backward(ctx, dx):
        from .. import _lib
        from .._seeds import seed_base
        from ..engine import _grad
        ids2, weight, padding_idx, p, seed = ctx.state
        n, c = weight.shape
        if dx.dim() == 3 and dx.dtype == torch.float32 and dx.stride(2) == 1 and not dx.is_contiguous():
            # the convolution stack's gradient is a strided view of its halo rows: read in place, no contiguous copy
            _lib.call("rtts_embedding_bwd_strided", ids2.data_ptr(), dx.data_ptr(), dx.stride(0), dx.stride(1), dx.shape[1],
                      dx.shape[0] * dx.shape[1], c, n, -1 if padding_idx is None else padding_idx, _grad(weight).data_ptr(), p, seed,
                      seed_base(dx.device).data_ptr(), torch.cuda.current_stream().cuda_stream)
            return None, None, None, None
        dx2 = dx.reshape(-1, c)
        dx2 = dx2 if dx2.dtype == torch.float32 and dx2.is_contiguous() else dx2.float().contiguous()
        _lib.call("rtts_embedding_bwd", ids2.data_ptr(), dx2.data_ptr(), dx2.shape[0], c, n, -1 if padding_idx is None else padding_idx,
                  _grad(weight).data_ptr(), p, seed, seed_base(dx.device).data_ptr(), torch.cuda.current_stream().cuda_stream)
        return None, None, None, None          # accumulated into weight.grad here (like every parameter of the fused path)


class EncoderPreNet(nn.Module):
    """``modules.py:8-61``: embedding -> 3 x [dropout, conv k5, BatchNorm, ReLU] -> dropout -> linear."""

    def __init__(self, num_embeddings: int, embedding_dim: int = 512, dropout: float = 0.5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.embed = nn.Embedding(num_embeddings, embedding_dim, padding_idx=0)
        self.projection = nn.Linear(embedding_dim, embedding_dim)
        layers = [("dropout0", nn.Dropout(dropout))]
        for i in (1, 2, 3):
            layers += [(f"conv{i}", nn.Conv1d(embedding_dim, embedding_dim, kernel_size=5, padding=2)),
                       (f"bn{i}", nn.BatchNorm1d(embedding_dim)), (f"relu{i}", nn.ReLU()),
                       (f"dropout{i}", nn.Dropout(dropout))]
        self.convolutions = nn.Sequential(OrderedDict(layers))

    use_fused = True

    def forward(self, input_, pe=None):
        """``pe``: when given (training on the GPU), the positional encoding is applied here, fused with the projection."""
        c = self.convolutions
        if self.use_fused and self.training and input_.is_cuda:
            x = _EmbeddingFn.apply(input_, self.embed.weight, self.embed.padding_idx, c.dropout0.p)     # dropout0 inside
        else:
            x = c.dropout0(self.embed(input_))                  # (B, L, C) channels-last throughout
        if self.use_fused and self.training and x.is_cuda and self.embedding_dim % 128 == 0 and (x.shape[0] * x.shape[1]) % 64 == 0:
            from ..edges import ConvStackFn, encoder_prenet_stack
            if getattr(self, "_stack", None) is None:
                self._stack = encoder_prenet_stack(self)
            z = ConvStackFn.apply(x, self._stack, True)            # the cast to bf16 rides in the stack's first kernel
            if pe is not None:
                from ..edges import proj_pe
                return proj_pe(z, self.projection, pe)               # projection + positional encoding fused
            return _bf16_linear(z, self.projection).float()
        if not self.training and self.use_fused and x.is_cuda and not torch.is_grad_enabled():
            # inference: the same kernels on the running BatchNorm statistics (edges.conv_stack_nograd), no library GEMM
            from ..edges import conv_stack_nograd, linear_nograd
            z = conv_stack_nograd(x, [(c.conv1, c.bn1, 1), (c.conv2, c.bn2, 1), (c.conv3, c.bn3, 1)])
            y = linear_nograd(z, self.projection).float()
            return y if pe is None else pe(y)
        if self.training and x.is_cuda:
            from .._lib import note_general_path
            note_general_path("encoder prenet", "use_fused is off" if not self.use_fused else
                              f"width {self.embedding_dim} / {x.shape[0] * x.shape[1]} rows outside the executor's envelope "
                              "(width % 128 == 0, rows % 64 == 0)")
        for conv, bn, drop in ((c.conv1, c.bn1, c.dropout1), (c.conv2, c.bn2, c.dropout2), (c.conv3, c.bn3, c.dropout3)):
            x = drop(F.relu(batch_norm_rows(conv1d_k5_rows(x, conv), bn)))
        y = _bf16_linear(x, self.projection).float()
        return y if pe is None else pe(y)


class DecoderPreNet(nn.Module):
    """``modules.py:64-100``: fc1-ReLU-drop-fc2-ReLU-drop-projection."""

    def __init__(self, input_size: int, output_size: int, hidden_size: int = 256, dropout: float = 0.5):
        super().__init__()
        self.input_size, self.output_size, self.hidden_size = input_size, output_size, hidden_size
        self.layer = nn.Sequential(OrderedDict([
            ("fc1", nn.Linear(input_size, hidden_size)), ("relu1", nn.ReLU()), ("dropout1", nn.Dropout(dropout)),
            ("fc2", nn.Linear(hidden_size, output_size)), ("relu2", nn.ReLU()), ("dropout2", nn.Dropout(dropout)),
            ("projection", nn.Linear(output_size, output_size))]))

    def forward(self, input_):
        l = self.layer
        if not self.training and input_.is_cuda and not torch.is_grad_enabled():
            from ..edges import linear_nograd            # inference: bias + ReLU in the GEMM epilogue, dropout is off
            x = linear_nograd(linear_nograd(input_, l.fc1, relu=True), l.fc2, relu=True)
            return linear_nograd(x, l.projection).float()
        x = l.dropout1(F.relu(_bf16_linear(input_, l.fc1)))
        x = l.dropout2(F.relu(_bf16_linear(x, l.fc2)))
        return _bf16_linear(x, l.projection).float()


class PostConvNet(nn.Module):
    """``modules.py:103-169``: [conv k5, BatchNorm, tanh, dropout] x depth + conv k5 (the mel postnet)."""

    def __init__(self, mel_size: int, num_hidden: int, dropout: float, depth: int):
        super().__init__()
        self.mel_size = mel_size
        layers = []
        for i in range(depth):
            layers += [(f"conv{i}", nn.Conv1d(mel_size if i == 0 else num_hidden, num_hidden, kernel_size=5, padding=2)),
                       (f"bn{i}", nn.BatchNorm1d(num_hidden)), (f"tanh{i}", nn.Tanh()), (f"dropout{i}", nn.Dropout(dropout))]
        layers += [("convend", nn.Conv1d(num_hidden, mel_size, kernel_size=5, padding=2))]
        self.layers = nn.Sequential(OrderedDict(layers))

    def forward(self, input_):
        x = input_                                              # (B, L, mel) channels-last throughout
        depth = (len(self.layers) - 1) // 4
        if not self.training and x.is_cuda and not torch.is_grad_enabled():
            from ..edges import conv_stack_nograd          # inference: implicit-GEMM convolutions, running BatchNorm statistics
            stack = [(getattr(self.layers, f"conv{i}"), getattr(self.layers, f"bn{i}"), 2) for i in range(depth)]
            return conv_stack_nograd(x, stack, convend=self.layers.convend).float()
        for i in range(depth):
            conv, bn, drop = getattr(self.layers, f"conv{i}"), getattr(self.layers, f"bn{i}"), getattr(self.layers, f"dropout{i}")
            x = drop(torch.tanh(batch_norm_rows(conv1d_k5_rows(x, conv), bn)))
        return conv1d_k5_rows(x, self.layers.convend).float()


class ScaledPositionalEncoding(nn.Module):
    """``modules.py:172-192``: x + alpha * dropout(table); the table interleaves [sin, cos] and the
    dropout mask is drawn once for the (T, d) table, i.e. shared over the batch."""

    def __init__(self, d_model, dropout):
        super().__init__()
        inv_freq = 1.0 / (10000 ** (torch.arange(0, d_model, 2).float() / d_model))
        self.register_buffer("inv_freq", inv_freq)
        self.alpha = nn.Parameter(torch.empty(1).normal_(0, 1))
        self.dropout = nn.Dropout(dropout)
        self._table = None

    def table(self, length: int, device):
        if self._table is None or self._table.shape[0] < length or self._table.device != device:
            # a captured hipGraph holds the ADDRESS of the table it was captured with: a table that is outgrown stays alive (and
            # unchanged) for as long as the module does, and the new one is sized generously so that this happens rarely
            if self._table is not None:
                self.__dict__.setdefault("_old_tables", []).append(self._table)
            rows = max(length, 1024)
            rows = 1 << (rows - 1).bit_length()
            pos = torch.arange(rows, device=device, dtype=self.inv_freq.dtype)
            ang = pos[:, None] * self.inv_freq.to(device)[None, :]
            self._table = torch.stack([ang.sin(), ang.cos()], dim=-1).reshape(rows, -1)
        return self._table[:length]

    def forward(self, input_):
        return input_ + self.alpha * self.dropout(self.table(input_.shape[1], input_.device))


class FeedForward(nn.Module):
    """``modules.py:195-207``: Linear-ReLU-Dropout-Linear (bf16 operands, fp32 accumulate)."""

    def __init__(self, dim=512, hidden=2048, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.ReLU(), nn.Dropout(dropout), nn.Linear(hidden, dim))

    def forward(self, x):
        h = F.relu(_bf16_linear(x, self.net[0]))
        h = self.net[2](h)
        return _bf16_linear(h, self.net[3]).float()
