"""Reversible residual stacks: forward without stored activations, backward with recompute.

Protocol of ``/root/reference/reformer_tts/model/reversible.py`` (``forward(x, **kw)`` under
no_grad, ``backward_pass(y, dy, **kw) -> (x, dx)``, parameter gradients accumulated as a side
effect), with these MI355X-first changes:
  * the two streams travel as two tensors -- no ``cat`` / ``chunk`` copy per block (the
    reference moves 50 MB per block and direction at decoder shape); the concatenated form is
    only assembled at the protocol boundary for callers that ask for it;
  * no global-RNG replay: a net that needs randomness to repeat (dropout) gets its CUDA RNG state
    captured per call, LSH layers re-use their saved sort permutation instead;
  * after every block's ``backward_pass`` an optional hook fires, which the data-parallel trainer
    uses to launch that block's gradient all-reduce while the next block recomputes.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import nn


def _has_dropout(m: nn.Module) -> bool:
    return any(isinstance(s, nn.Dropout) and s.p > 0 for s in m.modules())


class Deterministic(nn.Module):
    """Holds ``net`` (name kept for the state_dict).  ``record`` captures the device RNG state when
    the net contains active dropout; ``replay`` re-runs the net under that state."""

    def __init__(self, net: nn.Module):
        super().__init__()
        self.net = net
        self._rng = None

    def forward(self, *args, record_rng: bool = False, set_rng: bool = False, **kwargs):
        needs = self.training and _has_dropout(self.net)
        if set_rng:
            kwargs = dict(kwargs, recompute=True) if getattr(self.net, "accepts_recompute", False) else kwargs
            if needs and self._rng is not None:
                dev = args[0].device
                with torch.random.fork_rng(devices=[dev] if dev.type == "cuda" else []):
                    if dev.type == "cuda":
                        torch.cuda.set_rng_state(self._rng, dev)
                    else:
                        torch.set_rng_state(self._rng)
                    return self.net(*args, **kwargs)
            return self.net(*args, **kwargs)
        if record_rng and needs:
            dev = args[0].device
            self._rng = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else torch.get_rng_state()
        return self.net(*args, **kwargs)


class ReversibleBlock(nn.Module):
    """y1 = x1 + f(x2); y2 = x2 + g(y1)   (``reversible.py:46-98``)."""

    def __init__(self, f, g):
        super().__init__()
        self.f = Deterministic(f)
        self.g = Deterministic(g)

    def forward_halves(self, x1, x2, f_args={}, g_args={}):
        with torch.no_grad():
            y1 = x1 + self.f(x2, record_rng=self.training, **f_args)
            y2 = x2 + self.g(y1, record_rng=self.training, **g_args)
        return y1, y2

    def backward_halves(self, y1, y2, dy1, dy2, f_args={}, g_args={}):
        with torch.enable_grad():
            y1 = y1.detach().requires_grad_()
            gy1 = self.g(y1, set_rng=True, **g_args)
            torch.autograd.backward(gy1, dy2)
        with torch.no_grad():
            x2 = y2 - gy1
            del gy1, y2
            dx1 = dy1 + y1.grad
            y1.grad = None
        with torch.enable_grad():
            x2 = x2.detach().requires_grad_()
            fx2 = self.f(x2, set_rng=True, **f_args)
            torch.autograd.backward(fx2, dx1)
        with torch.no_grad():
            x1 = y1.detach() - fx2
            dx2 = dy2 + x2.grad
            x2.grad = None
        return x1, x2.detach(), dx1, dx2

    def forward(self, x, f_args={}, g_args={}):
        return torch.cat(self.forward_halves(*torch.chunk(x, 2, dim=2), f_args, g_args), dim=2)

    def backward_pass(self, y, dy, f_args={}, g_args={}):
        x1, x2, dx1, dx2 = self.backward_halves(*torch.chunk(y, 2, dim=2), *torch.chunk(dy, 2, dim=2), f_args, g_args)
        return torch.cat([x1, x2], dim=2), torch.cat([dx1, dx2], dim=2)


class ReversibleHalfResidual(nn.Module):
    """y1 = x1 + f(x2); x2 passes through   (``reversible.py:134-170``)."""

    def __init__(self, f):
        super().__init__()
        self.f = Deterministic(f)

    def forward_halves(self, x1, x2, **f_args):
        with torch.no_grad():
            y1 = x1 + self.f(x2, record_rng=self.training, **f_args)
        return y1, x2

    def backward_halves(self, y1, x2, dy1, dx2, **f_args):
        with torch.enable_grad():
            x2 = x2.detach().requires_grad_()
            fx2 = self.f(x2, set_rng=True, **f_args)
            torch.autograd.backward(fx2, dy1)
        with torch.no_grad():
            x1 = y1 - fx2
            dx2 = dx2 + x2.grad
            x2.grad = None
        return x1, x2.detach(), dy1, dx2

    def forward(self, x, **f_args):
        return torch.cat(self.forward_halves(*torch.chunk(x, 2, dim=2), **f_args), dim=2)

    def backward_pass(self, y, dy, **f_args):
        x1, x2, dx1, dx2 = self.backward_halves(*torch.chunk(y, 2, dim=2), *torch.chunk(dy, 2, dim=2), **f_args)
        return torch.cat([x1, x2], dim=2), torch.cat([dx1, dx2], dim=2)


class ReversibleSwap(nn.Module):
    """Exchange the two streams (``reversible.py:173-191``); free here: only the references swap."""

    def forward_halves(self, x1, x2, **_):
        return x2, x1

    def backward_halves(self, y1, y2, dy1, dy2, **_):
        return y2, y1, dy2, dy1

    def forward(self, x, **kwargs):
        x1, x2 = torch.chunk(x, 2, dim=2)
        return torch.cat([x2, x1], dim=2)

    def backward_pass(self, y, dy):
        x2, x1 = torch.chunk(y, 2, dim=2)
        dx2, dx1 = torch.chunk(dy, 2, dim=2)
        return torch.cat([x1, x2], dim=2), torch.cat([dx1, dx2], dim=2)


class _ReversibleFunction(torch.autograd.Function):
    """Keeps only the final pair of streams (``reversible.py:114-129``).

    ``context`` (the encoder output a decoder stack cross-attends to) is a differentiable input of
    its own: during the backward the blocks see a detached leaf, whose gradient is accumulated over
    all cross-attention blocks and handed back ONCE.  (In the reference every decoder layer's
    nested ``autograd.backward`` walks into the encoder graph, i.e. one full encoder reversible
    backward per decoder layer; the sum of those partial backward passes equals this single one.)"""

    @staticmethod
    def forward(ctx, x1, x2, context, seq, kwargs_list):
        for block, kwargs in zip(seq.blocks, kwargs_list):
            x1, x2 = block.forward_halves(x1, x2, **kwargs)
        ctx.y = (x1.detach(), x2.detach())
        ctx.context = None if context is None else context.detach()
        ctx.seq, ctx.kwargs_list = seq, kwargs_list
        return x1, x2

    @staticmethod
    def backward(ctx, dy1, dy2):
        y1, y2 = ctx.y
        ctx.y = None
        seq = ctx.seq
        leaf = None if ctx.context is None else ctx.context.requires_grad_()
        for i in range(len(seq.blocks) - 1, -1, -1):
            kwargs = ctx.kwargs_list[i]
            if leaf is not None and "key" in kwargs:
                kwargs = dict(kwargs, key=leaf, value=leaf)
            y1, y2, dy1, dy2 = seq.blocks[i].backward_halves(y1, y2, dy1, dy2, **kwargs)
            if seq.block_done_hook is not None:
                seq.block_done_hook(seq, i)
        return dy1, dy2, (None if leaf is None else leaf.grad), None, None


class ReversibleSequence(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = blocks
        self.block_done_hook: Optional[Callable] = None   # (sequence, block index) after its backward_pass
        self.use_fused = True      # training on the GPU: explicit executor (engine.py) when every block supports it
        self.fused_in_eval = False  # generation: the executor's forward in eval mode too (half the launches of the general path)
        self._program = None
        self._program_built = False
        self.manual = None          # dict while the trainer drives the backward by hand (see forward_sum), else None

    def forward_sum(self, x, kwargs_list=None, context=None):
        """Both streams start as ``x``; returns their sum after the stack (``reformer.py:81-93,139-158``)."""
        kwargs_list = kwargs_list if kwargs_list is not None else [{}] * len(self.blocks)
        # the cross-attention kernels hold 128 or 256 keys on chip at a time and walk longer texts in chunks (text is padded
        # to pad_base = 256); other lengths take the general path below (torch SDPA) instead of failing
        keys_ok = context is None or (context.shape[1] % 128 == 0 and 128 <= context.shape[1] <= 2048)
        if self.use_fused and keys_ok and x.is_cuda and (self.training or (self.fused_in_eval and not torch.is_grad_enabled())):
            if not self._program_built:
                from ..engine import build_program
                self._program, self._program_built = build_program(self), True
            if self._program is not None:
                from ..engine import FusedStackFn
                if self.manual is not None and self.training and torch.is_grad_enabled():
                    # the data-parallel trainer drives this stack's backward itself (engine.stack_backward_steps), one
                    # hipGraph per layer: the forward runs outside autograd and leaves (ctx, input, context, output) behind;
                    # the output is a leaf whose .grad the loss's backward fills
                    from ..engine import _ManualCtx
                    ctx = _ManualCtx()
                    with torch.no_grad():
                        out = FusedStackFn.forward(ctx, x, context, self, kwargs_list)
                    out.requires_grad_(True)
                    self.manual["call"] = (ctx, x, context, out)
                    return out
                return FusedStackFn.apply(x, context, self, kwargs_list)
            from .._lib import note_general_path
            note_general_path("reversible stack", "a block is outside the executors' envelope (e.g. feed-forward dropout > 0)")
        elif self.use_fused and x.is_cuda and not keys_ok:
            from .._lib import note_general_path
            note_general_path("decoder stack", f"{context.shape[1]} encoder keys: the cross-attention kernels take multiples of 128 up to 2048")
        y1, y2 = self.forward_halves(x, x, kwargs_list, context)
        return y1 + y2

    def forward_halves(self, x1, x2, kwargs_list=None, context=None):
        kwargs_list = kwargs_list if kwargs_list is not None else [{}] * len(self.blocks)
        return _ReversibleFunction.apply(x1, x2, context, self, kwargs_list)

    def forward(self, x, kwargs_list=None, **kwargs):
        x1, x2 = torch.chunk(x, 2, dim=2)
        return torch.cat(self.forward_halves(x1, x2, kwargs_list), dim=2)
