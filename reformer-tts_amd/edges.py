"""Explicit executors for the convolutional edges of the model and the loss.

``ConvStackFn``     chain of [Conv1d(k5) -> BatchNorm1d(train) -> act -> Dropout] on channels-last rows
                    (encoder prenet convolutions, ``/root/reference/reformer_tts/model/modules.py:33-48``)
``PostnetLossFn``   decoder output -> [mel | stop] heads (``model/reformer_tts.py:65-66``) -> PostConvNet
                    residual (``modules.py:146-169``, ``reformer_tts.py:139-143``) -> TTSLoss
                    (``model/loss.py:28-53``), forward values and the gradient w.r.t. the decoder output

A convolution is an IMPLICIT GEMM on the hand-written MFMA kernel (``rtts_conv1d_k5``, csrc/gemm_nt.hip): the activations of
a stack live in halo rows -- (B, L + 4, C) with two zero rows around every sequence, see ``Halo`` -- so the five taps of the
forward, of the input gradient and of the weight gradient are row-shifted reads of one array; no window (im2col) matrix
exists.  BatchNorm statistics / normalisation / backward, activations, dropout masks, weight re-layouts, weight gradients
(split-K ``rtts_gemm_tn`` on shifted views), the heads and the loss are kernels of ``csrc/edges.hip`` / ``gemm_tn.hip`` /
``gemm_nt.hip``.  The conv bias in front of a BatchNorm only shifts the batch mean: it is folded into the running mean
and its (exactly zero) gradient is not computed.
"""
from __future__ import annotations

import threading
from typing import List

import torch

from . import _lib
from . import engine as _engine
from .engine import WEIGHT_EPOCH, _WS, _bf16, _grad, bf16_twin, cast_colsum, gemm, wgrad

from ._seeds import _seed_counter, seed_base  # noqa: E402,F401  (shared with engine.py)


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ws(device, c: int) -> torch.Tensor:
    key = ("edge", device, c, torch.cuda.current_stream(device).cuda_stream)
    if key not in _WS._cache:
        _WS._cache[key] = torch.empty((2 * 256 + 2) * c, dtype=torch.float32, device=device)
    return _WS._cache[key]


def _pad128(c: int) -> int:
    return -(-c // 128) * 128


class Halo:
    """Halo rows of one (B, L) batch: rows m = b * (L + 4) + 2 + t carry sequence b, position t; the two rows in front of
    and behind every sequence, the ``LEAD`` rows before row 0 and everything behind row B * (L + 4) - 1 are zero.  ``mp`` =
    the row count the GEMMs run over (a multiple of 256: the 256 x 128 and 128 x 64 tiles of rtts_gemm_nt -- B*L itself is a
    multiple of 192 x 64 at the baseline shape, so the halo rows would spill a 192-row tiling into a second, almost empty
    wave of workgroups -- and of the weight gradient's 64-row stages); a buffer has ``alloc`` = mp + 2 * LEAD rows so that
    the taps' shifted reads (+-2 rows) stay inside it.  Every kernel that produces a halo array writes ALL its rows (zeros
    outside the valid set), so buffers come from ``torch.empty`` and carry no state from call to call."""
    H, LEAD, TILE = 2, 8, 256

    def __init__(self, b: int, l: int):
        self.b, self.l = b, l
        self.p = l + 2 * self.H
        self.rows = b * self.p
        self.mp = -(-self.rows // self.TILE) * self.TILE
        self.alloc = self.mp + 2 * self.LEAD

    def new(self, c: int, device, dtype=torch.bfloat16) -> torch.Tensor:
        return torch.empty(self.alloc, c, dtype=dtype, device=device)

    def body(self, buf: torch.Tensor, shift: int = 0) -> torch.Tensor:
        """(mp, C) view starting ``shift`` rows from halo row 0."""
        return buf[self.LEAD + shift:self.LEAD + shift + self.mp]

    def valid(self, body: torch.Tensor) -> torch.Tensor:
        """(B, L, C) strided view of the rows that carry data, of an array whose row 0 is halo row 0."""
        return body[:self.rows].view(self.b, self.p, -1)[:, self.H:self.H + self.l]


import os as _os
CONVS_PER_WGRAD_LAUNCH = int(_os.environ.get("RTTS_CONVS_PER_WGRAD", "4"))   # tap problems of up to 4 convolutions are launched together


def flush_conv_wgrad(queue=None) -> None:
    """The queued tap problems (five per convolution, dW_k = dY^T X shifted by k - 2) as ONE grouped split-K launch, on the
    stream they were queued on (``queue``: an engine._Queue; None: the current device's and stream's)."""
    queue = _engine._queue() if queue is None else queue
    q = queue.conv_gemms
    # problems whose N and K tile by 256 go together: a group of them is large enough for the 256 x 256 ring kernel
    # (rtts_gemm_tn_grouped takes it only when EVERY problem of the group tiles); the 128-wide first / last convolution of the
    # postnet would drag its neighbours down to the 128-tile kernel
    wide = [p for p in q if p[0][7] % 256 == 0 and p[0][8] % 256 == 0]
    rest = [p for p in q if not (p[0][7] % 256 == 0 and p[0][8] % 256 == 0)]
    del q[:]
    for part in (wide, rest):
        while part:
            chunk = part[:_lib.GEMM_TN_MAX_GROUP]
            del part[:len(chunk)]
            arr = (_lib.GemmTnProblem * len(chunk))()
            for dst, (fields, _keep) in zip(arr, chunk):
                dst.a, dst.lda, dst.b, dst.ldb, dst.c, dst.ldc, dst.M, dst.N, dst.K, dst.accumulate = fields
            ws = _engine._slab_ws(chunk[0][1][0].device, queue.stream)
            _lib.call("rtts_gemm_tn_grouped", arr, len(chunk), ws.data_ptr(), ws.numel(), queue.stream)


def flush_conv_dw(queue=None) -> None:
    """dw[co][ci][k] += dwp[co][k][ci] for every convolution whose backward has run since the last flush (one launch)."""
    queue = _engine._queue() if queue is None else queue
    flush_conv_wgrad(queue)
    pending = queue.conv_items
    while pending:
        chunk = pending[:_lib.CONV_PERM_MAX_GROUP]
        del pending[:len(chunk)]
        jobs = (_lib.ConvPermJob * len(chunk))()
        for j, (dwp, co, ci, cp, gw) in zip(jobs, chunk):
            j.w, j.wp, j.Co, j.Ci, j.CP = dwp.data_ptr(), gw.data_ptr(), co, ci, cp
        _lib.call("rtts_conv_dw_unperm_grouped", jobs, len(chunk), queue.stream)


class ConvK5:
    """One Conv1d(kernel 5, padding 2) on halo rows (implicit GEMM); forward, input gradient, weight gradient."""

    instances = None      # weakref.WeakSet of every executor: refresh_all() re-lays-out all their weights in one launch

    def __init__(self, conv: torch.nn.Conv1d):
        self.conv = conv
        self.co, self.ci = conv.out_channels, conv.in_channels
        self.cp = _pad128(self.ci)          # input channels padded (zero weights): the channel count of the input rows
        self.cop = _pad128(self.co)         # output channels padded likewise (zero rows)
        self._wp = None
        self._wp_version = None
        if ConvK5.instances is None:
            import weakref
            ConvK5.instances = weakref.WeakSet()
        ConvK5.instances.add(self)

    def _version(self):
        w = self.conv.weight
        return (w._version, WEIGHT_EPOCH[0], w.data_ptr())

    @staticmethod
    def refresh_all(device) -> None:
        """Re-layout the weights of every live executor on ``device`` whose cached copy is stale: one grouped launch at the
        start of a step (the weights only change in the optimizer step) instead of one launch per convolution."""
        stale = [c for c in (ConvK5.instances or ()) if c.conv.weight.device == device and c._wp_version != c._version()]
        for i in range(0, len(stale), _lib.CONV_PERM_MAX_GROUP):
            chunk = stale[i:i + _lib.CONV_PERM_MAX_GROUP]
            jobs = (_lib.ConvPermJob * len(chunk))()
            for j, c in zip(jobs, chunk):
                if c._wp is None or c._wp.device != device:
                    c._wp = torch.zeros(c.cop, 5 * c.cp, dtype=torch.bfloat16, device=device)
                j.w, j.wp, j.Co, j.Ci, j.CP = c.conv.weight.data_ptr(), c._wp.data_ptr(), c.co, c.ci, c.cp
            _lib.call("rtts_conv_w_perm_grouped", jobs, len(chunk), _s())
            for c in chunk:
                c._wp_version = c._version()

    def weight_perm(self) -> torch.Tensor:
        """(Cout_pad, 5*Cin_pad) bf16 with wp[co][k][ci] = w[co][ci][k]; rebuilt when the master changed."""
        w = self.conv.weight
        ver = (w._version, WEIGHT_EPOCH[0], w.data_ptr())
        if self._wp is None or self._wp_version != ver or self._wp.device != w.device:
            if self._wp is None or self._wp.device != w.device:
                self._wp = torch.zeros(self.cop, 5 * self.cp, dtype=torch.bfloat16, device=w.device)
            _lib.call("rtts_conv_w_perm", w.data_ptr(), self.co, self.ci, self.cp, self._wp.data_ptr(), _s())
            self._wp_version = ver
        return self._wp

    def forward(self, xh: torch.Tensor, g: Halo, bias=None) -> torch.Tensor:
        """xh: halo buffer (alloc, cp) bf16 -> y (mp, cop) fp32 (unrounded: a BatchNorm or the loss reads it)."""
        y = torch.empty(g.mp, self.cop, dtype=torch.float32, device=xh.device)
        wp = self.weight_perm()
        _lib.call("rtts_conv1d_k5", g.body(xh).data_ptr(), self.cp, wp.data_ptr(), 5 * self.cp, 0, g.mp, self.cop, self.cp, y.data_ptr(),
                  self.cop, None if bias is None else bias.data_ptr(), 1, _s())
        return y

    def forward_moments(self, xh: torch.Tensor, g: Halo):
        """``forward`` without a bias, plus the per-channel moments of y over the valid rows from the GEMM's own epilogue
        (``rtts_conv1d_k5_moments``) -> (y, partial rows of [sum y | sum y^2], their count): what the BatchNorm behind this
        convolution would otherwise re-read y for."""
        y = torch.empty(g.mp, self.cop, dtype=torch.float32, device=xh.device)
        nrows = _lib.load().rtts_gemm_nt_partial_rows(g.mp, self.cop)
        if nrows <= 0:
            raise _lib.RttsError(f"rtts_conv1d_k5_moments: no tile shape for {g.mp} x {self.cop}")
        partial = torch.empty(nrows, 2 * self.cop, dtype=torch.float32, device=xh.device)
        _lib.call("rtts_conv1d_k5_moments", g.body(xh).data_ptr(), self.cp, self.weight_perm().data_ptr(), 5 * self.cp, g.mp, self.cop, self.cp,
                  y.data_ptr(), self.cop, g.b, g.l, g.H, partial.data_ptr(), _s())
        return y, partial, nrows

    def backward(self, dyh: torch.Tensor, xh: torch.Tensor, g: Halo, need_dx: bool = True, dx_f32: bool = False):
        """dyh: halo buffer (alloc, cop) bf16, zero outside the valid set -> accumulates dW; returns dx (mp, cp) bf16 / fp32
        in halo rows (row 0 = halo row 0; rows outside the valid set hold no meaning)."""
        dev = dyh.device
        dwp = torch.empty(self.cop, 5 * self.cp, dtype=torch.float32, device=dev)
        # weight gradient of tap k = (dy)^T (x shifted by k - 2): five problems, queued so that up to three convolutions of a
        # stack share one grouped split-K launch (their operands are still in the Infinity Cache a layer or two later)
        dyb = g.body(dyh)
        dwq = _engine._queue()
        for k in range(5):
            xs, ck = g.body(xh, k - 2), dwp[:, k * self.cp:(k + 1) * self.cp]
            dwq.conv_gemms.append(((dyb.data_ptr(), self.cop, xs.data_ptr(), self.cp, ck.data_ptr(), 5 * self.cp, g.mp, self.cop, self.cp, 0),
                              (dyh, xh, dwp)))
        if len(dwq.conv_gemms) >= 5 * CONVS_PER_WGRAD_LAUNCH:
            flush_conv_wgrad(dwq)
        # dW goes back to nn.Conv1d's (Co, Ci, 5) layout with the other deferred gradient work of the backward: one grouped
        # launch for all convolutions (engine.flush_wgrad runs the hook before anything reads the gradients)
        dwq.conv_items.append((dwp, self.co, self.ci, self.cp, _grad(self.conv.weight)))
        if len(dwq.conv_items) >= _lib.CONV_PERM_MAX_GROUP:
            flush_conv_dw(dwq)
        else:
            from .engine import _queue_final_flush
            _queue_final_flush()
        if not need_dx:
            return None
        dx = torch.empty(g.mp, self.cp, dtype=torch.float32 if dx_f32 else torch.bfloat16, device=dev)
        _lib.call("rtts_conv1d_k5", dyb.data_ptr(), self.cop, self.weight_perm().data_ptr(), 5 * self.cp, 1, g.mp, self.cp, self.cop,
                  dx.data_ptr(), self.cp, None, int(dx_f32), _s())
        return dx


# Data-parallel BatchNorm statistics (the reference's BatchNorm layers, modules.py:29,127, normalise over the
# WHOLE batch, which data parallelism spreads over the ranks).  None: every rank normalises over its own rows (the default;
# DESIGN.md section 7 states the deviation).  A process group (``Trainer`` sets it from ``TTSTrainingConfig.sync_batchnorm``):
# the per-channel sums of both stages -- forward {sum y, sum y^2}, backward {sum g, sum g*yhat} -- and the row count are
# all-reduced between the stage that produces them and the stage that uses them: 2*C + 1 floats per BatchNorm layer and
# direction, six layers.  Eager steps only: a collective inside a hipGraph capture is not attempted.
SYNC_BN = None
CONV_MOMENTS = _os.environ.get("RTTS_CONV_MOMENTS", "1") != "0"      # A/B: the BatchNorm statistics from a second pass over y (rtts_bn_stats)


class ConvBNAct:
    """Conv1d(k5) -> BatchNorm1d (batch statistics) -> act (1 relu / 2 tanh) -> Dropout(p), on halo rows."""

    def __init__(self, conv, bn, act: int, p: float):
        self.c = ConvK5(conv)
        self.bn, self.act, self.p = bn, act, float(p)

    def forward(self, xh, g: Halo, plain_out: bool = False):
        """-> z (halo buffer (alloc, C) bf16, or (B*L, C) plain rows for the last layer of a stack), saved state."""
        fused = SYNC_BN is None and CONV_MOMENTS
        y, partial, nrows = self.c.forward_moments(xh, g) if fused else (self.c.forward(xh, g), None, 0)
        c = y.shape[1]
        dev = y.device
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        rstd = torch.empty(c, dtype=torch.float32, device=dev)
        bn = self.bn
        # running_mean tracks the mean of (y + conv bias): the bias is left out of y (BatchNorm cancels it) and shifts
        # only the running mean; num_batches_tracked is bumped by the same launch
        if fused:
            # the sums of y and y^2 left the convolution's epilogue as partial rows: one small launch finishes them
            _lib.call("rtts_bn_stats_from_partials", partial.data_ptr(), nrows, g.b, g.l, c, mean.data_ptr(), rstd.data_ptr(),
                      bn.running_mean.data_ptr(), bn.running_var.data_ptr(), self.c.conv.bias.data_ptr(), bn.num_batches_tracked.data_ptr(), _s())
        elif SYNC_BN is None:
            _lib.call("rtts_bn_stats", y.data_ptr(), g.b, g.l, g.H, c, mean.data_ptr(), rstd.data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), self.c.conv.bias.data_ptr(), bn.num_batches_tracked.data_ptr(), _ws(dev, c).data_ptr(), _s())
        else:
            import torch.distributed as dist
            mom = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
            _lib.call("rtts_bn_moments", y.data_ptr(), g.b, g.l, g.H, c, mom.data_ptr(), _ws(dev, c).data_ptr(), _s())
            mom[2 * c:].fill_(float(g.b * g.l))
            dist.all_reduce(mom, group=SYNC_BN)
            _lib.call("rtts_bn_from_moments", mom.data_ptr(), 0, c, mean.data_ptr(), rstd.data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), self.c.conv.bias.data_ptr(), bn.num_batches_tracked.data_ptr(), _s())
        seed = next(_seed_counter) * 2654435761 % (1 << 32)
        if plain_out:
            z = torch.empty(g.b * g.l, c, dtype=torch.bfloat16, device=dev)
            zargs = (0, 0, g.b * g.l)
        else:
            z = g.new(c, dev)
            zargs = (1, g.LEAD, g.alloc)
        _lib.call("rtts_bn_act_fwd", y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), self.act,
                  self.p, seed, seed_base(dev).data_ptr(), g.b, g.l, g.H, c, z.data_ptr(), *zargs, _s())
        return z, (xh, y, mean, rstd, seed)

    def backward(self, dz, dz_halo: bool, saved, g: Halo, need_dx=True, dx_f32=False):
        """dz: (mp, C) halo rows (row 0 = halo row 0) or (B*L, C) plain rows, bf16."""
        xh, y, mean, rstd, seed = saved
        c = y.shape[1]
        bn = self.bn
        dy = g.new(c, y.device)
        if SYNC_BN is None:
            _lib.call("rtts_bn_act_bwd", y.data_ptr(), dz.data_ptr(), int(dz_halo), mean.data_ptr(), rstd.data_ptr(), bn.weight.data_ptr(),
                      bn.bias.data_ptr(), self.act, self.p, seed, seed_base(y.device).data_ptr(), g.b, g.l, g.H, c, dy.data_ptr(), g.LEAD, g.alloc,
                      _grad(bn.weight).data_ptr(), _grad(bn.bias).data_ptr(), _ws(y.device, c).data_ptr(), _s())
        else:
            import torch.distributed as dist
            sums = torch.empty(2 * c + 1, dtype=torch.float32, device=y.device)
            _lib.call("rtts_bn_act_bwd_sums", y.data_ptr(), dz.data_ptr(), int(dz_halo), mean.data_ptr(), rstd.data_ptr(), bn.weight.data_ptr(),
                      bn.bias.data_ptr(), self.act, self.p, seed, seed_base(y.device).data_ptr(), g.b, g.l, g.H, c, sums.data_ptr(),
                      _grad(bn.weight).data_ptr(), _grad(bn.bias).data_ptr(), _ws(y.device, c).data_ptr(), _s())
            sums[2 * c:].fill_(float(g.b * g.l))
            dist.all_reduce(sums, group=SYNC_BN)
            _lib.call("rtts_bn_act_bwd_apply", y.data_ptr(), dz.data_ptr(), int(dz_halo), mean.data_ptr(), rstd.data_ptr(), bn.weight.data_ptr(),
                      bn.bias.data_ptr(), self.act, self.p, seed, seed_base(y.device).data_ptr(), g.b, g.l, g.H, c, sums.data_ptr(), 0,
                      dy.data_ptr(), g.LEAD, g.alloc, _s())
        _grad(self.c.conv.bias)       # exists (stays zero: the true gradient of a bias in front of BatchNorm is zero)
        return self.c.backward(dy, xh, g, need_dx, dx_f32)


class ConvStackFn(torch.autograd.Function):
    """x (B, L, C) fp32 | bf16 -> z (B, L, C) bf16 through a list of ConvBNAct."""

    @staticmethod
    def forward(ctx, x, stack: List[ConvBNAct], training: bool):
        b, l, c = x.shape
        g = Halo(b, l)
        x2 = x.detach().reshape(b * l, c)
        if x2.dtype not in (torch.float32, torch.bfloat16) or x2.stride(1) != 1:
            x2 = x2.float().contiguous()
        cur = g.new(c, x.device)
        _lib.call("rtts_to_halo", x2.data_ptr(), x2.stride(0), 0, c, int(x2.dtype == torch.float32), b, l, g.H, c, cur.data_ptr(), g.LEAD, g.alloc, _s())
        saved = []
        for i, layer in enumerate(stack):
            cur, s = layer.forward(cur, g, plain_out=(i == len(stack) - 1))
            saved.append(s)
        ctx.stack, ctx.saved_state, ctx.geom, ctx.in_dtype = stack, saved, g, x.dtype
        return cur.view(b, l, -1)

    @staticmethod
    def backward(ctx, dz):
        g = ctx.geom
        if ctx.saved_state is None:
            raise RuntimeError("ConvStackFn.backward: state already consumed")
        cur = dz.reshape(g.b * g.l, -1).to(torch.bfloat16).contiguous()
        halo = False
        n = len(ctx.stack)
        for i in range(n - 1, -1, -1):
            cur = ctx.stack[i].backward(cur, halo, ctx.saved_state[i], g, need_dx=True, dx_f32=(i == 0))
            halo = True
        ctx.saved_state = None
        dx = g.valid(cur)                     # (B, L, C) fp32 view of the halo rows that carry data
        return (dx if ctx.in_dtype == torch.float32 else dx.to(ctx.in_dtype)), None, None


def encoder_prenet_stack(prenet) -> List[ConvBNAct]:
    c = prenet.convolutions
    return [ConvBNAct(c.conv1, c.bn1, 1, c.dropout1.p), ConvBNAct(c.conv2, c.bn2, 1, c.dropout2.p),
            ConvBNAct(c.conv3, c.bn3, 1, c.dropout3.p)]


def segments(jobs):
    """[(dst, src, kind)] flat copies / additions (``_lib.SEG_*``) in one launch per ``_lib.SEGMENTS_MAX`` jobs; dst and src
    are contiguous and of equal size."""
    jobs = list(jobs)
    while jobs:
        group, jobs = jobs[:_lib.SEGMENTS_MAX], jobs[_lib.SEGMENTS_MAX:]
        arr = (_lib.Segment * len(group))()
        for j, (dst, src, kind) in zip(arr, group):
            if not (dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()):
                raise ValueError("segments: operands must be contiguous and of equal size")
            j.dst, j.src, j.count, j.kind = dst.data_ptr(), src.data_ptr(), dst.numel(), kind
        _lib.call("rtts_segments", arr, len(group), _s())


class PostnetLoss:
    """Heads + postnet + loss for the training step (forward values, then the gradient w.r.t. the decoder output)."""

    def __init__(self, model, loss_mod):
        self.model, self.loss_mod = model, loss_mod
        pn = model.postnet.layers
        depth = (len(pn) - 1) // 4
        self.layers = [ConvBNAct(getattr(pn, f"conv{i}"), getattr(pn, f"bn{i}"), 2, getattr(pn, f"dropout{i}").p) for i in range(depth)]
        self.convend = ConvK5(pn.convend)
        self.nm = model.num_mel_coeffs
        self.pos_weight = float(loss_mod.pos_weight)      # host copy: no device read inside a captured step
        self._wh = None

    def _operands(self):
        """-> (heads weight (128, d) bf16 = [mel | stop | 0], heads bias (128) fp32, the last convolution's bias padded to its
        GEMM width): persistent padded buffers refreshed from the parameters by ONE launch per step."""
        mel, stop = self.model.dec.mel_linear, self.model.dec.stop_linear
        d, dev = mel.weight.shape[1], mel.weight.device
        if self._wh is None or self._wh.device != dev:
            self._wh = torch.zeros(128, d, dtype=torch.bfloat16, device=dev)
            self._bh = torch.zeros(128, dtype=torch.float32, device=dev)
        c, nm = self.convend, self.nm
        if getattr(c, "_bias_pad", None) is None or c._bias_pad.device != dev:
            c._bias_pad = torch.zeros(c.cop, dtype=torch.float32, device=dev)
        wm, ws = _bf16(mel.weight), _bf16(stop.weight)
        kind_w = _lib.SEG_COPY_BF16 if wm.dtype == torch.bfloat16 else _lib.SEG_CAST_F32_BF16
        segments([(self._wh[:nm], wm.contiguous(), kind_w), (self._wh[nm:nm + 1], ws.contiguous(), kind_w),
                  (self._bh[:nm], mel.bias.detach(), _lib.SEG_COPY_F32), (self._bh[nm:nm + 1], stop.bias.detach(), _lib.SEG_COPY_F32),
                  (c._bias_pad[:c.co], c.conv.bias.detach(), _lib.SEG_COPY_F32)])
        return self._wh, self._bh, c._bias_pad

    def apply(self, y_dec, true_mel, true_stop, true_mask, valid_len=None):
        """-> (total, raw, postnet, stop) losses; only the total is differentiable (the others are reported values).
        ``valid_len``: int32 device word -- the batch's own length when the target / mask / stop tensors are buffers padded to a
        fixed length (Trainer's per-shape graph cache); None: their time dimension is the length."""
        return _PostnetLossFn.apply(y_dec, true_mel, true_stop, true_mask, self, valid_len)


class _PostnetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_dec, true_mel, true_stop, true_mask, ex: PostnetLoss, valid_len=None):
        b, lp, d = y_dec.shape                      # lp = padded decoder length
        l = true_mel.shape[1]                       # loss length (cutoff)
        nm, dev = ex.nm, y_dec.device
        m = b * lp
        g = Halo(b, lp)
        yb = bf16_twin(y_dec)                       # the stack's own bf16 copy of its output when there is one
        yb = yb.view(m, d) if yb is not None else y_dec.detach().reshape(m, d).to(torch.bfloat16)
        wh, bh, bias_end = ex._operands()
        heads = gemm(yb, wh, bias=bh, out_f32=True)                         # (M,128) fp32: [mel(80) | stop | 0...]
        x0 = g.new(128, dev)                        # the first convolution's weights are zero for channels >= nm
        _lib.call("rtts_to_halo", heads.data_ptr(), 128, 0, 128, 1, b, lp, g.H, 128, x0.data_ptr(), g.LEAD, g.alloc, _s())
        saved, cur = [], x0
        for layer in ex.layers:
            cur, s = layer.forward(cur, g)
            saved.append(s)
        res = ex.convend.forward(cur, g, bias=bias_end)                      # (mp, 128) fp32 halo rows; cols >= nm never read
        lm = ex.loss_mod
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        # the three stored gradients share one buffer so that the backward scales them with one launch; d_post is the output
        # gradient of the last convolution and is written in halo rows (the residual add post = raw + res happens in the kernel)
        n_raw, n_post = m * 128, g.alloc * 128
        gbuf = torch.empty(n_raw + n_post + m, dtype=torch.float32, device=dev)
        d_raw, d_post, g_stop = gbuf[:n_raw].view(m, 128), gbuf[n_raw:n_raw + n_post].view(g.alloc, 128), gbuf[n_raw + n_post:]
        pw = torch.empty(512 * 3, dtype=torch.float32, device=dev)
        if true_mel.dtype != torch.float32 or true_mel.stride(2) != 1 or true_mel.stride(1) != nm:
            true_mel = true_mel.float().contiguous()
        if valid_len is not None and not (valid_len.dtype == torch.int32 and valid_len.is_cuda and valid_len.numel() == 1):
            raise ValueError("PostnetLoss: valid_len must be a one-element int32 device tensor")
        # the targets are usually the view frames [1, L) of the batch: the kernel takes its batch stride, no copy
        _lib.call("rtts_tts_loss", heads.data_ptr(), None, 128, true_mel.data_ptr(), true_mask.contiguous().data_ptr(),
                  heads[:, nm:].data_ptr(), 128, true_stop.contiguous().data_ptr(), m, nm, lm.kind, ex.pos_weight,
                  float(lm.raw_pred_loss_weight), float(lm.post_pred_loss_weight), float(lm.stop_loss_weight), d_raw.data_ptr(),
                  d_post.data_ptr(), 128, g_stop.data_ptr(), losses.data_ptr(), pw.data_ptr(), lp, l, res.data_ptr(), 128, g.H, g.LEAD,
                  g.alloc, true_mel.stride(0), None if valid_len is None else valid_len.data_ptr(), 0, 0, _s())
        ctx.ex, ctx.state = ex, (yb, wh, saved, cur, gbuf, g, d)
        ctx.set_materialize_grads(False)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, g_total, g_raw, g_post, g_stop_loss):
        ex = ctx.ex
        if g_raw is not None or g_post is not None or g_stop_loss is not None:
            raise NotImplementedError("PostnetLoss: only the total loss is differentiable (the parts are reported values; "
                                      "model.TTSLoss differentiates all four)")
        if g_total is None:
            return None, None, None, None, None, None
        if ctx.state is None:
            raise RuntimeError("PostnetLoss.backward: state already consumed")
        yb, wh, saved, z_last, gbuf, g, d = ctx.state
        ctx.state = None
        nm, dev = ex.nm, yb.device
        b, lp = g.b, g.l
        m = b * lp
        # the stored gradients are those of the total loss for an upstream gradient of 1; the upstream scalar (1 in the trainer)
        # rides into the two kernels that read them (no scaling pass)
        up = g_total.detach().reshape(1)
        up = up if up.dtype == torch.float32 else up.float()
        n_raw, n_post = m * 128, g.alloc * 128
        d_raw, d_post, g_stop = gbuf[:n_raw].view(m, 128), gbuf[n_raw:n_raw + n_post].view(g.alloc, 128), gbuf[n_raw + n_post:]
        # convend: res = conv(z_last) + bias;  d_res = d_post.  Deterministic column sums (no ATen reduction), finalised with the
        # other deferred sums straight into the bias gradient
        dresb = cast_colsum(d_post, [(0, nm, _grad(ex.convend.conv.bias))], scale=up)
        dz = ex.convend.backward(dresb, z_last, g)
        for layer, s in zip(reversed(ex.layers[1:]), reversed(saved[1:])):
            dz = layer.backward(dz, True, s, g)
        dx0 = ex.layers[0].backward(dz, True, saved[0], g, need_dx=True, dx_f32=True)        # (mp, 128) fp32 halo rows
        dheads = torch.empty(m, 128, dtype=torch.float32, device=dev)
        _lib.call("rtts_heads_grad", d_raw.data_ptr(), d_post.data_ptr(), g.LEAD, dx0.data_ptr(), 128, g_stop.data_ptr(), b, lp, g.H, nm, 128,
                  dheads.data_ptr(), up.data_ptr(), _s())
        mel, stop = ex.model.dec.mel_linear, ex.model.dec.stop_linear
        dhb = cast_colsum(dheads, [(0, nm, _grad(mel.bias)), (nm, 1, _grad(stop.bias))])
        dwh = torch.empty(128, d, dtype=torch.float32, device=dev)
        wgrad(dwh, dhb, yb, accumulate=False)
        segments([(_grad(mel.weight), dwh[:nm], _lib.SEG_ADD_F32), (_grad(stop.weight), dwh[nm:nm + 1], _lib.SEG_ADD_F32)])
        dy = gemm(dhb, wh, kn=True, out_f32=True).view(b, lp, d)
        return dy, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# Linear stacks in front of the reversible stacks: (decoder prenet | encoder projection) + scaled positional encoding
# ------------------------------------------------------------------------------------------------------------------
def _seed() -> int:
    return next(_seed_counter) * 2654435761 % (1 << 32)


def _pe_ws(device) -> torch.Tensor:
    key = ("pe", device, torch.cuda.current_stream(device).cuda_stream)
    if key not in _WS._cache:
        _WS._cache[key] = torch.empty(512, dtype=torch.float32, device=device)
    return _WS._cache[key]


class _ProjPEFn(torch.autograd.Function):
    """z (B,L,K) bf16 -> out (B,L,d) fp32 = z W^T + b + alpha * dropout(table): the Linear that ends a prenet followed
    by ScaledPositionalEncoding (``modules.py:55-61,96-100,180-192``).  Bias gradient, dalpha and dW use the
    deterministic two-stage kernels (no ATen reductions, hipGraph-replayable)."""

    @staticmethod
    def forward(ctx, z, lin, pe, _track):
        # _track = lin.weight: makes the output require grad even when z does not (parameter gradients are
        # accumulated by this function itself, so None is returned for it)
        b, l, k = z.shape
        d = lin.out_features
        z2 = z.reshape(b * l, k)
        y = gemm(z2, _bf16(lin.weight), bias=lin.bias)
        table = pe.table(l, z.device).contiguous()
        out = torch.empty(b * l, d, dtype=torch.float32, device=z.device)
        p = pe.dropout.p if pe.training else 0.0
        seed = _seed()
        _lib.call("rtts_pe_add", y.data_ptr(), table.data_ptr(), pe.alpha.data_ptr(), float(p), seed, seed_base(z.device).data_ptr(), l, b * l, d,
                  out.data_ptr(), _s())
        ctx.mods, ctx.state = (lin, pe), (z2, table, p, seed, b, l, k, d)
        return out.view(b, l, d)

    @staticmethod
    def backward(ctx, dout):
        lin, pe = ctx.mods
        z2, table, p, seed, b, l, k, d = ctx.state
        dy = dout.reshape(b * l, d).float().contiguous()
        _lib.call("rtts_pe_dalpha", dy.data_ptr(), table.data_ptr(), float(p), seed, seed_base(dy.device).data_ptr(), l, b * l, d, _grad(pe.alpha).data_ptr(),
                  _pe_ws(dy.device).data_ptr(), _s())
        dyb = cast_colsum(dy, _grad(lin.bias))
        wgrad(_grad(lin.weight), dyb, z2)
        dz = gemm(dyb, _bf16(lin.weight), kn=True)
        return dz.view(b, l, k), None, None, None


def proj_pe(z, lin, pe):
    return _ProjPEFn.apply(z, lin, pe, lin.weight)


def _k_padded_weight(lin) -> torch.Tensor:
    """bf16 copy of a Linear weight whose input width is not a multiple of the GEMM's 64-deep K stage (the decoder prenet's
    80 mel channels), zero-padded to 128 columns; refreshed when the master changed (one small copy per optimizer step)."""
    w = lin.weight
    ver = (w._version, WEIGHT_EPOCH[0], w.data_ptr())
    cache = getattr(lin, "_rtts_kpad", None)
    if cache is None or cache[0] != ver or cache[1].device != w.device:
        buf = cache[1] if cache is not None and cache[1].device == w.device else torch.zeros(w.shape[0], 128, dtype=torch.bfloat16, device=w.device)
        buf[:, :w.shape[1]].copy_(_bf16(w))
        lin._rtts_kpad = cache = (ver, buf)
    return cache[1]


class _ReluDropLinearFn(torch.autograd.Function):
    """x (M,K) bf16 -> dropout_p(relu(x W^T + b)) bf16 (decoder prenet stages fc1/fc2, ``modules.py:82-94``); bias and ReLU in
    the GEMM's epilogue.  ``kpad``: x arrives zero-padded to 128 columns (K = 80 mel channels) and the weight is padded to match."""

    @staticmethod
    def forward(ctx, x, lin, p, kpad, _track):
        h = gemm(x, _k_padded_weight(lin) if kpad else _bf16(lin.weight), bias=lin.bias, relu=True)
        seed = _seed()
        if p > 0.0:
            _lib.call("rtts_relu_drop", h.data_ptr(), float(p), seed, seed_base(h.device).data_ptr(), h.numel(), _s())
        ctx.lin, ctx.state = lin, (x, h, p, kpad)
        return h

    @staticmethod
    def backward(ctx, dh):
        from .engine import colsum_bf16
        lin = ctx.lin
        x, h, p, kpad = ctx.state
        if dh.dtype != torch.bfloat16 or not dh.is_contiguous():
            dh = dh.to(torch.bfloat16).contiguous()
        gated = torch.empty_like(dh)                                # the incoming gradient is autograd's: not written in place
        colsum_bf16(dh, _grad(lin.bias), h, 1.0 / (1.0 - p), out=gated)      # gate (h > 0) * 1/(1-p) + bias gradient
        dh = gated
        if kpad:
            k = lin.weight.shape[1]
            tmp = torch.empty(lin.weight.shape[0], 128, dtype=torch.float32, device=dh.device)
            wgrad(tmp, dh, x, accumulate=False)
            _grad(lin.weight).add_(tmp[:, :k])
            return None, None, None, None, None                      # the padded input is data (the mel frames): no gradient
        wgrad(_grad(lin.weight), dh, x)
        return (gemm(dh, _bf16(lin.weight), kn=True) if ctx.needs_input_grad[0] else None), None, None, None, None


def decoder_prenet_pe(prenet, pe, spec):
    """DecoderPreNet + ScaledPositionalEncoding on (B, L, n_mels) fp32 -> (B, L, d) fp32."""
    b, l, nm = spec.shape
    lyr = prenet.layer
    training = prenet.training
    m = b * l
    kpad = nm % 64 != 0
    if kpad:
        if nm % 8 != 0 or nm > 128:
            raise NotImplementedError(f"decoder prenet: {nm} mel channels (supported: multiples of 8 up to 128, or multiples of 64)")
        # the input is usually the view frames [0, L-1) of the batch: the kernel takes its batch stride (no contiguous copy);
        # cast + zero padding of the channels in the same launch
        if spec.dtype != torch.float32 or spec.stride(2) != 1 or spec.stride(1) % 4 != 0 or spec.stride(0) % 4 != 0:
            spec = spec.float().contiguous()
        x = torch.empty(m, 128, dtype=torch.bfloat16, device=spec.device)
        _lib.call("rtts_to_halo", spec.data_ptr(), spec.stride(1), spec.stride(0), nm, 1, b, l, 0, 128, x.data_ptr(), 0, m, _s())
    else:
        x = spec.reshape(m, nm).to(torch.bfloat16)
    h = _ReluDropLinearFn.apply(x, lyr.fc1, lyr.dropout1.p if training else 0.0, kpad, lyr.fc1.weight)
    h = _ReluDropLinearFn.apply(h, lyr.fc2, lyr.dropout2.p if training else 0.0, False, lyr.fc2.weight)
    return proj_pe(h.view(b, l, -1), lyr.projection, pe)


# ------------------------------------------------------------------------------------------------------------------
# Inference (no-grad) forms of the edges: ``ReformerTTS.infer`` / ``Trainer.validate`` (reference reformer_tts.py:145-221,
# training/wrappers.py:107-140) run the prenets, the heads and the postnet in eval mode.  Same kernels as the training
# executors above -- rtts_gemm_nt for every Linear, the implicit-GEMM Conv1d on halo rows, rtts_bn_act_fwd with the RUNNING
# statistics in place of the batch statistics -- no autograd state, no library GEMM.
# ------------------------------------------------------------------------------------------------------------------
def _padded_linear(lin) -> tuple:
    """(weight (N rounded up to 64, K rounded up to 64) bf16 zero padded, bias (N padded) fp32 | None): PERSISTENT buffers on the
    module, refreshed IN PLACE when the master weight changed -- a captured hipGraph (``ReformerTTS.infer(use_graph=True)``) holds
    their addresses and must see the values of the current parameters (``refresh_eval_operands`` runs the refresh eagerly
    before replays)."""
    w = lin.weight
    ver = (w._version, WEIGHT_EPOCH[0], w.data_ptr(), None if lin.bias is None else lin.bias._version)
    cache = getattr(lin, "_rtts_padded", None)
    if cache is None or cache[1].device != w.device:
        n, k = w.shape
        npad, kpad = -(-n // 64) * 64, -(-k // 64) * 64
        wb = torch.zeros(npad, kpad, dtype=torch.bfloat16, device=w.device)
        bias = None if lin.bias is None else torch.zeros(npad, dtype=torch.float32, device=w.device)
        lin._rtts_padded = cache = [None, wb, bias]
    if cache[0] != ver:
        n, k = w.shape
        cache[1][:n, :k].copy_(w.detach())
        if lin.bias is not None:
            cache[2][:n].copy_(lin.bias.detach())
        cache[0] = ver
    return cache[1], cache[2]


def refresh_eval_operands(model) -> None:
    """Bring every cached inference operand of ``model`` (padded Linear weights, GEMM-layout convolution weights) up to date
    with the current parameters, eagerly: called at the start of ``ReformerTTS.infer`` so that graphs captured by an earlier
    call replay with today's weights."""
    for m in model.modules():
        if isinstance(m, torch.nn.Linear) and getattr(m, "_rtts_padded", None) is not None:
            _padded_linear(m)
        ex = getattr(m, "_rtts_k5", None)
        if ex is not None:
            ex.weight_perm()


def linear_nograd(x: torch.Tensor, lin, relu: bool = False, out_f32: bool = False) -> torch.Tensor:
    """y (..., N) = [relu](x W^T + b) on rtts_gemm_nt for a tensor that needs no gradient: rows are rounded up to 128 and the
    input width to 64 in one cast launch (rtts_to_halo with halo 0), the weight's output rows to 64.  bf16 (or fp32) result."""
    shape = x.shape
    k = shape[-1]
    x2 = x.detach().reshape(-1, k)
    m = x2.shape[0]
    wb, bias = _padded_linear(lin)
    n = lin.weight.shape[0]
    mp, kp = -(-m // 128) * 128, wb.shape[1]
    if x2.dtype == torch.bfloat16 and mp == m and kp == k and x2.stride(1) == 1 and x2.stride(0) % 8 == 0:
        a = x2
    else:
        if x2.dtype not in (torch.float32, torch.bfloat16) or x2.stride(1) != 1 or k % 8 != 0 or \
                x2.stride(0) % (4 if x2.dtype == torch.float32 else 8) != 0:
            x2 = torch.nn.functional.pad(x2.float(), (0, -k % 8)).contiguous()         # odd widths: align once (tiny inputs only)
        a = torch.empty(mp, kp, dtype=torch.bfloat16, device=x.device)
        _lib.call("rtts_to_halo", x2.data_ptr(), x2.stride(0), 0, x2.shape[1], int(x2.dtype == torch.float32), 1, m, 0, kp, a.data_ptr(), 0, mp, _s())
    if relu and bias is None:
        bias = torch.zeros(wb.shape[0], dtype=torch.float32, device=x.device)
    y = gemm(a, wb, bias=bias, relu=relu, out_f32=out_f32)
    return y[:m, :n].view(*shape[:-1], n)


def conv_stack_nograd(x: torch.Tensor, layers, convend=None) -> torch.Tensor:
    """Eval-mode convolution stack on halo rows: ``layers`` = [(conv, bn, act)] with act 1 = ReLU, 2 = tanh (BatchNorm on its
    running statistics, dropout off), optionally followed by ``convend`` (a bare Conv1d with bias).  x (B, L, C) fp32 | bf16
    -> (B, L, C_out): bf16 after the last activation, fp32 after ``convend``."""
    b, l, c = x.shape
    dev = x.device
    g = Halo(b, l)
    x2 = x.detach().reshape(b * l, c)
    if x2.dtype not in (torch.float32, torch.bfloat16) or x2.stride(1) != 1 or c % 8 != 0:
        x2 = torch.nn.functional.pad(x2.float(), (0, -c % 8)).contiguous()
    first = layers[0][0] if layers else convend
    ex0 = _conv_exec(first)
    cur = g.new(ex0.cp, dev)
    _lib.call("rtts_to_halo", x2.data_ptr(), x2.stride(0), 0, x2.shape[1], int(x2.dtype == torch.float32), b, l, g.H, ex0.cp, cur.data_ptr(),
              g.LEAD, g.alloc, _s())
    for i, (conv, bn, act) in enumerate(layers):
        ex = _conv_exec(conv)
        y = ex.forward(cur, g)                                    # (mp, cop) fp32, bias not added
        # eval: act(gamma * (y + bias - running_mean) * rsqrt(running_var + eps) + beta): the kernel's "mean" is running_mean - bias
        cop = ex.cop
        mean = torch.zeros(cop, dtype=torch.float32, device=dev)
        rstd = torch.zeros(cop, dtype=torch.float32, device=dev)
        mean[:ex.co] = bn.running_mean - conv.bias.detach()
        rstd[:ex.co] = torch.rsqrt(bn.running_var + bn.eps)
        gamma, beta = _pad_vec(bn.weight.detach(), cop), _pad_vec(bn.bias.detach(), cop)
        last = convend is None and i == len(layers) - 1
        if last:
            z = torch.empty(b * l, cop, dtype=torch.bfloat16, device=dev)
            zargs = (0, 0, b * l)
        else:
            z = g.new(cop, dev)
            zargs = (1, g.LEAD, g.alloc)
        _lib.call("rtts_bn_act_fwd", y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, 0.0, 0,
                  seed_base(dev).data_ptr(), b, l, g.H, cop, z.data_ptr(), *zargs, _s())
        if last:
            return z.view(b, l, cop)[..., :ex.co]
        cur = z
    ex = _conv_exec(convend)
    bias = _pad_vec(convend.bias.detach(), ex.cop)
    y = ex.forward(cur, g, bias=bias)                              # (mp, cop) fp32 halo rows
    return g.valid(y)[..., :ex.co]


def _conv_exec(conv) -> "ConvK5":
    ex = getattr(conv, "_rtts_k5", None)
    if ex is None:
        ex = conv._rtts_k5 = ConvK5(conv)
    return ex


def _pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    if v.shape[0] == n and v.dtype == torch.float32:
        return v.contiguous()
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.shape[0]].copy_(v)
    return out


def nograd_ok(x: torch.Tensor) -> bool:
    """The inference forms apply: a CUDA tensor, no gradient being recorded."""
    return x.is_cuda and not torch.is_grad_enabled()


_engine.FLUSH_HOOKS.append(flush_conv_dw)
