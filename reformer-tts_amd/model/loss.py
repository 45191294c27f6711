"""The reference's ``TTSLoss`` surface (``/root/reference/reformer_tts/model/loss.py:7-53``: constructor arguments, the four
returned values, the assertions) over ONE launch of ``rtts_tts_loss`` (csrc/edges.hip): masked MSE | L1 means over all
elements for the raw and the postnet prediction, BCE-with-logits(pos_weight) for the stop token, and the three gradients of the
weighted total in the same pass (two-stage deterministic reduction; no ATen reduction, which matters under hipGraph replay).

GPU only, like everything on the product path.  The training step proper does not come through here: ``edges.PostnetLoss``
calls the same kernel fused behind the heads and the postnet; this module serves callers that hold the three predictions
(validation, tests, the reference's ``LitReformerTTS.forward`` shape of code)."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor, nn

from .. import _lib

_KINDS = {"mse": 0, "l1": 1}


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, post, stop, true_mel, true_stop, true_mask, kind, pos_weight, weights):
        if not raw.is_cuda:
            raise _lib.RttsError("TTSLoss: GPU only (no CPU fallback for the HIP path)")
        b, l, nm = raw.shape
        rows, dev = b * l, raw.device
        f32 = dict(dtype=torch.float32, device=dev)
        r2, p2 = raw.detach().reshape(rows, nm).float().contiguous(), post.detach().reshape(rows, nm).float().contiguous()
        s1 = stop.detach().reshape(rows).float().contiguous()
        tgt, msk = true_mel.reshape(rows, nm).float().contiguous(), true_mask.reshape(rows, nm).float().contiguous()
        tst = true_stop.reshape(rows).float().contiguous()
        grads = torch.empty(2 * rows * nm + rows, **f32)
        d_raw, d_post, d_stop = grads[:rows * nm], grads[rows * nm:2 * rows * nm], grads[2 * rows * nm:]
        losses, ws = torch.empty(4, **f32), torch.empty(1536, **f32)
        _lib.call("rtts_tts_loss", r2.data_ptr(), p2.data_ptr(), nm, tgt.data_ptr(), msk.data_ptr(), s1.data_ptr(), 1, tst.data_ptr(),
                  rows, nm, kind, float(pos_weight), float(weights[0]), float(weights[1]), float(weights[2]), d_raw.data_ptr(),
                  d_post.data_ptr(), nm, d_stop.data_ptr(), losses.data_ptr(), ws.data_ptr(), l, l, None, 0, 0, 0, 0, 0,
                  None, 0, 0, torch.cuda.current_stream().cuda_stream)
        ctx.save_for_backward(grads)
        ctx.meta = (raw.shape, stop.shape, weights, raw.dtype, post.dtype, stop.dtype)
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, g_total, g_raw, g_post, g_stop):
        (grads,) = ctx.saved_tensors
        shape, stop_shape, w, dt_raw, dt_post, dt_stop = ctx.meta
        n = shape[0] * shape[1] * shape[2]

        def scale(g_part, weight):
            # the kernel stored d(total)/d(prediction) = weight * d(part)/d(prediction)
            if g_part is None or weight == 0.0:
                return g_total
            return g_total + g_part / weight

        d_raw = (grads[:n] * scale(g_raw, w[0])).view(shape).to(dt_raw)
        d_post = (grads[n:2 * n] * scale(g_post, w[1])).view(shape).to(dt_post)
        d_stop = (grads[2 * n:] * scale(g_stop, w[2])).view(stop_shape).to(dt_stop)
        return d_raw, d_post, d_stop, None, None, None, None, None, None


class TTSLoss(nn.Module):
    def __init__(self, pos_weight: Tensor, raw_pred_loss_weight: float = 1.0, post_pred_loss_weight: float = 1.0,
                 stop_loss_weight: float = 1.0, spectrogram_loss: str = "mse"):
        super().__init__()
        if spectrogram_loss not in _KINDS:
            raise RuntimeError(f"Unsupported loss type: {spectrogram_loss}")
        self.kind = _KINDS[spectrogram_loss]
        self.pos_weight = pos_weight
        self._pos_weight_host = float(pos_weight)          # read once: no device read inside a (possibly captured) step
        self.raw_pred_loss_weight, self.post_pred_loss_weight, self.stop_loss_weight = (
            float(raw_pred_loss_weight), float(post_pred_loss_weight), float(stop_loss_weight))

    def forward(self, raw_mel_out, postnet_mel_out, stop_out, true_mel, true_stop, true_mask) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """-> (total, raw, postnet, stop).  Predictions are masked, every mean runs over ALL elements (padded frames stay in
        the denominator), as the reference's does; unlike the reference nothing is modified in place."""
        assert raw_mel_out.shape == postnet_mel_out.shape == true_mask.shape == true_mel.shape
        assert stop_out.shape == true_stop.shape
        return _LossFn.apply(raw_mel_out, postnet_mel_out, stop_out, true_mel, true_stop, true_mask, self.kind,
                             self._pos_weight_host, (self.raw_pred_loss_weight, self.post_pred_loss_weight, self.stop_loss_weight))
