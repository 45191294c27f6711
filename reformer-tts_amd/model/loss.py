"""``/root/reference/reformer_tts/model/loss.py:7-53``."""
from typing import Tuple

from torch import Tensor, nn
from torch.nn.functional import binary_cross_entropy_with_logits


class TTSLoss(nn.Module):
    def __init__(self, pos_weight: Tensor, raw_pred_loss_weight: float = 1.0, post_pred_loss_weight: float = 1.0,
                 stop_loss_weight: float = 1.0, spectrogram_loss: str = "mse"):
        super().__init__()
        self.pos_weight = pos_weight
        self.raw_pred_loss_weight = raw_pred_loss_weight
        self.post_pred_loss_weight = post_pred_loss_weight
        self.stop_loss_weight = stop_loss_weight
        if spectrogram_loss == "mse":
            self.spectrogram_loss = nn.MSELoss()
        elif spectrogram_loss == "l1":
            self.spectrogram_loss = nn.L1Loss()
        else:
            raise RuntimeError(f"Unsupported loss type: {spectrogram_loss}")

    def forward(self, raw_mel_out, postnet_mel_out, stop_out, true_mel, true_stop, true_mask) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """Masked predictions, mean over ALL elements (padded frames stay in the denominator).
        The reference multiplies its arguments in place; the out-of-place product used here has
        the same value and gradient and does not invalidate the postnet's saved input."""
        assert raw_mel_out.shape == postnet_mel_out.shape == true_mask.shape == true_mel.shape
        assert stop_out.shape == true_stop.shape
        raw_mel_loss = self.spectrogram_loss(raw_mel_out * true_mask, true_mel)
        postnet_mel_loss = self.spectrogram_loss(postnet_mel_out * true_mask, true_mel)
        stop_loss = binary_cross_entropy_with_logits(stop_out, true_stop, pos_weight=self.pos_weight.to(stop_out.device))
        total = raw_mel_loss * self.raw_pred_loss_weight + postnet_mel_loss * self.post_pred_loss_weight \
            + stop_loss * self.stop_loss_weight
        return total, raw_mel_loss, postnet_mel_loss, stop_loss
