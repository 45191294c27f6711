"""CPU restatement of the SqueezeWave vocoder's INFERENCE path -- TEST INFRASTRUCTURE, not the product.

Follows ``/root/reference/reformer_tts/squeeze_wave/modules.py``: ``SqueezeWave.infer`` (:334-376), ``WN.forward``
(:203-235), ``InvertibleConv1d.reverse_forward`` (:65-85), ``DepthwiseSeparableConv1d`` (:88-122) and the gate
``fused_add_tanh_sigmoid_multiply`` (:10-24), functionally over a reference-named ``state_dict`` (weight-norm
parameters ``weight_g`` / ``weight_v``, BatchNorm running statistics) in fp32.  Pinned by goldens generated from the
reference's own module (``tests/golden/make_golden.py``, ``squeezewave_*.npz``).  Only ``tests/``, ``smoke()`` and
benchmark baselines may import this."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def default_cfg() -> dict:
    """``squeeze_wave/config.py:5-21`` (SqueezeWaveConfig / WNConfig defaults) + the dataset's 80 mel channels."""
    return dict(n_mel_channels=80, n_flows=12, n_audio_channels=128, early_return_interval=2, early_return_size=16,
                wn_config=dict(n_layers=8, n_channels=256, conv_kernel_size=3, mel_upsample_scale=2))


def small_cfg() -> dict:
    """4 flows, 16 audio channels (so the mel is upsampled 256 // 16 = 16 times: mel_upsample_scale must match), WN 2 x 32."""
    return dict(n_mel_channels=80, n_flows=4, n_audio_channels=16, early_return_interval=2, early_return_size=4,
                wn_config=dict(n_layers=2, n_channels=32, conv_kernel_size=3, mel_upsample_scale=16))


def _return_early(cfg, k):
    return k % cfg["early_return_interval"] == 0 and k > 0


def flow_channels(cfg):
    """(n_half, n_remaining) per flow and the final n_remaining (``modules.py:277-290``)."""
    n_half, n_rem, out = cfg["n_audio_channels"] // 2, cfg["n_audio_channels"], []
    for k in range(cfg["n_flows"]):
        if _return_early(cfg, k):
            n_half -= cfg["early_return_size"] // 2
            n_rem -= cfg["early_return_size"]
        out.append((n_half, n_rem))
    return out, n_rem


def _wn_weight(sd: SD, p: str) -> torch.Tensor:
    """torch.nn.utils.weight_norm(name='weight', dim=0): w = g * v / ||v|| per output channel; plain weight if removed."""
    if p + "weight" in sd:
        return sd[p + "weight"]
    v, g = sd[p + "weight_v"], sd[p + "weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def wn_forward(sd: SD, p: str, audio: torch.Tensor, mel: torch.Tensor, wn: dict) -> torch.Tensor:
    """``modules.py:203-235``: audio (B, n_half, L), mel (B, n_mel, Lm) -> (B, 2*n_half, L).  BatchNorm in eval mode."""
    c, nl = wn["n_channels"], wn["n_layers"]
    audio = F.conv1d(audio, _wn_weight(sd, p + "start_conv."), sd[p + "start_conv.bias"])
    cond = F.conv1d(mel, _wn_weight(sd, p + "cond_layer."), sd[p + "cond_layer.bias"])
    for i in range(nl):
        spec = cond[:, i * 2 * c:(i + 1) * 2 * c, :]
        if audio.shape[2] > spec.shape[2]:
            spec = F.interpolate(spec, scale_factor=wn["mel_upsample_scale"], mode="nearest")
        q = f"{p}in_layers.{i}.layer."
        x = F.batch_norm(audio, sd[q + "0.running_mean"], sd[q + "0.running_var"], sd[q + "0.weight"], sd[q + "0.bias"], False, 0.1, 1e-5)
        x = F.conv1d(x, sd[q + "1.weight"], sd[q + "1.bias"], padding=(wn["conv_kernel_size"] - 1) // 2, groups=c)
        x = F.conv1d(x, sd[q + "2.weight"], sd[q + "2.bias"])
        s = x + spec
        acts = torch.tanh(s[:, :c]) * torch.sigmoid(s[:, c:])
        audio = audio + F.conv1d(acts, _wn_weight(sd, f"{p}res_skip_layers.{i}."), sd[f"{p}res_skip_layers.{i}.bias"])
    return F.conv1d(audio, sd[p + "end_conv.weight"], sd[p + "end_conv.bias"])


def infer(sd: SD, cfg: dict, mel: torch.Tensor, sigma: float = 0.6,
          normal: Optional[Callable[[tuple], torch.Tensor]] = None) -> torch.Tensor:
    """``modules.py:334-376``: mel (B, n_mel, Lm) -> audio (B, 256 * Lm) clamped to [-1, 1].  ``normal(shape)`` supplies the
    Gaussian draws in the reference's order (initial noise, then one block per early-return flow, last flow first);
    the default draws from torch's global CPU generator exactly like the reference does."""
    if normal is None:
        normal = lambda shape: torch.empty(*shape).normal_()      # noqa: E731
    b = mel.shape[0]
    chans, n_rem = flow_channels(cfg)
    length = mel.shape[2] * (256 // cfg["n_audio_channels"])
    audio = normal((b, n_rem, length))
    for k in reversed(range(cfg["n_flows"])):
        half = audio.shape[1] // 2
        a0, a1 = audio[:, :half], audio[:, half:]
        out = wn_forward(sd, f"wn_layers.{k}.", a0, mel, cfg["wn_config"])
        s, bb = out[:, :half], out[:, half:]
        a1 = (a1 - bb) / torch.exp(s)
        audio = torch.cat([a0, a1], 1)
        w = sd[f"inv_conv_layers.{k}.conv.weight"].squeeze(-1)
        audio = F.conv1d(audio, w.float().inverse()[..., None])
        if _return_early(cfg, k):
            audio = torch.cat((sigma * normal((b, cfg["early_return_size"], length)), audio), 1)
    audio = audio.permute(0, 2, 1).contiguous().view(b, -1)
    return torch.clamp(audio, -1, 1)
