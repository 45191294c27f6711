"""``/root/reference/reformer_tts/squeeze_wave/config.py:5-21``: same fields, same defaults."""
from dataclasses import dataclass, field


@dataclass
class WNConfig:
    # in_audio_channels and in_mel_channels are passed explicitly
    n_layers: int = 8
    n_channels: int = 256
    conv_kernel_size: int = 3
    mel_upsample_scale: int = 2      # must equal 256 // n_audio_channels


@dataclass
class SqueezeWaveConfig:
    n_mel_channels: int
    n_flows: int = 12
    n_audio_channels: int = 128
    early_return_interval: int = 2
    early_return_size: int = 16
    wn_config: WNConfig = field(default_factory=WNConfig)
