from .config import SqueezeWaveConfig, WNConfig  # noqa: F401
from .modules import SqueezeWave  # noqa: F401
