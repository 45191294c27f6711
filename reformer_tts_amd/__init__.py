"""Import alias: the package directory is ``reformer-tts_amd/`` (not a valid Python
identifier), so ``import reformer_tts_amd`` resolves here and forwards its search path."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "reformer-tts_amd")]
from ._pkg import *  # noqa: F401,F403,E402
