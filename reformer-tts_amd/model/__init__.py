"""Host-side mirror of the reference's ``reformer_tts.model`` package (same class names,
constructor arguments, state_dict names and error behaviour) with the hot path on MI355X."""
from .config import *  # noqa: F401,F403
from .loss import TTSLoss  # noqa: F401
from .reformer_tts import ReformerTTS, pad_to_multiple  # noqa: F401
