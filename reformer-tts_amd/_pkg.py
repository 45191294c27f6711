"""MI355X-native Reformer-TTS training hot path."""
__version__ = "0.1.0"
