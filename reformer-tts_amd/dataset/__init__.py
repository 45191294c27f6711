from .utils import BatchPrefetcher, custom_sequence_padder  # noqa: F401
