from .trainer import Trainer, build_model, synthetic_batch  # noqa: F401
