"""Hyper-parameter dataclasses with the reference's field names and defaults
(``/root/reference/reformer_tts/model/config.py:5-84``, ``training/config.py:14-29``), plus a
small YAML override loader for files shaped like ``config/baseline.yml`` (the reference goes
through ``dacite``; only the ``model:`` and ``experiment.tts_training:`` sections matter here)."""
from __future__ import annotations

from dataclasses import asdict, dataclass, field, fields, is_dataclass
from typing import Optional


@dataclass
class FeedForwardConfig:
    hidden: int = 2048
    dropout: float = 0.0


@dataclass
class LSHSelfAttentionConfig:
    implementation: str = "hip"  # reference: "reformer_pytorch" | "huggingface_transformers"
    heads: int = 8
    bucket_size: int = 64
    n_hashes: int = 8
    add_local_attn_hash: bool = False
    attn_chunks: int = 1
    random_rotations_per_head: bool = False
    attend_across_buckets: bool = True
    allow_duplicate_attention: bool = True
    num_mem_kv: int = 0
    one_value_head: bool = False
    use_full_attn: bool = False
    full_attn_thres: Optional[int] = None
    return_attn: bool = False
    post_attn_dropout: float = 0.0
    dropout: float = 0.0


@dataclass
class MultiheadAttentionConfig:
    num_heads: int = 8
    dropout: float = 0.0
    bias: bool = True
    add_bias_kv: bool = False
    add_zero_attn: bool = False
    kdim: Optional[int] = None
    vdim: Optional[int] = None


@dataclass
class ReformerEncConfig:
    depth: int = 6
    ff_chunks: int = 100
    attn_kwargs: LSHSelfAttentionConfig = field(default_factory=LSHSelfAttentionConfig)
    ff_kwargs: FeedForwardConfig = field(default_factory=FeedForwardConfig)


@dataclass
class ReformerDecConfig:
    depth: int = 6
    ff_chunks: int = 100
    attn_kwargs: MultiheadAttentionConfig = field(default_factory=MultiheadAttentionConfig)
    self_attn_kwargs: LSHSelfAttentionConfig = field(default_factory=LSHSelfAttentionConfig)
    ff_kwargs: FeedForwardConfig = field(default_factory=FeedForwardConfig)


@dataclass
class EncoderPreNetConfig:
    dropout: float = 0.5


@dataclass
class DecoderPreNetConfig:
    hidden_size: int = 256
    dropout: float = 0.5


@dataclass
class PostConvNetConfig:
    depth: int = 4
    dropout: float = 0.0


@dataclass
class ReformerTTSConfig:
    num_mel_coeffs: int = 80
    dict_size: int = 76
    embedding_dim: int = 512
    pad_base: int = 128
    scp_encoding_dropout: float = 0.05
    enc_prenet_kwargs: EncoderPreNetConfig = field(default_factory=EncoderPreNetConfig)
    enc_reformer_kwargs: ReformerEncConfig = field(default_factory=ReformerEncConfig)
    dec_prenet_kwargs: DecoderPreNetConfig = field(default_factory=DecoderPreNetConfig)
    dec_reformer_kwargs: ReformerDecConfig = field(default_factory=ReformerDecConfig)
    postnet_kwargs: PostConvNetConfig = field(default_factory=PostConvNetConfig)


@dataclass
class LRSchedulerConfig:
    """``training/config.py:5-10``: exponential decay from ``initial_lr`` to ``final_lr`` between two epochs."""
    initial_lr: float = 1e-4
    final_lr: float = 1e-4
    start_schedule_epoch: int = 0
    end_schedule_epoch: Optional[int] = None


@dataclass
class TTSTrainingConfig:
    batch_size: int = 8
    learning_rate: float = 1e-4
    positive_stop_weight: float = 5.0
    weight_decay: float = 1e-4
    accumulate_grad_batches: int = 1
    gradient_clip_val: float = 0.0
    lr_scheduler: Optional[LRSchedulerConfig] = None
    max_epochs: Optional[int] = None       # experiment.max_epochs: the schedule's end when end_schedule_epoch is None
    warmup_steps: Optional[int] = None
    raw_pred_loss_weight: float = 1.0
    post_pred_loss_weight: float = 1.0
    stop_loss_weight: float = 1.0
    spectrogram_loss: str = "mse"
    # this package's addition (not a key of the reference's TTSTrainingConfig): under data parallelism, BatchNorm statistics
    # over the GLOBAL batch (what the reference's single-process BatchNorm sees) instead of each rank's own rows
    sync_batchnorm: bool = False
    # this package's addition: what the reversible backward recomputes (``engine.RECOMPUTE_MODES``).  "full" is the reference
    # (``model/reversible.py:114-129``: only each stack's output survives the forward, every block is re-run in the backward);
    # "attention-stash" / "output-stash" / "projection-stash" keep the attention outputs / + the block outputs f(x) / + the
    # projections; "stash" keeps the streams too and recomputes nothing (~1 GB more at the baseline shapes, of 288 GB).  A mode
    # whose estimated footprint does not fit the free HBM is lowered, with one log line (``Trainer.resolve_recompute``).
    recompute: str = "stash"


def _merge(dc, overrides: dict):
    """Override-only merge with strict keys (dacite's ``strict=True`` behaviour)."""
    names = {f.name for f in fields(dc)}
    for k, val in overrides.items():
        if k not in names:
            raise KeyError(f"unknown config key {k!r} for {type(dc).__name__}")
        cur = getattr(dc, k)
        if is_dataclass(cur) and isinstance(val, dict):
            _merge(cur, val)
        else:
            setattr(dc, k, val)
    return dc


def model_config_from_dict(d: dict) -> ReformerTTSConfig:
    return _merge(ReformerTTSConfig(), d)


def load_yaml(path: str):
    """-> (ReformerTTSConfig, TTSTrainingConfig) from a reference-style experiment YAML."""
    import yaml
    with open(path) as fh:
        raw = yaml.safe_load(fh) or {}
    model = model_config_from_dict(raw.get("model", {}))
    exp = raw.get("experiment", {})
    tr = dict(exp.get("tts_training", {}))
    for k in ("num_visualizations", "early_stopping_epochs", "noise_std"):      # plotting / early stopping / input noise: the
        tr.pop(k, None)                                                         # harness around the step, not the step
    sched = tr.pop("lr_scheduler", None)
    cfg = _merge(TTSTrainingConfig(), tr)
    if sched is not None:
        cfg.lr_scheduler = _merge(LRSchedulerConfig(), dict(sched))
    if exp.get("max_epochs") is not None:
        cfg.max_epochs = int(exp["max_epochs"])
    return model, cfg


def baseline_model_config() -> ReformerTTSConfig:
    """``config/baseline.yml`` model section resolved."""
    return model_config_from_dict(dict(
        dict_size=76, num_mel_coeffs=80, scp_encoding_dropout=0.05, pad_base=256,
        enc_prenet_kwargs=dict(dropout=0.05), dec_prenet_kwargs=dict(dropout=0.05),
        enc_reformer_kwargs=dict(depth=3),
        dec_reformer_kwargs=dict(depth=3, self_attn_kwargs=dict(bucket_size=128)),
        postnet_kwargs=dict(depth=2, dropout=0.1)))


def baseline_training_config() -> TTSTrainingConfig:
    return _merge(TTSTrainingConfig(), dict(batch_size=12, weight_decay=1e-7, accumulate_grad_batches=5,
                                            gradient_clip_val=1.0, warmup_steps=320))


def long_sequence_model_config() -> ReformerTTSConfig:
    """``config/bucket-size-64-18-06.yml`` model section resolved (BASELINE config #4)."""
    return model_config_from_dict(dict(
        dict_size=76, num_mel_coeffs=80, scp_encoding_dropout=0.05, pad_base=256,
        enc_prenet_kwargs=dict(dropout=0.05), dec_prenet_kwargs=dict(dropout=0.05),
        enc_reformer_kwargs=dict(attn_kwargs=dict(post_attn_dropout=0.15)),
        dec_reformer_kwargs=dict(self_attn_kwargs=dict(post_attn_dropout=0.15), attn_kwargs=dict(dropout=0.15)),
        postnet_kwargs=dict(depth=2, dropout=0.3)))


def as_kwargs(cfg: ReformerTTSConfig) -> dict:
    """``asdict`` form the reference passes to ``ReformerTTS(**...)`` (``training/wrappers.py:37``)."""
    return asdict(cfg)
