"""Host-side logic of the training harness against the oracle's restatement of the reference (no GPU): the learning-rate
schedule (``/root/reference/reformer_tts/training/wrappers.py:258-294``: exponential per-epoch decay + warm-up hook), the
stop-token MAE (``wrappers.py:74-80``) and the YAML -> config mapping of the ``lr_scheduler`` section
(``training/config.py:5-10,24``)."""
import os
import tempfile

import pytest
import torch

from oracle import optim_ref


def _trainer(cfg):
    """A Trainer shell that has only what the schedule needs (no model, no device buffers)."""
    from reformer_tts_amd.training import Trainer
    tr = Trainer.__new__(Trainer)
    tr.cfg, tr.global_step, tr.epoch = cfg, 0, 0
    sch = cfg.lr_scheduler
    tr._base_lr = float(cfg.learning_rate if sch is None else sch.initial_lr)
    tr._lr = tr._base_lr
    return tr


@pytest.mark.parametrize("warmup,sched", [
    (None, None), (7, None),
    (5, dict(initial_lr=3e-4, final_lr=3e-6, start_schedule_epoch=1, end_schedule_epoch=6)),
    (30, dict(initial_lr=1e-3, final_lr=1e-5, start_schedule_epoch=2, end_schedule_epoch=None)),      # warm-up spans epochs
    (None, dict(initial_lr=1e-4, final_lr=1e-4, start_schedule_epoch=1, end_schedule_epoch=3)),
])
def test_learning_rate_follows_the_reference_schedule(warmup, sched):
    from reformer_tts_amd.model.config import LRSchedulerConfig, TTSTrainingConfig
    cfg = TTSTrainingConfig(learning_rate=2e-4, warmup_steps=warmup, max_epochs=9,
                            lr_scheduler=None if sched is None else LRSchedulerConfig(**sched))
    tr = _trainer(cfg)
    steps_per_epoch, epochs = 11, 9
    want = optim_ref.lr_trajectory(steps_per_epoch, epochs, 2e-4, warmup, sched, max_epochs=9)
    got = []
    for _ in range(epochs):
        for _ in range(steps_per_epoch):
            peek = tr.lr_now_for(tr.global_step)                 # reading the rate (logging) changes nothing ...
            state = tr._lr
            assert tr.lr_now_for(tr.global_step) == peek and tr._lr == state
            got.append(tr.take_lr(tr.global_step))               # ... taking the step does (what set_step_hyper calls)
            assert got[-1] == peek
            tr.global_step += 1
        tr.end_epoch()
    assert len(got) == len(want)
    rel = max(abs(a - b) / b for a, b in zip(got, want))
    print(f"\n[lr schedule warmup={warmup} sched={sched}] max rel difference {rel:.2e} over {len(got)} steps; "
          f"first {got[0]:.3e}, last {got[-1]:.3e}")
    assert rel < 1e-12
    if sched is not None and sched["final_lr"] != sched["initial_lr"] and (warmup or 0) < steps_per_epoch:
        end = sched["end_schedule_epoch"] or 9
        # after the schedule's last epoch the rate has arrived at final_lr * exp(-gamma) ** (extra closed end point) -- what
        # the reference's inclusive range start <= epoch <= end gives: (end - start + 1) factors for (end - start) intervals
        gamma = (torch.log(torch.tensor(sched["initial_lr"])) - torch.log(torch.tensor(sched["final_lr"]))) / (end - sched["start_schedule_epoch"])
        assert abs(got[-1] - sched["final_lr"] * float(torch.exp(-gamma))) / got[-1] < 1e-5


def test_schedule_rejects_a_start_before_epoch_one():
    from reformer_tts_amd.model.config import LRSchedulerConfig, TTSTrainingConfig
    tr = _trainer(TTSTrainingConfig(lr_scheduler=LRSchedulerConfig(start_schedule_epoch=0, end_schedule_epoch=3)))
    with pytest.raises(AssertionError, match="start_schedule_epoch has to be >= 1"):
        tr.end_epoch()


def test_schedule_is_validated_before_training_starts():
    """The reference asserts on the scheduler configuration in configure_optimizers (wrappers.py:258-279), before the first
    step; so does Trainer.__init__ (not only end_epoch, a whole epoch later)."""
    from reformer_tts_amd.model.config import LRSchedulerConfig, TTSTrainingConfig, model_config_from_dict
    from reformer_tts_amd.training import Trainer, build_model
    from oracle import model_ref
    cfg = model_ref.small_cfg()
    cfg["enc_reformer_kwargs"]["attn_kwargs"]["implementation"] = "hip"
    cfg["dec_reformer_kwargs"]["self_attn_kwargs"]["implementation"] = "hip"
    model = build_model(model_config_from_dict(cfg))
    for sched, exc, msg in ((LRSchedulerConfig(start_schedule_epoch=0, end_schedule_epoch=3), AssertionError, "has to be >= 1"),
                            (LRSchedulerConfig(start_schedule_epoch=2, end_schedule_epoch=None), ValueError, "max_epochs must be set"),
                            (LRSchedulerConfig(start_schedule_epoch=3, end_schedule_epoch=3), ValueError, "must end after it starts")):
        with pytest.raises(exc, match=msg):
            Trainer(model, TTSTrainingConfig(lr_scheduler=sched), "cpu")


def test_stop_mae_matches_the_reference_formula():
    from reformer_tts_amd.training.trainer import stop_mae
    g = torch.Generator().manual_seed(0)
    for b, l in ((1, 5), (4, 37), (12, 1024)):
        for case in range(6):
            logits = torch.randn(b, l, generator=g) - (2.0 if case % 2 else 0.0)
            if case == 2:
                logits = -logits.abs()                    # no positive logit anywhere: the reference lands on frame 0
            if case == 3:
                logits = logits.abs()                     # every logit positive: frame 0
            if case == 4:
                logits[:, : l // 2] = -1.0                # first positive somewhere in the second half
            tok = torch.zeros(b, l)
            tok[torch.arange(b), torch.randint(0, l, (b,), generator=g)] = 1.0
            assert torch.equal(stop_mae(logits, tok), optim_ref.stop_mae(logits.unsqueeze(-1), tok)), (b, l, case)


def test_yaml_lr_scheduler_section_is_kept():
    from reformer_tts_amd.model.config import load_yaml
    text = """
experiment:
  max_epochs: 40
  tts_training:
    batch_size: 20
    learning_rate: 0.0003
    warmup_steps: 320
    num_visualizations: 2
    lr_scheduler:
      initial_lr: 0.0003
      final_lr: 0.00001
      start_schedule_epoch: 3
model:
  pad_base: 256
"""
    with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as fh:
        fh.write(text)
    try:
        _, tr = load_yaml(fh.name)
    finally:
        os.unlink(fh.name)
    assert tr.lr_scheduler is not None and tr.lr_scheduler.final_lr == 1e-5 and tr.lr_scheduler.start_schedule_epoch == 3
    assert tr.lr_scheduler.end_schedule_epoch is None and tr.max_epochs == 40 and tr.batch_size == 20
    with pytest.raises(KeyError, match="unknown config key"):
        from reformer_tts_amd.model.config import LRSchedulerConfig, _merge
        _merge(LRSchedulerConfig(), dict(initial=1.0))


def test_recompute_mode_is_a_yaml_loadable_configuration_field():
    """``tts_training.recompute`` (this package's addition to the reference's TTSTrainingConfig): what the reversible backward
    recomputes -- "full" = ``/root/reference/reformer_tts/model/reversible.py:114-129`` -- is a configuration the Trainer reads,
    not a module global somebody pokes; unknown names are refused; the engine's named modes round-trip."""
    from reformer_tts_amd import engine
    from reformer_tts_amd.model.config import TTSTrainingConfig, load_yaml
    assert TTSTrainingConfig().recompute == "stash"
    text = "experiment:\n  tts_training:\n    batch_size: 12\n    recompute: full\nmodel:\n  pad_base: 256\n"
    with tempfile.NamedTemporaryFile("w", suffix=".yml", delete=False) as fh:
        fh.write(text)
    try:
        _, tr = load_yaml(fh.name)
    finally:
        os.unlink(fh.name)
    assert tr.recompute == "full"
    before = engine.recompute_mode()
    try:
        for mode in engine.RECOMPUTE_MODES:
            engine.set_recompute(mode)
            assert engine.recompute_mode() == mode
            assert engine._streams_kept() == (mode == "stash")
        assert (engine.STASH_ATTENTION, engine.STASH_BLOCK_OUTPUT, engine.STASH_PROJECTIONS, engine.STASH_STREAMS) == (True,) * 4
        engine.set_recompute("full")
        assert not (engine.STASH_ATTENTION or engine.STASH_BLOCK_OUTPUT or engine.STASH_PROJECTIONS or engine.STASH_STREAMS)
        with pytest.raises(ValueError, match="recompute mode"):
            engine.set_recompute("some")
    finally:
        engine.set_recompute(before)


def test_stash_footprint_estimate_of_the_baseline_model():
    """``engine.stash_bytes`` on the executors' program of config/baseline.yml at B = 12 (rows 3072 / 12288): "full" holds
    nothing, every higher mode holds more, and "stash" is the ~1 GB DESIGN.md section 4 quotes (peak_hbm_gb of the two bench legs)."""
    from reformer_tts_amd import engine
    from reformer_tts_amd.model.config import baseline_model_config
    from reformer_tts_amd.training import build_model
    model = build_model(baseline_model_config())
    enc, dec = engine.build_program(model.enc.reformer.layers), engine.build_program(model.dec.reformer.layers)
    assert enc is not None and dec is not None
    tot = {m: engine.stash_bytes(enc, 12 * 256, 512, m) + engine.stash_bytes(dec, 12 * 1024, 512, m, 12 * 256) for m in engine.RECOMPUTE_MODES}
    print("\n[stash footprint, baseline.yml B=12] " + ", ".join(f"{m}: {v / 2**20:.0f} MiB" for m, v in tot.items()))
    assert tot["full"] == 0
    vals = [tot[m] for m in engine.RECOMPUTE_MODES]
    assert vals == sorted(vals) and len(set(vals)) == len(vals)
    assert 0.6 * 2**30 < tot["stash"] < 1.6 * 2**30
    # one decoder LSH sublayer in attention-stash: the attention output (bf16) + one fp32 logsumexp per token.head
    lsh_only = [("half", dec[0][1], None)]
    assert engine.stash_bytes(lsh_only, 12288, 512, "attention-stash") == 12288 * 512 * 2 + 12288 * 8 * 4
