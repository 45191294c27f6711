#!/usr/bin/env python3
"""Trainer.fit over REAL (ragged) batches with the per-shape graph cache against (a) replay() of one captured full-size step
and (b) eager fit, config/baseline.yml on one MI355X: ms per optimizer step and mel-frames/s of VALID frames.

Synthetic LJSpeech-shaped epoch: every batch has 12 utterances, text lengths U{50..200}, mel lengths U{L_lo..L_hi} with the
batch maximum drawn per batch (so the padded shapes vary: 1-4 multiples of 256 frames), collated by
dataset.custom_sequence_padder into pinned host tensors.  accumulate_grad_batches = 1 here so that a fit step and a replay
are the same amount of work when the batch is full-size.

    python scripts/fit_bench.py > gpurun_out/fit_bench.json"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd.dataset import custom_sequence_padder  # noqa: E402
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402


def epoch(n_batches, gen, full=False):
    out = []
    for _ in range(n_batches):
        top = 1024 if full else int(torch.randint(300, 1025, (1,), generator=gen))
        items = []
        for k in range(12):
            lm = top if k == 0 else int(torch.randint(max(100, top // 2), top + 1, (1,), generator=gen))
            lp = 200 if (full and k == 0) else int(torch.randint(50, 201, (1,), generator=gen))
            items.append(dict(phonemes=torch.randint(1, 77, (lp,), generator=gen),
                              spectrogram=(torch.randn(lm, 80, generator=gen) * 2 - 5).clamp(-11.5, 2.0)))
        out.append(custom_sequence_padder(items, pin_memory=True))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    res = {"workload": "config/baseline.yml, B=12, accumulate_grad_batches 1, one MI355X"}

    def trainer():
        tcfg = baseline_training_config()
        tcfg.batch_size, tcfg.accumulate_grad_batches, tcfg.lr_scheduler = 12, 1, None
        return Trainer(build_model(baseline_model_config(), dev, seed=42), tcfg, dev)

    def timed_fit(tr, batches, graphs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = tr.fit(batches, graphs=graphs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        frames = sum(int(b["loss_mask"][:, :, 0].sum()) for b in batches)
        return dict(ms_per_step=round(1e3 * dt / len(batches), 3), valid_frames_per_s=round(frames / dt, 1), steps=len(batches),
                    final_loss=round(float(losses[-1]), 4))

    # (1) full-size batches: fit over the graph cache against replay() of the same shape
    full = epoch(args.batches, gen, full=True)
    tr = trainer()
    tr.fit(full[:3], graphs=True)                      # captures (one shape) + warm-up
    res["fit_graphs_full_size"] = timed_fit(tr, full, True)
    tr2 = trainer()
    tr2.capture(synthetic_batch(12, 200, 1024, device=dev))
    for _ in range(3):
        tr2.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.batches):
        tr2.replay()
    torch.cuda.synchronize()
    res["replay_fixed_batch"] = dict(ms_per_step=round(1e3 * (time.perf_counter() - t0) / args.batches, 3), steps=args.batches)
    res["fit_over_replay"] = round(res["fit_graphs_full_size"]["ms_per_step"] / res["replay_fixed_batch"]["ms_per_step"], 4)
    # (2) a ragged epoch: graph cache against eager
    ragged = epoch(args.batches, gen)
    tr = trainer()
    tr.fit(ragged, graphs=True)                        # first pass: captures every padded shape of the epoch
    res["padded_shapes_captured"] = sorted(str(k) for k, v in tr._shape_graphs.items() if v is not None)
    res["fit_graphs_ragged"] = timed_fit(tr, ragged, True)
    tr3 = trainer()
    tr3.fit(ragged[:3], graphs=False)
    res["fit_eager_ragged"] = timed_fit(tr3, ragged, False)
    res["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
