#!/usr/bin/env python3
"""Soak: N replays of the captured training step on one fixed synthetic batch (config/baseline.yml, B=12): the loss must
stay finite and go down (fresh LSH rotations and dropout masks every step).
    python scripts/soak.py --steps 400 > gpurun_out/soak.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
args = ap.parse_args()
dev = torch.device("cuda:0")
tcfg = baseline_training_config()
tcfg.batch_size = 12
model = build_model(baseline_model_config(), dev, seed=42)
tr = Trainer(model, tcfg, dev)
batch = synthetic_batch(12, 200, 1024, seed=42, device=dev)
tr.capture(batch)
losses = []
for i in range(args.steps):
    out = tr.replay()
    if i % 20 == 0 or i == args.steps - 1:
        losses.append((tr.global_step, [round(float(x), 4) for x in out]))
torch.cuda.synchronize()
finite = all(all(v == v and abs(v) < 1e9 for v in l) for _, l in losses)
print(json.dumps({"workload": "config/baseline.yml, B=12, mel 1024, one fixed synthetic batch, hipGraph replay",
                  "steps": args.steps, "finite": finite, "loss_total_raw_post_stop_every_20_steps": losses,
                  "params_finite": bool(torch.isfinite(tr.flat_p).all())}))
assert finite and losses[-1][1][0] < losses[0][1][0]
