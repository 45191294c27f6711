#!/usr/bin/env python3
"""One process, one GPU: the step as ONE hipGraph (the N = 1 bench form: encoder beside decoder, weight gradients handed over)
against the CHAIN of graphs the data-parallel trainer replays (no collectives here: world size 1) -- what the chain's serial
backward costs per step.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
batch = synthetic_batch(12, 200, 1024, seed=42, device=dev)
for mode in ("full", "stash"):
    res = {}
    for seg in (False, True):
        tcfg = baseline_training_config()
        tcfg.batch_size = 12
        tcfg.recompute = mode
        tr = Trainer(build_model(baseline_model_config(), dev, seed=42), tcfg, dev)
        tr.capture(batch, segmented=seg)
        for _ in range(5):
            tr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            tr.replay()
        torch.cuda.synchronize()
        res[seg] = (time.perf_counter() - t0) / 30 * 1e3
        del tr
    print(f"{mode:6s}: one graph {res[False]:.3f} ms/step, chain of graphs {res[True]:.3f} ms/step (+{res[True] - res[False]:.3f})", flush=True)
