#!/usr/bin/env python3
"""Timeline of an UNPROFILED replay of the one-graph training step (GPU box): marker kernels captured into the graph write the
device's 100 MHz wall clock (rtts_debug_stamp); after a few replays the markers of the last one are printed relative to the step's
first kernel.  Answers what a rocprofv3 trace cannot (its own launch overhead shifts the start of the second queue): when does each
branch of the forward / backward start, and which one is the decoder's first cross-attention waiting for?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd import engine  # noqa: E402
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tcfg = baseline_training_config()
    tcfg.batch_size = 12
    tcfg.recompute = os.environ.get("RTTS_PROBE_RECOMPUTE", "full")
    tr = Trainer(build_model(baseline_model_config(), dev, seed=42), tcfg, dev)
    batch = synthetic_batch(12, 200, 1024, seed=42, device=dev)
    buf = torch.zeros(64, dtype=torch.int64, device=dev)
    engine.STAMPS = (buf, {})
    tr.capture(batch, segmented=False)
    names = dict(engine.STAMPS[1])
    rows = []
    for _ in range(8):
        tr.replay()
        torch.cuda.synchronize()
        rows.append(buf.cpu().clone())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        tr.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"{a.elapsed_time(b) / 20:.3f} ms/step with {len(names)} marker kernels in the graph ({tcfg.recompute} recompute)")
    t0 = names["step: first kernel of the forward"]
    for k in (-3, -2, -1):
        r = rows[k]
        print(f"-- replay {len(rows) + k}")
        for name, slot in sorted(names.items(), key=lambda kv: int(r[kv[1]])):
            print(f"   {(int(r[slot]) - int(r[t0])) / 100.0:9.1f} us  {name}")


if __name__ == "__main__":
    main()
