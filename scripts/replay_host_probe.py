#!/usr/bin/env python3
"""Is the replayed step ever waiting for the HOST?  (GPU box.)

A rocprofv3 kernel trace of bench.py shows ~0.2 ms per step with no queue busy, always inside the first 0.3 ms of a step
(profiles/r04_step_timeline.log).  This probe separates the candidates without a profiler:
  a  trainer.replay() back to back, as bench.py does (per-step words written, two pinned copies, one graph launch)
  b  the graph alone, back to back (no per-step words: every step repeats the same dropout masks -- timing only)
  c  like a, but the host may run at most ONE step ahead (a synchronize every step on the step before)
and reports the host time per call next to the device time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tcfg = baseline_training_config()
    tcfg.batch_size = 12
    tcfg.recompute = os.environ.get("RTTS_PROBE_RECOMPUTE", "full")
    tr = Trainer(build_model(baseline_model_config(), dev, seed=42), tcfg, dev)
    batch = synthetic_batch(12, 200, 1024, seed=42, device=dev)
    tr.capture(batch, segmented=False)
    for _ in range(5):
        tr.replay()
    torch.cuda.synchronize()

    def run(name, fn, sync_prev=False):
        torch.cuda.synchronize()
        host = []
        prev = None
        t0 = time.perf_counter()
        for _ in range(steps):
            h0 = time.perf_counter()
            fn()
            host.append(time.perf_counter() - h0)
            if sync_prev:
                ev = torch.cuda.Event()
                ev.record()
                if prev is not None:
                    prev.synchronize()
                prev = ev
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        host.sort()
        print(f"{name:48s} {dt * 1e3:7.3f} ms/step   host per call: median {host[len(host) // 2] * 1e3:6.3f} ms, max {host[-1] * 1e3:6.3f} ms", flush=True)

    for rep in range(2):
        run("a  trainer.replay()", tr.replay)
        run("b  graph.replay() alone", tr._graph.replay)
        run("c  trainer.replay(), host one step ahead at most", tr.replay, sync_prev=True)


if __name__ == "__main__":
    main()
