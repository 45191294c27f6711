#!/usr/bin/env python3
"""Which ATen operators does one eager training step still call, and from which line of this package (GPU box only)?

A TorchDispatchMode (host side only: no profiler, no tracing library) records every operator that reaches the dispatcher
during ONE Trainer.train_step, forward and backward (the mode travels to the autograd worker thread with the thread-local
state), and prints them grouped by (operator, innermost frame inside reformer-tts_amd/).  Views and metadata operators
launch nothing and are listed separately.
    python scripts/aten_trace.py [--batch 12 --text-len 256 --mel-len 1024]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

NO_LAUNCH = {"view", "_unsafe_view", "reshape", "as_strided", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze",
             "squeeze", "detach", "alias", "empty", "empty_like", "empty_strided", "new_empty", "unbind", "split", "split_with_sizes",
             "narrow", "chunk", "sym_size", "sym_stride", "sym_numel", "is_pinned", "_local_scalar_dense", "lift_fresh", "unflatten",
             "new_empty_strided", "record_stream", "is_same_size", "stride", "size", "dim", "numel", "is_contiguous", "set_", "resize_",
             "_reshape_alias", "view_as", "flatten", "unfold", "result_type", "can_cast", "is_nonzero", "diagonal", "movedim", "swapaxes"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        where = "(outside the package)"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "reformer-tts_amd" in fr.filename and "aten_trace" not in fr.filename:
                where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                break
        cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
        self.seen[(name, where, cuda or name in ("zeros", "ones", "full", "arange", "randn", "empty"))] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--text-len", type=int, default=256)
    ap.add_argument("--mel-len", type=int, default=1024)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
    from reformer_tts_amd.training import Trainer, build_model, synthetic_batch
    tcfg = baseline_training_config()
    tcfg.batch_size = args.batch
    trainer = Trainer(build_model(baseline_model_config(), dev, seed=42), tcfg, dev)
    batch = synthetic_batch(args.batch, args.text_len, args.mel_len, seed=42, device=dev)
    for _ in range(2):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    log = Log()
    with log:
        trainer.train_step(batch)
    torch.cuda.synchronize()
    launch = [(k, n) for k, n in log.seen.items() if k[0] not in NO_LAUNCH and k[2]]
    quiet = sum(n for k, n in log.seen.items() if k[0] in NO_LAUNCH or not k[2])
    print(f"operators that can launch: {sum(n for _, n in launch)} calls in one step ({quiet} view / metadata / host calls not listed)")
    for (name, where, _), n in sorted(launch, key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"  {n:3d} x {name:28s} {where}")


if __name__ == "__main__":
    main()
