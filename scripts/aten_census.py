#!/usr/bin/env python3
"""Which Python lines still issue ATen copy/fill/add/... ops inside one eager training step (GPU box only).
    python scripts/aten_census.py > gpurun_out/aten_census.txt"""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from reformer_tts_amd.model.config import TTSTrainingConfig, baseline_model_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.as_strided", "aten.t.", "aten.transpose", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.empty", "aten.alias", "aten.reshape", "aten.permute", "aten.mm.",
        "aten.addmm", "aten._addmm", "aten.unbind", "aten.split", "aten.lift_fresh", "aten.is_", "aten.new_empty", "aten.empty_like")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "reformer-tts_amd" in fr.filename or "reformer_tts_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            big = 0
            for a in list(args) + list((kwargs or {}).values()):
                if isinstance(a, torch.Tensor):
                    big = max(big, a.numel() * a.element_size())
            self.c[(name, where, big)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
model = build_model(baseline_model_config(), dev)
tr = Trainer(model, TTSTrainingConfig(batch_size=12), dev)
batch = synthetic_batch(12, 200, 1024, device=dev)
model.train()
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
with Census() as cs:
    tr.train_step(batch)
    torch.cuda.synchronize()
for (name, where, big), n in sorted(cs.c.items(), key=lambda kv: (kv[0][1], -kv[1])):
    print(f"{n:4d}  {name:34s} {where:24s} largest operand {big / 1e6:8.3f} MB")
