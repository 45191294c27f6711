"""Deterministic synthetic parameters (TEST INFRASTRUCTURE, see ``oracle/__init__.py``).

Weights are drawn from numpy's legacy ``RandomState`` (bit-stable across numpy
versions) keyed by the parameter *name*, so the golden generator (which runs
next to the reference, in the build container) and the tests (which run
anywhere) agree on every parameter without shipping a checkpoint.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    shape = tuple(int(s) for s in shape)
    n = rs.standard_normal(shape).astype(np.float32)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("inv_freq"):
        d = 2 * shape[0]
        return 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    if leaf == "alpha":
        return torch.from_numpy(0.5 + 0.25 * n)
    if "norm." in name or ".bn" in name:          # LayerNorm / BatchNorm affine
        return torch.from_numpy((1.0 if leaf == "weight" else 0.0) + 0.1 * n)
    if leaf in ("bias", "in_proj_bias"):
        return torch.from_numpy(0.1 * n)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    return torch.from_numpy(n * np.float32(1.0 / np.sqrt(fan_in)))


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """``shapes``: name -> shape for every *parameter* and the ``inv_freq`` buffers
    (BatchNorm running statistics are not part of the training forward and are skipped)."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()
            if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))}
