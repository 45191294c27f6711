"""Dropout seeds of the explicit executors: a host counter gives every dropout site of a step its own constant, and a
device word (rewritten by the trainer before each step) is added inside the kernels so that a replayed hipGraph --
whose per-call constants are frozen -- still draws fresh masks every step."""
from __future__ import annotations

import torch

_state = [0]
_SEED_BASE = {}


def seed_base(device) -> torch.Tensor:
    if device not in _SEED_BASE:
        _SEED_BASE[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return _SEED_BASE[device]


def next_seed() -> int:
    _state[0] += 1
    return _state[0] * 2654435761 % (1 << 32)


def reset(counter: int = 0) -> None:
    """Restart the host counter (tests: two runs that must draw the same masks)."""
    _state[0] = counter


class _Counter:
    """itertools.count-like view of the shared counter (edges.py draws from the same sequence)."""

    def __next__(self):
        _state[0] += 1
        return _state[0]


_seed_counter = _Counter()
