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


def drop_hash(seed: int, idx):
    """numpy twin of the kernels' counter hash (csrc/rtts_common.h ``rtts_drop_hash``): uint32 arithmetic, element-wise over idx."""
    import numpy as np
    m = np.uint64(0xFFFFFFFF)

    def u32(x):
        return np.asarray(x, dtype=np.uint64) & m

    s = u32(seed)
    s ^= s >> np.uint64(16); s = u32(s * np.uint64(0x85ebca6b)); s ^= s >> np.uint64(13); s = u32(s * np.uint64(0xc2b2ae35)); s ^= s >> np.uint64(16)
    x = u32(u32(idx) * np.uint64(0x9E3779B1) + s)
    x ^= x >> np.uint64(16); x = u32(x * np.uint64(0x7feb352d))
    x ^= u32((s << np.uint64(13)) | (s >> np.uint64(19)))
    x ^= x >> np.uint64(15); x = u32(x * np.uint64(0x846ca68b)); x ^= x >> np.uint64(16)
    return x.astype(np.uint64)


def attention_keep_scales(seed: int, p: float, heads: int, chunks: int, bucket_size: int):
    """(heads, chunks, bucket_size, 2 * bucket_size) float32 keep-scales of the LSH attention's probability dropout as the
    kernels draw them: pair = ((head * chunks + chunk) * bucket_size + query row) * 2 * bucket_size + key row, kept iff
    hash(seed, pair) >= p * 2^32."""
    import numpy as np
    import torch
    n = heads * chunks * bucket_size * 2 * bucket_size
    assert n < 2 ** 32
    h = drop_hash(seed, np.arange(n, dtype=np.uint64))
    thresh = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    keep = (h >= thresh).astype(np.float32) / np.float32(1.0 - np.float32(p))
    return torch.from_numpy(keep.reshape(heads, chunks, bucket_size, 2 * bucket_size))
