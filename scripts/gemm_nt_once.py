#!/usr/bin/env python3
"""A few rtts_gemm_nt launches per shape, for rocprofv3 runs (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_nt_probe import gemm_nt  # noqa: E402

dev = torch.device("cuda:0")
for m, n, k in [(12288, 512, 512), (12288, 2048, 512), (12288, 512, 2048), (3072, 512, 512)]:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev).bfloat16()
    wkn = w.t().contiguous()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        gemm_nt(x, w, out=out)
        gemm_nt(x, wkn, kn=True, out=out)
torch.cuda.synchronize()
