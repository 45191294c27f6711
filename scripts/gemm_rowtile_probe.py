#!/usr/bin/env python3
"""What would a tile that spans the whole row of an N = 512 product cost?  (GPU box; RTTS_LIB = a build with -DGN_EXPERIMENT_ROWTILE.)
A LayerNorm in the epilogue of the out-projections (to_out, out_proj, FeedForward net.3) needs every column of a row in ONE workgroup:
a 64 x 512 tile, one wave per 64 columns, 192 workgroups at M = 12288.  This times that tile with the plain epilogue against the
library's 192 x 128 pick and checks that the results agree."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from reformer_tts_amd import _lib, engine
from reformer_tts_amd._graphs import capturing

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, calls=20, replays=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with capturing(gr):
        for _ in range(calls):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (calls * replays)


for m, n, k in ((12288, 512, 512), (12288, 512, 2048), (3072, 512, 512), (3072, 512, 2048)):
    a = (torch.randn(m, k, device=dev, generator=g)).bfloat16()
    w = (torch.randn(n, k, device=dev, generator=g) * k ** -0.5).bfloat16()
    outs, res = {}, {0: [], 1: []}
    for rnd in range(5):
        for mode in (0, 1):
            _lib.call("rtts_debug_set_gemm_mode", mode)
            try:
                if rnd == 0:
                    outs[mode] = engine.gemm(a, w).float()
                res[mode].append(timed(lambda: engine.gemm(a, w)))
            finally:
                _lib.call("rtts_debug_set_gemm_mode", 0)
    err = float((outs[0] - outs[1]).abs().max())
    flop = 2.0 * m * n * k
    print(f"M={m} N={n} K={k}: library pick {statistics.median(res[0]):6.1f} us ({flop / statistics.median(res[0]) / 1e6:4.0f} TF)   "
          f"64x512 row tile {statistics.median(res[1]):6.1f} us ({flop / statistics.median(res[1]) / 1e6:4.0f} TF)   max |diff| {err:.3e}", flush=True)
