#!/usr/bin/env python3
"""A/B of the launch forms of csrc/gemm_nt.hip in ONE process on one box (interleaved rounds, medians):
  * several tiles per CU: one tile per workgroup (round 3) vs persistent workgroups on the 2-deep / the deep ring;
  * the cross-attention's q | kv projections and dxn | dkeys input gradients: two launches (+ the add) vs one grouped launch;
  * the to_out input gradient + rtts_lsh_bwd_delta vs the delta epilogue.
    python scripts/gemm_mode_ab.py > gpurun_out/r04_gemm_nt_persistent_ab.log"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from reformer_tts_amd import _lib, engine
from reformer_tts_amd._graphs import capturing

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, scale=1.0):
    return torch.randn(*s, device=dev, generator=g) * scale


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
    return 1e3 * a.elapsed_time(b) / (calls * replays)        # us per call


def ab(title, variants, rounds=5, flop=None):
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for k, fn in variants.items():
            res[k].append(fn())
    line = f"{title}: " + "  ".join(f"{k} {statistics.median(v):6.1f} us (min {min(v):6.1f})" + (f" {flop / statistics.median(v) / 1e6:5.0f} TF" if flop else "")
                                    for k, v in res.items())
    print(line, flush=True)


# ---- persistent tiles
for m, n, k, kind in ((12288, 2048, 512, "bias+relu"), (12288, 1024, 512, "plain"), (12288, 2048, 512, "kn+gate"), (12288, 1024, 512, "kn"),
                      (12288, 512, 2048, "plain"), (12288, 512, 512, "plain")):
    a = rnd(m, k).bfloat16()
    w = rnd(n, k, scale=k ** -0.5).bfloat16()
    wkn = w.t().contiguous()
    bias = rnd(n)
    words = engine.gate_words(m, n, dev)
    if kind == "kn+gate":
        engine.gemm(a, w, bias=bias, relu=True, words=words)

    def call():
        if kind == "plain":
            engine.gemm(a, w)
        elif kind == "kn":
            engine.gemm(a, wkn, kn=True)
        elif kind == "bias+relu":
            engine.gemm(a, w, bias=bias, relu=True, words=words)
        else:
            engine.gemm(a, wkn, kn=True, gate=True, words=words)

    def variant(mode):
        def run():
            _lib.call("rtts_debug_set_gemm_mode", mode)
            try:
                return timed(call)
            finally:
                _lib.call("rtts_debug_set_gemm_mode", 0)
        return run
    ab(f"M={m} N={n} K={k} {kind:9s}", {"one tile/WG": variant(1), "persistent 2-deep x2": variant(2), "persistent deep x1": variant(3)}, flop=2.0 * m * n * k)

# ---- store forms of the bf16 epilogues: inside a chain (the consumer of a GEMM's output is the next launch: what a dirty L2 costs
#      shows at the kernel boundary, so the GEMM is timed together with a streaming reader of its output)
for m, n, k in ((12288, 512, 512), (12288, 1024, 512), (12288, 2048, 512), (12288, 512, 2048), (3072, 512, 512)):
    a = rnd(m, k).bfloat16()
    w = rnd(n, k, scale=k ** -0.5).bfloat16()

    def form(f):
        def run():
            _lib.call("rtts_debug_set_gemm_mode", 10 + f)
            try:
                return timed(lambda: engine.gemm(a, w))
            finally:
                _lib.call("rtts_debug_set_gemm_mode", 9)
        return run
    ab(f"store form M={m} N={n} K={k}", {"8-byte": form(0), "16-byte": form(1), "16-byte write-through": form(2)}, flop=2.0 * m * n * k)

# ---- grouped pairs
xn, keys = rnd(12288, 512).bfloat16(), rnd(3072, 512).bfloat16()
w = rnd(1536, 512, scale=512 ** -0.5).bfloat16()
bias = rnd(1536)
ab("xattn q | kv projections", {
    "two launches": lambda: timed(lambda: (engine.gemm(xn, w[:512], bias=bias[:512]), engine.gemm(keys, w[512:], bias=bias[512:]))),
    "grouped": lambda: timed(lambda: engine.gemm_group([dict(a=xn, w=w[:512], bias=bias[:512]), dict(a=keys, w=w[512:], bias=bias[512:])]))})
dq, dkv = rnd(12288, 512).bfloat16(), rnd(3072, 1024).bfloat16()
dkeys = torch.zeros(3072, 512, device=dev)
ab("xattn dxn | dkeys", {
    "two launches + add": lambda: timed(lambda: (engine.gemm(dq, w[:512], kn=True), engine.residual(dkeys, engine.gemm(dkv, w[512:], kn=True), None, 1.0))),
    "grouped, accumulating": lambda: timed(lambda: engine.gemm_group([dict(a=dq, w=w[:512]), dict(a=dkv, w=w[512:], into=dkeys)], kn=True))})
# ---- delta epilogue
for b, t in ((12, 1024), (12, 256)):
    m = b * t
    dy, wo, out = rnd(m, 512).bfloat16(), rnd(512, 512, scale=512 ** -0.5).bfloat16(), rnd(m, 512).bfloat16()
    delta = torch.empty(b * 8, t, device=dev)

    def separate():
        do = engine.gemm(dy, wo, kn=True)
        _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), 512, do.data_ptr(), 512, b, 8, t, 64, delta.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ab(f"to_out input gradient + delta, M={m}", {"dgrad + delta launch": lambda: timed(separate),
                                                 "delta epilogue": lambda: timed(lambda: engine.gemm_dgrad_delta(dy, wo, out, t, 8))})
