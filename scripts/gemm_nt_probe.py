#!/usr/bin/env python3
"""rtts_gemm_nt against the library GEMM (hipBLASLt through torch.mm) at every shape of the training step (GPU box only):
correctness of each layout / epilogue against an fp32 product of the same bf16 operands, then time per call inside a
replayed hipGraph (20 back-to-back calls per graph, random operands)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reformer_tts_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def _s():
    return torch.cuda.current_stream().cuda_stream


def gemm_nt(a, w, kn=False, bias=None, epi=0, gate=None, want_colsum=False, out=None):
    m, k = a.shape
    n = w.shape[1] if kn else w.shape[0]
    c = out if out is not None else torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    cs = None
    if want_colsum:
        rows = _lib.load().rtts_gemm_nt_partial_rows(m, n)
        cs = torch.empty(rows, n, dtype=torch.float32, device=a.device)
    _lib.call("rtts_gemm_nt", a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), int(kn), m, n, k, c.data_ptr(), c.stride(0),
              None if bias is None else bias.data_ptr(), epi, None if gate is None else gate.data_ptr(),
              0 if gate is None else gate.stride(0), None if cs is None else cs.data_ptr(), _s())
    return (c, cs) if want_colsum else c


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3


def check(m, n, k):
    g = torch.Generator(device=dev).manual_seed(m + n + k)
    a = torch.randn(m, k, device=dev, generator=g).bfloat16()
    w = (torch.randn(n, k, device=dev, generator=g) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev, generator=g)
    ref = a.float() @ w.float().t()
    scale = ref.abs().max().item()
    res = {}
    res["nt"] = (gemm_nt(a, w).float() - ref).abs().max().item() / scale
    res["nt+bias"] = (gemm_nt(a, w, bias=bias, epi=1).float() - (ref + bias)).abs().max().item() / scale
    res["nt+bias+relu"] = (gemm_nt(a, w, bias=bias, epi=2).float() - torch.relu(ref + bias)).abs().max().item() / scale
    wkn = w.t().contiguous()                        # [K][N]
    res["kn"] = (gemm_nt(a, wkn, kn=True).float() - ref).abs().max().item() / scale
    gate = torch.randn(m, n, device=dev, generator=g).bfloat16()
    c, cs = gemm_nt(a, wkn, kn=True, epi=3, gate=gate, want_colsum=True)
    gref = ref * (gate.float() > 0)
    res["kn+gate"] = (c.float() - gref).abs().max().item() / scale
    csum = cs.sum(0)
    res["colsum"] = ((csum - gref.sum(0)).abs().max() / gref.sum(0).abs().max()).item()
    ok = all(v < 1e-2 for v in res.values())
    print(f"check M={m} N={n} K={k}: " + "  ".join(f"{kk} {v:.2e}" for kk, v in res.items()) + ("  OK" if ok else "  FAIL"), flush=True)
    return ok


def main():
    shapes = [(12288, 512, 512), (12288, 1024, 512), (12288, 2048, 512), (12288, 512, 2048), (12288, 512, 1024),
              (3072, 512, 512), (3072, 1024, 512), (3072, 2048, 512), (3072, 512, 2048), (3072, 512, 1024),
              (16384, 512, 512), (16384, 2048, 512), (16384, 512, 2048)]
    ok = True
    for m, n, k in [(192, 128, 64), (384, 256, 192), (256, 128, 128), (96, 64, 64), (128, 64, 256), (512, 256, 128), (256, 512, 64), (49152, 256, 192)] + shapes[:10]:
        ok &= check(m, n, k)
    if not ok:
        print("CORRECTNESS FAILED", flush=True)
        sys.exit(1)
    quick = os.environ.get("GEMM_PROBE_QUICK") == "1"       # only the hand-written kernel's plain forms (A/B of build variants)
    for m, n, k in shapes:
        if quick:
            x = torch.randn(m, k, device=dev).bfloat16()
            w = torch.randn(n, k, device=dev).bfloat16()
            wkn = w.t().contiguous()
            out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
            r = {"rtts nt": t(lambda: gemm_nt(x, w, out=out)), "rtts kn": t(lambda: gemm_nt(x, wkn, kn=True, out=out))}
            fl = 2.0 * m * n * k
            print(f"M={m} N={n} K={k}: " + "  ".join(f"{kk} {v:6.1f} us ({fl / v / 1e6:5.0f} TF)" for kk, v in r.items()), flush=True)
            continue
        x = torch.randn(m, k, device=dev).bfloat16()
        w = torch.randn(n, k, device=dev).bfloat16()
        wkn = w.t().contiguous()
        bias = torch.randn(n, device=dev)
        biasb = bias.bfloat16()
        gate = torch.randn(m, n, device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        r = {}
        r["lib NT"] = t(lambda: torch.mm(x, w.t(), out=out))
        r["lib NN"] = t(lambda: torch.mm(x, wkn, out=out))
        r["rtts nt"] = t(lambda: gemm_nt(x, w, out=out))
        r["rtts kn"] = t(lambda: gemm_nt(x, wkn, kn=True, out=out))
        r["lib bias+relu"] = t(lambda: torch._addmm_activation(biasb, x, w.t(), use_gelu=False))
        r["rtts bias+relu"] = t(lambda: gemm_nt(x, w, bias=bias, epi=2, out=out))
        r["rtts kn+gate"] = t(lambda: gemm_nt(x, wkn, kn=True, epi=3, gate=gate, out=out))
        fl = 2.0 * m * n * k
        print(f"M={m} N={n} K={k}: " + "  ".join(f"{kk} {v:6.1f} us ({fl / v / 1e6:5.0f} TF)" for kk, v in r.items()), flush=True)


if __name__ == "__main__":
    main()
