#!/usr/bin/env python3
"""What the library GEMMs of the step cost at their shapes, in the layouts the executor could feed them (GPU box only)."""
import torch

dev = torch.device("cuda:0")


def t(fn, n=50):
    for _ in range(5):
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


for m, n, k in [(12288, 512, 512), (12288, 1024, 512), (12288, 2048, 512), (12288, 512, 2048), (12288, 512, 1024), (3072, 1024, 512), (3072, 2048, 512),
                (12288, 512, 2560), (12288, 2560, 512), (12288, 512, 640), (12288, 128, 2560), (12288, 128, 512), (12288, 512, 128),
                (3072, 512, 2560), (3072, 2560, 512), (3072, 512, 512), (3072, 512, 2048)][:int(__import__('os').environ.get('GEMM_PROBE_SHAPES', '99'))]:
    x = torch.randn(m, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev).bfloat16()       # Linear weight (out, in)
    wt = w.t().contiguous()                            # (in, out)
    bias = torch.randn(n, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    r = {}
    r["mm(x, w.t())"] = t(lambda: torch.mm(x, w.t()))
    r["mm(x, wt)"] = t(lambda: torch.mm(x, wt))
    r["mm out="] = t(lambda: torch.mm(x, w.t(), out=out))
    r["addmm"] = t(lambda: torch.addmm(bias, x, w.t()))
    r["addmm+relu NT"] = t(lambda: torch._addmm_activation(bias, x, w.t(), use_gelu=False))
    r["addmm+relu NN"] = t(lambda: torch._addmm_activation(bias, x, wt, use_gelu=False))
    r["transpose"] = t(lambda: w.t().contiguous())
    fl = 2.0 * m * n * k
    print(f"M={m} N={n} K={k}: " + "  ".join(f"{kk} {v:6.1f} us ({fl / v / 1e6:5.0f} TF/s)" for kk, v in r.items()), flush=True)
