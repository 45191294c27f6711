#!/usr/bin/env python3
"""Where a rtts_gemm_nt launch spends its time (GPU box; needs the diagnostic build: scripts/build_ab.sh stamps -DGN_STAMPS
-DRTTS_GEMM_NT_AB, RTTS_LIB=.../librtts_stamps.so).  Wave 0 of every workgroup stamps the shader clock and the chip-wide
100 MHz real-time clock at: kernel entry | first stage landed | main loop done | stores issued | stores retired."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reformer_tts_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def run(m, n, k, kn=False, reps=6):
    x = torch.randn(m, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev).bfloat16()
    if kn:
        w = w.t().contiguous()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    stamps = torch.zeros(4096, 12, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    recs = []
    for _ in range(reps):            # back-to-back launches; each overwrites the stamps, the last one is read
        _lib.call("rtts_gemm_nt", x.data_ptr(), k, w.data_ptr(), w.stride(0), int(kn), m, n, k, out.data_ptr(), n, None, 0, None, 0,
                  stamps.data_ptr(), s)
    torch.cuda.synchronize()
    st = stamps.cpu()
    st = st[st[:, 0] != 0]
    c = st[:, 0:5].double()
    r = st[:, 5:10].double()
    t0 = r[:, 0].min()
    names = ["entry->stage0", "main loop", "epilogue issue", "store drain"]
    clk = ((c[:, 4] - c[:, 0]) / ((r[:, 4] - r[:, 0]) / 100.0)).median().item()      # cycles per us
    print(f"M={m} N={n} K={k} kn={int(kn)}: {st.shape[0]} workgroups, shader clock {clk / 1e3:.2f} GHz", flush=True)
    print(f"  kernel span (first entry -> last retire): {(r[:, 4].max() - t0) / 100.0:.2f} us;  entry skew {(r[:, 0].max() - t0) / 100.0:.2f} us;"
          f"  workgroup lifetime median {((r[:, 4] - r[:, 0]) / 100.0).median():.2f} max {((r[:, 4] - r[:, 0]) / 100.0).max():.2f} us", flush=True)
    for i, nm in enumerate(names):
        d = (c[:, i + 1] - c[:, i])
        print(f"  {nm:16s} cycles median {d.median():8.0f}  p10 {d.quantile(0.1):8.0f}  p90 {d.quantile(0.9):8.0f}  max {d.max():8.0f}"
              f"   ({d.median() / clk:.2f} us)", flush=True)
    # per XCD: when did its workgroups retire
    for x_ in range(8):
        sel = st[:, 10] == x_
        if sel.any():
            print(f"    xcd {x_}: {int(sel.sum())} wgs, retire median {((r[sel, 4] - t0) / 100.0).median():.2f} max {((r[sel, 4] - t0) / 100.0).max():.2f} us")


for shape in [(12288, 512, 512), (12288, 512, 2048), (12288, 2048, 512), (3072, 512, 512)]:
    run(*shape)
run(12288, 512, 512, kn=True)
