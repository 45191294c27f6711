#!/usr/bin/env python3
"""Cross-attention kernels at the decoder shape of config/baseline.yml (GPU box only): us per launch, HIP events.
RTTS_LIB=<alternative build> python scripts/kbench_xattn.py   # A/B against another build on the same box"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd import _lib  # noqa: E402
from reformer_tts_amd._seeds import seed_base  # noqa: E402

dev = torch.device("cuda:0")
b, h, t, tk, e = 12, 8, 1024, 256, 512
g = torch.Generator().manual_seed(0)
q = torch.randn(b * t, e, generator=g).bfloat16().to(dev)
kv = torch.randn(b * tk, 2 * e, generator=g).bfloat16().to(dev)
do = torch.randn(b * t, e, generator=g).bfloat16().to(dev)
kvalid = torch.ones(b, tk, dtype=torch.uint8, device=dev)
kvalid[:, 200:] = 0
o = torch.empty(b * t, e, dtype=torch.bfloat16, device=dev)
lse = torch.empty(b * h, t, dtype=torch.float32, device=dev)
delta = torch.empty(b * h, t, dtype=torch.float32, device=dev)
dq = torch.empty(b * t, e, dtype=torch.bfloat16, device=dev)
part = torch.empty(t // 128, b * tk, 2 * e, dtype=torch.bfloat16, device=dev)
nkc = _lib.load().rtts_xattn_key_chunks(tk)
dq_chunks = torch.empty(nkc, b * t, e, dtype=torch.bfloat16, device=dev) if nkc > 1 else None
s = torch.cuda.current_stream().cuda_stream
sb = seed_base(dev)


def fwd():
    _lib.call("rtts_xattn_fwd", q.data_ptr(), e, kv.data_ptr(), 2 * e, kvalid.data_ptr(), b, h, t, tk, e // h, o.data_ptr(), e,
              lse.data_ptr(), 0.0, 0, sb.data_ptr(), s)


def bwd():
    _lib.call("rtts_xattn_bwd", q.data_ptr(), e, kv.data_ptr(), 2 * e, kvalid.data_ptr(), do.data_ptr(), e, lse.data_ptr(),
              delta.data_ptr(), b, h, t, tk, e // h, dq.data_ptr(), e, part.data_ptr(), 0.0, 0, sb.data_ptr(),
              None if dq_chunks is None else dq_chunks.data_ptr(), s)


fwd()
_lib.call("rtts_lsh_bwd_delta", o.data_ptr(), e, do.data_ptr(), e, b, h, t, e // h, delta.data_ptr(), s)
for name, fn in (("xattn_fwd", fwd), ("xattn_bwd", bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    z.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(z) / 50 * 1e3:7.1f} us   checksum {float(dq.float().abs().sum()) + float(o.float().abs().sum()):.6e}", flush=True)
