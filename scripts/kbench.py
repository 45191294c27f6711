#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the LSH attention path at the baseline shapes (GPU box only).
    python scripts/kbench.py [--iters 50] [--only fwd,bwd,...]
Prints average launch time (HIP events on the launch stream) and algorithmic TFLOP/s or GB/s."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd import _lib, ops  # noqa: E402


GRAPH = [False]


def timeit(fn, iters):
    """us per call.  Eager launches are host-bound below ~15 us per call (Python wrapper + ctypes); with --graph the
    calls are captured into one hipGraph (20 per graph) and the replay is timed: device time only."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if GRAPH[0]:
        from reformer_tts_amd._graphs import capturing
        g = torch.cuda.CUDAGraph()
        with capturing(g):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        reps = max(1, iters // 20)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (reps * 20) * 1e3
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--graph", action="store_true", help="time hipGraph replays (device time; eager timing is host-bound for small kernels)")
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    GRAPH[0] = args.graph
    dev = torch.device("cuda:0")
    if only and "wgrad" in only:
        # split-K weight-gradient GEMM at the shapes of one decoder layer (M = 12288 tokens; kv: 3072 text rows)
        ws = torch.empty(16 * 1024 * 1024, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for nm, (m, n, k) in dict(to_out=(12288, 512, 512), qkv=(12288, 1024, 512), ffn1=(12288, 2048, 512), ffn2=(12288, 512, 2048),
                                  xkv=(3072, 1024, 512), enc_qkv=(3072, 1024, 512), enc_ffn=(3072, 2048, 512)).items():
            a = torch.randn(m, n, device=dev).bfloat16()
            bb = torch.randn(m, k, device=dev).bfloat16()
            c = torch.zeros(n, k, device=dev)
            us = timeit(lambda: _lib.call("rtts_gemm_tn", a.data_ptr(), a.stride(0), bb.data_ptr(), bb.stride(0), m, n, k, c.data_ptr(),
                                          c.stride(0), 1, ws.data_ptr(), ws.numel(), s), args.iters)
            us_lib = timeit(lambda: torch.mm(a.t(), bb, out_dtype=torch.float32), args.iters)
            print(f"wgrad {nm:8s} M={m} N={n} K={k}: gemm_tn+reduce {us:7.1f} us = {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s   "
                  f"hipBLASLt TN fp32-out {us_lib:7.1f} us", flush=True)
        # the seven deferred weight gradients of one decoder layer in one grouped launch
        shapes = [(12288, 512, 512), (12288, 1024, 512), (12288, 512, 512), (12288, 512, 512), (3072, 1024, 512), (12288, 2048, 512),
                  (12288, 512, 2048)]
        tens = [(torch.randn(m, n, device=dev).bfloat16(), torch.randn(m, k, device=dev).bfloat16(), torch.zeros(n, k, device=dev))
                for m, n, k in shapes]
        arr = (_lib.GemmTnProblem * len(tens))()
        for q, (a, bb, c) in zip(arr, tens):
            q.a, q.lda, q.b, q.ldb, q.c, q.ldc = a.data_ptr(), a.stride(0), bb.data_ptr(), bb.stride(0), c.data_ptr(), c.stride(0)
            q.M, q.N, q.K, q.accumulate = a.shape[0], a.shape[1], bb.shape[1], 1
        us = timeit(lambda: _lib.call("rtts_gemm_tn_grouped", arr, len(tens), ws.data_ptr(), ws.numel(), s), args.iters)
        fl = sum(2.0 * m * n * k for m, n, k in shapes)
        print(f"wgrad decoder-layer group (7 problems, {fl / 1e9:.1f} GFLOP): {us:7.1f} us = {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
        return
    for name, (b, h, t, bs, nh, causal) in dict(dec=(12, 8, 1024, 128, 8, True), enc=(12, 8, 256, 64, 8, False),
                                                 long=(4, 8, 4096, 64, 8, True)).items():
        dh = 64
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(b, t, 2 * h * dh, generator=g).bfloat16().to(dev)
        qk, v = qkv[..., :h * dh], qkv[..., h * dh:]
        rot = torch.randn(1, dh, nh, t // bs // 2, generator=g).to(dev)
        mask = torch.ones(b, t, dtype=torch.uint8, device=dev)
        mask[0, t - t // 4:] = 0
        st, _, _ = ops.lsh_hash_sort(qk, rot, h, bs)
        o, lse = ops.lsh_attn_fwd(qk, v, st, h, bs, causal, mask)
        out, lse_tot = ops.lsh_combine_fwd(o, lse, b, h)
        dout = torch.randn(b, t, h * dh, generator=g).bfloat16().to(dev)
        chunks = b * h * nh * (t // bs)
        pair_flops = 2.0 * bs * 2 * bs * dh * chunks
        tok = b * h * t
        res = {}
        if not only or "hash" in only:
            us = timeit(lambda: ops.lsh_hash_sort(qk, rot, h, bs), args.iters)
            res["hash_sort"] = (us, f"{tok * (128 + nh * 4) / us / 1e3:8.1f} GB/s alg")
        if not only or "fwd" in only:
            us = timeit(lambda: ops.lsh_attn_fwd(qk, v, st, h, bs, causal, mask), args.iters)
            res["attn_fwd"] = (us, f"{2 * pair_flops / us / 1e6:8.1f} TFLOP/s")
        if not only or "combine" in only:
            us = timeit(lambda: ops.lsh_combine_fwd(o, lse, b, h), args.iters)
            res["combine"] = (us, f"{tok * (nh * 132 + 132) / us / 1e3:8.1f} GB/s alg")
        if not only or "bwd" in only:
            delta = torch.empty(b * h, t, device=dev)
            dqk_part = torch.empty(_lib.load().rtts_lsh_bwd_qk_slots(), b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
            dv_part = torch.empty(2, b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
            flags = torch.ones(b * h, nh, t, dtype=torch.uint8, device=dev)
            walks = _lib.load().rtts_lsh_attn_bwd_run_length(b, h, t, nh, bs) > 0
            ld = qkv.stride(1)

            def bwd():
                s = torch.cuda.current_stream().cuda_stream      # inside: under --graph the capture stream is current
                _lib.call("rtts_lsh_attn_bwd", qk.data_ptr(), v.data_ptr(), ld, st.data_ptr(), mask.data_ptr(), dout.data_ptr(),
                          dout.stride(1), lse_tot.data_ptr(), delta.data_ptr(), b, h, t, dh, nh, bs, int(causal),
                          dqk_part.data_ptr(), dv_part.data_ptr(), flags.data_ptr(), 0.0, 0, None, s)
            _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), out.stride(1), dout.data_ptr(), dout.stride(1), b, h, t, dh,
                      delta.data_ptr(), torch.cuda.current_stream().cuda_stream)
            us = timeit(bwd, args.iters)
            res["attn_bwd"] = (us, f"{5 * pair_flops / us / 1e6:8.1f} TFLOP/s")
            dqk, dv = torch.empty_like(qk.contiguous()), torch.empty_like(qk.contiguous())

            def red():
                s = torch.cuda.current_stream().cuda_stream
                _lib.call("rtts_lsh_bwd_reduce", dqk_part.data_ptr(), dv_part.data_ptr(), b, h, t, dh, nh, dqk.data_ptr(),
                          dv.data_ptr(), dqk.stride(1), flags.data_ptr() if walks else None, s)
            us = timeit(red, args.iters)
            res["bwd_reduce"] = (us, f"{tok * (4 * nh * 128 + 256) / us / 1e3:8.1f} GB/s alg")
        for k, (us, extra) in res.items():
            print(f"{name:5s} {k:14s} {us:9.1f} us  {extra}", flush=True)


if __name__ == "__main__":
    main()
