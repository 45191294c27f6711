#!/usr/bin/env python3
"""How does a replayed hipGraph run two parallel branches -- one of many small kernels, one of few large ones?  (GPU box only.)
Times the replay of: branch S alone, branch L alone, both in one graph with S created first, both with L created first."""
import os

import torch

dev = torch.device("cuda:0")
small = torch.zeros(4096, device=dev)
big = torch.zeros(256 * 1024 * 1024 // 4, device=dev)          # 256 MB: one pass ~ 120 us
side = torch.cuda.Stream()


MODE = os.environ.get("PROBE_S", "tiny")
chain_in = torch.randn(1 << 16, device=dev)
chain_out = torch.empty_like(chain_in)
chain_idx = torch.empty(1 << 16, dtype=torch.long, device=dev)


def S(n=200):
    if MODE == "tiny":
        for _ in range(n):
            small.add_(1.0)
    else:                                      # few workgroups, long: latency-bound kernels that cannot fill the chip
        for _ in range(12):
            torch.sort(chain_in, out=(chain_out, chain_idx))


ma = torch.randn(4096, 4096, device=dev).bfloat16()
mb = torch.randn(4096, 4096, device=dev).bfloat16()
mc = torch.empty(4096, 4096, device=dev, dtype=torch.bfloat16)
MODE_L = os.environ.get("PROBE_L", "stream")


def L(n=8):
    if MODE_L == "stream":
        for _ in range(n):
            big.add_(1.0)
    else:                                      # 256 workgroups of a library GEMM: one per CU, registers and LDS to spare
        for _ in range(n):
            torch.mm(ma, mb, out=mc)


def capture(order):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        small.add_(0.0)                       # common root
        if order in ("S", "L"):
            (S if order == "S" else L)()
        elif order == "I":                    # both branches, their launches INTERLEAVED in program (= node creation) order
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            for i in range(24):
                with torch.cuda.stream(side):
                    if MODE == "tiny":
                        for _ in range(8):
                            small.add_(1.0)
                    elif i % 2 == 0:
                        torch.sort(chain_in, out=(chain_out, chain_idx))
                if i % 3 == 0:
                    L(1)
            main.wait_stream(side)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            first, second = (S, L) if order == "SL" else (L, S)
            if order == "SL":
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    first()
                second()
            else:
                first()                       # L on the capture stream first, then S on the side stream
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    second()
            main.wait_stream(side)
        small.add_(0.0)                       # join
    return g


def timeit(g, reps=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def eager_two_streams(reps=20):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            S()
        L()
        main.wait_stream(side)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


import os
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), " GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
eager_two_streams(2)
print(f"eager, S on a side stream beside L: {eager_two_streams():8.1f} us (host-bound for S: 200 launches)", flush=True)
for order in ("S", "L", "SL", "LS", "I"):
    S(3); L(1)
    torch.cuda.synchronize()
    g = capture(order)
    print(f"{order:3s}: {timeit(g):8.1f} us per replay", flush=True)


if os.environ.get("PROBE_SHORT"):
    raise SystemExit(0)
# two single-branch graphs launched on two streams
gS, gL = capture("S"), capture("L")
pstream = torch.cuda.Stream()


def two_graphs(reps=20):
    main = torch.cuda.current_stream()
    for _ in range(3):
        gS.replay(); gL.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        pstream.wait_stream(main)
        with torch.cuda.stream(pstream):
            gS.replay()
        gL.replay()
        main.wait_stream(pstream)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f"graph S on a second stream beside graph L on the first: {two_graphs():8.1f} us per pair", flush=True)


# does the choice of the second stream matter (streams share a few hardware queues)?
streams = [torch.cuda.Stream() for _ in range(6)]
mainS = torch.cuda.Stream()
for i, st in enumerate(streams):
    pstream = st
    with torch.cuda.stream(mainS):
        t = two_graphs(10)
    print(f"graph L on a created stream, graph S on created stream #{i}: {t:8.1f} us per pair", flush=True)
