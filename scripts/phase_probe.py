#!/usr/bin/env python3
"""Where the cycles of one lsh_attn_bwd workgroup go (GPU box only).

Builds a PRIVATE copy of csrc/lsh_attn_bwd.hip with -DAB_PHASE_TIMING (wave 0 of every workgroup stamps the shader
clock at the phase boundaries) into reformer-tts_amd/lib/librtts_probe.so, runs it once at the decoder shape and
prints the median / mean cycles per phase.  The product library is not touched.
    python scripts/phase_probe.py --build     # here (hipcc cross-compiles)
    python scripts/phase_probe.py             # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "reformer-tts_amd", "csrc")
UNROLL = next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--unroll"), "1")
PROBE = os.path.join(ROOT, "reformer-tts_amd", "lib", f"librtts_probe_u{UNROLL}.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=on",
           "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-DAB_PHASE_TIMING", f"-DAB_UNROLL={UNROLL}", "-shared", "-x", "hip", os.path.join(CSRC, "lsh_attn_bwd.hip"), os.path.join(CSRC, "rtts_api.cpp"),
           "-o", PROBE]
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_fwd():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=on",
           "-fno-slp-vectorize", "-DAF_PHASE_TIMING", "-shared", "-x", "hip", os.path.join(CSRC, "lsh_attn_fwd.hip"),
           os.path.join(CSRC, "rtts_api.cpp"), "-o", PROBE.replace("probe_u", "probe_fwd_u")]
    print(" ".join(cmd[-6:]), flush=True)
    subprocess.check_call(cmd)


def main_fwd():
    import numpy as np
    import torch
    from reformer_tts_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = C.CDLL(PROBE.replace("probe_u", "probe_fwd_u"))
    lib.rtts_lsh_attn_fwd.argtypes = _lib.SIGNATURES["rtts_lsh_attn_fwd"]
    lib.rtts_debug_af_phases.argtypes = [C.c_void_p]
    b, h, t, bs, nh, causal, dh = 12, 8, 1024, 128, 8, True, 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(b, t, 2 * h * dh, generator=g).bfloat16().to(dev)
    qk, v = qkv[..., :h * dh], qkv[..., h * dh:]
    rot = torch.randn(1, dh, nh, t // bs // 2, generator=g).to(dev)
    mask = torch.ones(b, t, dtype=torch.uint8, device=dev)
    mask[0, t - t // 4:] = 0
    st, _, _ = ops.lsh_hash_sort(qk, rot, h, bs)
    o = torch.empty(b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b * h, nh, t, dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        rc = lib.rtts_lsh_attn_fwd(qk.data_ptr(), v.data_ptr(), qkv.stride(1), st.data_ptr(), mask.data_ptr(), b, h, t, dh, nh, bs,
                                   int(causal), o.data_ptr(), lse.data_ptr(), 0.0, 0, None, s)
        assert rc == 0
    torch.cuda.synchronize()
    buf = np.zeros(16 * 8192, dtype=np.uint64)
    assert lib.rtts_debug_af_phases(buf.ctypes.data) == 0
    allp = buf.reshape(8192, 16)[: b * h * nh * (t // bs)].astype(np.int64)
    run = lib.rtts_lsh_attn_fwd_run_length(b, h, t, nh, bs)
    if run > 0:
        # walking form: stamps of step AF_WSTEP (default 1) of every run, wave 0 (own keys) and wave NQT (looked-back keys)
        allw = buf.reshape(8192, 16)[: b * h * nh * (t // bs) // run].astype(np.int64)
        own = ["Q fragments + 4 key tiles", "wait barrier 1 (tiles of all waves)", "(nothing)", "wait barrier 2 (partials written)",
               "merge + stage + row stores issued", "wait barrier 3", "wait barrier 4 (next chunk placed)"]
        back = ["Q fragments + 4 key tiles", "wait barrier 1 (tiles of all waves)", "partials -> LDS, next rows requested",
                "wait barrier 2", "wait barrier 3 (own waves merged + stored)", "next rows -> LDS (waits for the loads)", "wait barrier 4"]
        for half, off, names in (("own waves (wave 0)", 0, own), ("back waves (wave NQT)", 8, back)):
            ph = allw[:, off:off + 8]
            ph = ph[ph[:, 7] > 0]
            d = np.diff(ph, axis=1)
            print(f"{half}: run length {run}, workgroups {len(ph)}")
            for i, nm in enumerate(names):
                print(f"  {nm:44s} median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
            print(f"  one step                                     median {np.median(ph[:, 7] - ph[:, 0]):8.0f}")
        return
    names = ["gather (positions -> rows -> LDS image)", "wait barrier 1", "Q fragments + 4 key tiles (online softmax)",
             "wait merge barrier", "merge + stage + row stores issued"]
    for half, off in (("own keys (wave 0)", 0), ("looked-back keys (wave NQT)", 8)):
        ph = allp[:, off:off + 6]
        ph = ph[ph[:, 3] > 0]
        d = np.diff(ph, axis=1)
        print(f"{half}: workgroups {len(ph)}")
        for i, nm in enumerate(names):
            if off and i == 4:
                break           # the second half returns after handing its partial results over
            print(f"  {nm:44s} median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
        if not off:
            print(f"  total                                        median {np.median(ph[:, 5] - ph[:, 0]):8.0f}")


def main():
    if "--fwd" in sys.argv:
        if "--build" in sys.argv:
            build_fwd()
        else:
            main_fwd()
        return
    if "--build" in sys.argv:
        build()
        return
    import numpy as np
    import torch
    from reformer_tts_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = C.CDLL(PROBE)
    lib.rtts_lsh_attn_bwd.argtypes = _lib.SIGNATURES["rtts_lsh_attn_bwd"]
    lib.rtts_debug_ab_phases.argtypes = [C.c_void_p]
    b, h, t, bs, nh, causal, dh = 12, 8, 1024, 128, 8, True, 64
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
    delta = torch.empty(b * h, t, device=dev)
    dqk_part = torch.empty(_lib.load().rtts_lsh_bwd_qk_slots(), b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
    dv_part = torch.empty(2, b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
    flags = torch.empty(b * h, nh, t, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), out.stride(1), dout.data_ptr(), dout.stride(1), b, h, t, dh, delta.data_ptr(), s)

    def run():
        rc = lib.rtts_lsh_attn_bwd(qk.data_ptr(), v.data_ptr(), qkv.stride(1), st.data_ptr(), mask.data_ptr(), dout.data_ptr(),
                                   dout.stride(1), lse_tot.data_ptr(), delta.data_ptr(), b, h, t, dh, nh, bs, int(causal),
                                   dqk_part.data_ptr(), dv_part.data_ptr(), flags.data_ptr(), 0.0, 0, None, s)
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    e.record()
    torch.cuda.synchronize()
    print(f"probe kernel: {a.elapsed_time(e) / 10 * 1e3:.1f} us per launch", flush=True)
    buf = np.zeros(32 * 8192, dtype=np.uint64)
    rc = lib.rtts_debug_ab_phases(buf.ctypes.data)
    assert rc == 0, rc
    allp = buf.reshape(8192, 32).astype(np.int64)
    allp = allp[allp[:, 0] > 0]
    walk = os.environ.get("RTTS_LSH_BWD_WALK", "")
    if walk not in ("", "0"):
        # walking kernel: the stamps are those of step 2 of every run (a step whose operands were prefetched)
        names = ["wait barrier (operands of this step on chip)", "issue prefetch of the next chunk", "key tile: K fragments, V rows arrive",
                 "main loop", "wait barrier (dS)", "dQ product + park", "wait barrier", "row stores issued", "stores acknowledged"]
        ph = allp[:, :10]
        d = np.diff(ph, axis=1)
        tot = ph[:, 9] - ph[:, 0]
        print(f"runs {len(ph)}; one prefetched step: median {np.median(tot):.0f} mean {tot.mean():.0f} cycles")
        for i, nm in enumerate(names):
            print(f"  {nm:46s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
        return
    allp = allp[: b * h * nh * (t // bs)]
    ph = allp[:, :10]
    d = np.diff(ph, axis=1)
    names = ["gather + LDS image", "wait barrier 1", "key consts + main loop", "dV rows staged + stores issued", "wait barrier 2",
             "dQ phase + park", "wait barrier 3", "dK epilogue + stores issued", "stores acknowledged"]
    tot = ph[:, 9] - ph[:, 0]
    print(f"workgroups {len(ph)}; total per WG: median {np.median(tot):.0f} mean {tot.mean():.0f} cycles (s_memtime ticks)")
    for i, nm in enumerate(names):
        print(f"  {nm:34s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
    print(f"  inside the gather (wave 0): positions arrive {np.median(allp[:, 10] - ph[:, 0]):.0f} | rows arrive +{np.median(allp[:, 11] - allp[:, 10]):.0f}"
          f" | LDS image written +{np.median(ph[:, 1] - allp[:, 11]):.0f}")
    nw = bs * 4 // 64
    print("  per wave, cycles after barrier 1:  main loop done | dV rows out (arrival at barrier 2)")
    for w in range(nw):
        print(f"    wave {w}: {np.median(allp[:, 16 + w] - ph[:, 2]):8.0f} | {np.median(allp[:, 24 + w] - ph[:, 2]):8.0f}")
    span = ph[:, 9].max() - ph[:, 0].min()
    print(f"  first start -> last end: {span} ticks; sum of WG totals / 256 CUs: {tot.sum() / 256:.0f}")


if __name__ == "__main__":
    main()
