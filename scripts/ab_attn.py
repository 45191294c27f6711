#!/usr/bin/env python3
"""A/B timing of lsh_attn_bwd (or, with --fwd, lsh_attn_fwd) build variants on ONE box (box-to-box spread is larger than most kernel changes).

    python scripts/ab_attn.py --build "-DAB_UNROLL=1" "-DAB_UNROLL=2" "-DAB_UNROLL=4"    # here: one private .so per flag set
    python scripts/ab_attn.py "-DAB_UNROLL=1" "-DAB_UNROLL=2" "-DAB_UNROLL=4"            # GPU box: interleaved timing
Each variant is csrc/lsh_attn_bwd.hip + rtts_api.cpp compiled with the product flags plus the given ones; the product
library is not touched."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "reformer-tts_amd", "csrc")


KERNEL = "fwd" if "--fwd" in sys.argv else "bwd"


def lib_path(flags):
    return os.path.join(ROOT, "reformer-tts_amd", "lib", "librtts_ab_" + KERNEL + "_" + hashlib.md5(flags.encode()).hexdigest()[:8] + ".so")


def build(flags):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=on",
           "-fno-slp-vectorize", *(["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if KERNEL == "bwd" and "sched-strategy" not in flags else []), *flags.split(), "-shared", "-x", "hip", os.path.join(CSRC, f"lsh_attn_{KERNEL}.hip"),
           os.path.join(CSRC, "rtts_api.cpp"), "-o", lib_path(flags)]
    print(" ".join(cmd[-8:]), flush=True)
    subprocess.check_call(cmd)


def main():
    args = [a for a in sys.argv[1:] if a not in ("--build", "--fwd")]
    if "--build" in sys.argv:
        for f in args:
            build(f)
        return
    import torch
    from reformer_tts_amd import _lib, ops
    dev = torch.device("cuda:0")
    libs = []
    for f in args:
        lib = C.CDLL(lib_path(f))
        getattr(lib, f"rtts_lsh_attn_{KERNEL}").argtypes = _lib.SIGNATURES[f"rtts_lsh_attn_{KERNEL}"]
        libs.append(lib)
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
        delta = torch.empty(b * h, t, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), out.stride(1), dout.data_ptr(), dout.stride(1), b, h, t, dh, delta.data_ptr(), s)
        outs = []
        tot = [0.0] * len(libs)
        reps = 5
        for rep in range(reps + 1):
            for i, lib in enumerate(libs):
                dqk = torch.zeros(2, b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
                dv = torch.zeros(2, b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)
                flags = torch.zeros(b * h, nh, t, dtype=torch.uint8, device=dev)     # row_flags of the walking form
                if KERNEL == "fwd":
                    dqk = torch.zeros(b * h, nh, t, dh, dtype=torch.bfloat16, device=dev)   # o
                    dv = torch.zeros(b * h, nh, t, dtype=torch.float32, device=dev)        # lse

                def run():
                    if KERNEL == "fwd":
                        rc = lib.rtts_lsh_attn_fwd(qk.data_ptr(), v.data_ptr(), qkv.stride(1), st.data_ptr(), mask.data_ptr(), b, h, t, dh,
                                                   nh, bs, int(causal), dqk.data_ptr(), dv.data_ptr(), 0.0, 0, None, s)
                        assert rc == 0
                        return
                    rc = lib.rtts_lsh_attn_bwd(qk.data_ptr(), v.data_ptr(), qkv.stride(1), st.data_ptr(), mask.data_ptr(),
                                               dout.data_ptr(), dout.stride(1), lse_tot.data_ptr(), delta.data_ptr(), b, h, t, dh, nh,
                                               bs, int(causal), dqk.data_ptr(), dv.data_ptr(), flags.data_ptr(), 0.0, 0, None, s)
                    assert rc == 0
                run()
                torch.cuda.synchronize()
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    run()
                e.record()
                torch.cuda.synchronize()
                if rep:
                    tot[i] += a.elapsed_time(e) / 20 * 1e3
                elif len(outs) < len(libs):
                    outs.append((dqk.float().clone(), dv.float().clone()))
        for i, f in enumerate(args):
            diff = max(float((outs[i][k] - outs[0][k]).abs().max()) for k in range(2))
            scale = max(float(outs[0][k].abs().max()) for k in range(2))
            print(f"{name:5s} {f:40s} {tot[i] / reps:8.1f} us   max |diff to variant 0| = {diff:.3e} (scale {scale:.2e})", flush=True)


if __name__ == "__main__":
    main()
