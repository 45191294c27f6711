"""Build librtts_hip.so (gfx950) in-tree with hipcc, and the oracle's C twin with gcc.

hipcc cross-compiles without a GPU.  The shared object lands in
``reformer-tts_amd/lib/`` (git-ignored, but shipped to the GPU box by gpurun)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librtts_hip.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_lsh.so")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_hip(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "rtts.h")]
    if not force and _newer(LIB, deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=on",
             # no SLP packing of adjacent f32 ops into v_pk_*_f32: beside MFMAs a packed op costs several times two plain ones
             "-fno-slp-vectorize"]
    # per-file extras, measured A/B on one box (scripts/ab_attn.py): the ILP-first machine scheduler buys the LSH backward
    # 2-7 % (its tile loop alternates MFMA bursts and softmax arithmetic of two waves per SIMD); neutral on the forward
    extra = {"lsh_attn_bwd.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + [d for d in deps if d.endswith(".h")]):
            continue
        cmd = [hipcc, *flags, *extra.get(os.path.basename(s), []), "-x", "hip", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_oracle(force: bool = False, verbose: bool = True) -> str:
    src = os.path.join(ROOT, "oracle", "lsh_int.c")
    if not force and _newer(ORACLE_LIB, [src]):
        return ORACLE_LIB
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", ORACLE_LIB, src, "-lm"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ORACLE_LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    build_oracle(force="--force" in sys.argv)
