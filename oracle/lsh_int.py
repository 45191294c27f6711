"""ctypes binding of ``oracle/lsh_int.c`` (TEST INFRASTRUCTURE, see ``oracle/__init__.py``)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "lsh_int.c")
_LIB = os.path.join(_HERE, "_build", "liboracle_lsh.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
            os.makedirs(os.path.dirname(_LIB), exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", _LIB, _SRC, "-lm"])
        _lib = C.CDLL(_LIB)
    return _lib


def hash_buckets(qk: np.ndarray, rot: np.ndarray) -> np.ndarray:
    """qk (BH,T,dh) f32, rot (1|BH, dh, R, half) f32 -> buckets (BH,R,T) i32 (fixed-order fmaf)."""
    qk = np.ascontiguousarray(qk, dtype=np.float32)
    rot = np.ascontiguousarray(rot, dtype=np.float32)
    bh, t, dh = qk.shape
    rows, dh2, r, half = rot.shape
    assert dh2 == dh and rows in (1, bh)
    out = np.empty((bh, r, t), dtype=np.int32)
    _load().oracle_lsh_hash(qk.ctypes.data_as(C.c_void_p), rot.ctypes.data_as(C.c_void_p), rows, bh, t, dh, r, half,
                            out.ctypes.data_as(C.c_void_p))
    return out


def sort_buckets(buckets: np.ndarray, n_buckets: int):
    """buckets (BH,R,T) i32 -> st (slot -> t), undo (t -> slot), both (BH,R,T) i32."""
    buckets = np.ascontiguousarray(buckets, dtype=np.int32)
    bh, r, t = buckets.shape
    st = np.empty_like(buckets)
    undo = np.empty_like(buckets)
    _load().oracle_lsh_sort(buckets.ctypes.data_as(C.c_void_p), bh, t, r, n_buckets, st.ctypes.data_as(C.c_void_p),
                            undo.ctypes.data_as(C.c_void_p))
    return st, undo
