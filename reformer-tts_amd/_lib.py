"""ctypes binding of librtts_hip.so (C ABI declared in include/rtts.h).

The product path has no CPU fallback: if the library has not been built this
module raises, loudly, at first use."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTTS_LIB: an alternative build of the same library (kernel A/B runs on one box); it must exist, there is no fallback
LIB_PATH = os.environ.get("RTTS_LIB") or os.path.join(_HERE, "lib", "librtts_hip.so")

_i64, _i32, _vp, _f32, _u32 = C.c_int64, C.c_int, C.c_void_p, C.c_float, C.c_uint32

class GemmTnProblem(C.Structure):
    """rtts_gemm_tn_problem of include/rtts.h."""
    _fields_ = [("a", _vp), ("lda", _i64), ("b", _vp), ("ldb", _i64), ("c", _vp), ("ldc", _i64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("accumulate", C.c_int32)]


GEMM_TN_MAX_GROUP = 16


class GemmNtProblem(C.Structure):
    """rtts_gemm_nt_problem of include/rtts.h."""
    _fields_ = [("a", _vp), ("lda", _i64), ("w", _vp), ("ldw", _i64), ("c", _vp), ("ldc", _i64), ("bias", _vp), ("aux", _vp),
                ("ld_aux", _i64), ("aux_out", _vp), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("epilogue", C.c_int32),
                ("T", C.c_int32), ("H", C.c_int32), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


GEMM_NT_MAX_GROUP = 4


class ColsumJob(C.Structure):
    """rtts_colsum_job of include/rtts.h."""
    _fields_ = [("partial", _vp), ("out", _vp), ("nrows", C.c_int32), ("n", C.c_int32), ("ld", C.c_int32), ("reserved", C.c_int32)]


COLSUM_MAX_GROUP = 48


class Segment(C.Structure):
    """rtts_segment of include/rtts.h."""
    _fields_ = [("dst", _vp), ("src", _vp), ("count", C.c_int64), ("kind", C.c_int32), ("reserved", C.c_int32)]


SEGMENTS_MAX = 12
SEG_COPY_F32, SEG_COPY_BF16, SEG_ADD_F32, SEG_CAST_F32_BF16 = 0, 1, 2, 3


class ConvPermJob(C.Structure):
    """rtts_conv_perm_job of include/rtts.h."""
    _fields_ = [("w", _vp), ("wp", _vp), ("Co", C.c_int32), ("Ci", C.c_int32), ("CP", C.c_int32), ("reserved", C.c_int32)]


CONV_PERM_MAX_GROUP = 8

# name -> argtypes, exactly the prototypes of include/rtts.h
SIGNATURES = {
    "rtts_lsh_hash_sort": [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "rtts_lsh_attn_fwd": [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f32, _u32, _vp, _vp],
    "rtts_lsh_hash_sort_launches": [_i32],
    "rtts_lsh_attn_fwd_run_length": [_i32, _i32, _i32, _i32, _i32],
    "rtts_lsh_combine_fwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp],
    "rtts_lsh_bwd_delta": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp],
    "rtts_lsh_attn_bwd": [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _u32, _vp,
                          _vp],
    "rtts_lsh_bwd_qk_slots": [],
    "rtts_lsh_attn_bwd_run_length": [_i32, _i32, _i32, _i32, _i32],
    "rtts_debug_set_walk": [_i32, _i32],
    "rtts_lsh_bwd_reduce": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp],
    "rtts_grad_clip_scale": [_vp, _i64, _f32, _f32, _vp, _vp, _vp],
    "rtts_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "rtts_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f32, _u32, _vp, _vp],
    "rtts_ln_bwd_to": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f32, _u32, _vp, _vp],
    "rtts_ln_bwd_join": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f32, _u32, _vp, _vp],
    "rtts_cast_colsum": [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _u32, _vp, _vp, _vp],
    "rtts_colsum_bf16": [_vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp],
    "rtts_sum_streams": [_vp, _vp, _i64, _vp, _vp, _vp],
    "rtts_residual_epilogue": [_vp, _vp, _vp, _f32, _vp, _i64, _i32, _f32, _u32, _vp, _vp],
    "rtts_colsum_partial_rows": [_i32],
    "rtts_colsum_final_grouped": [C.POINTER(ColsumJob), _i32, _vp],
    "rtts_residual_ln": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _u32, _vp, _vp, _vp],
    "rtts_bias_act": [_vp, _vp, _i64, _i32, _i32, _vp],
    "rtts_cast_f32_bf16": [_vp, _vp, _i64, _vp],
    "rtts_xattn_fwd": [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _f32, _u32, _vp, _vp],
    "rtts_xattn_bwd": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _f32, _u32, _vp, _vp, _vp],
    "rtts_xattn_key_chunks": [_i32],
    "rtts_sum_slabs": [_vp, _i32, _i64, _vp, _vp],
    "rtts_conv1d_k5": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _vp],
    "rtts_to_halo": [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i64, _vp],
    "rtts_heads_grad": [_vp, _vp, _i32, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "rtts_segments": [C.POINTER(Segment), _i32, _vp],
    "rtts_conv_w_perm": [_vp, _i32, _i32, _i32, _vp, _vp],
    "rtts_conv_dw_unperm": [_vp, _i32, _i32, _i32, _vp, _vp],
    "rtts_conv_w_perm_grouped": [C.POINTER(ConvPermJob), _i32, _vp],
    "rtts_conv_dw_unperm_grouped": [C.POINTER(ConvPermJob), _i32, _vp],
    "rtts_conv1d_k5_moments": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp],
    "rtts_bn_stats_from_partials": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rtts_bn_stats": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rtts_bn_act_fwd": [_vp, _vp, _vp, _vp, _vp, _i32, _f32, _u32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i64, _vp],
    "rtts_bn_act_bwd": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _u32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp],
    "rtts_bn_moments": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "rtts_bn_from_moments": [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "rtts_bn_act_bwd_sums": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _u32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "rtts_bn_act_bwd_apply": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _u32, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _i64, _vp],
    "rtts_tts_loss": [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _i32,
                      _vp, _i64, _i32, _i32, _i64, _i64, _vp, _i64, _i64, _vp],
    "rtts_pe_add": [_vp, _vp, _vp, _f32, _u32, _vp, _i32, _i64, _i32, _vp, _vp],
    "rtts_debug_stamp": [_vp, _i32, _vp],
    "rtts_batch_masks": [_vp, _i64, _i32, _i32, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "rtts_pe_dalpha": [_vp, _vp, _f32, _u32, _vp, _i32, _i64, _i32, _vp, _vp, _vp],
    "rtts_relu_drop": [_vp, _f32, _u32, _vp, _i64, _vp],
    "rtts_embedding_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _u32, _vp, _vp],
    "rtts_embedding_bwd_strided": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _u32, _vp, _vp],
    "rtts_embedding_fwd": [_vp, _vp, _i32, _i32, _i32, _f32, _u32, _vp, _vp, _vp],
    "rtts_gemm_tn": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _i32, _vp, _i64, _vp],
    "rtts_gemm_tn_grouped": [C.POINTER(GemmTnProblem), _i32, _vp, _i64, _vp],
    "rtts_gemm_nt": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp],
    "rtts_gemm_nt_partial_rows": [_i32, _i32],
    "rtts_gemm_nt_grouped": [C.POINTER(GemmNtProblem), _i32, _i32, _vp],
    "rtts_debug_set_gemm_mode": [_i32],
    "rtts_gemm_nt_gate_words": [_i32, _i32],
    "rtts_gemm_nt_gated": [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _vp, _vp],
    "rtts_peak_copy": [_vp, _vp, _i64, _vp],
    "rtts_comm_probe": [_vp, _vp, _i64, _i32, _i32, _vp],
    "rtts_peak_mfma": [_vp, _i32, _i32, _vp],
    "rtts_sw_depthwise_k3": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "rtts_sw_gate": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "rtts_sw_coupling_inv": [_vp, _i64, _vp, _i64, _i32, _vp],
    "rtts_sw_coupling_inv1x1": [_vp, _i64, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _vp],
    "rtts_adamw_step": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _f32, _f32, _vp, _vp],
}

_lib = None


class RttsError(RuntimeError):
    pass


_NOTED = set()
PATHS_LEFT = []        # (where, reason) of every distinct departure from the explicit executors, in order (tests read this)


def note_general_path(where: str, reason: str) -> None:
    """Say ONCE per (where, reason) that a stack / edge / loss runs on the general eager path instead of the explicit HIP
    executors: a user can tell which path produced a number (logger ``reformer_tts_amd``, level WARNING)."""
    key = (where, reason)
    if key in _NOTED:
        return
    _NOTED.add(key)
    PATHS_LEFT.append(key)
    import logging
    logging.getLogger("reformer_tts_amd").warning("%s: general (eager) path instead of the explicit HIP executor -- %s", where, reason)


def log_once(key: str, message: str) -> None:
    """One WARNING per ``key`` through the package logger (decisions the user did not ask for but should see)."""
    if key in _NOTED:
        return
    _NOTED.add(key)
    import logging
    logging.getLogger("reformer_tts_amd").warning("%s", message)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RttsError(
                f"{LIB_PATH} is missing: build it with `python reformer-tts_amd/build.py` "
                "(or __graft_entry__.build()). There is no CPU fallback for the HIP path.")
        # Load order matters: librtts_hip.so needs libamdhip64.so.7, and PyTorch ships its own copy of the HIP runtime.
        # With torch imported first the loader binds this library to the runtime torch uses (one runtime per process:
        # torch's streams and device pointers are what the entry points receive).  Loaded the other way round, the
        # system copy under /opt/rocm comes in first and the first launch fails with "no ROCm-capable device".
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.rtts_version.restype = C.c_int
        lib.rtts_last_error.restype = C.c_char_p
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.rtts_gemm_nt_gate_words.restype = C.c_int64
        _lib = lib
        # A/B scripts may name a run length in the environment: read ONCE, here -- the library's launch path reads none
        fw, bw = os.environ.get("RTTS_LSH_FWD_WALK"), os.environ.get("RTTS_LSH_BWD_WALK")
        if fw is not None or bw is not None:
            _WALK[:] = [-1 if fw is None else int(fw), -1 if bw is None else int(bw)]
            lib.rtts_debug_set_walk(*_WALK)
        gs = os.environ.get("RTTS_GEMM_STORE")            # A/B runs: store form of the GEMM's bf16 epilogues (0 / 1 / 2)
        if gs is not None:
            lib.rtts_debug_set_gemm_mode(10 + int(gs))
    return _lib


_WALK = [-1, -1]


class forced_walk:
    """TEST-ONLY context manager: force the run length of the walking LSH attention kernels (``rtts_debug_set_walk``:
    None = leave as it is, -1 = the library's pick, 0 = the one-chunk kernel, n = runs of n chunks)."""

    def __init__(self, fwd=None, bwd=None):
        self.want = (fwd, bwd)

    def __enter__(self):
        self.old = list(_WALK)
        new = [o if w is None else int(w) for o, w in zip(self.old, self.want)]
        call("rtts_debug_set_walk", *new)
        _WALK[:] = new
        return self

    def __exit__(self, *exc):
        call("rtts_debug_set_walk", *self.old)
        _WALK[:] = self.old
        return False


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RttsError(lib.rtts_last_error().decode() or f"{name} failed with code {rc}")
