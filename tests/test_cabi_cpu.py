"""CPU-side checks: the C-ABI library loads and exports every symbol include/rtts.h declares
(no compute calls without a GPU), and the C twin of the integer stages agrees with the PyTorch
restatement and with HuggingFace's fixtures."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import lsh_int, lsh_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "rtts.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from reformer_tts_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), s
    assert set(_lib.SIGNATURES) <= set(syms)
    assert lib.rtts_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    from reformer_tts_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librtts_hip.so")
    with pytest.raises(_lib.RttsError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.parametrize("bh,t,bs,nh,per_head", [(3, 256, 64, 4, False), (2, 1024, 128, 8, True), (2, 512, 64, 2, False)])
def test_c_twin_vs_pytorch_restatement(bh, t, bs, nh, per_head):
    g = torch.Generator().manual_seed(t + nh)
    qk = torch.randn(bh, t, 64, generator=g).bfloat16().float()
    rot = torch.randn(bh if per_head else 1, 64, nh, t // bs // 2, generator=g)
    b_c = lsh_int.hash_buckets(qk.numpy(), rot.numpy())
    b_t = lsh_ref.hash_vectors(qk, rot).reshape(bh, nh, t).numpy()
    assert (b_c != b_t).mean() < 1e-3          # only exact near-ties may differ (einsum order)
    st_c, undo_c = lsh_int.sort_buckets(b_c, t // bs)
    sticker, undo = lsh_ref.sort_buckets(torch.from_numpy(b_c.reshape(bh, nh * t).astype(np.int64)), t)
    assert np.array_equal(st_c.reshape(bh, -1), (sticker % t).numpy())
    flat_undo = undo_c + (np.arange(nh, dtype=np.int32) * t)[None, :, None]
    assert np.array_equal(flat_undo.reshape(bh, -1), undo.numpy())


def test_c_twin_vs_huggingface_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "hf_lsh_int.npz"))
    for tag in ("a", "b", "c"):
        heads, t, dh, chunk, nh = (int(x) for x in z[f"{tag}/meta"])
        vec = np.random.RandomState(5).standard_normal((2, heads, t, dh)).astype(np.float32)
        rot = np.tile(z[f"{tag}/rot"], (2, 1, 1, 1))
        buckets = lsh_int.hash_buckets(vec.reshape(2 * heads, t, dh), rot)
        exp = z[f"{tag}/buckets"].astype(np.int32).reshape(2 * heads, nh, t)
        assert (buckets != exp).mean() < 1e-4, tag
        st, _ = lsh_int.sort_buckets(exp, t // chunk)
        exp_sorted = z[f"{tag}/sorted_idx"].astype(np.int32).reshape(2 * heads, nh, t) % t
        assert np.array_equal(st, exp_sorted), tag


def test_model_refuses_to_run_off_the_gpu():
    """No CPU fallback: a forward (or generation) on a CPU-resident model raises the library's own error at the door."""
    import torch
    from reformer_tts_amd import _lib
    from reformer_tts_amd.model.config import baseline_model_config
    from reformer_tts_amd.training import build_model, synthetic_batch
    cfg = baseline_model_config()
    cfg.enc_reformer_kwargs.depth = 1
    cfg.dec_reformer_kwargs.depth = 1
    model = build_model(cfg)
    batch = synthetic_batch(1, 20, 256)
    with pytest.raises(_lib.RttsError, match="GPU only"):
        model(batch["phonemes"], batch["spectrogram"][:, :-1])
    with pytest.raises(_lib.RttsError, match="GPU only"):
        model.infer(batch["phonemes"], max_len=90)
