#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE.

Runs only in the build container (needs ``/root/reference``); the fixtures it
writes are plain data (inputs, parameters, expected outputs) and are committed,
so the tests never need the reference.  Nothing of the reference's source is
stored -- the script imports its modules, feeds them seeded inputs and records
what they return.

    python tests/golden/make_golden.py

Fixtures
  pieces.npz       one entry group per reference component run on its own
                   
  model_small.npz  whole-model wiring: the reference's ReformerTTS + TTSLoss with
                   ``reformer_pytorch`` (absent here) replaced by
                   ``oracle.lsh_ref.LSHSelfAttention`` -- "reference wiring x
                   restated LSH" (SURVEY.md 8c (2)); parameters come from
                   ``oracle.synth`` and are therefore not stored.
  model_cfg1.npz   the same wiring fixture at BASELINE config #1's values: config/baseline.yml with 1 + 1 layers, d = 512,
                   8 heads, buckets 64 / 128, pad_base 256, feed-forward 2048 (B = 1, 150 phonemes, 800 frames)
  infer_small.npz  the reference's ReformerTTS.infer (autoregressive loop of full eval-mode forwards) on the same
                   wiring, three strategies; rotations recorded in call order
  squeezewave_{small,full}.npz  the reference's SqueezeWave.infer (vocoder)
  hf_lsh_int.npz   integer stages (hash, stable sort) from HuggingFace's
                   independent implementation of the same paper (cross-check,
                   not the reference)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import lsh_ref, model_ref, synth  # noqa: E402

shim = types.ModuleType("reformer_pytorch")
shim.LSHSelfAttention = lsh_ref.LSHSelfAttention
sys.modules["reformer_pytorch"] = shim

from reformer_tts.model import modules as R_modules  # noqa: E402
from reformer_tts.model import reformer as R_reformer  # noqa: E402
from reformer_tts.model import reversible as R_rev  # noqa: E402
from reformer_tts.model.loss import TTSLoss  # noqa: E402
from reformer_tts.model.reformer_tts import ReformerTTS, pad_to_multiple  # noqa: E402
from reformer_tts.dataset.utils import custom_sequence_padder  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def put_sd(out, tag, module):
    for k, v in module.state_dict().items():
        out[f"{tag}/sd/{k}"] = npy(v)


def pieces():
    out = {}
    g = torch.Generator().manual_seed(1234)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    # a4 ScaledPositionalEncoding
    torch.manual_seed(1)
    m = R_modules.ScaledPositionalEncoding(16, 0.0).train()
    x = rnd(2, 10, 16)
    put_sd(out, "pe", m); out["pe/x"] = npy(x); out["pe/y"] = npy(m(x))

    # a2 EncoderPreNet (train mode: batch statistics; dropout 0)
    m = R_modules.EncoderPreNet(13, 16, dropout=0.0).train()
    ids = torch.randint(0, 13, (3, 11), generator=g)
    put_sd(out, "encpre", m); out["encpre/ids"] = npy(ids); out["encpre/y"] = npy(m(ids))

    # a3 DecoderPreNet
    m = R_modules.DecoderPreNet(8, 16, hidden_size=12, dropout=0.0).train()
    x = rnd(2, 7, 8)
    put_sd(out, "decpre", m); out["decpre/x"] = npy(x); out["decpre/y"] = npy(m(x))

    # a16 PostConvNet
    m = R_modules.PostConvNet(8, 16, dropout=0.0, depth=2).train()
    x = rnd(2, 9, 8)
    put_sd(out, "postnet", m); out["postnet/x"] = npy(x); out["postnet/y"] = npy(m(x))

    # a8 + a9 Chunk(WithNorm(LayerNorm, FeedForward))
    ff = R_modules.FeedForward(16, hidden=32, dropout=0.0)
    m = R_reformer.Chunk(5, R_reformer.WithNorm(torch.nn.LayerNorm, 16, ff), along_dim=-2).train()
    with torch.no_grad():
        m.fn.norm.weight.add_(0.1 * rnd(16)); m.fn.norm.bias.add_(0.1 * rnd(16))
    x = rnd(2, 12, 16)
    put_sd(out, "ffn", m); out["ffn/x"] = npy(x); out["ffn/y"] = npy(m(x))

    # a10 WithNorm(LayerNorm, MultiheadAttentionWrapper) in eval mode to capture the weights too
    mats = []
    mha = R_reformer.MultiheadAttentionWrapper(16, mats, num_heads=2, dropout=0.0)
    m = R_reformer.WithNorm(torch.nn.LayerNorm, 16, mha).eval()
    x, keys = rnd(2, 6, 16), rnd(2, 5, 16)
    kpm = torch.tensor([[False, False, False, True, True], [False] * 5])
    y = m(x, key=keys, value=keys, key_padding_mask=kpm)
    put_sd(out, "xattn", m); out["xattn/x"] = npy(x); out["xattn/keys"] = npy(keys)
    out["xattn/kpm"] = npy(kpm); out["xattn/y"] = npy(y); out["xattn/w"] = npy(mats[0])

    # a12 ReversibleBlock stack and a13 HalfResidual/Swap chain: forward + grads
    def lin_block():
        return R_reformer.WithNorm(torch.nn.LayerNorm, 8, torch.nn.Linear(8, 8))
    blocks = torch.nn.ModuleList([R_rev.ReversibleBlock(lin_block(), lin_block()) for _ in range(2)])
    seq = R_rev.ReversibleSequence(blocks).train()
    x = rnd(2, 5, 16).requires_grad_()
    y = seq(x, kwargs_list=[dict(f_args={}, g_args={}) for _ in range(2)])
    dy = rnd(2, 5, 16)
    y.backward(dy)
    put_sd(out, "revblock", seq); out["revblock/x"] = npy(x); out["revblock/y"] = npy(y)
    out["revblock/dy"] = npy(dy); out["revblock/dx"] = npy(x.grad)
    for k, p in seq.named_parameters():
        out[f"revblock/grad/{k}"] = npy(p.grad)

    chain = []
    for _ in range(2):
        chain += [R_rev.ReversibleHalfResidual(lin_block()), R_rev.ReversibleSwap()]
    seq = R_rev.ReversibleSequence(torch.nn.ModuleList(chain)).train()
    x = rnd(2, 5, 16).requires_grad_()
    y = seq(x, kwargs_list=[dict() for _ in chain])
    y.backward(dy)
    put_sd(out, "revhalf", seq); out["revhalf/x"] = npy(x); out["revhalf/y"] = npy(y)
    out["revhalf/dy"] = npy(dy); out["revhalf/dx"] = npy(x.grad)
    for k, p in seq.named_parameters():
        out[f"revhalf/grad/{k}"] = npy(p.grad)

    # a17 TTSLoss (clones: the reference multiplies its arguments in place)
    raw, post, stop = rnd(2, 7, 4), rnd(2, 7, 4), rnd(2, 7)
    mel, tstop = rnd(2, 7, 4), (torch.rand(2, 7, generator=g) > 0.7).float()
    mask = torch.ones(2, 7, 4); mask[1, 4:] = 0
    for kind in ("mse", "l1"):
        res = TTSLoss(torch.tensor(5.0), 1.0, 0.5, 2.0, kind)(raw.clone(), post.clone(), stop, mel, tstop, mask)
        out[f"loss/{kind}"] = np.array([float(r) for r in res], dtype=np.float32)
    for k, v in dict(raw=raw, post=post, stop=stop, mel=mel, tstop=tstop, mask=mask).items():
        out[f"loss/{k}"] = npy(v)

    # a1 pad_to_multiple, a18/next-2 collate
    x = rnd(2, 5, 3)
    out["pad/x"] = npy(x); out["pad/y4"] = npy(pad_to_multiple(x, 4)); out["pad/y5"] = npy(pad_to_multiple(x, 5))
    items = [dict(phonemes=torch.randint(1, 9, (n,), generator=g), spectrogram=rnd(l, 4)) for n, l in ((3, 5), (6, 2), (4, 7))]
    col = custom_sequence_padder(items)
    for i, it in enumerate(items):
        out[f"collate/in{i}/phonemes"] = npy(it["phonemes"]); out[f"collate/in{i}/spectrogram"] = npy(it["spectrogram"])
    for k, v in col.items():
        out[f"collate/out/{k}"] = npy(v)
    np.savez_compressed(os.path.join(HERE, "pieces.npz"), **out)
    print("pieces.npz:", len(out), "arrays")


small_cfg = model_ref.small_cfg


def model_small():
    _whole_model("model_small.npz", small_cfg(), model_ref.synthetic_batch(2, 40, 150, ragged=True, seed=1), store_outputs=True)


def model_cfg1():
    """The reference's ReformerTTS + TTSLoss at BASELINE config #1's values (``oracle.model_ref.cfg1``: config/baseline.yml with
    1 + 1 layers at d = 512 / 8 heads, buckets 64 / 128, pad_base 256, feed-forward 2048 in 100 ``Chunk`` pieces), B = 1, one
    LJSpeech-shaped utterance of 150 phonemes / 800 frames (padded to 256 / 1024 by the model): pins the encoder-64 /
    decoder-128 bucket asymmetry, pad_base = 2 x bucket and the 94-piece feed-forward split of the ORACLE'S wiring against the
    reference's.  The 800 x 80 outputs are stored as float16 (the comparison is at 1e-3), gradients as norms."""
    _whole_model("model_cfg1.npz", model_ref.cfg1(), model_ref.synthetic_batch(1, 150, 800, seed=5), store_outputs="f16")


def _whole_model(fname, cfg, batch, store_outputs):
    torch.manual_seed(7)
    m = ReformerTTS(**cfg).train()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=3)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(("running" in k or "num_batches" in k) for k in missing.missing_keys), missing
    rot_log = []
    for mod in m.modules():
        if isinstance(mod, lsh_ref.LSHSelfAttention):
            mod.rotation_log = rot_log
    spec = batch["spectrogram"]
    torch.manual_seed(11)
    raw, post, stop, _ = m(batch["phonemes"], spec[:, :-1], batch["loss_mask"].mean(-1))
    n_fwd = len(rot_log)
    res = TTSLoss(torch.tensor(5.0))(raw.clone(), post.clone(), stop.view(stop.shape[0], -1), spec[:, 1:],
                                     batch["stop_tokens"], batch["loss_mask"])
    res[0].backward()
    # the reversible backward replays the RNG, so the rotations drawn during recompute repeat the forward ones
    for i, r in enumerate(rot_log[n_fwd:]):
        assert any(torch.equal(r, f) for f in rot_log[:n_fwd]), i
    out = {f"shape/{k}": np.array(s, dtype=np.int64) for k, s in shapes.items()}
    for k, v in batch.items():
        out[f"batch/{k}"] = npy(v)
    for i, r in enumerate(rot_log[:n_fwd]):
        out[f"rot/{i}"] = npy(r)
    cast = (lambda a: a.astype(np.float16)) if store_outputs == "f16" else (lambda a: a)
    out["out/raw"], out["out/post"], out["out/stop"] = cast(npy(raw)), cast(npy(post)), cast(npy(stop))
    out["out/loss"] = np.array([float(r) for r in res], dtype=np.float32)
    for k, p in m.named_parameters():
        out[f"gradnorm/{k}"] = np.array(float(p.grad.norm()), dtype=np.float32)
    for k in ("dec.mel_linear.weight", "enc.positional_encoding.alpha", "dec.positional_encoding.alpha",
              "enc.reformer.layers.blocks.0.f.net.fn.layer.to_out.bias",
              "dec.reformer.layers.blocks.0.f.net.norm.weight", "dec.reformer.layers.blocks.2.f.net.fn.layer.in_proj_bias"):
        out[f"grad/{k}"] = npy(dict(m.named_parameters())[k].grad)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, ": loss", out["out/loss"], "layers", n_fwd, "padded lengths", [tuple(r.shape) for r in rot_log[:n_fwd]])


STOP_T = float(os.environ.get("GOLDEN_STOP_T", "0.995"))   # synthetic weights give large stop logits


def infer_small():
    """the reference's own ReformerTTS.infer (reformer_tts.py:145-221) in eval mode with
    non-trivial BatchNorm running statistics; every rotation the LSH shim draws is recorded in call order."""
    cfg = small_cfg()
    torch.manual_seed(7)
    m = ReformerTTS(**cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=3)
    m.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(21)
    out = {}
    for name, buf in m.named_buffers():
        if name.endswith("running_mean"):
            buf.copy_(0.2 * torch.randn(buf.shape, generator=g))
        elif name.endswith("running_var"):
            buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        if "running" in name:
            out[f"buf/{name}"] = npy(buf)
    m.eval()
    phonemes = torch.randint(1, 77, (2, 23), generator=g)
    out["phonemes"] = npy(phonemes)
    # max_len must exceed n_mels = 80: the loop's guard is max(spectrogram.shape) > max_len (reformer_tts.py:212), and
    # the shape includes the mel axis, so anything smaller ends after one forward
    for strategy, cs, kw in (("concat", "concat", dict(max_len=86, stop_at_stop_token=False)),
                             ("concat_stop", "concat", dict(max_len=90, stop_threshold=STOP_T)),
                             ("replace", "replace", dict(max_len=82, stop_at_stop_token=False)),
                             ("replace_stop", "replace", dict(max_len=84, stop_threshold=STOP_T)),
                             ("short", "concat", dict(max_len=7, stop_at_stop_token=False))):
        rot_log = []
        for mod in m.modules():
            if isinstance(mod, lsh_ref.LSHSelfAttention):
                mod.rotation_log = rot_log
        torch.manual_seed(5)
        with torch.no_grad():
            spec, stop = m.infer(phonemes, combine_strategy=cs, **kw)
        out[f"{strategy}/spectrogram"], out[f"{strategy}/stop"] = npy(spec), npy(stop)
        for i, r in enumerate(rot_log):
            out[f"{strategy}/rot/{i}"] = npy(r)
        out[f"{strategy}/n_rot"] = np.array(len(rot_log))
        out[f"{strategy}/kw"] = np.array([kw["max_len"], kw.get("stop_threshold", 0.25), float(kw.get("stop_at_stop_token", True))])
        print("infer", strategy, tuple(spec.shape), stop.tolist(), "forwards", len(rot_log) // 2)
    np.savez_compressed(os.path.join(HERE, "infer_small.npz"), **out)


def squeezewave():
    """the reference's SqueezeWave.infer (squeeze_wave/modules.py:334-376), eval mode, weight norm
    and BatchNorm in place (NOT folded), random non-degenerate parameters; the Gaussian draws come from torch's global
    generator under a recorded seed.  'small' stores its whole state_dict; 'full' (the default 12-flow / 256-channel
    configuration, 23.7 M parameters) takes its parameters from oracle.synth by name and stores only the BatchNorm
    running statistics."""
    sys.modules.setdefault("dacite", types.ModuleType("dacite"))
    from reformer_tts.squeeze_wave.config import WNConfig
    from reformer_tts.squeeze_wave.modules import SqueezeWave
    from oracle import squeezewave_ref as sw_ref
    for tag, cfg, mel_len, batch in (("small", sw_ref.small_cfg(), 24, 2), ("full", sw_ref.default_cfg(), 32, 1)):
        wn = cfg["wn_config"]
        torch.manual_seed(3)
        m = SqueezeWave(cfg["n_flows"], cfg["n_audio_channels"], cfg["n_mel_channels"], cfg["early_return_interval"],
                        cfg["early_return_size"], WNConfig(wn["n_layers"], wn["n_channels"], wn["conv_kernel_size"], wn["mel_upsample_scale"]))
        g = torch.Generator().manual_seed(17)
        out = {}
        sd = m.state_dict()
        if tag == "full":
            shapes = {k: tuple(v.shape) for k, v in sd.items()}
            syn = synth.synth_state_dict(shapes, seed=11)
            for k in sd:
                if "inv_conv_layers" in k or "num_batches" in k or k not in syn:
                    continue                       # keep the orthonormal initialisation (well conditioned inverse)
                sd[k] = syn[k] * (0.05 if "end_conv" in k else 1.0)      # log-scales near 0: the flow stays finite
            for k, s_ in shapes.items():
                out[f"shape/{k}"] = np.array(s_, dtype=np.int64)
        else:
            for k, v in sd.items():
                if v.dtype.is_floating_point and "inv_conv_layers" not in k:
                    sd[k] = (0.02 if "end_conv" in k else 0.3) * torch.randn(v.shape, generator=g) + \
                        (1.0 if k.endswith("weight_g") or k.endswith("0.weight") else 0.0)
        for k in sd:
            if k.endswith("running_var"):
                sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
            elif k.endswith("running_mean"):
                sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
            if tag == "small" or "running" in k or "inv_conv_layers" in k:
                out[f"sd/{k}"] = npy(sd[k])
        m.load_state_dict(sd)
        m.eval()
        mel = (torch.randn(batch, cfg["n_mel_channels"], mel_len, generator=g) * 2.0 - 5.0).clamp(-11.5, 2.0)
        torch.manual_seed(99)
        with torch.no_grad():
            audio = m.infer(mel, sigma=0.6)
        out["mel"], out["audio"], out["seed"] = npy(mel), npy(audio), np.array(99)
        np.savez_compressed(os.path.join(HERE, f"squeezewave_{tag}.npz"), **out)
        print("squeezewave", tag, tuple(audio.shape), "clamped fraction", float((audio.abs() >= 1).float().mean()))


def hf_lsh_int():
    """HuggingFace LSHSelfAttention integer stages on seeded vectors (per-head rotations)."""
    from transformers import ReformerConfig
    from transformers.models.reformer.modeling_reformer import LSHSelfAttention as HF
    out = {}
    for tag, (heads, t, dh, chunk, n_hashes) in dict(a=(2, 256, 64, 64, 4), b=(3, 512, 64, 64, 8), c=(2, 1024, 64, 128, 2)).items():
        nb = t // chunk
        cfg = ReformerConfig(hidden_size=heads * dh, num_attention_heads=heads, attention_head_size=dh, num_hashes=n_hashes,
                             lsh_attn_chunk_length=chunk, num_buckets=nb, hash_seed=99, attn_layers=["lsh"])
        layer = HF(cfg)
        layer.num_buckets = nb
        vec = torch.from_numpy(np.random.RandomState(5).standard_normal((2, heads, t, dh)).astype(np.float32))
        buckets = layer._hash_vectors(vec, n_hashes, None)                      # (B, heads, n_hashes*T)
        torch.manual_seed(99)
        rotations = torch.randn(heads, dh, n_hashes, nb // 2)                   # what _hash_vectors drew
        sidx, undo = layer._get_sorted_bucket_idx_and_undo_sorted_bucket_idx(t, buckets, n_hashes)
        out[f"{tag}/rot"] = npy(rotations)      # vec is regenerated from RandomState(5) by the test
        out[f"{tag}/buckets"] = npy(buckets).astype(np.int16)
        out[f"{tag}/sorted_idx"] = npy(sidx).astype(np.int16); out[f"{tag}/undo"] = npy(undo).astype(np.int16)
        out[f"{tag}/meta"] = np.array([heads, t, dh, chunk, n_hashes], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "hf_lsh_int.npz"), **out)
    print("hf_lsh_int.npz:", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(4)
    pieces()
    model_small()
    model_cfg1()
    infer_small()
    squeezewave()
    hf_lsh_int()
