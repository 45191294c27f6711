"""Row kernels of the stack executor called through the C-ABI, against float64 torch (reference: torch.nn.LayerNorm's backward,
reformer_tts/model/reformer.py:85-93 for the two streams that start as one input)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("m,d", [(3072, 512), (1000, 128)])
@pytest.mark.parametrize("form", ["in place", "out of place", "joining"])
def test_layer_norm_backward_forms(gpu, m, d, form):
    """rtts_ln_bwd_join: dx_out = dx_in + addend + dLN(dxn); the LayerNorm parameter gradients through the partial rows.
    'joining' is the last update of a reversible stack's backward: d(input) = g1 + g2 rides in the pass that completes g2."""
    from reformer_tts_amd import _lib
    torch.manual_seed(m + d)
    x = torch.randn(m, d, device=gpu) * 2.0 + 0.3
    gamma = torch.rand(d, device=gpu) + 0.5
    dxn = torch.randn(m, d, device=gpu).bfloat16()
    dx_in = torch.randn(m, d, device=gpu)
    addend = torch.randn(m, d, device=gpu)
    mean = x.mean(-1)
    rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    # float64 reference
    x64 = x.double().requires_grad_()
    g64 = gamma.double().requires_grad_()
    b64 = torch.zeros(d, dtype=torch.float64, device=gpu, requires_grad=True)
    y = torch.nn.functional.layer_norm(x64, (d,), g64, b64, 1e-5)
    y.backward(dxn.double())
    want = dx_in.double() + x64.grad + (addend.double() if form == "joining" else 0.0)
    out = dx_in.clone() if form == "in place" else torch.empty_like(dx_in)
    src = out if form == "in place" else dx_in
    dgamma = torch.zeros(d, device=gpu)
    dbeta = torch.zeros(d, device=gpu)
    ws = torch.empty(2 * 256 * d, device=gpu)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.call("rtts_ln_bwd_join", dxn.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), src.data_ptr(),
              addend.data_ptr() if form == "joining" else None, out.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), m, d,
              None, None, 0.0, 0, None, stream)
    torch.cuda.synchronize()
    assert float((out.double() - want).abs().max() / want.abs().max()) < 2e-6
    assert float((dgamma.double() - g64.grad).norm() / g64.grad.norm()) < 1e-5
    assert float((dbeta.double() - b64.grad).norm() / b64.grad.norm()) < 1e-5
    if form == "joining":
        with pytest.raises(_lib.RttsError):          # the addend must not be the output
            _lib.call("rtts_ln_bwd_join", dxn.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), src.data_ptr(),
                      out.data_ptr(), out.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), m, d, None, None, 0.0, 0, None, stream)


@pytest.mark.parametrize("b,lp,lm", [(3, 200, 1024), (2, 256, 777), (1, 1, 1)])
def test_batch_masks_match_the_reference_sequence(gpu, b, lp, lm):
    """rtts_batch_masks against the operator sequence of reformer_tts.py:119-125 with the frame mask of wrappers.py:60
    (loss_mask.mean(-1)): padded phonemes, phoneme mask and its inverse, padded frame mask -- exact."""
    from reformer_tts_amd.model.reformer_tts import ReformerTTS, pad_to_multiple
    torch.manual_seed(b * 1000 + lp)
    pad_base, n_mels = 256, 80
    phon = torch.randint(0, 5, (b, lp), device=gpu)                        # zeros inside the text too: mask = (id != 0), not a length
    loss_mask = (torch.rand(b, lm, 1, device=gpu) > 0.3).float().expand(b, lm, n_mels).contiguous()
    loss_mask[:, lm // 2:] *= torch.rand(b, lm - lm // 2, n_mels, device=gpu)    # fractional rows: mean != 0 unless the whole row is 0
    spec = torch.randn(b, lm, n_mels, device=gpu)

    class _Stub:
        pad_base = 256
        _require_gpu = staticmethod(lambda: None)
    got = ReformerTTS._encode_inputs_fused(_Stub, phon, spec, loss_mask)
    want = ReformerTTS._encode_inputs(_Stub, phon, spec, loss_mask.mean(dim=-1))
    for g, w in zip(got, want):
        assert g.shape == w.shape and g.dtype == w.dtype and torch.equal(g, w)
    assert torch.equal(got[1]._rtts_not, ~want[1])
    # a strided view of the masks (frames [0, L-1) of a longer tensor), as the trainer hands it over
    longer = torch.cat([loss_mask, loss_mask[:, :1]], dim=1)
    got2 = ReformerTTS._encode_inputs_fused(_Stub, phon, spec, longer[:, :-1])
    assert torch.equal(got2[2], want[2])


def test_embedding_backward_reads_a_strided_gradient_in_place(gpu):
    """rtts_embedding_bwd_strided: dx as a (B, L, C) view of halo rows (what the convolution stack's backward hands to the embedding:
    edges.Halo.valid) gives bit for bit the gradient of the contiguous copy through rtts_embedding_bwd, with and without the
    dropout mask (keyed by the LOGICAL row); reference modules.py:17,22,56 (nn.Embedding(padding_idx=0) + Dropout)."""
    from reformer_tts_amd import _lib
    from reformer_tts_amd._seeds import seed_base
    torch.manual_seed(3)
    b, l, c, n = 5, 200, 128, 77
    halo = torch.randn(b, l + 4, c + 64, device=gpu)                 # a larger array: row stride c + 64, batch stride (l + 4) rows
    dx = halo[:, 2:2 + l, :c]
    assert not dx.is_contiguous() and dx.stride(2) == 1
    ids = torch.randint(0, n, (b * l,), device=gpu)
    s = torch.cuda.current_stream().cuda_stream
    for p, seed in ((0.0, 0), (0.1, 4242)):
        d_ref, d_str = torch.zeros(n, c, device=gpu), torch.zeros(n, c, device=gpu)
        flat = dx.reshape(-1, c).contiguous()
        _lib.call("rtts_embedding_bwd", ids.data_ptr(), flat.data_ptr(), b * l, c, n, 0, d_ref.data_ptr(), p, seed, seed_base(gpu).data_ptr(), s)
        _lib.call("rtts_embedding_bwd_strided", ids.data_ptr(), dx.data_ptr(), dx.stride(0), dx.stride(1), l, b * l, c, n, 0, d_str.data_ptr(), p, seed,
                  seed_base(gpu).data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(d_ref, d_str)
        assert float(d_ref[0].abs().max()) == 0.0 and float(d_ref[1:].abs().max()) > 0.0
