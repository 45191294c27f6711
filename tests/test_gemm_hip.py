"""rtts_gemm_nt (csrc/gemm_nt.hip) -- the hand-written MFMA GEMM behind every projection and feed-forward layer of the
stacks -- through the C ABI:
  * every layout / epilogue against a float64 product of the SAME bf16 operands (separates kernel error from bf16
    rounding of the inputs: what is left is fp32 accumulation order + ONE rounding of the result to bf16);
  * the FeedForward executor (forward, reconstruction, input and parameter gradients) against the CPU oracle
    ``oracle.model_ref.feed_forward`` -- itself pinned to the reference's ``Chunk(WithNorm(FeedForward))``
    (/root/reference/reformer_tts/model/modules.py:195-207, reformer.py:25-45) by tests/golden/pieces.npz ``ffn/*`` in
    tests/test_oracle_golden.py -- at the baseline shape (12288, 512) -> 2048 -> 512 and at the T = 4096 shape.
Every test prints the error it achieved next to its tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import __graft_entry__
    __graft_entry__.build()
    return torch.device("cuda:0")


def _gemm(a, w, **kw):
    from reformer_tts_amd import engine
    return engine.gemm(a, w, **kw)


def _rel(x, ref):
    """(max abs error / max |ref|, rel-L2 error) in float64."""
    x, ref = x.double(), ref.double()
    return ((x - ref).abs().max() / ref.abs().max()).item(), ((x - ref).norm() / ref.norm()).item()


# one bf16 rounding of the result: half an ulp is between 2^-9 (top of a binade) and 2^-8 (bottom) of the value, so against
# max |ref| the bound is 2^-8 = 3.9e-3 (+ fp32 accumulation noise, orders of magnitude smaller); rel-L2 of a rounding to 8
# significant bits ~ 2^-9 / sqrt(3) * (binade average) = 1.6e-3 .. 1.7e-3 measured
BF16_HALF_ULP = 2.0 ** -8


@pytest.mark.parametrize("m,n,k", [(192, 128, 64), (384, 256, 192), (256, 128, 128), (96, 64, 64), (128, 64, 256),
                                   (12288, 512, 512), (12288, 1024, 512), (12288, 2048, 512), (12288, 512, 2048),
                                   (12288, 512, 1024), (3072, 1024, 512), (3072, 512, 1024), (16384, 2048, 512),
                                   (16384, 512, 2048), (768, 384, 384), (512, 256, 128), (256, 512, 64), (49152, 256, 192)])
def test_gemm_nt_layouts_and_epilogues_vs_float64(gpu, m, n, k):
    g = torch.Generator(device=gpu).manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, device=gpu, generator=g).bfloat16()
    w = (torch.randn(n, k, device=gpu, generator=g) / k ** 0.5).bfloat16()
    bias = torch.randn(n, device=gpu, generator=g)
    gate = torch.randn(m, n, device=gpu, generator=g).bfloat16()
    ref = a.double() @ w.double().t()
    wkn = w.t().contiguous()
    out = {
        "nt": (_gemm(a, w), ref),
        "nt+bias": (_gemm(a, w, bias=bias), ref + bias.double()),
        "nt+bias+relu": (_gemm(a, w, bias=bias, relu=True), torch.relu(ref + bias.double())),
        "kn": (_gemm(a, wkn, kn=True), ref),
        "kn+gate": (_gemm(a, wkn, kn=True, gate=gate), ref * (gate.double() > 0)),
    }
    msgs = []
    for name, (got, want) in out.items():
        emax, el2 = _rel(got, want)
        msgs.append(f"{name} max {emax:.2e} l2 {el2:.2e}")
        assert emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (name, emax, el2)
    # the partial column sums of the gated product (fp32, summed before the rounding to bf16)
    from reformer_tts_amd import engine
    db = torch.zeros(n, device=gpu)
    engine.gemm(a, wkn, kn=True, gate=gate, gate_bias_grad=db)
    engine.flush_wgrad()
    cref = (ref * (gate.double() > 0)).sum(0)
    ecs = ((db.double() - cref).abs().max() / cref.abs().max()).item()
    assert ecs <= 1e-5, ecs
    print(f"\n[gemm_nt {m}x{n}x{k}] " + "; ".join(msgs) + f"; colsum {ecs:.2e} (tol: max {1.02 * BF16_HALF_ULP:.2e}, l2 2e-3, colsum 1e-5)")


@pytest.mark.parametrize("m,n,k", [(12288, 2048, 512), (3072, 2048, 512), (768, 384, 384), (256, 128, 128), (96, 64, 64)])
def test_one_bit_relu_gate_equals_the_activation_gate(gpu, m, n, k):
    """The feed-forward pair with the 1-bit gate: the words the forward's bias + ReLU epilogue leaves say exactly "this output is a
    positive bf16", so the input gradient gated by them is bit-identical to the one gated by the activation, column sums too
    (every tile shape that has a word form; the forward's output is unchanged by writing them)."""
    from reformer_tts_amd import _lib, engine
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).bfloat16().to(gpu)
    w1 = (torch.randn(n, k, generator=g) / k ** 0.5).bfloat16().to(gpu)
    b1 = torch.randn(n, generator=g).to(gpu)
    dy = torch.randn(m, 512, generator=g).bfloat16().to(gpu)
    w2 = (torch.randn(512, n, generator=g) / 512 ** 0.5).bfloat16().to(gpu)          # (K, N) of the input gradient dh = dy W2
    words = engine.gate_words(m, n, gpu)
    assert words is not None and words.numel() == _lib.load().rtts_gemm_nt_gate_words(m, n)
    h_plain = _gemm(x, w1, bias=b1, relu=True)
    h = _gemm(x, w1, bias=b1, relu=True, words=words)
    assert torch.equal(h, h_plain)
    rows = _lib.load().rtts_gemm_nt_partial_rows(m, n)
    outs = []
    for kw in (dict(gate=h), dict(gate=True, words=words)):
        cs = torch.zeros(n, device=gpu)
        c = _gemm(dy, w2, kn=True, gate_bias_grad=cs, **kw)
        engine.flush_colsum()
        torch.cuda.synchronize()
        outs.append((c, cs))
    assert rows > 0 and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    frac = float((h > 0).float().mean())
    print(f"\n[one-bit gate {m}x{n}x{k}] {words.numel()} words for {m * n} outputs ({frac:.2f} positive): input gradient and bias sums "
          f"bit-identical to the activation-gated ones")


def test_gemm_nt_rejects_shapes_it_cannot_tile(gpu):
    from reformer_tts_amd import _lib
    a = torch.zeros(100, 64, device=gpu, dtype=torch.bfloat16)
    w = torch.zeros(64, 64, device=gpu, dtype=torch.bfloat16)
    with pytest.raises(_lib.RttsError, match="tiles by none"):
        _gemm(a, w)
    a = torch.zeros(128, 72, device=gpu, dtype=torch.bfloat16)
    w = torch.zeros(64, 72, device=gpu, dtype=torch.bfloat16)
    with pytest.raises(_lib.RttsError, match="multiple of 64"):
        _gemm(a, w)


def _r16(x):
    """Round to bfloat16 (values kept in the input dtype)."""
    return x.to(torch.bfloat16).to(x.dtype)


class _Ste(torch.autograd.Function):
    """Forward: round to bf16 (or identity); backward: round the gradient to bf16 (or identity) -- places the executor's
    roundings of activations and of activation gradients into a float64 autograd graph."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _r16(x) if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (_r16(g) if ctx.bwd else g), None, None


def _ffn_case(gpu, b, t, d=512, hidden=2048, seed=0):
    """FFNExec forward + backward against (a) oracle.model_ref.feed_forward under fp32 autograd and (b) a float64 model of
    the executor's own arithmetic: the same function with bf16 roundings where the executor stores bf16 (LayerNorm output,
    hidden activations, block output, output gradient, hidden gradient, LayerNorm-input gradient; weights rounded once)."""
    from oracle import model_ref
    from reformer_tts_amd import engine
    from reformer_tts_amd.model.modules import FeedForward
    from reformer_tts_amd.model.reformer import Chunk, WithNorm
    torch.manual_seed(seed)
    mod = Chunk(100, WithNorm(torch.nn.LayerNorm, d, FeedForward(d, hidden)), along_dim=-2)
    with torch.no_grad():
        mod.fn.norm.weight.add_(0.1 * torch.randn(d))
        mod.fn.norm.bias.add_(0.1 * torch.randn(d))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in mod.state_dict().items()}
    x = torch.randn(b * t, d)
    acc0 = torch.randn(b * t, d)
    dy = torch.randn(b * t, d) / d ** 0.5
    # (a) oracle: gradients of <dy, ffn(x)> w.r.t. x and the parameters
    xo = x.clone().requires_grad_(True)
    fo = model_ref.feed_forward(sd, "fn.", xo.view(b, t, d)).reshape(b * t, d)
    (fo * dy).sum().backward()
    # (b) the same function in float64 with the executor's roundings
    sd64 = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
    w1, w2 = _r16(sd64["fn.fn.net.0.weight"]), _r16(sd64["fn.fn.net.3.weight"])     # identity gradient: dW of the rounded weight
    w1 = sd64["fn.fn.net.0.weight"] + (w1 - sd64["fn.fn.net.0.weight"]).detach()
    w2 = sd64["fn.fn.net.3.weight"] + (w2 - sd64["fn.fn.net.3.weight"]).detach()
    x64 = x.double().requires_grad_(True)
    xn = _Ste.apply(torch.nn.functional.layer_norm(x64, (d,), sd64["fn.norm.weight"], sd64["fn.norm.bias"], 1e-5), True, True)
    z = _Ste.apply(xn @ w1.t() + sd64["fn.fn.net.0.bias"], False, True)              # d(loss)/dz = the gated hidden gradient, bf16
    h = _Ste.apply(torch.relu(z), True, False)
    f64 = _Ste.apply(h @ w2.t(), True, False) + sd64["fn.fn.net.3.bias"]
    (f64 * _r16(dy.double())).sum().backward()

    mod = mod.to(gpu)
    ex = engine.FFNExec(mod)
    acc = acc0.to(gpu).clone()
    inp = x.to(gpu).clone()
    ex.forward(acc, inp, b, t)
    torch.cuda.synchronize()
    fwd = acc.cpu() - acc0
    res = {"forward": _rel(fwd, fo.detach())}
    sharp = {"forward": _rel(fwd, f64.detach())}
    d_acc = dy.to(gpu).clone()
    d_inp = torch.zeros(b * t, d, device=gpu)
    for p in mod.parameters():
        p.grad = None
    ex.backward(acc, inp, d_acc, d_inp, b, t)
    engine.flush_wgrad()
    torch.cuda.synchronize()
    res["reconstruction"] = _rel(acc.cpu(), acc0)                       # acc -= f(inp): back to the stream before the block
    res["d_input"] = _rel(d_inp.cpu(), xo.grad)
    sharp["d_input"] = _rel(d_inp.cpu(), x64.grad)
    for name, p in mod.state_dict(keep_vars=True).items():
        res["d_" + name] = _rel(p.grad.cpu(), sd[name].grad)
        sharp["d_" + name] = _rel(p.grad.cpu(), sd64[name].grad)
    return res, sharp


@pytest.mark.parametrize("b,t", [(12, 1024), (4, 4096)])
def test_ffn_executor_vs_oracle_feed_forward(gpu, b, t):
    res, sharp = _ffn_case(gpu, b, t)
    print(f"\n[ffn executor B={b} T={t} d=512 hidden=2048] vs fp32 oracle: " + "; ".join(f"{k} max {v[0]:.2e} l2 {v[1]:.2e}" for k, v in res.items()))
    print(f"[ffn executor B={b} T={t}] vs float64 model of the bf16 arithmetic: " + "; ".join(f"{k} max {v[0]:.2e} l2 {v[1]:.2e}" for k, v in sharp.items()))
    # Sharp: identical roundings in float64 -- what is left is fp32 accumulation order and the occasional bf16 tie: 2e-3.
    for k, v in sharp.items():
        assert v[1] <= 2e-3, ("vs float64 model", k, v)
    # Against the fp32 oracle the forward carries the bf16 roundings of two chained GEMMs (~4e-3).  The gradients that pass
    # the ReLU carry, on top, the gate's sign flips: a pre-activation within the bf16 error of zero (~0.25 % of them at a
    # relative error of 3e-3) has its gate flipped, and a fraction f of flipped elements is a rel-L2 error of sqrt(2 f) ~ 4e-2
    # whatever the kernel -- which is why the sharp comparison above exists.
    assert res["forward"][1] <= 1e-2, res["forward"]
    assert res["reconstruction"][1] <= 1e-6, res["reconstruction"]       # subtracts exactly what the forward added (stash)
    for k, v in res.items():
        if k.startswith("d_"):
            assert v[1] <= (6e-2 if ("net.0" in k or "norm" in k or k == "d_input") else 1e-2), (k, v)


# ------------------------------------------------------------------ round 4: grouped launches, persistent tiles, delta / accumulate epilogues
def test_grouped_launch_vs_float64(gpu):
    """``rtts_gemm_nt_grouped``: independent problems in ONE launch, each against float64 on its own bf16 operands -- the
    cross-attention's q and k|v projections (reformer.py:161-186; M = 12288 / 3072, bias epilogue), their input-gradient pair (dxn
    bf16 beside dkeys fp32 += ..., different K), three encoder-sized problems on the small tile, and a group whose members need
    different epilogues (bf16 / bias / fp32)."""
    from reformer_tts_amd import engine
    g = torch.Generator(device=gpu).manual_seed(5)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, device=gpu, generator=g) * scale)

    msgs = []
    # (1) q | kv projections
    xn, keys = rnd(12288, 512).bfloat16(), rnd(3072, 512).bfloat16()
    w = rnd(1536, 512, scale=512 ** -0.5).bfloat16()
    bias = rnd(1536)
    q, kv = engine.gemm_group([dict(a=xn, w=w[:512], bias=bias[:512]), dict(a=keys, w=w[512:], bias=bias[512:])])
    for name, got, ref in (("q", q, xn.double() @ w[:512].double().t() + bias[:512].double()),
                           ("kv", kv, keys.double() @ w[512:].double().t() + bias[512:].double())):
        emax, el2 = _rel(got, ref)
        msgs.append(f"{name} max {emax:.2e} l2 {el2:.2e}")
        assert got.dtype == torch.bfloat16 and emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (name, emax, el2)
    # the same two products as separate launches: bit-identical (same tile arithmetic, only the grid differs)
    assert torch.equal(q, _gemm(xn, w[:512], bias=bias[:512])) and torch.equal(kv, _gemm(keys, w[512:], bias=bias[512:]))
    # (2) dxn beside dkeys += dkv W_kv (fp32, accumulated into what is there)
    dq, dkv = rnd(12288, 512).bfloat16(), rnd(3072, 1024).bfloat16()
    dkeys0 = rnd(3072, 512)
    dkeys = dkeys0.clone()
    dxn, same = engine.gemm_group([dict(a=dq, w=w[:512]), dict(a=dkv, w=w[512:], into=dkeys)], kn=True)
    assert same is dkeys
    emax, el2 = _rel(dxn, dq.double() @ w[:512].double())
    assert emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (emax, el2)
    ref = dkeys0.double() + dkv.double() @ w[512:].double()
    e32 = ((dkeys.double() - ref).abs().max() / ref.abs().max()).item()
    msgs.append(f"dxn max {emax:.2e}; dkeys (fp32, accumulated) max {e32:.2e}")
    assert e32 <= 2e-6, e32
    # (3) three small problems (the 96 x 64 tile), mixed epilogues: bf16, bias, fp32
    a1, a2, a3 = rnd(768, 256).bfloat16(), rnd(384, 128).bfloat16(), rnd(192, 64).bfloat16()
    w1, w2, w3 = rnd(128, 256, scale=0.1).bfloat16(), rnd(320, 128, scale=0.1).bfloat16(), rnd(64, 64, scale=0.1).bfloat16()
    b2 = rnd(320)
    o1, o2, o3 = engine.gemm_group([dict(a=a1, w=w1), dict(a=a2, w=w2, bias=b2), dict(a=a3, w=w3, out_f32=True)])
    for name, got, ref, tol in (("small/plain", o1, a1.double() @ w1.double().t(), 1.02 * BF16_HALF_ULP),
                                ("small/bias", o2, a2.double() @ w2.double().t() + b2.double(), 1.02 * BF16_HALF_ULP),
                                ("small/f32", o3, a3.double() @ w3.double().t(), 2e-6)):
        emax, _ = _rel(got, ref)
        msgs.append(f"{name} max {emax:.2e}")
        assert emax <= tol, (name, emax)
    assert o3.dtype == torch.float32
    torch.cuda.synchronize()
    print("\n[grouped gemm_nt vs float64] " + "; ".join(msgs))
    with pytest.raises(Exception, match="1\\.\\."):
        engine.gemm_group([dict(a=a1, w=w1)] * 5)


@pytest.mark.parametrize("m,n,k,epi", [(12288, 2048, 512, "bias+relu"), (12288, 1024, 512, "plain"), (12288, 2048, 512, "kn+gate"),
                                       (12288, 1024, 512, "kn"), (6144, 2048, 256, "plain"), (12288, 4096, 512, "bias")])
def test_persistent_tiles_are_bit_identical_to_one_tile_per_workgroup(gpu, m, n, k, epi):
    """MODE 2 of csrc/gemm_nt.hip: resident workgroups walk the tiles of a problem with several tiles per CU (N >= 1024 at
    M = 12288), the LDS ring turning across tile boundaries.  The arithmetic of a tile is the same instruction sequence in
    every form, so the results -- including the 1-bit gate words and the partial column sums of the gated input gradient --
    must agree BIT FOR BIT with one tile per workgroup, on the 2-deep ring (two workgroups per CU) and on the deep ring."""
    from reformer_tts_amd import _lib, engine
    g = torch.Generator(device=gpu).manual_seed(m + n + k)
    a = torch.randn(m, k, device=gpu, generator=g).bfloat16()
    w = (torch.randn(n, k, device=gpu, generator=g) / k ** 0.5).bfloat16()
    bias = torch.randn(n, device=gpu, generator=g)
    wkn = w.t().contiguous()
    outs = {}
    for mode in (1, 2, 3, 0):
        _lib.call("rtts_debug_set_gemm_mode", mode)
        try:
            if epi == "plain":
                res = (_gemm(a, w),)
            elif epi == "bias":
                res = (_gemm(a, w, bias=bias),)
            elif epi == "kn":
                res = (_gemm(a, wkn, kn=True),)
            elif epi == "bias+relu":
                words = engine.gate_words(m, n, gpu)
                res = (_gemm(a, w, bias=bias, relu=True, words=words), words.clone())
            else:
                words = engine.gate_words(m, n, gpu)
                _gemm(a, w, bias=bias, relu=True, words=words)
                db = torch.zeros(n, device=gpu)
                res = (_gemm(a, wkn, kn=True, gate=True, words=words, gate_bias_grad=db),)
                engine.flush_wgrad()
                res = res + (db.clone(),)
            torch.cuda.synchronize()
        finally:
            _lib.call("rtts_debug_set_gemm_mode", 0)
        outs[mode] = res
    for mode in (2, 3, 0):
        for x, y in zip(outs[mode], outs[1]):
            assert torch.equal(x, y), f"launch form {mode} differs from one tile per workgroup"
    ref = a.double() @ w.double().t()
    if epi in ("plain", "kn"):
        emax, el2 = _rel(outs[0][0], ref)
        assert emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (emax, el2)


@pytest.mark.parametrize("b,t,heads,k", [(12, 1024, 8, 512), (12, 256, 8, 512), (2, 256, 2, 128), (1, 384, 4, 256)])
def test_dgrad_with_delta_epilogue_vs_float64(gpu, b, t, heads, k):
    """Epilogue 5: dout = dy W (the input gradient of to_out / out_proj) and delta[b*H + h][t] = sum over head h's 64 columns of
    out * dout in ONE launch, against float64 (dout: one bf16 rounding; delta: fp32 sums of the ROUNDED dout, as the attention
    backward needs them) and against the separate ``rtts_lsh_bwd_delta`` launch it replaces."""
    from reformer_tts_amd import _lib, engine
    m, n = b * t, 64 * heads
    g = torch.Generator(device=gpu).manual_seed(b * t + heads)
    dy = torch.randn(m, k, device=gpu, generator=g).bfloat16()
    w = (torch.randn(k, n, device=gpu, generator=g) / k ** 0.5).bfloat16()
    out = torch.randn(m, n, device=gpu, generator=g).bfloat16()
    assert engine.dgrad_delta_ok(m, n)
    dout, delta = engine.gemm_dgrad_delta(dy, w, out, t, heads)
    plain = _gemm(dy, w, kn=True)
    torch.cuda.synchronize()
    assert torch.equal(dout, plain), "the delta epilogue must not change dout"
    emax, el2 = _rel(dout, dy.double() @ w.double())
    assert emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (emax, el2)
    ref = (out.double() * dout.double()).view(b, t, heads, 64).sum(-1).permute(0, 2, 1).reshape(b * heads, t)
    ed = ((delta.double() - ref).abs().max() / ref.abs().max()).item()
    sep = torch.empty(b * heads, t, dtype=torch.float32, device=gpu)
    _lib.call("rtts_lsh_bwd_delta", out.data_ptr(), n, dout.data_ptr(), n, b, heads, t, 64, sep.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    es = ((delta - sep).abs().max() / sep.abs().max()).item()
    print(f"\n[dgrad + delta B={b} T={t} H={heads} K={k}] dout max {emax:.2e} l2 {el2:.2e}; delta vs float64 {ed:.2e}, vs rtts_lsh_bwd_delta {es:.2e} (tol 1e-5)")
    assert ed <= 1e-5 and es <= 1e-5, (ed, es)


@pytest.mark.parametrize("m,n,k", [(12288, 512, 512), (3072, 1024, 512), (768, 384, 384), (12288, 2048, 512), (256, 128, 128)])
def test_store_forms_of_the_bf16_epilogues_are_bit_identical(gpu, m, n, k):
    """The bf16 epilogues store 8 bytes per lane straight from the accumulators (form 0), or 16 bytes after the lane pair
    (l, l ^ 16) has traded halves of a column-tile pair (form 1), or the same as write-through stores (form 2: the output
    leaves L2 while the kernel runs instead of being flushed at its end).  Same values at the same addresses: every layout /
    epilogue must agree bit for bit across the forms, gate words, column sums and the delta epilogue included."""
    from reformer_tts_amd import _lib, engine
    g = torch.Generator(device=gpu).manual_seed(m + 3 * n + k)
    a = torch.randn(m, k, device=gpu, generator=g).bfloat16()
    w = (torch.randn(n, k, device=gpu, generator=g) / k ** 0.5).bfloat16()
    bias = torch.randn(n, device=gpu, generator=g)
    wkn = w.t().contiguous()
    aux = torch.randn(m, n, device=gpu, generator=g).bfloat16()
    outs = {}
    for form in (0, 1, 2):
        _lib.call("rtts_debug_set_gemm_mode", 10 + form)
        try:
            words = engine.gate_words(m, n, gpu)
            res = [_gemm(a, w), _gemm(a, w, bias=bias), _gemm(a, wkn, kn=True)]
            if words is not None:
                res.append(_gemm(a, w, bias=bias, relu=True, words=words))
                db = torch.zeros(n, device=gpu)
                res.append(_gemm(a, wkn, kn=True, gate=True, words=words, gate_bias_grad=db))
                engine.flush_wgrad()
                res += [words.clone(), db]
            if n % 64 == 0 and engine.dgrad_delta_ok(m, n) and m % 256 == 0:
                res += list(engine.gemm_dgrad_delta(a, wkn, aux, 256, n // 64))
            res += engine.gemm_group([dict(a=a, w=w, bias=bias), dict(a=a, w=w)])
            torch.cuda.synchronize()
        finally:
            _lib.call("rtts_debug_set_gemm_mode", 9)
        outs[form] = res
    for form in (1, 2):
        assert len(outs[form]) == len(outs[0])
        for idx, (x, y) in enumerate(zip(outs[form], outs[0])):
            assert torch.equal(x, y), f"store form {form}: result {idx} differs from the 8-byte form"
    emax, el2 = _rel(outs[2][0], a.double() @ w.double().t())
    assert emax <= 1.02 * BF16_HALF_ULP and el2 <= 2e-3, (emax, el2)
