"""GPU parity of the fused FFN kernel (C ABI codetr_ffn_relu_f16) against a plain PyTorch fp32 reference of the
same op, y = x + relu(x W1^T + b1) W2^T + b2, with the hidden activation rounded to fp16 between the products (what
the kernel -- and the reference's fp16 path -- does).  Tolerance: fp32 accumulation on both sides; differences are
summation order + the two fp16 roundings of the FFN branch: |err| <= 2 ulp(fp16) of the branch magnitude + 1 ulp
of the result."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w1, b1, w2, b2):
    h = torch.relu(x.float() @ w1.float().t() + b1.float()).half().float()
    y = (h @ w2.float().t() + b2.float()).half().float()
    return (y + x.float())


@pytest.mark.parametrize("M,hidden", [(1, 64), (128, 2048), (129, 2048), (900, 2048), (5000, 512), (30785, 2048),
                                      (128 * 256 + 4999, 256)])  # one full round of 256 tiles + a tail of 64-row tiles
def test_ffn_fused_vs_fp32(M, hidden):
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(M + hidden)
    x = torch.randn(M, 256, device=DEV, generator=g).half()
    w1 = (torch.randn(hidden, 256, device=DEV, generator=g) / 16).half()
    b1 = (torch.randn(hidden, device=DEV, generator=g) * 0.5).half()
    w2 = (torch.randn(256, hidden, device=DEV, generator=g) / hidden ** 0.5).half()
    b2 = (torch.randn(256, device=DEV, generator=g) * 0.5).half()
    before = _cabi.CALLS["ffn_fused"]
    y = hip_ops.ffn_fused(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    assert _cabi.CALLS["ffn_fused"] == before + 1 and y.shape == x.shape and y.dtype == torch.float16
    ref = _ref(x, w1, b1, w2, b2)
    branch = (ref - x.float()).abs()
    tol = 2.0 ** -10 * (2 * branch + ref.abs()) + 2e-3
    err = (y.float() - ref).abs()
    assert (err <= tol).all(), f"max err {float(err.max())} at tol {float(tol.flatten()[err.argmax()])}"


def test_ffn_module_takes_fused_path_and_matches_two_gemm_path():
    from codetr import _cabi, hip_ops
    from codetr.transformer_layers import FFN

    torch.manual_seed(0)
    ffn = FFN(embed_dims=256, feedforward_channels=2048).to(DEV).half().eval()
    x = torch.randn(2, hip_ops.FFN_FUSED_MIN_ROWS // 2 + 77, 256, device=DEV).half()
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        y = ffn(x)
        fc1, fc2 = ffn.layers[0][0], ffn.layers[1]
        h = hip_ops.linear(x, fc1.weight, fc1.bias, act="relu")
        y2 = hip_ops.linear(h, fc2.weight, fc2.bias, residual=x)
    assert _cabi.CALLS["ffn_fused"] == before["ffn_fused"] + 1
    torch.testing.assert_close(y.float(), y2.float(), rtol=2e-3, atol=4e-3)
    # explicit identity / no identity keep the two-GEMM path (different residual semantics)
    with torch.no_grad():
        y3 = ffn(x, identity=torch.zeros_like(x))
    torch.testing.assert_close(y3.float(), (y2.float() - x.float()), rtol=2e-3, atol=6e-3)
    # short inputs (the decoder's 900 queries) stay on the two-GEMM path: one fused block walks the whole hidden dim
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        ys = ffn(x[:, :450])
    assert _cabi.CALLS["ffn_fused"] == before["ffn_fused"] and _cabi.CALLS["linear"] == before["linear"] + 2
    torch.testing.assert_close(ys.float(), y2[:, :450].float(), rtol=2e-3, atol=4e-3)


def test_ffn_layernorm_pos_epilogue_matches_three_kernels():
    """codetr_ffn_relu_ln_f16 == fused FFN -> codetr_layernorm_f16 -> fp16 add.  The epilogue normalises out of the
    accumulator layout: the same two-pass fp32 statistics as layernorm_kernel but summed in a different order (64 in-lane
    values, then the four lanes of a row), so a few outputs differ in the last fp16 bit; the `+ pos` output is exactly
    the kernel's own normalised output + pos."""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(3)
    M = 128 * 256 + 3333  # full round + 64-row tail launch
    x = torch.randn(M, 256, device=DEV, generator=g).half()
    pos = torch.randn(M, 256, device=DEV, generator=g).half()
    w1 = (torch.randn(2048, 256, device=DEV, generator=g) / 16).half()
    b1 = torch.randn(2048, device=DEV, generator=g).half()
    w2 = (torch.randn(256, 2048, device=DEV, generator=g) / 45).half()
    b2 = torch.randn(256, device=DEV, generator=g).half()
    gam = (1 + 0.1 * torch.randn(256, device=DEV, generator=g)).half()
    bet = (0.1 * torch.randn(256, device=DEV, generator=g)).half()
    y0 = hip_ops.ffn_fused(x, w1, b1, w2, b2)
    n0 = hip_ops.layer_norm(y0, gam, bet, 1e-5)
    before = _cabi.CALLS["layernorm"]
    n1 = hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5))
    n2, q2 = hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5), pos=pos)
    assert _cabi.CALLS["layernorm"] == before
    assert torch.equal(n1, n2)
    assert (n1 != n0).float().mean().item() < 0.02
    torch.testing.assert_close(n1.float(), n0.float(), rtol=0, atol=4e-3)   # 1 fp16 ulp at |value| < 4
    assert torch.equal(q2, n2 + pos)
    y3, q3 = hip_ops.ffn_fused(x, w1, b1, w2, b2, pos=pos)  # second output without the norm
    assert torch.equal(y3, y0) and torch.equal(q3, y0 + pos)
    # input LayerNorm folded in: == layer_norm -> fused FFN (+ LN + pos).  The statistics are summed in a different
    # lane order than layernorm_kernel's, so x1 may differ in the last fp16 bit on a few elements: 1-ulp tolerance on
    # the normalised output scale instead of bit equality
    gi = (1 + 0.1 * torch.randn(256, device=DEV, generator=g)).half()
    bi = (0.1 * torch.randn(256, device=DEV, generator=g)).half()
    t = (x * 3 + 1).half()
    x1 = hip_ops.layer_norm(t, gi, bi, 1e-5)
    r0, rq0 = hip_ops.ffn_fused(x1, w1, b1, w2, b2, ln=(gam, bet, 1e-5), pos=pos)
    before = _cabi.CALLS["layernorm"]
    r1, rq1 = hip_ops.ffn_fused(t, w1, b1, w2, b2, ln=(gam, bet, 1e-5), pos=pos, ln_in=(gi, bi, 1e-5))
    assert _cabi.CALLS["layernorm"] == before
    torch.testing.assert_close(r1.float(), r0.float(), rtol=0, atol=4e-3)
    torch.testing.assert_close(rq1.float(), rq0.float(), rtol=0, atol=8e-3)
    assert (r1 != r0).float().mean() < 0.02   # the vast majority of elements are bit-identical
    r2 = hip_ops.ffn_fused(t, w1, b1, w2, b2, ln_in=(gi, bi, 1e-5))    # input norm only
    torch.testing.assert_close(r2.float(), hip_ops.ffn_fused(x1, w1, b1, w2, b2).float(), rtol=0, atol=2e-2)


def test_encoder_layers_chain_through_the_fused_epilogue():
    """A post-norm (self_attn, norm, ffn, norm) encoder: LN2 and the next layer's `+ query_pos` come out of the FFN
    kernel (no separate layer-norm / add launches for them) and the result equals the unfused walk."""
    from codetr import _cabi, hip_ops
    from codetr.transformer import DetrTransformerEncoder

    torch.manual_seed(0)
    cfg = dict(type="BaseTransformerLayer",
               attn_cfgs=dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=2, dropout=0.0),
               feedforward_channels=2048, ffn_dropout=0.0, operation_order=("self_attn", "norm", "ffn", "norm"))
    enc = DetrTransformerEncoder(transformerlayers=cfg, num_layers=3).to(DEV).half().eval()
    shapes = [(160, 160), (40, 40)]
    S = sum(h * w for h, w in shapes)
    assert S >= hip_ops.FFN_FUSED_MIN_ROWS
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(1, S, 256, device=DEV, generator=g).half()
    pos = torch.randn(1, S, 256, device=DEV, generator=g).half()
    ref = torch.rand(1, S, 2, 2, device=DEV, generator=g).half()
    ss = torch.tensor(shapes, device=DEV)
    ls = torch.tensor([0, shapes[0][0] * shapes[0][1]], device=DEV)
    mask = torch.zeros(1, S, dtype=torch.bool, device=DEV)
    kw = dict(reference_points=ref, spatial_shapes=ss, level_start_index=ls)
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        out = enc.forward_bf(q, pos, mask, **kw)
    assert _cabi.CALLS["ffn_fused"] - before["ffn_fused"] == 3
    assert _cabi.CALLS["layernorm"] - before["layernorm"] == 0      # both norms of every layer ride in the FFN kernel
    saved = hip_ops.FFN_FUSED_MIN_ROWS
    try:
        hip_ops.FFN_FUSED_MIN_ROWS = 1 << 60                      # two-GEMM FFN, separate norms and adds
        with torch.no_grad():
            out0 = enc.forward_bf(q, pos, mask, **kw)
    finally:
        hip_ops.FFN_FUSED_MIN_ROWS = saved
    torch.testing.assert_close(out.float(), out0.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,hidden", [(129, 2048), (5000, 512), (30785, 2048)])
def test_ffn_fused_bf16_vs_fp32(M, hidden):
    """the bf16 instantiation (codetr_ffn_relu_ln2_bf16): hidden activation rounded to bf16 like the two-GEMM path"""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(M + hidden)
    bf = torch.bfloat16
    x = torch.randn(M, 256, device=DEV, generator=g).to(bf)
    w1 = (torch.randn(hidden, 256, device=DEV, generator=g) / 16).to(bf)
    b1 = (torch.randn(hidden, device=DEV, generator=g) * 0.5).to(bf)
    w2 = (torch.randn(256, hidden, device=DEV, generator=g) / hidden ** 0.5).to(bf)
    b2 = (torch.randn(256, device=DEV, generator=g) * 0.5).to(bf)
    before = _cabi.CALLS["ffn_fused"]
    y = hip_ops.ffn_fused(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    assert _cabi.CALLS["ffn_fused"] == before + 1 and y.dtype == bf
    h = torch.relu(x.float() @ w1.float().t() + b1.float()).to(bf).float()
    ref = (h @ w2.float().t() + b2.float()).to(bf).float() + x.float()
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=3e-2)   # bf16: 8 significant bits


def test_ffn_bf16_layernorm_pos_epilogue_matches_three_kernels():
    """bf16: fused (input LN, FFN, LN, + pos) against layer_norm -> fused FFN -> layer_norm -> add"""
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(5)
    bf = torch.bfloat16
    M = 128 * 64 + 77
    mk = lambda *s, sc=1.0: (torch.randn(*s, device=DEV, generator=g) * sc).to(bf)  # noqa: E731
    x, pos = mk(M, 256), mk(M, 256)
    w1, b1, w2, b2 = mk(2048, 256, sc=1 / 16), mk(2048), mk(256, 2048, sc=1 / 45), mk(256)
    gam, bet = (1 + mk(256, sc=0.1).float()).to(bf), mk(256, sc=0.1)
    y0 = hip_ops.ffn_fused(x, w1, b1, w2, b2)
    n0 = hip_ops.layer_norm(y0, gam, bet, 1e-5)
    n1, q1 = hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5), pos=pos)
    assert (n1 != n0).float().mean().item() < 0.02 and torch.equal(q1, n1 + pos)   # (summation order: see the f16 test)
    torch.testing.assert_close(n1.float(), n0.float(), rtol=0, atol=3.2e-2)         # 1 bf16 ulp at |value| < 4
    x1 = hip_ops.layer_norm(x, gam, bet, 1e-5)
    r0 = hip_ops.ffn_fused(x1, w1, b1, w2, b2, ln=(gam, bet, 1e-5))
    r1 = hip_ops.ffn_fused(x, w1, b1, w2, b2, ln=(gam, bet, 1e-5), ln_in=(gam, bet, 1e-5))
    torch.testing.assert_close(r1.float(), r0.float(), rtol=0, atol=6e-2)
    assert (r1 != r0).float().mean() < 0.05


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,with_ln_in,with_pos", [(128 * 256 + 3333, True, True), (30785, True, False), (128 * 3 + 5, False, False)])
def test_ffn_with_output_projection_folded_in(M, with_ln_in, with_pos, dtype):
    """codetr_ffn_oproj_relu_ln2_*: x0 = identity + E(attn Wo^T + bo); x1 = LN_in(x0); y = LN(x1 + ffn(x1)) (+ pos) in ONE
    launch == codetr_linear_* with the residual epilogue, then the fused FFN kernel with ln_in.  The out-projection
    accumulates in a different order than the GEMM kernel and the norm statistics in a different lane order, so a few x1
    values differ in their last bit: 1 ulp of the normalised output scale, the vast majority bit-identical; and against the
    fp32 composition of the same operands."""
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(M)
    r = lambda *s, k=1.0: (torch.randn(*s, device=DEV, generator=g) * k).to(dtype)   # noqa: E731
    attn, ident, pos = r(M, 256), r(M, 256, k=2.0), r(M, 256)
    wo, bo = r(256, 256, k=1 / 16), r(256, k=0.5)
    w1, b1 = r(2048, 256, k=1 / 16), r(2048)
    w2, b2 = r(256, 2048, k=1 / 45), r(256)
    gam, bet = (1 + 0.1 * torch.randn(256, device=DEV, generator=g)).to(dtype), r(256, k=0.1)
    gi, bi = (1 + 0.1 * torch.randn(256, device=DEV, generator=g)).to(dtype), r(256, k=0.1)
    ln, ln_in = (gam, bet, 1e-5), ((gi, bi, 1e-5) if with_ln_in else None)
    with torch.no_grad():
        x0 = hip_ops.linear(attn, wo, bo, residual=ident)
        ref = hip_ops.ffn_fused(x0, w1, b1, w2, b2, ln=ln, pos=pos if with_pos else None, ln_in=ln_in)
        before = dict(_cabi.CALLS)
        got = hip_ops.ffn_oproj_fused(attn, wo, bo, ident, w1, b1, w2, b2, ln, pos=pos if with_pos else None, ln_in=ln_in)
        torch.cuda.synchronize()
        assert _cabi.CALLS["ffn_oproj_fused"] == before["ffn_oproj_fused"] + 1
        assert _cabi.CALLS["linear"] == before["linear"] and _cabi.CALLS["layernorm"] == before["layernorm"]
    if not with_pos:
        ref, got = (ref,), (got,)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    for a, b, scale in zip(got, ref, (4.0, 8.0)):
        assert a.shape == b.shape and a.dtype == dtype
        err = (a.float() - b.float()).abs()
        assert float(err.max()) <= 4 * ulp * scale, float(err.max())
        assert float((a != b).float().mean()) < (0.03 if dtype == torch.float16 else 0.06)
    if with_pos:
        assert torch.equal(got[1], got[0] + pos)
    # fp32 composition of the same operands (roundings at the same places as the kernels': E after the out-projection,
    # after the residual, after the norm, after the hidden activation, after the second product, after the last residual)
    E = lambda t: t.to(dtype).float()   # noqa: E731
    f0 = E(E(attn.float() @ wo.float().t() + bo.float()) + ident.float())
    f1 = E(torch.nn.functional.layer_norm(f0, (256,), gi.float(), bi.float(), 1e-5)) if with_ln_in else f0
    h = E(torch.relu(f1 @ w1.float().t() + b1.float()))
    f2 = E(E(h @ w2.float().t() + b2.float()) + f1)
    fy = torch.nn.functional.layer_norm(f2, (256,), gam.float(), bet.float(), 1e-5)
    err = (got[0].float() - fy).abs()
    # a 1-ulp flip of x1 moves the normalised output by a few ulp on the rows it touches: bound the mean tightly, the max loosely
    assert float(err.mean()) < ulp * 1.0 and float(err.max()) < 24 * ulp * 4.0, (float(err.mean()), float(err.max()))


def test_encoder_layer_defers_the_output_projection_into_the_ffn_launch():
    """DetrTransformerEncoder on the packed MSDA path: per layer ONE projection launch, the MSDA kernel, ONE FFN launch
    (output_proj, norm1, FFN, norm2, + query_pos inside) -- and the same result as with the switch off."""
    from codetr import _cabi, hip_ops
    from codetr.transformer import DetrTransformerEncoder

    torch.manual_seed(0)
    cfg = dict(type="BaseTransformerLayer",
               attn_cfgs=dict(type="MultiScaleDeformableAttention", embed_dims=256, num_levels=5, dropout=0.0),
               feedforward_channels=2048, ffn_dropout=0.0, operation_order=("self_attn", "norm", "ffn", "norm"))
    enc = DetrTransformerEncoder(transformerlayers=cfg, num_layers=2).to(DEV).half().eval()
    shapes = [(160, 200), (80, 100), (40, 50), (20, 25), (10, 13)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(1, S, 256, device=DEV, generator=g).half()
    pos = torch.randn(1, S, 256, device=DEV, generator=g).half()
    from codetr.transformer import _shape_tensors

    ss, ls = _shape_tensors(shapes, torch.device(DEV))
    mask = torch.zeros(1, S, dtype=torch.bool, device=DEV)
    vr = torch.ones(1, 5, 2, device=DEV, dtype=torch.float16)
    _, ref, _, _ = hip_ops.encoder_geometry(vr, mask, shapes)
    # fp32 reference points from the valid pixel counts (the packed kernel's contract): an unpadded image
    ref._codetr_valid_counts = torch.tensor([[[w, h] for h, w in shapes]], dtype=torch.float32, device=DEV)
    kw = dict(reference_points=ref, spatial_shapes=ss, level_start_index=ls)
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        out = enc.forward_bf(q, pos, mask, **kw)
    d = {k: _cabi.CALLS[k] - before[k] for k in ("ffn_oproj_fused", "encoder_projections", "msda_encoder_packed", "layernorm")}
    assert d == {"ffn_oproj_fused": 2, "encoder_projections": 2, "msda_encoder_packed": 2, "layernorm": 0}, d
    hip_ops.FFN_OPROJ = False
    try:
        with torch.no_grad():
            out0 = enc.forward_bf(q, pos, mask, **kw)
    finally:
        hip_ops.FFN_OPROJ = True
    torch.testing.assert_close(out.float(), out0.float(), rtol=0, atol=1.6e-2)
    assert float((out.float() - out0.float()).abs().mean()) < 1e-3
