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
