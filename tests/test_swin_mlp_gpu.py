"""GPU parity of the fused Swin MLP (C ABI codetr_swin_mlp_{f16,bf16}, csrc/swin_mlp.hip) against a plain PyTorch fp32
reference of the same op, y = x + fc2(gelu(fc1(layer_norm(x)))) (reference codetr/swin.py:331-352), with the tensors the
separate launches would have written -- norm2's output and the hidden activation -- rounded to the storage type between the
steps, and fc2's output rounded before the identity is added.  Tolerance: fp32 accumulation on both sides; the differences are
summation order and which side of a rounding boundary the three intermediate roundings fall: |err| <= 2 ulp of the branch
magnitude + 1 ulp of the result + 2e-3 (3 ulp + 3e-2 for bf16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, g, b, eps, w1, b1, w2, b2):
    dt = x.dtype
    ln = torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), g.float(), b.float(), eps).to(dt).float()
    h = torch.nn.functional.gelu(ln @ w1.float().t() + b1.float()).to(dt).float()
    y = (h @ w2.float().t() + b2.float()).to(dt).float()
    return y + x.float()


def _inputs(M, C, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s, k=1.0: (torch.randn(*s, device=DEV, generator=g) * k).to(dtype)  # noqa: E731
    x = r(M, C, k=1.5) + r(1, C, k=0.5)            # rows with a common offset: the LayerNorm has something to remove
    gam, bet = (1.0 + 0.2 * torch.randn(C, device=DEV, generator=g)).to(dtype), r(C, k=0.2)
    w1, b1 = r(4 * C, C, k=C ** -0.5), r(4 * C, k=0.3)
    w2, b2 = r(C, 4 * C, k=(4 * C) ** -0.5), r(C, k=0.3)
    return x, gam, bet, w1, b1, w2, b2


@pytest.mark.parametrize("C", [192, 384])
@pytest.mark.parametrize("M", [1, 127, 128, 129, 4999, 128 * 256 + 77, 153600])   # tails, one full round + a tail, a whole stage
def test_swin_mlp_vs_fp32(C, M):
    from codetr import _cabi, hip_ops

    x, gam, bet, w1, b1, w2, b2 = _inputs(M, C, torch.float16, seed=M + C)
    before = _cabi.CALLS["swin_mlp"]
    y = hip_ops.swin_mlp(x, gam, bet, 1e-5, w1, b1, w2, b2)
    torch.cuda.synchronize()
    assert _cabi.CALLS["swin_mlp"] == before + 1 and y.shape == x.shape and y.dtype == torch.float16
    ref = _ref(x, gam, bet, 1e-5, w1, b1, w2, b2)
    branch = (ref - x.float()).abs()
    tol = 2.0 ** -10 * (2 * branch + ref.abs()) + 2e-3
    err = (y.float() - ref).abs()
    bad = ~(err <= tol)
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outside; max err {float(err.nan_to_num(1e9).max())}"


@pytest.mark.parametrize("C", [192, 384])
def test_swin_mlp_bf16(C):
    from codetr import hip_ops

    x, gam, bet, w1, b1, w2, b2 = _inputs(40000, C, torch.bfloat16, seed=C)
    y = hip_ops.swin_mlp(x, gam, bet, 1e-5, w1, b1, w2, b2)
    torch.cuda.synchronize()
    ref = _ref(x, gam, bet, 1e-5, w1, b1, w2, b2)
    branch = (ref - x.float()).abs()
    tol = 2.0 ** -7 * (3 * branch + ref.abs()) + 3e-2
    assert ((y.float() - ref).abs() <= tol).all()


def test_repeats_are_bit_identical_and_rows_are_independent():
    """a missed counted wait / an early ring refill shows as a sporadic difference; a row's result must not depend on the tile
    it sits in"""
    from codetr import hip_ops

    x, gam, bet, w1, b1, w2, b2 = _inputs(128 * 300 + 5, 384, torch.float16, seed=9)
    first = hip_ops.swin_mlp(x, gam, bet, 1e-5, w1, b1, w2, b2)
    for _ in range(6):
        assert torch.equal(hip_ops.swin_mlp(x, gam, bet, 1e-5, w1, b1, w2, b2), first)
    shifted = hip_ops.swin_mlp(x[37:].contiguous(), gam, bet, 1e-5, w1, b1, w2, b2)
    assert torch.equal(shifted, first[37:])


def test_swin_block_takes_the_fused_mlp_and_matches_the_separate_launches():
    from codetr import _cabi, hip_ops
    from codetr.swin import SwinBlock

    torch.manual_seed(0)
    blk = SwinBlock(192, 6, 768, window_size=12).to(DEV).half().eval()
    H, W = 192, 180                                   # 34 560 tokens: above SWIN_MLP_MIN_ROWS
    x = torch.randn(1, H * W, 192, device=DEV).half()
    before = dict(_cabi.CALLS)
    with torch.no_grad():
        y = blk(x, (H, W))
    assert _cabi.CALLS["swin_mlp"] == before["swin_mlp"] + 1
    hip_ops.SWIN_MLP = False
    try:
        with torch.no_grad():
            y2 = blk(x, (H, W))
    finally:
        hip_ops.SWIN_MLP = True
    assert _cabi.CALLS["swin_mlp"] == before["swin_mlp"] + 1
    torch.testing.assert_close(y.float(), y2.float(), rtol=2e-3, atol=6e-3)


def test_contract():
    from codetr import _cabi

    lib = _cabi.load()
    assert lib.codetr_swin_mlp_supported(1000, 192, 768) == 1 and lib.codetr_swin_mlp_supported(1000, 384, 1536) == 1
    assert lib.codetr_swin_mlp_supported(1000, 256, 1024) == 0 and lib.codetr_swin_mlp_supported(1000, 192, 512) == 0
    x = torch.zeros(256, 256, dtype=torch.float16, device=DEV)
    st = _cabi.current_stream_ptr(x.device)
    p = x.data_ptr()
    assert lib.codetr_swin_mlp_f16(st, p, p, p, 1e-5, p, p, p, p, p, 256, 256) == -4
    assert lib.codetr_swin_mlp_f16(st, None, p, p, 1e-5, p, p, p, p, p, 256, 192) == -1
