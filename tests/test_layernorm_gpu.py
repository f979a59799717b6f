"""GPU parity of the native LayerNorm (C ABI codetr_layernorm_{f16,bf16}) against a plain PyTorch fp32
reference of the same op.  Tolerance: 1 ulp of the output dtype (single rounding of an fp32 result) +
1e-3 abs for the bf16/fp16 rounding of gamma*xhat near zero."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(1, 8), (7, 192), (1000, 256), (513, 384), (300, 768), (77, 1536), (33, 3072),
                                     (5, 4096), (153600, 192), (2, 1000)])
def test_layernorm_vs_fp32(rows, C, dtype):
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(rows + C)
    x = (torch.randn(rows, C, device=DEV, generator=g) * 3 + 1.5).to(dtype)
    w = (1 + 0.2 * torch.randn(C, device=DEV, generator=g)).to(dtype)
    b = (0.3 * torch.randn(C, device=DEV, generator=g)).to(dtype)
    y = hip_ops.layer_norm(x, w, b, 1e-5)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float(), (C,), w.float(), b.float(), 1e-5)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = (y.float() - ref).abs()
    assert (err <= ulp * ref.abs() + 1e-5 + ulp * 1e-2).all(), float(err.max())
    assert y.dtype == dtype and y.shape == x.shape


def test_layernorm_large_offset_rows_two_pass():
    """rows with |mean| >> std: a one-pass E[x^2]-E[x]^2 formulation loses the variance in fp32."""
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(0)
    x = (100.0 + 0.05 * torch.randn(64, 256, device=DEV, generator=g)).half()
    w = torch.ones(256, device=DEV).half()
    b = torch.zeros(256, device=DEV).half()
    y = hip_ops.layer_norm(x, w, b, 1e-5)
    ref = torch.nn.functional.layer_norm(x.float(), (256,), None, None, 1e-5)
    torch.testing.assert_close(y.float(), ref, rtol=2e-3, atol=2e-3)


def test_layernorm_3d_and_noncontiguous():
    from codetr import hip_ops

    x = torch.randn(4, 50, 192, device=DEV).half()
    w = torch.randn(192, device=DEV).half()
    b = torch.randn(192, device=DEV).half()
    ref = torch.nn.functional.layer_norm(x.float(), (192,), w.float(), b.float(), 1e-5)
    torch.testing.assert_close(hip_ops.layer_norm(x, w, b).float(), ref, rtol=2e-3, atol=2e-3)
    xt = x.transpose(0, 1)
    torch.testing.assert_close(hip_ops.layer_norm(xt, w, b).float(), ref.transpose(0, 1), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("B,H,W,C", [(1, 8, 12, 192), (2, 7, 9, 192), (1, 20, 30, 384), (2, 10, 14, 768), (1, 5, 6, 64)])
def test_patch_merge_layernorm_equals_gather_then_layernorm(B, H, W, C):
    """codetr_patch_merge_layernorm_f16 == (2x2 gather in (ky, kx, c) order, zero-padded) -> codetr_layernorm_f16,
    bit for bit (same statistics arithmetic on the same values)."""
    import torch.nn.functional as F
    from codetr import _cabi, hip_ops

    g = torch.Generator(device=DEV).manual_seed(H * 31 + W)
    x = torch.randn(B, H * W, C, device=DEV, generator=g).half()
    gam = (1 + 0.1 * torch.randn(4 * C, device=DEV, generator=g)).half()
    bet = (0.1 * torch.randn(4 * C, device=DEV, generator=g)).half()
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    x4 = F.pad(x.view(B, H, W, C), (0, 0, 0, W % 2, 0, H % 2))
    merged = x4.view(B, H2, 2, W2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H2 * W2, 4 * C)
    ref = hip_ops.layer_norm(merged, gam, bet, 1e-5)
    before = _cabi.CALLS["patch_merge_layernorm"]
    y = hip_ops.patch_merge_layernorm(x, (H, W), gam, bet, 1e-5)      # odd sizes: the kernel pads with zeros itself
    assert _cabi.CALLS["patch_merge_layernorm"] == before + 1
    assert y.shape == ref.shape and torch.equal(y, ref)


def test_patch_merge_layernorm_bf16():
    """bf16 instantiation of the merge + LayerNorm kernel against gather + F.layer_norm in fp32"""
    import torch.nn.functional as F
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(8)
    B, H, W, C = 2, 14, 10, 96
    x = torch.randn(B, H, W, C, device=DEV, generator=g).bfloat16()
    gam = (1 + 0.1 * torch.randn(4 * C, device=DEV, generator=g)).bfloat16()
    bet = (0.1 * torch.randn(4 * C, device=DEV, generator=g)).bfloat16()
    out = _cabi.patch_merge_layernorm(x, gam, bet, 1e-5)
    torch.cuda.synchronize()
    m = x.float().view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, (H // 2) * (W // 2), 4 * C)
    ref = F.layer_norm(m, (4 * C,), gam.float(), bet.float(), 1e-5)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=2e-2)
