"""Swin stem as patch gather + GEMM (csrc/patch_embed.hip, codetr_patch_im2col_b16 + codetr_linear_*): compared with
the fp32 convolution it replaces (mmdet PatchEmbed: Conv2d(3, E, 4, stride 4) on the image zero-padded to the right /
bottom, reference codetr/swin.py:567, transformer_mmcv.py:100-210).  The gather is integer / byte work and is checked
bit-exactly against the ATen unfold of the same image; the GEMM output within fp16 rounding of the fp32 result."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.mark.parametrize("B,C,H,W", [(2, 3, 64, 96), (1, 3, 61, 83), (3, 3, 30, 29), (1, 4, 16, 260), (1, 1, 7, 5)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_patch_gather_bit_exact(B, C, H, W, dtype):
    from codetr import _cabi

    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H)
    x = torch.randn(B, C, H, W, generator=g).to(DEV).to(dtype)
    Hp, Wp = -(-H // 4), -(-W // 4)
    out = torch.full((B * Hp * Wp, 64), 7.0, dtype=dtype, device=DEV)
    before = _cabi.CALLS["patch_im2col"]
    _cabi.patch_im2col(x, 4, 64, out)
    torch.cuda.synchronize()
    assert _cabi.CALLS["patch_im2col"] == before + 1
    xp = F.pad(x.float(), (0, Wp * 4 - W, 0, Hp * 4 - H))
    want = F.unfold(xp, 4, stride=4).transpose(1, 2).reshape(B * Hp * Wp, C * 16).to(dtype)   # (c, ky, kx) columns
    assert torch.equal(out[:, :C * 16].view(torch.int16), want.view(torch.int16))
    assert (out[:, C * 16:] == 0).all()


@pytest.mark.parametrize("H,W", [(128, 192), (250, 333)])
def test_patch_embed_module_matches_fp32_conv(H, W):
    from codetr import _cabi
    from codetr.swin import PatchEmbed

    torch.manual_seed(3)
    pe = PatchEmbed(in_channels=3, embed_dims=192, kernel_size=4, stride=4, norm_cfg=dict(type="LN")).to(DEV).eval()
    x = torch.randn(2, 3, H, W, device=DEV)
    Hp, Wp = -(-H // 4), -(-W // 4)
    with torch.no_grad():
        ref = F.conv2d(F.pad(x, (0, Wp * 4 - W, 0, Hp * 4 - H)), pe.projection.weight, pe.projection.bias, stride=4)
        ref = F.layer_norm(ref.flatten(2).transpose(1, 2), (192,), pe.norm.weight, pe.norm.bias, pe.norm.eps)
        before = dict(_cabi.CALLS)
        out, hw = pe.half()(x.half())
    assert hw == (Hp, Wp) and out.shape == (2, Hp * Wp, 192)
    assert _cabi.CALLS["patch_im2col"] == before["patch_im2col"] + 1 and _cabi.CALLS["linear"] == before["linear"] + 1
    err = (out.float() - ref).abs().max().item()
    assert err < 2e-2, err      # LayerNorm output is O(1); fp16 operands, fp32 accumulation


def test_large_m_takes_the_short_k_kernel():
    """the stem GEMM at a model-sized token count (M >= 32768, K = 64, N = 192) against the fp32 product"""
    from codetr import hip_ops

    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(40000, 64, device=DEV, generator=g).half()
    w = (torch.randn(192, 64, device=DEV, generator=g) * 0.1).half()
    b = torch.randn(192, device=DEV, generator=g).half()
    y = hip_ops.linear(x, w, b)
    ref = x.float() @ w.float().t() + b.float()
    assert (y.float() - ref).abs().max().item() < 1e-2


@pytest.mark.parametrize("B,H,W,C,k,s,p", [(2, 40, 60, 64, 3, 2, 1), (1, 5, 7, 8, 3, 2, 1), (3, 9, 9, 16, 3, 1, 1),
                                           (1, 8, 6, 24, 2, 2, 0)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_im2col_tokens_bit_exact(B, H, W, C, k, s, p, dtype):
    """token-major k x k patches (the neck's extra 3x3 / stride-2 level) against F.unfold of the NCHW view"""
    from codetr import hip_ops

    g = torch.Generator(device="cpu").manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, generator=g).to(DEV).to(dtype)
    cols = hip_ops.im2col_tokens(x, k, s, p)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    u = F.unfold(x.float().permute(0, 3, 1, 2), k, stride=s, padding=p)          # [B, C*k*k, Ho*Wo], (c, ky, kx)
    want = u.view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B, Ho * Wo, k * k * C).to(dtype)
    assert cols.shape == want.shape
    assert torch.equal(cols.view(torch.int16), want.view(torch.int16))
