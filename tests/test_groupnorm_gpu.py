"""GPU parity of the token-major GroupNorm (C ABI codetr_groupnorm_tokens_f16) against a plain PyTorch fp32
reference: F.group_norm on the NCHW view of the same data.  Tolerance: 1 fp16 ulp of the output + 2e-3 abs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,H,W", [(1, 1, 1), (2, 7, 9), (1, 80, 120), (3, 33, 17), (1, 320, 480)])
def test_groupnorm_tokens_into_slice(B, H, W):
    from codetr import hip_ops

    C, G = 256, 32
    g = torch.Generator(device=DEV).manual_seed(H * W)
    x = (torch.randn(B, H * W, C, device=DEV, generator=g) * 2 + 0.7).half()
    gamma = (1 + 0.2 * torch.randn(C, device=DEV, generator=g)).half()
    beta = (0.3 * torch.randn(C, device=DEV, generator=g)).half()
    S = H * W + 13
    dest = torch.full((B, S, C), 7.0, device=DEV, dtype=torch.float16)
    hip_ops.groupnorm_tokens_into(x, gamma, beta, G, 1e-5, dest, 5)
    torch.cuda.synchronize()
    nchw = x.float().view(B, H, W, C).permute(0, 3, 1, 2)
    ref = torch.nn.functional.group_norm(nchw, G, gamma.float(), beta.float(), 1e-5).permute(0, 2, 3, 1).reshape(B, H * W, C)
    got = dest[:, 5:5 + H * W].float()
    assert ((got - ref).abs() <= 2.0 ** -10 * ref.abs() + 2e-3).all(), float((got - ref).abs().max())
    assert (dest[:, :5] == 7).all() and (dest[:, 5 + H * W:] == 7).all()  # neighbours untouched


def test_groupnorm_tokens_bf16():
    """bf16 storage instantiation against F.group_norm in fp32"""
    import torch.nn.functional as F
    from codetr import _cabi

    g = torch.Generator(device=DEV).manual_seed(4)
    B, HW, C = 2, 777, 256
    x = (torch.randn(B, HW, C, device=DEV, generator=g) * 2 + 0.5).bfloat16()
    gam = (1 + 0.1 * torch.randn(C, device=DEV, generator=g)).bfloat16()
    bet = (0.1 * torch.randn(C, device=DEV, generator=g)).bfloat16()
    out = torch.empty(B, HW + 50, C, dtype=torch.bfloat16, device=DEV)
    _cabi.groupnorm_tokens(x, gam, bet, 32, 1e-5, out[0, 50:], out.shape[1] * C)
    torch.cuda.synchronize()
    ref = F.group_norm(x.float().transpose(1, 2), 32, gam.float(), bet.float(), 1e-5).transpose(1, 2)
    torch.testing.assert_close(out[:, 50:].float(), ref, rtol=1e-2, atol=2e-2)
